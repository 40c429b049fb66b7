/*
 * msk_render.h — C ABI of the batched camera pipeline (depth + segmentation).
 *
 * Stands in for what ManiSkill reaches through `sapien.render` on its camera hot path
 * (mani_skill/envs/scene.py:382-427 update_render, :1026-1110 GPU render setup and camera groups;
 * mani_skill/utils/structs/render_camera.py:160-182,269-273 take_picture / get_picture_cuda;
 * mani_skill/render/shaders.py:68-84,141-145: the `minimal` shader pack's `PositionSegmentation`
 * texture, r16g16b16a16sint = camera-space OpenGL xyz in millimetres + per-scene segmentation id,
 * background 0).  Render shapes are attached to the bodies of the env template (one description,
 * every env draws it at its own poses, read straight from the simulator's env records — the
 * `set_cuda_poses(px.cuda_rigid_body_data)` coupling of scene.py:1026-1037 without a copy).
 *
 * Conventions (mani_skill/utils/sapien_utils.py:320-324, structs/render_camera.py:140-141):
 * camera frame x forward, y left, z up; OpenGL frame x right, y up, -z forward; a pixel belongs to
 * the nearest surface whose projection covers the pixel centre (i + 0.5, j + 0.5), row 0 at the top.
 */
#ifndef MSK_RENDER_H
#define MSK_RENDER_H

#include "msk_physx.h"

#ifdef __cplusplus
extern "C" {
#endif

#define MSK_MAX_RENDER_VERTS 4096
#define MSK_MAX_RENDER_TRIS 8192
#define MSK_MAX_RENDER_SHAPES 128
#define MSK_MAX_CAMERAS 4

/* RenderShapeTriangleMesh / Box / Plane attached to a body (RenderBodyComponent.attach,
 * building/actor_builder.py:166-191).  body = -1: static (env frame).  verts: nverts*3 floats in shape-local
 * coordinates; tris: ntris*3 vertex indices, counter-clockwise seen from outside.  seg_id = Entity.per_scene_id
 * written to the segmentation channel (sapien_env.py:1254-1265).  Call after msk_finalize, before
 * msk_render_finalize.  Returns the render shape index. */
int msk_render_add_mesh(msk_ctx* ctx, int body, const float local_pose[7], const float* verts, int nverts,
                        const int32_t* tris, int ntris, int seg_id);
/* RenderMaterial(base_color=rgba) of a render shape (building/actor_builder.py:166-191); default (0.8, 0.8, 0.8, 1). */
int msk_render_set_base_color(msk_ctx* ctx, int render_shape, const float rgba[4]);
/* RenderMaterial.base_color_texture = RenderTexture2D(...) on a RenderShapeTriangleMesh with uvs (utils/building/ground.py:62-108: the
 * grid texture of every scene's floor; building/actor_builder.py visuals with textured materials): `rgba` = height x width x 4 bytes, row 0
 * first, `uvs` = two floats per vertex of the render shape in the order they were handed to msk_render_add_mesh; (0, 0) is the top-left
 * texel corner, addressing repeats.  The Color texture then shows, per pixel, the texel under the pixel centre (perspective-correct
 * interpolation of the uvs; nearest texel of the mip level floor(log2(pixel footprint in texels)), the levels being 2 x 2 box averages
 * built here) times the triangle's flat shade -- texel * shade / 255 per channel, rounded -- instead
 * of the base colour alone; texel bytes are used as stored (no sRGB decoding).  At most MSK_MAX_TEXTURES textures of MSK_MAX_TEXELS texels in
 * total per context, mip levels included (4 / 3 of the images).  Only k_render_splat (the default kernel) samples textures; under MSK_RENDER_MODE=0 shapes keep their base colour.
 * Call after msk_render_add_mesh, before msk_render_finalize. */
#define MSK_MAX_TEXTURES 8
#define MSK_MAX_TEXELS (1 << 20)
int msk_render_set_texture(msk_ctx* ctx, int render_shape, const uint8_t* rgba, int width, int height, const float* uvs);
/* The render shape draws a per-env box instance (msk_declare_env_box): its vertices — give it the unit box, corners at +-1 — are
 * multiplied component-wise by the env's half sizes of `shape`, and its local position is the env's. */
int msk_render_bind_env_box(msk_ctx* ctx, int render_shape, int shape);
/* scene.set_ambient_light + add_directional_light (envs/sapien_env.py:849-853; envs/scene.py:566-718): at most 4 directional
 * lights, directions in the sub-scene frame.  Default: ManiSkill's default lighting (ambient 0.3; (1, 1, -1) and (0, 0, -1), white). */
int msk_render_set_lights(msk_ctx* ctx, const float ambient[3], int ndir, const float* directions, const float* colors);
/* scene.add_point_light / add_spot_light (envs/scene.py:582-640; RenderPointLightComponent, RenderSpotLightComponent): at most
 * MSK_MAX_LOCAL_LIGHTS lights at fixed places of the sub-scene frame, MSK_LOCAL_LIGHT_FLOATS floats each:
 *   [0..2] position, [3..5] axis the light looks along (spot), [6..8] colour (radiant intensity: irradiance = colour / distance^2),
 *   [9] inner_fov, [10] outer_fov in radians (full cone angles; inner_fov = 0: a point light), [11] reserved.
 * Flat shading evaluates them at each triangle's centroid; no shadows (none of the lights casts one here). */
#define MSK_MAX_LOCAL_LIGHTS 8
#define MSK_LOCAL_LIGHT_FLOATS 12
int msk_render_set_local_lights(msk_ctx* ctx, int n, const float* lights);
/* RenderSystemGroup creation + set_cuda_poses (scene.py:1026-1037): uploads the geometry. */
int msk_render_finalize(msk_ctx* ctx);
/* RenderCameraComponent(width, height) + set_fovy(fovy, compute_x=True) + near / far + local pose
 * (scene.py:198-297; sensors/camera.py:126-186).  mount_body = -1: the camera is fixed in the env frame.
 * width and height must be multiples of 16, at most 4096 tiles of 8 x 8 pixels (512 x 512: the human-render cameras).  Returns the camera id. */
int msk_camera_create(msk_ctx* ctx, int width, int height, float fovy, float near_plane, float far_plane,
                      int mount_body, const float local_pose[7]);
/* camera_group.get_picture_cuda("PositionSegmentation"): device pointer to int16 [num_envs][height][width][4],
 * valid until msk_destroy; shape gets the four extents. */
void* msk_camera_buffer(msk_ctx* ctx, int camera, int64_t shape[4]);
/* Camera.get_obs(depth=True, segmentation=True) under the minimal pack's texture transform
 * (sensors/camera.py:190-242; render/shaders.py:141-145: depth = -position[..., [2]], segmentation = position[..., [3]]):
 * device pointers to int16 [num_envs][height][width][1] planes (uint8 x 4 for MSK_CAM_COLOR) that msk_camera_take_picture fills together with the
 * PositionSegmentation texture, so the observation needs no gather pass over the texture. */
/* MSK_CAM_COLOR: the `Color` texture of the minimal pack, r8g8b8a8unorm = uint8 [num_envs][height][width][4]
 * (render/shaders.py:68-74,141-144: rgb = Color[..., :3]); background (0, 0, 0, 0); rendered from the first request of this
 * buffer on (a camera that only serves depth / segmentation skips shading and the store).  The pack's GLSL is not in the
 * reference tree; this backend shades flat per triangle: base_color * min(1, ambient + sum_l light_l * max(0, n . -dir_l) + point / spot lights
 * at the triangle's centroid, inverse-square, see msk_render_set_local_lights). */
enum msk_camera_plane { MSK_CAM_DEPTH = 0, MSK_CAM_SEGMENTATION = 1, MSK_CAM_COLOR = 2 };
void* msk_camera_obs_buffer(msk_ctx* ctx, int camera, int which, int64_t shape[4]);
/* Which outputs msk_camera_take_picture fills from now on.  position_texture = 0: the caller only reads the depth / segmentation planes (Camera.get_obs with
 * position=False -- every ManiSkill obs mode but the ones with `position`, sensors/camera.py:190-242) and the int16 x 4 PositionSegmentation texture is neither
 * computed (camera-space x, y) nor stored: 8 of the 12 bytes a pixel costs; the texture's contents are then undefined until a picture is taken with
 * position_texture = 1 (the default).  The planes are the same bits either way.  The argument is a bit mask: MSK_CAM_OUT_POSITION_TEXTURE (1) as above,
 * MSK_CAM_OUT_NO_COLOR (2): Color is neither shaded nor stored although its buffer was asked for (SAPIEN renders every texture of the shader pack whatever the
 * observation mode reads; a caller that knows the mode -- depth + segmentation: BASELINE config 3 -- switches the rest off); the buffer keeps its old contents. */
enum { MSK_CAM_OUT_POSITION_TEXTURE = 1, MSK_CAM_OUT_NO_COLOR = 2 };
int msk_camera_set_outputs(msk_ctx* ctx, int camera, int position_texture);
/* render_system_group.update_render() + camera_group.take_picture(): rasterises every env. */
int msk_camera_take_picture(msk_ctx* ctx, int camera, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MSK_RENDER_H */
