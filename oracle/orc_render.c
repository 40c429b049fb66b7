/*
 * orc_render.c — scalar CPU restatement of the depth + segmentation camera (TEST INFRASTRUCTURE ONLY).
 *
 * Stands in for `render_system_group.update_render()` + `camera_group.take_picture()` +
 * `get_picture_cuda("PositionSegmentation")` (mani_skill/envs/scene.py:382-427,
 * mani_skill/utils/structs/render_camera.py:160-182,269-273) under the `minimal` shader contract of
 * mani_skill/render/shaders.py:68-84,141-145: int16 x 4 per pixel = camera-space OpenGL position in
 * millimetres + segmentation id, background 0.  The arithmetic behind those calls (SAPIEN's Vulkan
 * rasteriser) is not in the reference tree: parity unpinned against it.  What is restated here is the
 * textbook pipeline the HIP kernels implement — camera-frame vertices, near-plane clip, projection
 * with the reference's camera conventions (x forward / y left / z up, pixel centres at +0.5, row 0 on
 * top), back-face cull, edge functions, nearest surface per pixel — one env, one pixel, one triangle
 * at a time, in triangle order.
 */
#include <stdlib.h>
#include <string.h>

#include "../include/msk_render.h"
#include "orc_sim.h"

#define ORC_EXPORT __attribute__((visibility("default")))

typedef struct { int body, seg; pose local; float color[4]; int xs; int tex; /* texture id or -1 */ int v0; /* first vertex */ } r_shape;
typedef struct { int w, h; uint32_t* texels; /* r in the low byte */ } r_texture;
#define ORC_MAX_LIGHTS 4
typedef struct { int v0, v1, v2, shape; } r_tri;
typedef struct {
  int W, H, mount;
  float fx, fy, cx, cy, near_, far_;
  pose local;
  int16_t* out;
  int16_t *depth, *seg;
  uint32_t* color;   /* Color r8g8b8a8unorm, r in the low byte */
  int no_color;      /* msk_camera_set_outputs: MSK_CAM_OUT_NO_COLOR -- the Color buffer keeps its contents */
} r_camera;
typedef struct {
  int nv, nt, ns, finalized, ncams;
  r_shape shapes[MSK_MAX_RENDER_SHAPES];
  v3 verts[MSK_MAX_RENDER_VERTS];
  float vuv[MSK_MAX_RENDER_VERTS][2];
  int ntex, ntexels;
  r_texture tex[MSK_MAX_TEXTURES];
  unsigned char vshape[MSK_MAX_RENDER_VERTS];
  r_tri tris[MSK_MAX_RENDER_TRIS];
  r_camera cams[MSK_MAX_CAMERAS];
  float ambient[3];
  int nlights;
  float ldir[ORC_MAX_LIGHTS][3], lcol[ORC_MAX_LIGHTS][3];
  /* point and spot lights (positions / axes in the env frame): msk_render_set_local_lights */
  int nlocal;
  float ppos[MSK_MAX_LOCAL_LIGHTS][3], pdir[MSK_MAX_LOCAL_LIGHTS][3], pcol[MSK_MAX_LOCAL_LIGHTS][3], pcone[MSK_MAX_LOCAL_LIGHTS][2];
} r_model;

typedef struct {
  float A0, B0, C0, A1, B1, C1, A2, B2, C2, Aw, Bw, Cw;
  int seg, prim, x0, x1, y0, y1;
  uint32_t color;
  int tex;                          /* texture id or -1 */
  float Au, Bu, Cu, Av, Bv, Cv;     /* u / depth and v / depth are affine on the screen, like 1 / depth */
} r_setup;

/* inspection for tools/oracle_render_stats.py: the pixel bounding boxes of one env's screen triangles of the next picture */
static int32_t* g_dbg_boxes = NULL;
static int g_dbg_env = -1, g_dbg_max = 0, g_dbg_n = 0;
ORC_EXPORT void orc_render_debug_capture(int env, int32_t* boxes, int max_boxes) { g_dbg_env = env; g_dbg_boxes = boxes; g_dbg_max = max_boxes; g_dbg_n = 0; }
ORC_EXPORT int orc_render_debug_count(void) { return g_dbg_n; }

static int rfail(orc_ctx* c, int code, const char* msg) {
  strncpy(c->err, msg, sizeof(c->err) - 1);
  return code;
}
static pose r_pose7(const float* p) {
  pose r;
  r.p = v3_make(p[0], p[1], p[2]);
  r.q = quat_normalize(quat_make(p[3], p[4], p[5], p[6]));
  return r;
}

ORC_EXPORT int orc_render_add_mesh(orc_ctx* c, int body, const float local_pose[7], const float* verts, int nverts,
                                   const int32_t* tris, int ntris, int seg_id) {
  if (!c->finalized) return rfail(c, MSK_ERR_INVALID, "render shapes are added after finalize");
  if (!c->render) { /* ManiSkill's default lighting (envs/sapien_env.py:849-853) */
    c->render = calloc(1, sizeof(r_model));
    r_model* r0 = (r_model*)c->render;
    r0->ambient[0] = r0->ambient[1] = r0->ambient[2] = 0.3f;
    r0->nlights = 2;
    const float inv3 = 1.0f / sqrtf(3.0f);
    r0->ldir[0][0] = inv3; r0->ldir[0][1] = inv3; r0->ldir[0][2] = -inv3;
    r0->ldir[1][0] = 0.0f; r0->ldir[1][1] = 0.0f; r0->ldir[1][2] = -1.0f;
    for (int l = 0; l < 2; ++l) r0->lcol[l][0] = r0->lcol[l][1] = r0->lcol[l][2] = 1.0f;
  }
  r_model* r = (r_model*)c->render;
  if (r->finalized) return rfail(c, MSK_ERR_INVALID, "render_add_mesh after render_finalize");
  if (r->ns >= MSK_MAX_RENDER_SHAPES || r->nv + nverts > MSK_MAX_RENDER_VERTS || r->nt + ntris > MSK_MAX_RENDER_TRIS)
    return rfail(c, MSK_ERR_CAPACITY, "render geometry capacity exceeded");
  r->shapes[r->ns].body = body;
  r->shapes[r->ns].seg = seg_id;
  r->shapes[r->ns].local = r_pose7(local_pose);
  r->shapes[r->ns].color[0] = r->shapes[r->ns].color[1] = r->shapes[r->ns].color[2] = 0.8f;
  r->shapes[r->ns].color[3] = 1.0f;
  r->shapes[r->ns].xs = -1;
  r->shapes[r->ns].tex = -1;
  r->shapes[r->ns].v0 = r->nv;
  for (int i = 0; i < nverts; ++i) {
    r->verts[r->nv + i] = v3_make(verts[3 * i], verts[3 * i + 1], verts[3 * i + 2]);
    r->vshape[r->nv + i] = (unsigned char)r->ns;
  }
  for (int i = 0; i < ntris; ++i) {
    r_tri* t = &r->tris[r->nt + i];
    t->v0 = r->nv + tris[3 * i]; t->v1 = r->nv + tris[3 * i + 1]; t->v2 = r->nv + tris[3 * i + 2]; t->shape = r->ns;
  }
  r->nv += nverts; r->nt += ntris;
  return r->ns++;
}

ORC_EXPORT int orc_render_set_base_color(orc_ctx* c, int render_shape, const float rgba[4]) {
  r_model* r = (r_model*)c->render;
  if (!r || render_shape < 0 || render_shape >= r->ns) return rfail(c, MSK_ERR_INVALID, "bad render shape");
  if (r->finalized) return rfail(c, MSK_ERR_INVALID, "render_set_base_color after render_finalize");
  for (int k = 0; k < 4; ++k) r->shapes[render_shape].color[k] = rgba[k];
  return MSK_OK;
}

ORC_EXPORT int orc_render_set_texture(orc_ctx* c, int render_shape, const uint8_t* rgba, int width, int height, const float* uvs) {
  r_model* r = (r_model*)c->render;
  if (!r || render_shape < 0 || render_shape >= r->ns) return rfail(c, MSK_ERR_INVALID, "bad render shape");
  if (r->finalized) return rfail(c, MSK_ERR_INVALID, "render_set_texture after render_finalize");
  if (!rgba || !uvs || width <= 0 || height <= 0 || width > 4096 || height > 4096) return rfail(c, MSK_ERR_INVALID, "render_set_texture: bad texture");
  /* the texture and its mip chain (each level the 2 x 2 box average of the one before, rounded; edges clamp), level after level */
  int total = 0;
  for (int w = width, h = height;; w = w > 1 ? w / 2 : 1, h = h > 1 ? h / 2 : 1) { total += w * h; if (w == 1 && h == 1) break; }
  if (r->ntex >= MSK_MAX_TEXTURES || r->ntexels + total > MSK_MAX_TEXELS) return rfail(c, MSK_ERR_CAPACITY, "texture capacity exceeded");
  r_texture* t = &r->tex[r->ntex];
  t->w = width; t->h = height;
  t->texels = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)total);
  for (int i = 0; i < width * height; ++i)
    t->texels[i] = (uint32_t)rgba[4 * i] | ((uint32_t)rgba[4 * i + 1] << 8) | ((uint32_t)rgba[4 * i + 2] << 16) | ((uint32_t)rgba[4 * i + 3] << 24);
  {
    uint32_t* src = t->texels;
    for (int w = width, h = height; !(w == 1 && h == 1);) {
      const int w2 = w > 1 ? w / 2 : 1, h2 = h > 1 ? h / 2 : 1;
      uint32_t* dst = src + (size_t)w * h;
      for (int y = 0; y < h2; ++y)
        for (int x = 0; x < w2; ++x) {
          const int x0 = w > 1 ? 2 * x : 0, x1 = w > 1 ? 2 * x + 1 : 0, y0 = h > 1 ? 2 * y : 0, y1 = h > 1 ? 2 * y + 1 : 0;
          uint32_t o = 0;
          for (int ch = 0; ch < 4; ++ch) {
            const uint32_t sum = ((src[y0 * w + x0] >> (8 * ch)) & 0xFFu) + ((src[y0 * w + x1] >> (8 * ch)) & 0xFFu) +
                                 ((src[y1 * w + x0] >> (8 * ch)) & 0xFFu) + ((src[y1 * w + x1] >> (8 * ch)) & 0xFFu);
            o |= ((sum + 2u) >> 2) << (8 * ch);
          }
          dst[y * w2 + x] = o;
        }
      src = dst; w = w2; h = h2;
    }
  }
  r->ntexels += total;
  const int v0 = r->shapes[render_shape].v0, v1 = render_shape + 1 < r->ns ? r->shapes[render_shape + 1].v0 : r->nv;
  for (int i = v0; i < v1; ++i) { r->vuv[i][0] = uvs[2 * (i - v0)]; r->vuv[i][1] = uvs[2 * (i - v0) + 1]; }
  r->shapes[render_shape].tex = r->ntex;
  return r->ntex++;
}

ORC_EXPORT int orc_render_bind_env_box(orc_ctx* c, int render_shape, int shape) {
  r_model* r = (r_model*)c->render;
  if (!r || render_shape < 0 || render_shape >= r->ns) return rfail(c, MSK_ERR_INVALID, "bad render shape");
  if (r->finalized) return rfail(c, MSK_ERR_INVALID, "render_bind_env_box after render_finalize");
  if (shape < 0 || shape >= c->ns || c->xs_slot[shape] < 0) return rfail(c, MSK_ERR_INVALID, "render_bind_env_box: shape was not declared");
  r->shapes[render_shape].xs = c->xs_slot[shape];
  return MSK_OK;
}

ORC_EXPORT int orc_render_set_lights(orc_ctx* c, const float ambient[3], int ndir, const float* directions, const float* colors) {
  r_model* r = (r_model*)c->render;
  if (!r) return rfail(c, MSK_ERR_INVALID, "no render shapes");
  if (r->finalized) return rfail(c, MSK_ERR_INVALID, "render_set_lights after render_finalize");
  if (ndir < 0 || ndir > ORC_MAX_LIGHTS) return rfail(c, MSK_ERR_CAPACITY, "too many directional lights");
  for (int k = 0; k < 3; ++k) r->ambient[k] = ambient[k];
  r->nlights = ndir;
  for (int l = 0; l < ndir; ++l) {
    const float x = directions[3 * l], y = directions[3 * l + 1], z = directions[3 * l + 2];
    const float len = sqrtf(x * x + y * y + z * z);
    if (!(len > 0.0f)) return rfail(c, MSK_ERR_INVALID, "zero light direction");
    r->ldir[l][0] = x / len; r->ldir[l][1] = y / len; r->ldir[l][2] = z / len;
    for (int k = 0; k < 3; ++k) r->lcol[l][k] = colors[3 * l + k];
  }
  return MSK_OK;
}

ORC_EXPORT int orc_render_set_local_lights(orc_ctx* c, int n, const float* lights) {
  r_model* r = (r_model*)c->render;
  if (!r) return rfail(c, MSK_ERR_INVALID, "no render shapes");
  if (r->finalized) return rfail(c, MSK_ERR_INVALID, "render_set_local_lights after render_finalize");
  if (n < 0 || n > MSK_MAX_LOCAL_LIGHTS) return rfail(c, MSK_ERR_CAPACITY, "too many point / spot lights");
  for (int l = 0; l < n; ++l) {
    const float* p = lights + l * MSK_LOCAL_LIGHT_FLOATS;
    const float len = sqrtf(p[3] * p[3] + p[4] * p[4] + p[5] * p[5]);
    const int spot = p[9] > 0.0f;
    if (spot && !(len > 0.0f)) return rfail(c, MSK_ERR_INVALID, "zero spot-light axis");
    if (spot && !(p[9] <= p[10] && p[10] < 3.1415927f * 2.0f)) return rfail(c, MSK_ERR_INVALID, "spot light: 0 < inner_fov <= outer_fov < 2 pi");
    for (int k = 0; k < 3; ++k) { r->ppos[l][k] = p[k]; r->pdir[l][k] = spot ? p[3 + k] / len : 0.0f; r->pcol[l][k] = p[6 + k]; }
    /* cosines of the half angles; a point light passes every direction (cos >= -2) */
    r->pcone[l][0] = spot ? (float)cos(0.5 * (double)p[9]) : -2.0f;
    r->pcone[l][1] = spot ? (float)cos(0.5 * (double)p[10]) : -3.0f;
  }
  r->nlocal = n;
  return MSK_OK;
}

/* flat shading of one triangle (camera-frame corners, counter-clockwise seen from outside): per channel
 * base * min(1, ambient + sum_l light_l * max(0, n . -dir_l) + sum_p light_p * cone_p * max(0, n . l_p) / |x_p - centre|^2),
 * rounded to 8 bits; alpha = 255.  Point and spot lights are evaluated at the triangle's centroid (l_p: unit vector towards the
 * light); cone_p = clamp((cos(angle to the axis) - cos(outer / 2)) / (cos(inner / 2) - cos(outer / 2)), 0, 1), 1 for a point light. */
static uint32_t shade_triangle(v3 p0, v3 p1, v3 p2, const float* base, const float* ambient, int nl, const float* ldir_cam, const float* lcol,
                               int np, const float* ppos_cam, const float* pdir_cam, const float* pcol, const float* pcone) {
  v3 n = v3_cross(v3_sub(p1, p0), v3_sub(p2, p0));
  const float l = v3_len(n);
  n = (l > 0.0f) ? v3_scale(n, 1.0f / l) : v3_make(0, 0, 0);
  float lit[3] = {ambient[0], ambient[1], ambient[2]};
  for (int k = 0; k < nl; ++k) {
    const float d = fmaxf(0.0f, -(n.x * ldir_cam[k * 3] + n.y * ldir_cam[k * 3 + 1] + n.z * ldir_cam[k * 3 + 2]));
    lit[0] = fmaf(lcol[k * 3], d, lit[0]); lit[1] = fmaf(lcol[k * 3 + 1], d, lit[1]); lit[2] = fmaf(lcol[k * 3 + 2], d, lit[2]);
  }
  if (np > 0) {
    const float third = 1.0f / 3.0f;
    const v3 cen = v3_make((p0.x + p1.x + p2.x) * third, (p0.y + p1.y + p2.y) * third, (p0.z + p1.z + p2.z) * third);
    for (int k = 0; k < np; ++k) {
      const v3 L = v3_make(ppos_cam[k * 3] - cen.x, ppos_cam[k * 3 + 1] - cen.y, ppos_cam[k * 3 + 2] - cen.z);
      const float d2 = fmaf(L.x, L.x, fmaf(L.y, L.y, L.z * L.z));
      if (!(d2 > 1e-12f)) continue;
      const float inv = 1.0f / sqrtf(d2);
      float a = fmaxf(0.0f, fmaf(n.x, L.x, fmaf(n.y, L.y, n.z * L.z)) * inv) / d2;
      if (pcone[k * 2] > -1.5f) { /* spot: the light looks along its axis */
        const float cs = -fmaf(pdir_cam[k * 3], L.x, fmaf(pdir_cam[k * 3 + 1], L.y, pdir_cam[k * 3 + 2] * L.z)) * inv;
        const float span = pcone[k * 2] - pcone[k * 2 + 1];
        const float f = span > 1e-6f ? fminf(fmaxf((cs - pcone[k * 2 + 1]) / span, 0.0f), 1.0f) : (cs >= pcone[k * 2] ? 1.0f : 0.0f);
        a = a * f;
      }
      lit[0] = fmaf(pcol[k * 3], a, lit[0]); lit[1] = fmaf(pcol[k * 3 + 1], a, lit[1]); lit[2] = fmaf(pcol[k * 3 + 2], a, lit[2]);
    }
  }
  uint32_t out = 0xFF000000u;
  for (int ch = 0; ch < 3; ++ch) {
    const float v = fminf(fmaxf(base[ch] * fminf(lit[ch], 1.0f), 0.0f), 1.0f);
    out |= ((uint32_t)rintf(v * 255.0f) & 0xFFu) << (8 * ch);
  }
  return out;
}

ORC_EXPORT int orc_render_finalize(orc_ctx* c) {
  if (!c->render) return rfail(c, MSK_ERR_INVALID, "no render shapes");
  ((r_model*)c->render)->finalized = 1;
  return MSK_OK;
}

ORC_EXPORT int orc_camera_create(orc_ctx* c, int width, int height, float fovy, float near_plane, float far_plane, int mount_body,
                                 const float local_pose[7]) {
  r_model* r = (r_model*)c->render;
  if (!r || !r->finalized) return rfail(c, MSK_ERR_INVALID, "camera_create before render_finalize");
  if (r->ncams >= MSK_MAX_CAMERAS) return rfail(c, MSK_ERR_CAPACITY, "too many cameras");
  if (width % 16 || height % 16 || width <= 0 || height <= 0) return rfail(c, MSK_ERR_INVALID, "camera size must be a multiple of 16");
  r_camera* cam = &r->cams[r->ncams];
  cam->W = width; cam->H = height; cam->mount = mount_body;
  cam->fy = (float)(0.5 * height / tan(0.5 * (double)fovy));
  cam->fx = cam->fy;
  cam->cx = 0.5f * width; cam->cy = 0.5f * height;
  cam->near_ = near_plane; cam->far_ = far_plane;
  cam->local = r_pose7(local_pose);
  cam->out = (int16_t*)calloc((size_t)c->num_envs * width * height * 4, sizeof(int16_t));
  cam->color = (uint32_t*)calloc((size_t)c->num_envs * width * height, sizeof(uint32_t));
  cam->depth = (int16_t*)calloc((size_t)c->num_envs * width * height, sizeof(int16_t));
  cam->seg = (int16_t*)calloc((size_t)c->num_envs * width * height, sizeof(int16_t));
  return r->ncams++;
}

ORC_EXPORT void* orc_camera_buffer(orc_ctx* c, int camera, int64_t shape[4]) {
  r_model* r = (r_model*)c->render;
  if (!r || camera < 0 || camera >= r->ncams) return NULL;
  shape[0] = c->num_envs; shape[1] = r->cams[camera].H; shape[2] = r->cams[camera].W; shape[3] = 4;
  return r->cams[camera].out;
}

/* Camera.get_obs planes (sensors/camera.py:190-242 with render/shaders.py:141-145) */
ORC_EXPORT void* orc_camera_obs_buffer(orc_ctx* c, int camera, int which, int64_t shape[4]) {
  r_model* r = (r_model*)c->render;
  if (!r || camera < 0 || camera >= r->ncams || which < MSK_CAM_DEPTH || which > MSK_CAM_COLOR) return NULL;
  shape[0] = c->num_envs; shape[1] = r->cams[camera].H; shape[2] = r->cams[camera].W; shape[3] = which == MSK_CAM_COLOR ? 4 : 1;
  if (which == MSK_CAM_COLOR) return r->cams[camera].color;
  return which == MSK_CAM_DEPTH ? (void*)r->cams[camera].depth : (void*)r->cams[camera].seg;
}

/* msk_camera_set_outputs (include/msk_render.h): with position_texture = 0 the texture's contents are undefined -- this restatement keeps using it as its
 * working picture, so it simply stays defined here; the planes are the same either way */
ORC_EXPORT int orc_camera_set_outputs(orc_ctx* c, int camera, int position_texture) {
  r_model* r = (r_model*)c->render;
  if (!r || camera < 0 || camera >= r->ncams) return rfail(c, MSK_ERR_INVALID, "bad camera");
  r->cams[camera].no_color = (position_texture & MSK_CAM_OUT_NO_COLOR) != 0;   /* (the texture itself is always filled here: its contents are undefined there) */
  return MSK_OK;
}

static void project_point(const r_camera* cam, v3 p, float* u, float* v, float* w) {
  const float iw = 1.0f / p.x;
  *u = fmaf(cam->fx, -p.y * iw, cam->cx);
  *v = fmaf(-cam->fy, p.z * iw, cam->cy);
  *w = iw;
}

/* uv: texture coordinates of the three corners (u0 v0 u1 v1 u2 v2) or NULL */
static int setup_triangle(const r_camera* cam, v3 p0, v3 p1, v3 p2, int seg, int prim, r_setup* t, const float* uv) {
  float u0, v0, w0, u1, v1, w1, u2, v2, w2;
  project_point(cam, p0, &u0, &v0, &w0);
  project_point(cam, p1, &u1, &v1, &w1);
  project_point(cam, p2, &u2, &v2, &w2);
  /* rows grow downwards: counter-clockwise (seen from outside) = negative area; others are back faces */
  const float area = fmaf(u1 - u0, v2 - v0, -((v1 - v0) * (u2 - u0)));
  if (!(area < -1e-12f)) return 0;
  float t_; t_ = u1; u1 = u2; u2 = t_; t_ = v1; v1 = v2; v2 = t_; t_ = w1; w1 = w2; w2 = t_;
  const float a2 = -area;
  const float umin = fminf(u0, fminf(u1, u2)), umax = fmaxf(u0, fmaxf(u1, u2));
  const float vmin = fminf(v0, fminf(v1, v2)), vmax = fmaxf(v0, fmaxf(v1, v2));
  if (!(umin < 1e9f && umax > -1e9f && vmin < 1e9f && vmax > -1e9f)) return 0;
  int x0 = (int)ceilf(umin - 0.5f), x1 = (int)floorf(umax - 0.5f);
  int y0 = (int)ceilf(vmin - 0.5f), y1 = (int)floorf(vmax - 0.5f);
  if (x0 < 0) x0 = 0;
  if (y0 < 0) y0 = 0;
  if (x1 > cam->W - 1) x1 = cam->W - 1;
  if (y1 > cam->H - 1) y1 = cam->H - 1;
  if (x0 > x1 || y0 > y1) return 0;
  t->A0 = -(v1 - v0); t->B0 = u1 - u0; t->C0 = -fmaf(t->A0, u0, t->B0 * v0);
  t->A1 = -(v2 - v1); t->B1 = u2 - u1; t->C1 = -fmaf(t->A1, u1, t->B1 * v1);
  t->A2 = -(v0 - v2); t->B2 = u0 - u2; t->C2 = -fmaf(t->A2, u2, t->B2 * v2);
  const float ia = 1.0f / a2;
  t->Aw = fmaf(t->A1, w0, fmaf(t->A2, w1, t->A0 * w2)) * ia;
  t->Bw = fmaf(t->B1, w0, fmaf(t->B2, w1, t->B0 * w2)) * ia;
  t->Cw = fmaf(t->C1, w0, fmaf(t->C2, w1, t->C0 * w2)) * ia;
  t->seg = seg; t->prim = prim;
  t->x0 = x0; t->x1 = x1; t->y0 = y0; t->y1 = y1;
  t->tex = -1;
  if (uv) { /* (corners 1 and 2 were exchanged above) */
    const float a0 = uv[0] * w0, a1 = uv[4] * w1, a2 = uv[2] * w2, b0 = uv[1] * w0, b1 = uv[5] * w1, b2 = uv[3] * w2;
    t->Au = fmaf(t->A1, a0, fmaf(t->A2, a1, t->A0 * a2)) * ia;
    t->Bu = fmaf(t->B1, a0, fmaf(t->B2, a1, t->B0 * a2)) * ia;
    t->Cu = fmaf(t->C1, a0, fmaf(t->C2, a1, t->C0 * a2)) * ia;
    t->Av = fmaf(t->A1, b0, fmaf(t->A2, b1, t->A0 * b2)) * ia;
    t->Bv = fmaf(t->B1, b0, fmaf(t->B2, b1, t->B0 * b2)) * ia;
    t->Cv = fmaf(t->C1, b0, fmaf(t->C2, b1, t->C0 * b2)) * ia;
  }
  return 1;
}

static v3 lerp_near(v3 a, v3 b, float near_) {
  const float s = (near_ - a.x) / (b.x - a.x);
  return v3_make(near_, fmaf(s, b.y - a.y, a.y), fmaf(s, b.z - a.z, a.z));
}
/* the same point's texture coordinates */
static void lerp_near_uv(v3 a, v3 b, float near_, const float* ua, const float* ub, float* out) {
  const float s = (near_ - a.x) / (b.x - a.x);
  out[0] = fmaf(s, ub[0] - ua[0], ua[0]); out[1] = fmaf(s, ub[1] - ua[1], ua[1]);
}
/* the texel under (u, v) whose footprint of one pixel is `rho` texels of level 0: mip level floor(log2 rho) (level 0 below two texels per
 * pixel), nearest texel of that level, repeating; (0, 0) is the top-left corner of texel (0, 0).  -> index into the texture's mip chain */
static int texel_index(int w, int h, float u, float v, float rho) {
  int level = 0;
  if (rho >= 2.0f && rho < 3.0e38f) { union { float f; uint32_t i; } b; b.f = rho; level = (int)((b.i >> 23) & 0xFFu) - 127; }
  else if (!(rho < 2.0f)) level = 30;
  int ofs = 0;
  for (int l = 0; l < level && !(w == 1 && h == 1); ++l) { ofs += w * h; w = w > 1 ? w / 2 : 1; h = h > 1 ? h / 2 : 1; }
  const float fu = u * (float)w, fv = v * (float)h;
  int iu = (fabsf(fu) < 1.0e9f) ? (int)floorf(fu) : 0, iv = (fabsf(fv) < 1.0e9f) ? (int)floorf(fv) : 0;
  iu %= w; if (iu < 0) iu += w;
  iv %= h; if (iv < 0) iv += h;
  return ofs + iv * w + iu;
}
static uint32_t modulate(uint32_t texel, uint32_t shade) { /* per channel texel * shade / 255, rounded; alpha = the shade's */
  uint32_t out = shade & 0xFF000000u;
  for (int ch = 0; ch < 3; ++ch) {
    const uint32_t a = (texel >> (8 * ch)) & 0xFFu, b = (shade >> (8 * ch)) & 0xFFu;
    out |= ((a * b + 127u) / 255u) << (8 * ch);
  }
  return out;
}

static int16_t to_mm(float x) {
  float r = rintf(x * 1000.0f);
  r = fminf(fmaxf(r, -32768.0f), 32767.0f);
  return (int16_t)r;
}

ORC_EXPORT int orc_camera_take_picture(orc_ctx* c, int camera, void* stream) {
  (void)stream;
  r_model* r = (r_model*)c->render;
  if (!r || camera < 0 || camera >= r->ncams) return rfail(c, MSK_ERR_INVALID, "bad camera");
  const r_camera* cam = &r->cams[camera];
  const float wmin = 1.0f / cam->far_;
  v3* cv = (v3*)malloc(sizeof(v3) * (size_t)(r->nv > 0 ? r->nv : 1));
  r_setup* st = (r_setup*)malloc(sizeof(r_setup) * (size_t)(2 * r->nt + 1));
  float* bw = (float*)malloc(sizeof(float) * (size_t)cam->W * cam->H);
  for (int e = 0; e < c->num_envs; ++e) {
    orc_env* env = &c->envs[e];
    orc_forward_kinematics(c, env);
    pose Tc = cam->local;
    if (cam->mount >= 0) Tc = pose_mul(env->bpose[cam->mount], cam->local);
    const pose Tci = pose_inv(Tc);
    pose shapeT[MSK_MAX_RENDER_SHAPES];
    v3 shapeS[MSK_MAX_RENDER_SHAPES];
    for (int s = 0; s < r->ns; ++s) {
      pose L = r->shapes[s].local;
      shapeS[s] = v3_make(1.0f, 1.0f, 1.0f);
      if (r->shapes[s].xs >= 0) { /* follows a per-env box instance */
        const float* x = c->xshape + ((size_t)e * c->nxs + r->shapes[s].xs) * 8;
        shapeS[s] = v3_make(x[0], x[1], x[2]);
        L.p = v3_make(x[4], x[5], x[6]);
      }
      pose T = L;
      if (r->shapes[s].body >= 0) T = pose_mul(env->bpose[r->shapes[s].body], L);
      shapeT[s] = pose_mul(Tci, T);
    }
    for (int i = 0; i < r->nv; ++i) {
      const v3 sc = shapeS[r->vshape[i]], vl = r->verts[i];
      cv[i] = pose_apply(shapeT[r->vshape[i]], v3_make(vl.x * sc.x, vl.y * sc.y, vl.z * sc.z));
    }
    float light_cam[ORC_MAX_LIGHTS * 3];
    for (int l = 0; l < r->nlights; ++l) {
      const v3 d = quat_rotate(Tci.q, v3_make(r->ldir[l][0], r->ldir[l][1], r->ldir[l][2]));
      light_cam[l * 3] = d.x; light_cam[l * 3 + 1] = d.y; light_cam[l * 3 + 2] = d.z;
    }
    float ppos_cam[MSK_MAX_LOCAL_LIGHTS * 3 + 1], pdir_cam[MSK_MAX_LOCAL_LIGHTS * 3 + 1];
    for (int l = 0; l < r->nlocal; ++l) {
      const v3 x = pose_apply(Tci, v3_make(r->ppos[l][0], r->ppos[l][1], r->ppos[l][2]));
      const v3 d = quat_rotate(Tci.q, v3_make(r->pdir[l][0], r->pdir[l][1], r->pdir[l][2]));
      ppos_cam[l * 3] = x.x; ppos_cam[l * 3 + 1] = x.y; ppos_cam[l * 3 + 2] = x.z;
      pdir_cam[l * 3] = d.x; pdir_cam[l * 3 + 1] = d.y; pdir_cam[l * 3 + 2] = d.z;
    }
    int ns = 0;
    for (int ti = 0; ti < r->nt; ++ti) {
      const r_tri* tr = &r->tris[ti];
      const v3 p[3] = {cv[tr->v0], cv[tr->v1], cv[tr->v2]};
      const int seg = r->shapes[tr->shape].seg;
      const uint32_t col = shade_triangle(p[0], p[1], p[2], r->shapes[tr->shape].color, r->ambient, r->nlights, light_cam, &r->lcol[0][0],
                                          r->nlocal, ppos_cam, pdir_cam, &r->pcol[0][0], &r->pcone[0][0]);
      const int in0 = p[0].x >= cam->near_, in1 = p[1].x >= cam->near_, in2 = p[2].x >= cam->near_;
      const int nin = in0 + in1 + in2;
      const int tex = r->shapes[tr->shape].tex;
      const int vid[3] = {tr->v0, tr->v1, tr->v2};
      v3 q[4];
      float quv[4][2];
      int nq = 0;
      if (nin == 3) {
        q[0] = p[0]; q[1] = p[1]; q[2] = p[2]; nq = 3;
        for (int k = 0; k < 3; ++k) { quv[k][0] = r->vuv[vid[k]][0]; quv[k][1] = r->vuv[vid[k]][1]; }
      } else if (nin > 0) {
        for (int k = 0; k < 3; ++k) {
          const v3 a = p[k], b = p[(k + 1) % 3];
          const float* ua = r->vuv[vid[k]]; const float* ub = r->vuv[vid[(k + 1) % 3]];
          const int ia = a.x >= cam->near_, ib = b.x >= cam->near_;
          if (ia) { quv[nq][0] = ua[0]; quv[nq][1] = ua[1]; q[nq++] = a; }
          if (ia != ib) {
            if (ia) { lerp_near_uv(a, b, cam->near_, ua, ub, quv[nq]); q[nq++] = lerp_near(a, b, cam->near_); }
            else { lerp_near_uv(b, a, cam->near_, ub, ua, quv[nq]); q[nq++] = lerp_near(b, a, cam->near_); }
          }
        }
      }
      for (int sub = 0; sub + 2 < nq; ++sub) {
        const float uv6[6] = {quv[0][0], quv[0][1], quv[sub + 1][0], quv[sub + 1][1], quv[sub + 2][0], quv[sub + 2][1]};
        if (setup_triangle(cam, q[0], q[sub + 1], q[sub + 2], seg, ti * 2 + sub, &st[ns], tex >= 0 ? uv6 : NULL)) { st[ns].color = col; st[ns].tex = tex; ns++; }
      }
    }
    if (e == g_dbg_env && g_dbg_boxes) {
      g_dbg_n = ns < g_dbg_max ? ns : g_dbg_max;
      for (int k = 0; k < g_dbg_n; ++k) { g_dbg_boxes[4 * k] = st[k].x0; g_dbg_boxes[4 * k + 1] = st[k].x1; g_dbg_boxes[4 * k + 2] = st[k].y0; g_dbg_boxes[4 * k + 3] = st[k].y1; }
    }
    int16_t* img = cam->out + (size_t)e * cam->W * cam->H * 4;
    uint32_t* cimg = cam->color + (size_t)e * cam->W * cam->H;
    memset(img, 0, sizeof(int16_t) * (size_t)cam->W * cam->H * 4);
    if (!cam->no_color) memset(cimg, 0, sizeof(uint32_t) * (size_t)cam->W * cam->H);
    for (int i = 0; i < cam->W * cam->H; ++i) bw[i] = 0.0f;
    /* triangles in primitive order; strictly nearer wins, so equal depths keep the lower primitive id */
    for (int k = 0; k < ns; ++k) {
      const r_setup* t = &st[k];
      for (int py = t->y0; py <= t->y1; ++py)
        for (int px = t->x0; px <= t->x1; ++px) {
          const float x = (float)px + 0.5f, y = (float)py + 0.5f;
          const float e0 = fmaf(t->A0, x, fmaf(t->B0, y, t->C0));
          const float e1 = fmaf(t->A1, x, fmaf(t->B1, y, t->C1));
          const float e2 = fmaf(t->A2, x, fmaf(t->B2, y, t->C2));
          if (!(e0 >= 0.0f && e1 >= 0.0f && e2 >= 0.0f)) continue;
          const float w = fmaf(t->Aw, x, fmaf(t->Bw, y, t->Cw));
          if (!(w >= wmin) || !(w > bw[py * cam->W + px])) continue;
          bw[py * cam->W + px] = w;
          const float d = 1.0f / w;
          int16_t* o = img + ((size_t)py * cam->W + px) * 4;
          o[0] = to_mm((x - cam->cx) / cam->fx * d);
          o[1] = to_mm(-(y - cam->cy) / cam->fy * d);
          o[2] = to_mm(-d);
          o[3] = (int16_t)t->seg;
          uint32_t col = t->color;
          if (t->tex >= 0) { /* texel under the pixel centre: u = (u / depth) * depth */
            const float uu = fmaf(t->Au, x, fmaf(t->Bu, y, t->Cu)) * d, vv = fmaf(t->Av, x, fmaf(t->Bv, y, t->Cv)) * d;
            const r_texture* tx = &r->tex[t->tex];
            /* one pixel step moves (u, v) by d/dx (U / w) = (Au - u Aw) / w and so on: the footprint in level-0 texels picks the mip level */
            const float ux = fmaf(-uu, t->Aw, t->Au) * d, uy = fmaf(-uu, t->Bw, t->Bu) * d;
            const float vx = fmaf(-vv, t->Aw, t->Av) * d, vy = fmaf(-vv, t->Bw, t->Bv) * d;
            const float rho = fmaxf(fmaxf(fabsf(ux), fabsf(uy)) * (float)tx->w, fmaxf(fabsf(vx), fabsf(vy)) * (float)tx->h);
            col = modulate(tx->texels[texel_index(tx->w, tx->h, uu, vv, rho)], col);
          }
          if (!cam->no_color) cimg[py * cam->W + px] = col;
        }
    }
    for (int i = 0; i < cam->W * cam->H; ++i) {
      cam->depth[(size_t)e * cam->W * cam->H + i] = (int16_t)(-(int)img[4 * i + 2]);
      cam->seg[(size_t)e * cam->W * cam->H + i] = img[4 * i + 3];
    }
  }
  free(cv); free(st); free(bw);
  return MSK_OK;
}
