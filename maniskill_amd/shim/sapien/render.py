"""``sapien.render``: the camera half of the shim.  Render bodies / shapes / cameras / lights are host-side records; the batched
path (``RenderSystemGroup`` + ``create_camera_group`` + ``take_picture`` + ``get_picture_cuda``) compiles sub-scene 0's render
bodies into the C-ABI rasteriser (include/msk_render.h; hand-written HIP tile rasteriser) and serves the ``minimal`` shader
pack's textures ``Color`` (uint8 x 4) and ``PositionSegmentation`` (int16 x 4) as zero-copy torch views.

Reference call sites: mani_skill/envs/scene.py:198-297 (RenderCameraComponent), :382-427 (update_render), :1026-1110
(RenderSystemGroup, set_cuda_poses, create_camera_group); utils/structs/render_camera.py:160-182,269-273; render/shaders.py:68-84.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import numpy as np

from ._core import Component
from ._pose import Pose
from . import _mesh

_log_level = "warn"
_globals = dict(camera_shader_dir="minimal", viewer_shader_dir="minimal", picture_format={}, rt=dict())


def set_log_level(level):
    global _log_level
    _log_level = level


def set_camera_shader_dir(d):
    _globals["camera_shader_dir"] = d


def get_camera_shader_dir():
    return _globals["camera_shader_dir"]


def set_viewer_shader_dir(d):
    _globals["viewer_shader_dir"] = d


def get_viewer_shader_dir():
    return _globals["viewer_shader_dir"]


def set_picture_format(name, fmt):
    _globals["picture_format"][name] = fmt


def set_ray_tracing_samples_per_pixel(v):
    _globals["rt"]["spp"] = v


def set_ray_tracing_path_depth(v):
    _globals["rt"]["path_depth"] = v


def set_ray_tracing_denoiser(v):
    _globals["rt"]["denoiser"] = v


def set_ray_tracing_dof_num_blades(v): _globals["rt"]["dof_blades"] = v
def set_ray_tracing_dof_rotation(v): _globals["rt"]["dof_rot"] = v
def set_ray_tracing_dof_ratio(v): _globals["rt"]["dof_ratio"] = v


def get_device_summary():
    return "MI355X HIP rasteriser (maniskill_amd)"


def enable_vr():
    raise RuntimeError("VR is not available")


# ----------------------------------------------------------------------------------------------------------- materials
class RenderTexture2D:
    def __init__(self, filename=None, mipmap_levels=1, filter_mode="linear", address_mode="repeat", srgb=True, **kw):
        self.filename = None if filename is None else str(filename)
        self.mipmap_levels, self.filter_mode, self.address_mode, self.srgb = mipmap_levels, filter_mode, address_mode, srgb

    _cache: dict = {}

    def _pixels(self, max_side=256):
        """uint8 (h, w, 4) of the image, reduced to at most max_side texels per side (the rasteriser's textures share 2^20 texels per
        context, mip levels included; pictures are 128 to 512 pixels wide), or None if the file cannot be read."""
        if not self.filename or not os.path.exists(self.filename):
            return None
        key = (self.filename, max_side)
        if key not in RenderTexture2D._cache:
            try:
                from PIL import Image
                im = Image.open(self.filename).convert("RGBA")
                if max(im.size) > max_side:
                    k = max_side / float(max(im.size))
                    im = im.resize((max(1, int(round(im.size[0] * k))), max(1, int(round(im.size[1] * k)))), Image.BOX)
                RenderTexture2D._cache[key] = np.ascontiguousarray(np.asarray(im, dtype=np.uint8))
            except Exception:
                RenderTexture2D._cache[key] = None
        return RenderTexture2D._cache[key]

    def _mean_color(self):
        """Flat shading has no texture lookup: a textured material is drawn in the texture's mean colour."""
        if not self.filename or not os.path.exists(self.filename):
            return None
        try:
            from PIL import Image
            im = np.asarray(Image.open(self.filename).convert("RGB"), dtype=np.float64) / 255.0
            return (im.reshape(-1, 3).mean(0) ** (2.2 if self.srgb else 1.0)).tolist()
        except Exception:
            return None


class RenderCubemap:
    def __init__(self, *a, **k):
        pass


class RenderMaterial:
    def __init__(self, emission=(0, 0, 0, 1), base_color=(1, 1, 1, 1), specular=0.0, roughness=1.0, metallic=0.0, transmission=0.0,
                 ior=1.45, transmission_roughness=0.0):
        self.emission = list(emission)
        self.base_color = [float(x) for x in base_color]
        self.specular, self.roughness, self.metallic = specular, roughness, metallic
        self.transmission, self.ior, self.transmission_roughness = transmission, ior, transmission_roughness
        self.base_color_texture = None
        self.diffuse_texture = None
        self.normal_texture = self.roughness_texture = self.metallic_texture = self.emission_texture = self.transmission_texture = None

    def set_base_color(self, c):
        self.base_color = [float(x) for x in c]

    def get_base_color(self):
        return self.base_color

    def set_emission(self, c): self.emission = list(c)
    def set_specular(self, v): self.specular = v
    def set_roughness(self, v): self.roughness = v
    def set_metallic(self, v): self.metallic = v
    def set_transmission(self, v): self.transmission = v
    def set_ior(self, v): self.ior = v
    def set_base_color_texture(self, t): self.base_color_texture = t
    def set_diffuse_texture(self, t): self.diffuse_texture = t
    def set_normal_texture(self, t): self.normal_texture = t
    def set_roughness_texture(self, t): self.roughness_texture = t
    def set_metallic_texture(self, t): self.metallic_texture = t
    def set_emission_texture(self, t): self.emission_texture = t
    def set_transmission_texture(self, t): self.transmission_texture = t

    def _flat_color(self):
        tex = self.base_color_texture or self.diffuse_texture
        if tex is not None:
            c = tex._mean_color()
            if c is not None:
                return [c[0] * self.base_color[0], c[1] * self.base_color[1], c[2] * self.base_color[2], self.base_color[3]]
        return list(self.base_color)


# ----------------------------------------------------------------------------------------------------------- shapes
class RenderShape:
    def __init__(self, material: Optional[RenderMaterial] = None):
        self.material = material if material is not None else RenderMaterial(base_color=(0.8, 0.8, 0.8, 1))
        self.local_pose = Pose()
        self.name = ""
        self.scale = np.ones(3, dtype=np.float32)
        self.gpu_pose_batch_index = -1
        self.shade_flat = False
        self.per_scene_id = 0

    def set_gpu_pose_batch_index(self, i):
        self.gpu_pose_batch_index = int(i)

    def get_gpu_pose_batch_index(self):
        return self.gpu_pose_batch_index

    def get_local_pose(self):
        return self.local_pose

    def set_local_pose(self, p):
        self.local_pose = p

    def get_material(self):
        return self.material

    def set_material(self, m):
        self.material = m

    @property
    def parts(self):
        return [self]

    def get_parts(self):
        return self.parts

    def _triangles(self):
        """-> list of (vertices [n,3] in the shape frame, faces [m,3] ccw from outside, rgba)."""
        raise NotImplementedError


class RenderShapeBox(RenderShape):
    def __init__(self, half_size, material=None):
        super().__init__(material)
        self.half_size = np.array(half_size, dtype=np.float32).reshape(3)

    def _triangles(self):
        v = np.array([[x, y, z] for x in (-1, 1) for y in (-1, 1) for z in (-1, 1)], dtype=np.float64) * self.half_size
        return [(v, _mesh.hull_faces(v), self.material._flat_color())]


class RenderShapeSphere(RenderShape):
    def __init__(self, radius, material=None):
        super().__init__(material)
        self.radius = float(radius)

    def _triangles(self):
        v = _icosphere(1) * self.radius
        return [(v, _mesh.hull_faces(v), self.material._flat_color())]


class RenderShapeCapsule(RenderShape):
    def __init__(self, radius, half_length, material=None):
        super().__init__(material)
        self.radius, self.half_length = float(radius), float(half_length)

    def _triangles(self):
        s = _icosphere(1) * self.radius
        v = np.concatenate([s[s[:, 0] <= 1e-9] - [self.half_length, 0, 0], s[s[:, 0] >= -1e-9] + [self.half_length, 0, 0]])
        return [(v, _mesh.hull_faces(v), self.material._flat_color())]


class RenderShapeCylinder(RenderShape):
    def __init__(self, radius, half_length, material=None):
        super().__init__(material)
        self.radius, self.half_length = float(radius), float(half_length)

    def _triangles(self):
        v = _mesh.prism(self.radius, self.half_length, sides=24).astype(np.float64)
        return [(v, _mesh.hull_faces(v), self.material._flat_color())]


class RenderShapePlane(RenderShape):
    def __init__(self, scale, material=None):
        super().__init__(material)
        self.scale = np.array(scale, dtype=np.float32).reshape(3)

    def _triangles(self):
        # plane through the origin with normal +x (SAPIEN convention), half extents scale[1], scale[2]
        L1, L2 = float(self.scale[1]), float(self.scale[2])
        v = np.array([[0, -L1, -L2], [0, L1, -L2], [0, L1, L2], [0, -L1, L2]], dtype=np.float64)
        return [(v, np.array([[0, 1, 2], [0, 2, 3]]), self.material._flat_color())]


class RenderShapeTriangleMeshPart:
    def __init__(self, vertices, triangles, material, uvs=None):
        self.vertices = np.asarray(vertices, dtype=np.float32)
        self.triangles = np.asarray(triangles, dtype=np.uint32)
        self.material = material
        self.uvs = None if uvs is None else np.asarray(uvs, dtype=np.float32).reshape(-1, 2)
        if self.uvs is not None and len(self.uvs) != len(self.vertices):
            self.uvs = None

    def get_uvs(self): return self.uvs

    def get_vertices(self): return self.vertices
    def get_triangles(self): return self.triangles
    def get_material(self): return self.material


class RenderShapeTriangleMesh(RenderShape):
    """RenderShapeTriangleMesh(filename, scale, material) or RenderShapeTriangleMesh(vertices, triangles, normals, uvs, material)."""

    def __init__(self, filename=None, scale=(1, 1, 1), material=None, vertices=None, triangles=None, normals=None, uvs=None):
        super().__init__(material)
        self.scale = np.array(scale, dtype=np.float32).reshape(-1)
        if self.scale.size == 1:
            self.scale = np.full(3, float(self.scale[0]), dtype=np.float32)
        self.filename = None
        self.double_sided = False
        if vertices is not None or (filename is not None and not isinstance(filename, (str, os.PathLike))):
            if vertices is None:       # positional (vertices, triangles, normals, uvs, material)
                vertices, triangles = filename, scale
                self.scale = np.ones(3, dtype=np.float32)
            self._parts = [RenderShapeTriangleMeshPart(vertices, triangles, self.material, uvs)]
            self.double_sided = True   # procedurally generated sheets (ground grid) are seen from both sides
        else:
            self.filename = str(filename)
            self._parts = []
            if not os.path.exists(self.filename):     # a visual that cannot be loaded is skipped, as a renderer does: the body stays
                import warnings
                warnings.warn(f"visual mesh not found, skipped: {self.filename}")
                return
            for p in _mesh.load_mesh_parts(self.filename):
                mat = material if material is not None else RenderMaterial(base_color=p["base_color"])
                self._parts.append(RenderShapeTriangleMeshPart(p["vertices"], p["faces"], mat))

    @property
    def parts(self):
        return self._parts

    def _triangles(self):
        return [(v, f, rgba) for v, f, rgba, _, _ in self._textured_parts(want_textures=False)]

    def _textured_parts(self, want_textures=True):
        """-> list of (vertices, faces, rgba, uvs or None, texture pixels or None): a part whose material has a base-colour texture that can
        be read, on a mesh that came with uvs, is drawn with it (then rgba is the material's own base colour, which multiplies the texels);
        any other textured part in the texture's mean colour."""
        out = []
        for p in self._parts:
            v = p.vertices.astype(np.float64) * self.scale
            f = p.triangles.astype(np.int64)
            if self.double_sided:
                f = np.concatenate([f, f[:, ::-1]])
            tex = p.material.base_color_texture or p.material.diffuse_texture
            pix = tex._pixels() if (want_textures and tex is not None and getattr(p, "uvs", None) is not None) else None
            if pix is not None:
                out.append((v, f, list(p.material.base_color), p.uvs.astype(np.float64), pix))
            else:
                out.append((v, f, p.material._flat_color(), None, None))
        return out


def _simplify(v, f, max_tris=256):
    """Dense mesh -> something the template holds: a planar sheet becomes the fan of its 2-D convex outline (both windings, as
    sheets are drawn double sided, exact); anything else is simplified by vertex clustering to `max_tris` triangles (the surface
    moves by at most the returned bound, recorded in RenderSystemGroup.simplification)."""
    from scipy.spatial import ConvexHull
    v = np.asarray(v, dtype=np.float64)
    c = v.mean(0)
    u, sv, vt = np.linalg.svd(v - c, full_matrices=False)
    if sv[2] < 1e-9 * max(sv[0], 1e-30):
        uv = (v - c) @ vt[:2].T
        h = ConvexHull(uv)
        ring = h.vertices                      # counter-clockwise in the (vt[0], vt[1]) plane
        _simplify.last_ring = ring
        pts = v[ring]
        fan = np.array([[0, k, k + 1] for k in range(1, len(ring) - 1)], dtype=np.int64)
        return pts, np.concatenate([fan, fan[:, ::-1]])
    _simplify.last_ring = None
    nv, nf, bound = _mesh.cluster_simplify(v, f, max_tris)
    _simplify.last_bound = bound
    return nv, nf


def _icosphere(subdiv):
    t = (1.0 + 5 ** 0.5) / 2.0
    v = np.array([[-1, t, 0], [1, t, 0], [-1, -t, 0], [1, -t, 0], [0, -1, t], [0, 1, t], [0, -1, -t], [0, 1, -t],
                  [t, 0, -1], [t, 0, 1], [-t, 0, -1], [-t, 0, 1]], dtype=np.float64)
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    for _ in range(subdiv):
        from scipy.spatial import ConvexHull
        h = ConvexHull(v)
        mids = set()
        for a, b, c in h.simplices:
            for i, j in ((a, b), (b, c), (c, a)):
                mids.add((min(i, j), max(i, j)))
        m = np.array([(v[i] + v[j]) for i, j in sorted(mids)])
        v = np.concatenate([v, m / np.linalg.norm(m, axis=1, keepdims=True)])
    return v


# ----------------------------------------------------------------------------------------------------------- components
class RenderBodyComponent(Component):
    def __init__(self):
        super().__init__()
        self.render_shapes: list[RenderShape] = []
        self.visibility = 1.0
        self.shading_mode = 0
        self.is_render_id_disabled = False

    def attach(self, shape: RenderShape):
        self.render_shapes.append(shape)
        return self

    def get_render_shapes(self):
        return self.render_shapes

    def set_visibility(self, v):
        self.visibility = float(v)

    def get_visibility(self):
        return self.visibility

    def set_property(self, name, value):
        pass

    def disable_render_id(self):
        self.is_render_id_disabled = True

    def enable_render_id(self):
        self.is_render_id_disabled = False

    def _on_add_to_scene(self, scene):
        if scene.render_system is not None:
            scene.render_system.render_bodies.append(self)

    def _on_remove_from_scene(self, scene):
        if scene.render_system is not None and self in scene.render_system.render_bodies:
            scene.render_system.render_bodies.remove(self)


class _Light(Component):
    def __init__(self):
        super().__init__()
        self.color = [1.0, 1.0, 1.0]
        self.shadow = False
        self.shadow_near, self.shadow_far, self.shadow_map_size, self.shadow_half_size = 0.1, 10.0, 2048, 10.0
        self.local_pose = Pose()
        self._pose = Pose()

    @property
    def pose(self):
        return self._pose

    @pose.setter
    def pose(self, p):
        self._pose = p

    def _on_add_to_scene(self, scene):
        if scene.render_system is not None:
            scene.render_system.lights.append(self)

    def _on_remove_from_scene(self, scene):
        if scene.render_system is not None and self in scene.render_system.lights:
            scene.render_system.lights.remove(self)


class RenderDirectionalLightComponent(_Light):
    @property
    def direction(self):
        return self._pose.to_transformation_matrix()[:3, 0]     # a light looks along its +x


class RenderPointLightComponent(_Light):
    pass


class RenderSpotLightComponent(_Light):
    inner_fov = outer_fov = 0.0

    @property
    def direction(self):
        return self._pose.to_transformation_matrix()[:3, 0]     # a light looks along its +x


class RenderTexturedLightComponent(RenderSpotLightComponent):
    pass


class RenderParallelogramLightComponent(_Light):
    def set_shape(self, half_width, half_height, angle=1.5707963):
        self.half_width, self.half_height = half_width, half_height


class RenderCameraComponent(Component):
    def __init__(self, width, height, shader_dir=""):
        super().__init__()
        self.width, self.height = int(width), int(height)
        self.near, self.far = 0.01, 100.0
        self.local_pose = Pose()
        self.fx = self.fy = 0.5 * self.height       # fovy = 90 degrees until set
        self.cx, self.cy = 0.5 * self.width, 0.5 * self.height
        self.skew = 0.0
        self.gpu_pose_batch_index = -1
        self._group = None
        self._props = {}

    # intrinsics ------------------------------------------------------------------------------------------------------
    def set_fovy(self, fovy, compute_x=True):
        self.fy = 0.5 * self.height / np.tan(0.5 * float(fovy))
        if compute_x:
            self.fx = self.fy

    def set_fovx(self, fovx, compute_y=True):
        self.fx = 0.5 * self.width / np.tan(0.5 * float(fovx))
        if compute_y:
            self.fy = self.fx

    @property
    def fovy(self):
        return float(2 * np.arctan(0.5 * self.height / self.fy))

    @property
    def fovx(self):
        return float(2 * np.arctan(0.5 * self.width / self.fx))

    def set_focal_lengths(self, fx, fy):
        self.fx, self.fy = float(fx), float(fy)

    def set_principal_point(self, cx, cy):
        self.cx, self.cy = float(cx), float(cy)

    def set_skew(self, s):
        self.skew = float(s)

    def set_perspective_parameters(self, near, far, fx, fy, cx, cy, skew):
        self.near, self.far, self.fx, self.fy, self.cx, self.cy, self.skew = near, far, fx, fy, cx, cy, skew

    def set_near(self, v): self.near = float(v)
    def set_far(self, v): self.far = float(v)
    def get_near(self): return self.near
    def get_far(self): return self.far
    def get_width(self): return self.width
    def get_height(self): return self.height
    def get_skew(self): return self.skew
    def get_local_pose(self): return self.local_pose
    def set_local_pose(self, p): self.local_pose = p

    def set_gpu_pose_batch_index(self, i):
        self.gpu_pose_batch_index = int(i)

    def set_property(self, name, value):
        self._props[name] = value

    def set_texture(self, name, tex): self._props[name] = tex
    def set_texture_array(self, name, texs): self._props[name] = texs

    # matrices (render_camera.py's CPU branch) ------------------------------------------------------------------------------
    def get_intrinsic_matrix(self):
        return np.array([[self.fx, self.skew, self.cx], [0, self.fy, self.cy], [0, 0, 1]], dtype=np.float32)

    @property
    def global_pose(self):
        return (self.entity.pose if self.entity is not None else Pose()) * self.local_pose

    def get_global_pose(self):
        return self.global_pose

    def get_extrinsic_matrix(self):
        ros2opencv = np.array([[0, -1, 0, 0], [0, 0, -1, 0], [1, 0, 0, 0], [0, 0, 0, 1]], dtype=np.float32)
        return (ros2opencv @ self.global_pose.inv().to_transformation_matrix())[:3, :4]

    def get_model_matrix(self):
        return (self.global_pose * Pose([0, 0, 0], [-0.5, -0.5, 0.5, 0.5])).to_transformation_matrix()

    def get_projection_matrix(self):
        n, f, w, h = self.near, self.far, self.width, self.height
        M = np.zeros((4, 4), dtype=np.float32)
        M[0, 0], M[1, 1] = 2 * self.fx / w, -2 * self.fy / h
        M[0, 2], M[1, 2] = -(2 * self.cx / w - 1), -(2 * self.cy / h - 1)
        M[2, 2], M[2, 3] = -f / (f - n), -f * n / (f - n)
        M[3, 2] = -1
        return M

    # single-camera picture API (CPU sim path of render_camera.py) -----------------------------------------------------------
    def take_picture(self):
        self._single_group().take_picture()

    def get_picture(self, name):
        t = self._single_group().get_picture_cuda(name).torch()[0]
        return t.detach().cpu().numpy()

    def get_picture_cuda(self, name):
        g = self._single_group()
        return _Picture(g.get_picture_cuda(name).torch()[0])

    def get_picture_names(self):
        return ["Color", "PositionSegmentation"]

    def _single_group(self):
        if self._group is None:
            rs = self.entity.scene.render_system
            grp = rs._own_group()
            self._group = grp.create_camera_group([self], ["Color", "PositionSegmentation"])
        return self._group

    def _on_add_to_scene(self, scene):
        if scene.render_system is not None:
            scene.render_system.cameras.append(self)

    def _on_remove_from_scene(self, scene):
        if scene.render_system is not None and self in scene.render_system.cameras:
            scene.render_system.cameras.remove(self)


class _Picture:
    def __init__(self, t):
        self._t = t

    def torch(self):
        return self._t


# ----------------------------------------------------------------------------------------------------------- systems
class RenderSystem:
    """One per sub-scene (sapien_env.py:1194-1196)."""

    def __init__(self, device=None):
        self.device = device
        self.render_bodies: list[RenderBodyComponent] = []
        self.cameras: list[RenderCameraComponent] = []
        self.lights: list[_Light] = []
        self.ambient_light = [0.0, 0.0, 0.0]
        self.cubemap = None
        self._scene = None
        self._group = None

    def get_render_bodies(self): return self.render_bodies
    def get_cameras(self): return self.cameras
    def get_lights(self): return self.lights
    def get_ambient_light(self): return self.ambient_light
    def set_ambient_light(self, c): self.ambient_light = list(c)
    def set_cubemap(self, c): self.cubemap = c

    def _own_group(self):
        if self._group is None:
            self._group = RenderSystemGroup([self])
            px = self._scene.physx_system
            if hasattr(px, "_ensure"):
                px._ensure()
            self._group.set_cuda_poses(px.cuda_rigid_body_data)
        return self._group

    def _update_render_single(self):
        pass     # poses are read from the simulator's own state by the rasteriser

    def step(self):
        pass


class RenderCameraGroup:
    """``camera_group.take_picture()`` / ``get_picture_cuda(name).torch()`` for one camera across all sub-scenes.  With several
    structural groups every group renders its own sub-scenes (its own rasteriser template and camera); the textures are handed out
    in sub-scene order."""

    def __init__(self, group: "RenderSystemGroup", cameras, texture_names):
        self._g = group
        self.cameras = list(cameras)
        self.texture_names = list(texture_names)
        cam = self.cameras[0]
        for c in self.cameras:
            if (c.width, c.height) != (cam.width, cam.height):
                raise RuntimeError("all cameras of a group must have the same size")
        if abs(cam.fx - cam.fy) > 1e-4 * cam.fy or abs(cam.cx - 0.5 * cam.width) > 1e-3 or abs(cam.cy - 0.5 * cam.height) > 1e-3 or cam.skew != 0:
            raise RuntimeError("only centred pinhole cameras with square pixels are supported")
        from maniskill_amd import _native as N
        self._cams = []          # (engine, camera id, sub-scene indices)
        for eng, mine, _gi in group._parts:
            c0 = self.cameras[mine[0]] if len(self.cameras) > mine[0] else cam
            mount_body, local = group._camera_mount(c0)
            L, ctx = eng.lib, eng.ctx
            cid = L.camera_create(ctx, c0.width, c0.height, float(c0.fovy), float(c0.near), float(c0.far), int(mount_body),
                                  N._fa(list(local._p) + list(local._q), 7))
            if cid < 0:
                msg = L.last_error(ctx)
                raise RuntimeError(f"failed to create camera buffer: {msg.decode() if msg else cid}")
            self._cams.append((eng, cid, mine))
        self.id = self._cams[0][1]
        self._tex = {}
        self._perm = None
        if len(self._cams) > 1:      # rows of the concatenated group textures -> sub-scene order
            order = [k for _, _, mine in self._cams for k in mine]
            inv = np.empty(len(order), dtype=np.int64)
            inv[np.asarray(order)] = np.arange(len(order))
            self._perm = inv

    def _buffer_of(self, eng, cid, name):
        import torch
        key = (id(eng), name)
        if key in self._tex:
            return self._tex[key]
        L, ctx = eng.lib, eng.ctx
        shape = (C.c_int64 * 4)()
        if name == "PositionSegmentation":
            ptr, ctype, typestr = L.camera_buffer(ctx, cid, shape), C.c_int16, "<i2"
        elif name == "Color":
            ptr, ctype, typestr = L.camera_obs_buffer(ctx, cid, 2, shape), C.c_uint8, "|u1"
        else:
            raise RuntimeError(f"the minimal shader pack provides Color and PositionSegmentation, not {name}")
        if not ptr:
            raise RuntimeError(f"no buffer for texture {name}")
        shp = tuple(int(s) for s in shape)
        if eng.host_memory:
            n = int(np.prod(shp))
            t = torch.from_numpy(np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ctype)), shape=(n,)).reshape(shp))
        else:
            from maniskill_amd.physx import _DevicePointer
            t = torch.as_tensor(_DevicePointer(ptr, shp, typestr), device=eng.device)
        self._tex[key] = t
        return t

    def _buffer(self, name):
        import torch
        parts = [self._buffer_of(eng, cid, name) for eng, cid, _ in self._cams]
        if self._perm is None:
            return parts[0]
        if not isinstance(self._perm, torch.Tensor):
            self._perm = torch.as_tensor(self._perm, device=parts[0].device)
        return torch.cat(parts, dim=0).index_select(0, self._perm)

    def take_picture(self):
        for eng, cid, _ in self._cams:
            for n in self.texture_names:       # a texture must have been requested before the picture that fills it
                if n in ("Color",):
                    self._buffer_of(eng, cid, n)
            eng.lib.check(eng.ctx, eng.lib.camera_take_picture(eng.ctx, cid, eng._stream()), "camera_take_picture")

    def get_picture_cuda(self, name):
        off = (name == "PositionSegmentation" and not getattr(self, "_position_texture", True)) or (name == "Color" and not getattr(self, "_color_on", True))
        if off:
            # pictures are taken without this texture (set_outputs): rendered for this request, from the current state, the mode put back
            was = (getattr(self, "_position_texture", True), getattr(self, "_color_on", True))
            self.set_outputs(True, True)
            try:
                self.take_picture()
            finally:
                self.set_outputs(*was)
        return _Picture(self._buffer(name))

    def get_picture_names(self):
        return ["Color", "PositionSegmentation"]

    # extension of this backend (not in SAPIEN; maniskill_amd/fused_step.py uses it when present): the rasteriser's own depth / segmentation planes --
    # what the minimal pack's texture transform (render/shaders.py:75-83: -data[..., [2]], data[..., [3]]) computes from PositionSegmentation, written by
    # the rasteriser's own store -- and pictures without the int16 x 4 texture (msk_camera_set_outputs)
    def planes(self):
        """-> (depth, segmentation) int16 [N, H, W, 1] of the last take_picture; one structural group only"""
        if len(self._cams) != 1:
            raise RuntimeError("planes(): sub-scenes of several structural groups render into several buffers")
        eng, cid, _ = self._cams[0]
        out = []
        for which in (0, 1):
            key = (id(eng), f"plane{which}")
            t = self._tex.get(key)
            if t is None:
                import torch
                shape = (C.c_int64 * 4)()
                ptr = eng.lib.camera_obs_buffer(eng.ctx, cid, which, shape)
                if not ptr:
                    raise RuntimeError("no plane buffer")
                shp = tuple(int(v) for v in shape)
                if eng.host_memory:
                    t = torch.from_numpy(np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_int16)), shape=(int(np.prod(shp)),)).reshape(shp))
                else:
                    from maniskill_amd.physx import _DevicePointer
                    t = torch.as_tensor(_DevicePointer(ptr, shp, "<i2"), device=eng.device)
                self._tex[key] = t
            out.append(t)
        return tuple(out)

    def set_outputs(self, position_texture: bool, color: bool = True):
        """position_texture False: take_picture fills the planes (and Color) only, not the PositionSegmentation texture; color False: Color is neither shaded
        nor stored although the pack lists it (include/msk_render.h: MSK_CAM_OUT_NO_COLOR)"""
        for eng, cid, _ in self._cams:
            eng.lib.check(eng.ctx, eng.lib.camera_set_outputs(eng.ctx, cid, int(bool(position_texture)) | (0 if color else 2)), "camera_set_outputs")
        self._position_texture = bool(position_texture)
        self._color_on = bool(color)


class RenderSystemGroup:
    """``RenderSystemGroup([s.render_system for s in sub_scenes])`` + ``set_cuda_poses(px.cuda_rigid_body_data)``
    (envs/scene.py:1026-1037): compiles the render bodies of sub-scene 0 into the rasteriser's template."""

    def __init__(self, systems):
        self.systems = list(systems)
        self._engine = None
        self._groups = []

    def set_cuda_poses(self, handle):
        px = self.systems[0]._scene.physx_system
        self._px = px
        self._engine = px._engine
        # one rasteriser template per structural group of sub-scenes (the groups the physics runs as, _system.py): compiled from the
        # group's first sub-scene; (engine, sub-scene indices in this RenderSystemGroup's order)
        self._parts = []
        if len(self.systems) == 1 or len(px._groups) == 1:
            self._parts.append((px._engine, list(range(len(self.systems))), 0))
        else:
            index = {px._scene_index(rs._scene): k for k, rs in enumerate(self.systems)}
            for gi, g in enumerate(px._groups):
                mine = [index[e] for e in g.envs if e in index]
                if mine:
                    self._parts.append((g.engine, mine, gi))
        self.simplification = dict(parts=0, max_surface_error=0.0, tris_before=0, tris_after=0)
        for eng, mine, gi in self._parts:
            self._compile(eng, self.systems[mine[0]], gi)

    def update_render(self):
        pass     # the rasteriser reads body poses straight from the simulator's state at take_picture

    def create_camera_group(self, cameras, texture_names):
        g = RenderCameraGroup(self, cameras, texture_names)
        self._groups.append(g)
        return g

    # -----------------------------------------------------------------------------------------------------------------
    def _body_of(self, entity):
        """(template body id or -1, pose to fold into local poses)"""
        pb = entity._physx_body() if entity is not None else None
        if pb is None or pb._body_id < 0:
            return -1, (entity._pose if entity is not None else Pose())
        return pb._body_id, Pose()

    def _camera_mount(self, cam):
        body, fold = self._body_of(cam.entity)
        return body, fold * cam.local_pose

    def _compile(self, eng, rs0, gi):
        from maniskill_amd import _native as N
        L, ctx = eng.lib, eng.ctx
        # triangle budget of the rasteriser's scene template (include/msk_render.h: 8192 triangles, 4096 vertices): parts above their
        # share are simplified; the largest surface displacement is kept for the record
        counts = [len(f) for rb in rs0.render_bodies for sh in rb.render_shapes for (_, f, _) in sh._triangles()]
        # MSK_RENDER_TRI_BUDGET / MSK_RENDER_PART_KEEP: the template's triangle budget for simplified parts and the size up to which a part is drawn as it is
        # (tools/camera_fidelity.py measures what a budget costs in segmentation IoU against the un-simplified geometry; a picture's cost grows with the triangles)
        KEEP = int(os.environ.get("MSK_RENDER_PART_KEEP", "400"))   # parts up to this size are drawn as they are (table.glb's four meshes)
        BUDGET = int(os.environ.get("MSK_RENDER_TRI_BUDGET", "2600"))   # (round 6: 5200 with cell means gave 99.36 % pixel agreement, 2600 with quadric minimisers 99.67 %: profiles/r06_camera_fidelity.log)
        small = sum(n for n in counts if n <= KEEP)
        dense = sum(1 for n in counts if n > KEEP)
        MAX_TRIS_PER_PART = KEEP if not dense else max(64, min(max(KEEP, BUDGET // 8), (BUDGET - small) // dense))   # template: 8192 triangles, 4096 vertices
        declared = dict(getattr(self._px, "_env_box_shapes_of_group", {}).get(gi, {}))
        skipped_pose_only = 0
        textures_used = {}
        for rb in rs0.render_bodies:
            if rb.visibility <= 0:
                continue
            ent = rb.entity
            pb = ent._physx_body() if ent is not None else None
            if pb is not None and pb._body_id <= -2:
                # pose-only actor (_system.py: kinematic, shape-less, beyond the engine's body capacity): the rasteriser takes body poses
                # from the engine's state, which has no row for it -- left out of the picture rather than drawn frozen where it was built
                skipped_pose_only += 1
                continue
            body, fold = self._body_of(ent)
            seg = int(ent.per_scene_id)
            for shape in rb.render_shapes:
                lp = fold * shape.local_pose
                follows = None
                if isinstance(shape, RenderShapeBox) and body in declared:
                    for sid, hs in declared[body]:
                        if np.allclose(hs, shape.half_size):
                            follows = sid
                parts = shape._textured_parts() if hasattr(shape, "_textured_parts") else [(v, f, c, None, None) for v, f, c in shape._triangles()]
                for v, f, rgba, uvs, pix in parts:
                    if follows is not None:
                        v = v / np.maximum(shape.half_size.astype(np.float64), 1e-12)
                    if len(v) > len(f) and uvs is None:     # exporters write three vertices per triangle: weld identical positions (exact)
                        uq, inv = np.unique(np.round(np.asarray(v, dtype=np.float64), 7), axis=0, return_inverse=True)
                        v, f = uq, inv.reshape(-1)[np.asarray(f, dtype=np.int64)]
                    if len(f) > KEEP:
                        # the rasteriser's template is small (include/msk_render.h capacities): dense visual meshes are drawn as
                        # their <= 64-vertex hull until the micro-triangle path exists; flat sheets (the ground grid,
                        # building/ground.py:46-119: 20 000 coplanar triangles) as their outline polygon, which is exact
                        n0 = len(f)
                        _simplify.last_bound = 0.0
                        v_in = np.asarray(v, dtype=np.float64)
                        v, f = _simplify(v, f, MAX_TRIS_PER_PART)
                        if uvs is not None:
                            # a flat sheet whose texture coordinates are affine in position (the ground: uv = xy * scale + offset) keeps its
                            # texture on the outline polygon exactly; anything else falls back to the texture's mean colour
                            ring = getattr(_simplify, "last_ring", None)
                            ok = False
                            if ring is not None:
                                Aff = np.concatenate([v_in, np.ones((len(v_in), 1))], axis=1)
                                M, *_ = np.linalg.lstsq(Aff, uvs, rcond=None)
                                ok = float(np.abs(Aff @ M - uvs).max()) < 1e-6 * max(1.0, float(np.abs(uvs).max()))
                            if ok:
                                uvs = uvs[ring]
                            else:
                                uvs, pix = None, None
                                rgba = shape.material._flat_color() if hasattr(shape, "material") else rgba
                        sm = self.simplification
                        sm["parts"] += 1
                        sm["max_surface_error"] = max(sm["max_surface_error"], _simplify.last_bound)
                        sm["tris_before"] += n0
                        sm["tris_after"] += len(f)
                    v32 = np.ascontiguousarray(v, dtype=np.float32)
                    f32 = np.ascontiguousarray(f, dtype=np.int32)
                    rsid = L.render_add_mesh(ctx, int(body), N._fa(list(lp._p) + list(lp._q), 7), v32.ctypes.data_as(C.POINTER(C.c_float)),
                                             len(v32), f32.ctypes.data_as(C.POINTER(C.c_int32)), len(f32), seg)
                    L.check(ctx, rsid, "render_add_mesh")
                    L.check(ctx, L.render_set_base_color(ctx, rsid, N._fa(rgba, 4)), "render_set_base_color")
                    if uvs is not None and pix is not None and len(uvs) == len(v32):
                        ntx = pix.shape[0] * pix.shape[1]      # (include/msk_render.h: 8 textures, 2^20 texels per context; beyond: base colour only)
                        if textures_used.get("texels", 0) + (4 * ntx) // 3 + 16 <= (1 << 20) and textures_used.get("count", 0) < 8:
                            uv32 = np.ascontiguousarray(uvs, dtype=np.float32)
                            L.check(ctx, L.render_set_texture(ctx, rsid, pix.ctypes.data_as(C.POINTER(C.c_uint8)), int(pix.shape[1]), int(pix.shape[0]),
                                                              uv32.ctypes.data_as(C.POINTER(C.c_float))), "render_set_texture")
                            textures_used["texels"] = textures_used.get("texels", 0) + (4 * ntx) // 3 + 16
                            textures_used["count"] = textures_used.get("count", 0) + 1
                    if follows is not None:
                        L.check(ctx, L.render_bind_env_box(ctx, rsid, follows), "render_bind_env_box")
        if skipped_pose_only:
            import warnings
            warnings.warn(f"{skipped_pose_only} pose-only actors per sub-scene (kinematic, no collision shape, beyond the engine's 63 bodies) "
                          "are not drawn by the camera")
        dirs, cols = [], []
        for l in rs0.lights:
            if isinstance(l, RenderDirectionalLightComponent) and len(dirs) < 4:
                dirs.append(l.direction)
                cols.append(l.color[:3])
        amb = np.asarray(rs0.ambient_light, dtype=np.float32)[:3]
        if dirs or amb.any():
            d = np.ascontiguousarray(dirs, dtype=np.float32).reshape(-1, 3)
            c = np.ascontiguousarray(cols, dtype=np.float32).reshape(-1, 3)
            fp = C.POINTER(C.c_float)
            L.check(ctx, L.render_set_lights(ctx, N._fa(amb, 3), len(d), d.ctypes.data_as(fp), c.ctypes.data_as(fp)), "render_set_lights")
        # point and spot lights (envs/scene.py:582-640): fixed places of the sub-scene frame, inverse-square, no shadows
        local = []
        for l in rs0.lights:
            if isinstance(l, RenderParallelogramLightComponent):
                continue                                        # area lights belong to the ray-traced packs
            if isinstance(l, (RenderPointLightComponent, RenderSpotLightComponent)) and len(local) < 8:
                spot = isinstance(l, RenderSpotLightComponent)
                ax = l.direction if spot else np.array([1.0, 0.0, 0.0])
                inner, outer = (float(l.inner_fov), float(l.outer_fov)) if spot else (0.0, 0.0)
                if spot and not inner > 0.0:
                    inner = 1e-3
                local.append(list(l._pose._p) + list(ax) + list(l.color[:3]) + [inner, max(outer, inner), 0.0])
        if local:
            a = np.ascontiguousarray(local, dtype=np.float32)
            L.check(ctx, L.render_set_local_lights(ctx, len(a), a.ctypes.data_as(C.POINTER(C.c_float))), "render_set_local_lights")
        L.check(ctx, L.render_finalize(ctx), "render_finalize")


# names the 3.1 render API would add are deliberately absent (GpuSyncManager, RenderManager): ManiSkill then stays on the 3.0 path
