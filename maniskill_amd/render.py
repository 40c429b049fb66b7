"""Host-side mirror of the ``sapien.render`` surface on ManiSkill's camera hot path.

Reference call sites (names and argument meaning kept):
  * ``RenderSystemGroup`` / ``set_cuda_poses`` / ``update_render``       mani_skill/envs/scene.py:382-427,1026-1037
  * ``create_camera_group`` / ``take_picture`` / ``get_picture_cuda``   scene.py:1087-1110; utils/structs/render_camera.py:160-182,269-273
  * ``CameraConfig`` (uid, pose, width, height, fov, near, far)         mani_skill/sensors/camera.py:32-67
  * ``look_at``                                                         mani_skill/utils/sapien_utils.py:317-364
  * texture transforms of the ``minimal`` shader pack                   mani_skill/render/shaders.py:68-84

All pixels are produced by the C-ABI library (include/msk_render.h: tile-binned HIP rasteriser); this module owns
handles, builds the triangle lists of the template's shapes once (cold path) and exposes zero-copy tensor views.

What is drawn: every collision shape of the template that belongs to a visible body (boxes, convex hulls, the
ground plane as a large quad).  The reference draws the robots' visual GLB meshes (~135 k triangles per Panda);
here the robot appears as its cooked collision hulls — same silhouettes to within the hull-vs-mesh
deviation, segmentation ids per link identical.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Optional, Sequence

import numpy as np
import torch

from .graph import const

from . import _native as N

GROUND_HALF_EXTENT = 50.0  # building/ground.py:46-119 draws a 100 x 100 m grid


@dataclass
class CameraConfig:
    """sensors/camera.py:32-67 (the fields the camera hot path reads)."""
    uid: str
    p: Sequence[float]
    q: Sequence[float]
    width: int = 128
    height: int = 128
    fov: float = np.pi / 2
    near: float = 0.01
    far: float = 100.0
    mount: int = -1            # template body id the camera rides on, -1 = fixed in the sub-scene frame


def look_at(eye, target, up=(0.0, 0.0, 1.0)):
    """sapien_utils.look_at: camera pose (p, q wxyz) with x forward, y left, z up."""
    eye, target, up = (np.asarray(v, dtype=np.float64) for v in (eye, target, up))
    f = target - eye
    f /= np.linalg.norm(f)
    left = np.cross(up / np.linalg.norm(up), f)
    left /= np.linalg.norm(left)
    upv = np.cross(f, left)
    R = np.stack([f, left, upv], axis=1)
    # rotation matrix -> quaternion (wxyz)
    t = np.trace(R)
    if t > 0:
        s = np.sqrt(t + 1.0) * 2
        q = np.array([0.25 * s, (R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s])
    else:
        i = int(np.argmax(np.diag(R)))
        j, k = (i + 1) % 3, (i + 2) % 3
        s = np.sqrt(1.0 + R[i, i] - R[j, j] - R[k, k]) * 2
        q = np.zeros(4)
        q[0] = (R[k, j] - R[j, k]) / s
        q[1 + i] = 0.25 * s
        q[1 + j] = (R[j, i] + R[i, j]) / s
        q[1 + k] = (R[k, i] + R[i, k]) / s
    if q[0] < 0:
        q = -q
    return eye.astype(np.float32), (q / np.linalg.norm(q)).astype(np.float32)


# --------------------------------------------------------------------------------------
# triangle lists of the primitive shapes (counter-clockwise seen from outside)
# --------------------------------------------------------------------------------------
def box_mesh(half):
    hx, hy, hz = [float(h) for h in half]
    v = np.array([[sx * hx, sy * hy, sz * hz] for sz in (-1, 1) for sy in (-1, 1) for sx in (-1, 1)], dtype=np.float32)
    quads = [(0, 2, 3, 1), (4, 5, 7, 6), (0, 1, 5, 4), (2, 6, 7, 3), (0, 4, 6, 2), (1, 3, 7, 5)]
    t = []
    for a, b, c, d in quads:
        t += [(a, b, c), (a, c, d)]
    return v, _outward(v, np.asarray(t, dtype=np.int32))


def plane_mesh(half_extent=GROUND_HALF_EXTENT):
    """Plane through the shape origin with normal +x (SAPIEN convention), as one quad."""
    L = float(half_extent)
    v = np.array([[0, -L, -L], [0, L, -L], [0, L, L], [0, -L, L]], dtype=np.float32)
    return v, np.array([[0, 1, 2], [0, 2, 3]], dtype=np.int32)   # counter-clockwise seen from +x


def hull_mesh(verts):
    from scipy.spatial import ConvexHull

    v = np.ascontiguousarray(verts, dtype=np.float32).reshape(-1, 3)
    h = ConvexHull(v.astype(np.float64))
    return v, _outward(v, h.simplices.astype(np.int32))


def _outward(v, tris):
    c = v.mean(axis=0)
    out = tris.copy()
    for i, (a, b, cc) in enumerate(tris):
        n = np.cross(v[b] - v[a], v[cc] - v[a])
        if np.dot(n, v[a] - c) < 0:
            out[i] = (a, cc, b)
    return out


class PictureHandle:
    """``camera_group.get_picture_cuda(name)``: an object with ``.torch()``."""

    def __init__(self, tensor):
        self._t = tensor

    def torch(self):
        return self._t


class _DevPtr:
    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {"shape": tuple(int(s) for s in shape), "typestr": typestr, "data": (int(ptr), False),
                                         "version": 2, "strides": None}


class RenderCameraGroup:
    """One batched camera: ``take_picture()`` rasterises every sub-scene, ``get_picture_cuda`` returns the
    ``PositionSegmentation`` texture (N, H, W, 4) int16."""

    def __init__(self, px, cfg: CameraConfig):
        self.px, self.cfg = px, cfg
        L = px.lib
        cam = L.check(px.ctx, L.camera_create(px.ctx, int(cfg.width), int(cfg.height), float(cfg.fov), float(cfg.near), float(cfg.far),
                                              int(cfg.mount), N._fa(list(cfg.p) + list(cfg.q), 7)), "camera_create")
        self.id = cam
        def wrap(ptr, shape, what, ctype=C.c_int16, typestr="<i2"):
            if not ptr:
                raise RuntimeError(f"{what} returned NULL")
            shp = tuple(int(s) for s in shape)
            if px.host_memory:
                n = int(np.prod(shp))
                return torch.from_numpy(np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ctype)), shape=(n,)).reshape(shp))
            t = torch.as_tensor(_DevPtr(ptr, shp, typestr), device=px.device)
            assert t.data_ptr() == ptr
            return t

        shape = (C.c_int64 * 4)()
        self._tex = wrap(L.camera_buffer(px.ctx, cam, shape), shape, "msk_camera_buffer")
        # depth / segmentation planes written by the rasteriser's own store (no gather pass over the texture)
        self._depth = wrap(L.camera_obs_buffer(px.ctx, cam, 0, shape), shape, "msk_camera_obs_buffer")
        self._seg = wrap(L.camera_obs_buffer(px.ctx, cam, 1, shape), shape, "msk_camera_obs_buffer")
        self._color_t = None   # Color r8g8b8a8unorm (N, H, W, 4) uint8: requested lazily (the backend only renders it once asked)
        self._position_texture = True      # msk_camera_set_outputs: False = only the depth / segmentation planes are rendered (set_outputs)
        self._wrap = wrap
        # intrinsics of set_fovy(fovy, compute_x=True) (scene.py:250-257)
        fy = 0.5 * cfg.height / np.tan(0.5 * cfg.fov)
        self.intrinsic_cv = torch.tensor([[fy, 0, 0.5 * cfg.width], [0, fy, 0.5 * cfg.height], [0, 0, 1]], dtype=torch.float32)
        self._local = None
        self._cached_extrinsic = None
        self._cached_intrinsic = None
        self._cached_model = None

    # ---- camera matrices (utils/structs/render_camera.py:77-155; sensors/camera.py:248-253) ---------------------
    @staticmethod
    def _pose_to_matrix(p: torch.Tensor, q: torch.Tensor) -> torch.Tensor:
        """(B,3), (B,4 wxyz) -> (B,4,4)"""
        w, x, y, z = q.unbind(-1)
        R = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w),
                         2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w),
                         2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], dim=-1).view(-1, 3, 3)
        T = torch.zeros(p.shape[0], 4, 4, dtype=torch.float32, device=p.device)
        T[:, :3, :3] = R
        T[:, :3, 3] = p
        T[:, 3, 3] = 1
        return T

    def get_global_pose(self) -> torch.Tensor:
        """(N, 7) camera pose in its sub-scene's frame: the local pose, or mount.pose * local_pose (render_camera.py:313-317)."""
        cfg, px = self.cfg, self.px
        if self._local is None:      # uploaded once: a host-to-device copy per call could not be captured into a step graph
            self._local = torch.tensor(list(cfg.p) + list(cfg.q), dtype=torch.float32, device=px.device)
        local = self._local
        N = px.num_envs
        if cfg.mount < 0:
            return local[None].repeat(N, 1)
        px.gpu_fetch_all()
        rbd = px.cuda_rigid_body_data.torch().view(N, px.bodies_per_env, 13)
        mp = rbd[:, cfg.mount, :3] - px.scene_offsets
        mq = rbd[:, cfg.mount, 3:7]
        # pose product: p = mp + R(mq) lp ; q = mq * lq
        Tm = self._pose_to_matrix(mp, mq)
        lp = local[:3]
        p = (Tm[:, :3, :3] @ lp) + mp
        w1, x1, y1, z1 = mq.unbind(-1)
        w2, x2, y2, z2 = local[3:7]
        q = torch.stack([w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2, w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2,
                         w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2, w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2], dim=-1)
        return torch.cat([p, q], dim=-1)

    def get_extrinsic_matrix(self) -> torch.Tensor:
        """(N, 3, 4) OpenCV extrinsic: ros2opencv @ inv(global pose)."""
        if self.cfg.mount < 0 and self._cached_extrinsic is not None:
            return self._cached_extrinsic
        g = self.get_global_pose()
        T = self._pose_to_matrix(g[:, :3], g[:, 3:7])
        # inverse of a rigid transform, written out (Pose.inv() in the reference; torch.linalg.inv checks its `info` on the host, which a
        # graph capture refuses): [R p]^-1 = [R^T  -R^T p]
        Tinv = torch.zeros_like(T)
        Rt = T[:, :3, :3].transpose(1, 2)
        Tinv[:, :3, :3] = Rt
        Tinv[:, :3, 3] = -(Rt @ T[:, :3, 3:4])[:, :, 0]
        Tinv[:, 3, 3] = 1
        ros2opencv = const(((0, 0, 1, 0), (-1, 0, 0, 0), (0, -1, 0, 0), (0, 0, 0, 1)), g.device).T
        res = (ros2opencv @ Tinv)[:, :3, :4]
        if self.cfg.mount < 0:
            self._cached_extrinsic = res
        return res

    def get_model_matrix(self) -> torch.Tensor:
        """(N, 4, 4) OpenGL camera-to-world: global pose * POSE_GL_TO_ROS (q = [-0.5, -0.5, 0.5, 0.5])."""
        if self.cfg.mount < 0 and self._cached_model is not None:
            return self._cached_model
        g = self.get_global_pose()
        gl = const((-0.5, -0.5, 0.5, 0.5), g.device)
        w1, x1, y1, z1 = g[:, 3:7].unbind(-1)
        w2, x2, y2, z2 = gl
        q = torch.stack([w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2, w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2,
                         w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2, w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2], dim=-1)
        res = self._pose_to_matrix(g[:, :3], q)
        if self.cfg.mount < 0:
            self._cached_model = res
        return res

    def get_intrinsic_matrix(self) -> torch.Tensor:
        if self._cached_intrinsic is None:
            self._cached_intrinsic = self.intrinsic_cv[None].repeat(self.px.num_envs, 1, 1).to(self.px.device)
        return self._cached_intrinsic

    def get_params(self) -> dict:
        """Camera.get_params (sensors/camera.py:248-253)."""
        return dict(extrinsic_cv=self.get_extrinsic_matrix(), cam2world_gl=self.get_model_matrix(), intrinsic_cv=self.get_intrinsic_matrix())

    def take_picture(self):
        L, px = self.px.lib, self.px
        L.check(px.ctx, L.camera_take_picture(px.ctx, self.id, px._stream()), "camera_take_picture")

    @property
    def _color(self):
        if self._color_t is None:
            shape = (C.c_int64 * 4)()
            self._color_t = self._wrap(self.px.lib.camera_obs_buffer(self.px.ctx, self.id, 2, shape), shape, "msk_camera_obs_buffer",
                                       C.c_uint8, "|u1")
        return self._color_t

    def enable_color(self):
        """Ask for the Color texture before the first take_picture that should fill it."""
        return self._color

    def set_outputs(self, position_texture: bool):
        """``position_texture=False``: the caller only reads ``get_obs(depth=..., segmentation=..., rgb=...)`` -- the int16 x 4 PositionSegmentation texture is
        neither computed nor stored from the next ``take_picture`` on (8 of the 12 bytes a pixel costs; the planes keep their bits).  Asking for the texture
        all the same (``get_picture_cuda``, ``get_obs(position=True)``) renders it for that request."""
        position_texture = bool(position_texture)
        if position_texture != self._position_texture:
            L = self.px.lib
            L.check(self.px.ctx, L.camera_set_outputs(self.px.ctx, self.id, int(position_texture)), "camera_set_outputs")
            self._position_texture = position_texture

    def _texture(self):
        """the PositionSegmentation texture of the CURRENT state.  With ``set_outputs(False)`` in force the pictures do not fill it (a captured step graph
        keeps the mode it was captured with, whatever is set later), so it is rendered here, for this request, and the configured mode is restored.
        SIDE EFFECT: that request is a full ``take_picture`` of the current physics state -- it also rewrites the depth, segmentation and colour planes, so
        planes handed out earlier with ``copy=False`` show the current state afterwards (ask for copies when the state has advanced in between)."""
        if not self._position_texture:
            L = self.px.lib
            L.check(self.px.ctx, L.camera_set_outputs(self.px.ctx, self.id, 1), "camera_set_outputs")
            try:
                self.take_picture()
            finally:
                L.check(self.px.ctx, L.camera_set_outputs(self.px.ctx, self.id, int(self._position_texture)), "camera_set_outputs")
        return self._tex

    def get_picture_cuda(self, name: str = "PositionSegmentation") -> PictureHandle:
        if name == "Color":
            return PictureHandle(self._color)
        if name != "PositionSegmentation":
            raise KeyError(f"the minimal shader pack provides Color and PositionSegmentation, not {name}")
        return PictureHandle(self._texture())

    def get_obs(self, depth=True, segmentation=True, position=False, copy=True, rgb=False):
        """Camera.get_obs (sensors/camera.py:190-242) with the minimal pack's texture transform.  ``copy=False`` hands
        out the rasteriser's own planes (overwritten by the next take_picture) instead of a snapshot."""
        out = {}
        if rgb:          # Color[..., :3] (render/shaders.py:74)
            out["rgb"] = self._color[..., :3].clone() if copy else self._color[..., :3]
        if position:
            out["position"] = self._texture()[..., :3]
        if depth:        # == -data[..., [2]]
            out["depth"] = self._depth.clone() if copy else self._depth
        if segmentation:  # == data[..., [3]]
            out["segmentation"] = self._seg.clone() if copy else self._seg
        return out


def add_render_mesh(px, body, pose7, verts, tris, seg, rgba=None, texture=None, uvs=None) -> int:
    """One RenderShapeTriangleMesh on body ``body`` (-1: static): msk_render_add_mesh + base colour + optional base-colour texture
    (``texture``: uint8 (h, w, 4), ``uvs``: (nverts, 2), (0, 0) = top-left texel corner).  Before msk_render_finalize."""
    L = px.lib
    v = np.ascontiguousarray(verts, dtype=np.float32).reshape(-1, 3)
    t = np.ascontiguousarray(tris, dtype=np.int32).reshape(-1, 3)
    rs = L.check(px.ctx, L.render_add_mesh(px.ctx, int(body), N._fa(pose7, 7), v.ctypes.data_as(C.POINTER(C.c_float)), len(v),
                                           t.ctypes.data_as(C.POINTER(C.c_int32)), len(t), int(seg)), "render_add_mesh")
    if rgba is not None:
        L.check(px.ctx, L.render_set_base_color(px.ctx, rs, N._fa(rgba, 4)), "render_set_base_color")
    if texture is not None:
        tex = np.ascontiguousarray(texture, dtype=np.uint8)
        uv = np.ascontiguousarray(uvs, dtype=np.float32).reshape(len(v), 2)
        assert tex.ndim == 3 and tex.shape[2] == 4
        L.check(px.ctx, L.render_set_texture(px.ctx, rs, tex.ctypes.data_as(C.POINTER(C.c_uint8)), tex.shape[1], tex.shape[0],
                                             uv.ctypes.data_as(C.POINTER(C.c_float))), "render_set_texture")
    return rs


def attach_template_visuals(px, template, hidden_bodies=(), lights=None, extra_meshes=()):
    """RenderBodyComponent per body from the template's collision shapes (building/actor_builder.py:166-191 attaches the
    visual records; this backend draws the collision geometry).  Segmentation id = body id + 1 (per_scene_id in
    add_entity order, 0 = background); the static ground gets bodies_per_env + 1.

    ``lights`` (optional, the `Color` texture's lighting instead of ManiSkill's default, envs/scene.py:566-718):
    ``dict(ambient=(r, g, b), directional=[(direction, colour), ...], point=[(position, colour), ...],
    spot=[(position, direction, inner_fov, outer_fov, colour), ...])``."""
    L = px.lib
    nb = px.bodies_per_env
    n = tot_v = tot_t = 0
    declared = {a[0] for op, a in template.ops if op == "declare_env_box"}   # per-env box instances: drawn at the env's own size
    si = -1
    for op, a in template.ops:
        if op not in ("add_shape", "add_visual"):
            continue
        if op == "add_shape":
            si += 1
        body, stype, pose7, params, verts = a[0], a[1], a[2], a[3], a[4]
        if body in hidden_bodies:
            continue
        follows = op == "add_shape" and si in declared
        if stype == N.SHAPE_BOX:
            v, t = box_mesh((1.0, 1.0, 1.0) if follows else params)
        elif stype == N.SHAPE_PLANE:
            v, t = plane_mesh()
        elif stype == N.SHAPE_CONVEX:
            v, t = hull_mesh(verts)
        else:
            continue
        seg = (body + 1) if body >= 0 else nb + 1
        v = np.ascontiguousarray(v, dtype=np.float32)
        t = np.ascontiguousarray(t, dtype=np.int32)
        rs = L.check(px.ctx, L.render_add_mesh(px.ctx, int(body), N._fa(pose7, 7), v.ctypes.data_as(C.POINTER(C.c_float)), len(v),
                                               t.ctypes.data_as(C.POINTER(C.c_int32)), len(t), int(seg)), "render_add_mesh")
        if follows:
            L.check(px.ctx, L.render_bind_env_box(px.ctx, rs, si), "render_bind_env_box")
        rgba = getattr(template, "body_colors", {}).get(int(body))
        if rgba is not None:
            L.check(px.ctx, L.render_set_base_color(px.ctx, rs, N._fa(rgba, 4)), "render_set_base_color")
        n += 1
        tot_v += len(v); tot_t += len(t)
    for em in extra_meshes:        # dicts of add_render_mesh's arguments (textured sheets and the like)
        add_render_mesh(px, **em)
        n += 1; tot_v += len(em["verts"]); tot_t += len(em["tris"])
    px.render_template_size = (n, tot_v, tot_t)      # shapes, vertices, triangles handed to the rasteriser
    if lights is not None:
        fp = C.POINTER(C.c_float)
        if "ambient" in lights or "directional" in lights:
            d = np.ascontiguousarray([x[0] for x in lights.get("directional", ())], dtype=np.float32).reshape(-1, 3)
            c = np.ascontiguousarray([x[1] for x in lights.get("directional", ())], dtype=np.float32).reshape(-1, 3)
            L.check(px.ctx, L.render_set_lights(px.ctx, N._fa(lights.get("ambient", (0.0, 0.0, 0.0)), 3), len(d), d.ctypes.data_as(fp),
                                                c.ctypes.data_as(fp)), "render_set_lights")
        rows = [list(p) + [1.0, 0.0, 0.0] + list(col) + [0.0, 0.0, 0.0] for p, col in lights.get("point", ())]
        rows += [list(p) + list(ax) + list(col) + [float(inner), float(outer), 0.0] for p, ax, inner, outer, col in lights.get("spot", ())]
        if rows:
            a = np.ascontiguousarray(rows, dtype=np.float32)
            L.check(px.ctx, L.render_set_local_lights(px.ctx, len(a), a.ctypes.data_as(fp)), "render_set_local_lights")
    L.check(px.ctx, L.render_finalize(px.ctx), "render_finalize")
    return n
