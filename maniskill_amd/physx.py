"""Host-side mirror of the ``sapien.physx`` surface ManiSkill uses on its hot path.

Reference call sites this mirrors (names and argument meaning kept):
  * ``physx.PhysxGpuSystem(device)``, ``.timestep``, ``.step()``            sapien_env.py:1187,1227; scene.py:379-380
  * ``px.gpu_init()``                                                     scene.py:910
  * ``px.cuda_rigid_body_data`` / ``cuda_articulation_*`` + ``.torch()``   structs/actor.py:352, articulation.py:726-797
  * ``px.gpu_apply_*`` / ``px.gpu_fetch_*`` / ``gpu_update_articulation_kinematics``   scene.py:950-986
  * ``px.gpu_create_contact_pair_impulse_query`` / ``gpu_query_contact_pair_impulses``  scene.py:771-781

All device work is done by the C-ABI library (include/msk_physx.h) whose kernels are
hand-written HIP for gfx950; this module only owns handles and zero-copy tensor views.
Differences from SAPIEN that are deliberate: all sub-scenes share one
``SceneTemplate`` (homogeneous envs) which is described once instead of N times, and buffer
rows are env-major (row = env * bodies_per_env + body id).
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import Optional, Sequence

import numpy as np
import torch

from . import _native as N


# --------------------------------------------------------------------------------------
# configuration (mirrors mani_skill/utils/structs/types.py:35-90)
# --------------------------------------------------------------------------------------
@dataclass
class SceneConfig:
    gravity: Sequence[float] = (0.0, 0.0, -9.81)
    bounce_threshold: float = 2.0
    sleep_threshold: float = 0.005
    contact_offset: float = 0.02
    rest_offset: float = 0.0
    solver_position_iterations: int = 15
    solver_velocity_iterations: int = 1
    enable_pcm: bool = True
    enable_tgs: bool = True
    enable_ccd: bool = False
    enable_enhanced_determinism: bool = False
    enable_friction_every_iteration: bool = True
    cpu_workers: int = 0
    contact_capacity: int = 1      # msk_config.contact_capacity: 1 = 128 points / 128 solver blocks per sub-scene (the wide solver class behind the others: the
                                   # default since round 5 -- 1.1 % of the headline, profiles/r05_contact_capacity_cost.log --, PegInsertionSide-v1 overran 48); 0 = 48 / 64


@dataclass
class DefaultMaterialsConfig:
    static_friction: float = 0.3
    dynamic_friction: float = 0.3
    restitution: float = 0.0


@dataclass
class SimConfig:
    spacing: float = 5.0
    sim_freq: int = 100
    control_freq: int = 20
    scene_config: SceneConfig = field(default_factory=SceneConfig)
    default_materials_config: DefaultMaterialsConfig = field(default_factory=DefaultMaterialsConfig)


# --------------------------------------------------------------------------------------
# template description (host only, cold path)
# --------------------------------------------------------------------------------------
def _pose7(p=(0, 0, 0), q=(1, 0, 0, 0)):
    return [float(x) for x in p] + [float(x) for x in q]


class SceneTemplate:
    """Description of ONE sub-scene; the backend replicates it ``num_envs`` times.

    Plays the role of the per-sub-scene loops in building/actor_builder.py:234-245 and
    building/articulation_builder.py:143-205, done once instead of N times.
    """

    def __init__(self):
        self.ops = []            # recorded C-ABI build calls
        self.body_names = []     # body id -> name
        self.body_masses = []    # body id -> template mass (kinematic actors: 0)
        self.body_kind = []
        self.art_names = []
        self.art_links = []      # per articulation: list of body ids (link order)
        self.art_active = []     # per articulation: list of body ids whose incoming joint is active (dof order)
        self.joint_names = {}    # body id -> incoming joint name
        self.joint_limits = {}   # body id -> (lo, hi)
        self.nshapes = 0

    # -- articulations ----------------------------------------------------------------
    def add_articulation(self, name, root_p=(0, 0, 0), root_q=(1, 0, 0, 0), floating=False) -> int:
        """floating: fix_root_link = False -- the root link gets six coordinates of its own (msk_set_articulation_floating)."""
        self.ops.append(("add_articulation", (_pose7(root_p, root_q),)))
        if floating:
            self.ops.append(("set_articulation_floating", (len(self.art_names),)))
        self.art_names.append(name)
        self.art_links.append([])
        self.art_active.append([])
        return len(self.art_names) - 1

    def add_link(self, art, name, parent, joint_type, joint_name="", pose_in_parent=None, pose_in_child=None,
                 limits=(-np.inf, np.inf), mass=0.0, com=(0, 0, 0), inertia6=(0,) * 6, disable_gravity=False,
                 armature=0.0, friction=0.0) -> int:
        bid = len(self.body_names)
        lo, hi = float(limits[0]), float(limits[1])
        lo = max(lo, -3.0e38)
        hi = min(hi, 3.0e38)
        self.ops.append(("add_link", (art, parent, joint_type, pose_in_parent or _pose7(), pose_in_child or _pose7(),
                                      lo, hi, float(mass), list(map(float, com)), list(map(float, inertia6)),
                                      int(bool(disable_gravity)), float(armature), float(friction))))
        self.body_names.append(name)
        self.body_masses.append(float(mass))
        self.body_kind.append(N.BODY_LINK)
        self.art_links[art].append(bid)
        self.joint_names[bid] = joint_name
        if parent >= 0 and joint_type != N.JOINT_FIXED:
            self.art_active[art].append(bid)
            self.joint_limits[bid] = (lo, hi)
        return bid

    def set_drive(self, link_body, stiffness, damping, force_limit=3.0e38, mode="force"):
        self.ops.append(("set_drive", (link_body, float(stiffness), float(damping), float(min(force_limit, 3.0e38)),
                                       1 if mode == "acceleration" else 0)))

    def add_tendon(self, link_a, link_b, coef_a, coef_b, rest_length=0.0, stiffness=1e5, damping=0.0):
        self.ops.append(("add_tendon", (link_a, link_b, float(coef_a), float(coef_b), float(rest_length),
                                        float(stiffness), float(damping))))

    # -- actors -----------------------------------------------------------------------
    def add_actor(self, name, kind, p=(0, 0, 0), q=(1, 0, 0, 0), mass=0.0, com=(0, 0, 0), inertia6=(1, 1, 1, 0, 0, 0),
                  linear_damping=0.0, angular_damping=0.05, disable_gravity=False) -> int:
        bid = len(self.body_names)
        self.ops.append(("add_actor", (kind, _pose7(p, q), float(mass), list(map(float, com)),
                                       list(map(float, inertia6)), float(linear_damping), float(angular_damping),
                                       int(bool(disable_gravity)))))
        self.body_names.append(name)
        self.body_masses.append(float(mass))
        self.body_kind.append(kind)
        return bid

    def set_locked_axes(self, body: int, axes) -> None:
        """PhysxRigidDynamicComponent.set_locked_motion_axes: six flags (linear x y z, angular x y z, world axes) of a dynamic actor."""
        mask = sum(1 << k for k, f in enumerate(axes) if f)
        self.ops.append(("set_locked_axes", (int(body), int(mask))))

    # -- shapes -----------------------------------------------------------------------
    def add_shape(self, body, shape_type, p=(0, 0, 0), q=(1, 0, 0, 0), params=(0, 0, 0), verts=None,
                  static_friction=0.3, dynamic_friction=0.3, restitution=0.0, groups=(1, 1, 0, 0),
                  patch_radius=0.0, min_patch_radius=0.0) -> int:
        v = None if verts is None else np.ascontiguousarray(verts, dtype=np.float32).reshape(-1, 3)
        self.ops.append(("add_shape", (body, shape_type, _pose7(p, q), list(map(float, params)), v,
                                       float(static_friction), float(dynamic_friction), float(restitution),
                                       [int(g) & 0xFFFFFFFF for g in groups], float(patch_radius),
                                       float(min_patch_radius))))
        self.nshapes += 1
        return self.nshapes - 1

    def add_visual(self, body, shape_type, p=(0, 0, 0), q=(1, 0, 0, 0), params=(0, 0, 0), verts=None):
        """Visual-only shape (builder.add_box_visual & co. without a collision record, building/actor_builder.py:166-191):
        consumed by render.attach_template_visuals, ignored by the physics."""
        v = None if verts is None else np.ascontiguousarray(verts, dtype=np.float32).reshape(-1, 3)
        self.ops.append(("add_visual", (body, shape_type, _pose7(p, q), list(map(float, params)), v)))

    def disable_collision(self, body_a, body_b):
        self.ops.append(("disable_collision", (body_a, body_b)))

    def declare_env_box(self, shape):
        """The box `shape` (index returned by add_shape) takes half sizes / local position from each env's own record
        (include/msk_physx.h: per-env instances; the reference builds one actor per sub-scene and merges them)."""
        self.ops.append(("declare_env_box", (int(shape),)))

    def declare_env_mass(self, body):
        self.ops.append(("declare_env_mass", (int(body),)))

    def set_body_color(self, body, rgba):
        """RenderMaterial(base_color=rgba) of every visual of `body` (-1: the static scene); consumed by
        render.attach_template_visuals for the Color texture."""
        if not hasattr(self, "body_colors"):
            self.body_colors = {}
        self.body_colors[int(body)] = [float(x) for x in rgba]

    def body_id(self, name) -> int:
        return self.body_names.index(name)


# --------------------------------------------------------------------------------------
# device buffer handles
# --------------------------------------------------------------------------------------
class _DevicePointer:
    """Exposes a raw device pointer through ``__cuda_array_interface__`` (v2) so that
    ``torch.as_tensor`` wraps it without a copy (ROCm builds of torch honour it)."""

    def __init__(self, ptr, shape, typestr="<f4"):
        self.__cuda_array_interface__ = {
            "shape": tuple(int(s) for s in shape), "typestr": typestr, "data": (int(ptr), False),
            "version": 2, "strides": None,
        }


class CudaArrayHandle:
    """What ``px.cuda_rigid_body_data`` & co. return in SAPIEN: an object with ``.torch()``."""

    def __init__(self, ptr, shape, device, host=False):
        self.ptr, self.shape, self.device = ptr, tuple(int(s) for s in shape), device
        if host:
            n = int(np.prod(self.shape))
            arr = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_float)), shape=(n,)).reshape(self.shape)
            self._t = torch.from_numpy(arr)
        else:
            self._t = torch.as_tensor(_DevicePointer(ptr, self.shape), device=device)
            assert self._t.numel() == 0 or self._t.data_ptr() == ptr, "zero-copy wrap of the device buffer failed"

    def torch(self) -> torch.Tensor:
        return self._t


class ContactPairImpulseQuery:
    def __init__(self, qid, handle):
        self.id = qid
        self.cuda_impulses = handle


# --------------------------------------------------------------------------------------
# the system
# --------------------------------------------------------------------------------------
class _TensorHandle:
    """``.torch()`` over a slice of a caller-owned tensor (bind_buffers)"""

    def __init__(self, t):
        self._t = t
        self.shape = tuple(t.shape)

    def torch(self):
        return self._t


def batch_call(engines, op: int, mask: int = 0):
    """``msk_batch``: one native call for the same boundary call on several contexts (structural groups of one scene)."""
    e0 = engines[0]
    arr = getattr(e0, "_batch_arr", None)
    if arr is None or arr[1] != len(engines):
        arr = e0._batch_arr = ((C.c_void_p * len(engines))(*[e.ctx for e in engines]), len(engines))
    e0.lib.check(e0.ctx, e0.lib.batch(arr[0], arr[1], op, mask, e0._stream()), "batch")


_WARNED = set()


class PhysxGpuSystem:
    """All sub-scenes of one process / one GPU (reference: one ``physx.PhysxGpuSystem``)."""

    host_memory = False  # the oracle test double overrides this

    def __init__(self, device: str | torch.device, template: SceneTemplate, num_envs: int,
                 sim_config: Optional[SimConfig] = None, lib: Optional[N.NativeLib] = None):
        self.lib = lib if lib is not None else N.default_lib()
        self.device = torch.device(device)
        self.template = template
        self.num_envs = int(num_envs)
        self.sim_config = sim_config or SimConfig()
        sc = self.sim_config.scene_config
        cfg = N.MskConfig()
        cfg.timestep = 1.0 / self.sim_config.sim_freq
        cfg.gravity = (C.c_float * 3)(*[float(g) for g in sc.gravity])
        cfg.solver_position_iterations = sc.solver_position_iterations
        cfg.solver_velocity_iterations = sc.solver_velocity_iterations
        cfg.contact_offset, cfg.rest_offset = sc.contact_offset, sc.rest_offset
        cfg.bounce_threshold, cfg.sleep_threshold = sc.bounce_threshold, sc.sleep_threshold
        cfg.enable_tgs, cfg.enable_pcm = int(sc.enable_tgs), int(sc.enable_pcm)
        cfg.contact_capacity = int(getattr(sc, "contact_capacity", 0))
        self._timestep = cfg.timestep
        dev_index = self.device.index if (self.device.type == "cuda" and self.device.index is not None) else 0
        self.ctx = self.lib.create(dev_index, C.byref(cfg))
        if not self.ctx:
            raise RuntimeError("msk_create failed")
        self._replay(template)
        self._initialized = False
        self._queries = []

    # -- build ------------------------------------------------------------------------
    def _replay(self, tpl: SceneTemplate):
        L, ctx = self.lib, self.ctx
        for op, a in tpl.ops:
            if op == "add_articulation":
                L.check(ctx, L.add_articulation(ctx, N._fa(a[0], 7)), op)
            elif op == "add_link":
                L.check(ctx, L.add_link(ctx, a[0], a[1], a[2], N._fa(a[3], 7), N._fa(a[4], 7), a[5], a[6], a[7],
                                        N._fa(a[8], 3), N._fa(a[9], 6), a[10], a[11], a[12]), op)
            elif op == "set_articulation_floating":
                L.check(ctx, L.set_articulation_floating(ctx, a[0]), op)
            elif op == "set_drive":
                L.check(ctx, L.set_drive(ctx, *a), op)
            elif op == "add_tendon":
                L.check(ctx, L.add_tendon(ctx, *a), op)
            elif op == "add_actor":
                L.check(ctx, L.add_actor(ctx, a[0], N._fa(a[1], 7), a[2], N._fa(a[3], 3), N._fa(a[4], 6), a[5], a[6], a[7]), op)
            elif op == "add_shape":
                verts = a[4]
                vp = verts.ctypes.data_as(C.POINTER(C.c_float)) if verts is not None else None
                nv = 0 if verts is None else verts.shape[0]
                groups = (C.c_uint32 * 4)(*a[8])
                L.check(ctx, L.add_shape(ctx, a[0], a[1], N._fa(a[2], 7), N._fa(a[3], 3), vp, nv, a[5], a[6], a[7],
                                         groups, a[9], a[10]), op)
            elif op == "disable_collision":
                L.check(ctx, L.disable_collision(ctx, *a), op)
            elif op == "set_locked_axes":
                L.check(ctx, L.set_locked_axes(ctx, *a), op)
            elif op == "add_visual":
                pass  # render.attach_template_visuals
            elif op == "declare_env_box":
                L.check(ctx, L.declare_env_box(ctx, a[0]), op)
            elif op == "declare_env_mass":
                L.check(ctx, L.declare_env_mass(ctx, a[0]), op)
            else:  # pragma: no cover
                raise AssertionError(op)

    @property
    def timestep(self) -> float:
        return self._timestep

    def _stream(self):
        if self.host_memory or self.device.type != "cuda":
            return None
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def gpu_init(self):
        """``px.gpu_init()`` (scene.py:910): freeze the template and allocate device state."""
        L, ctx = self.lib, self.ctx
        L.check(ctx, L.finalize(ctx, self.num_envs), "finalize")
        sizes = (C.c_int32 * 8)()
        L.get_sizes(ctx, sizes)
        self.bodies_per_env, self.arts_per_env, self.max_dof, self.nv, self.nshapes, self.npairs = list(sizes)[:6]

        def handle(buf_id):
            shape = (C.c_int64 * 2)()
            ptr = L.buffer(ctx, buf_id, shape)
            if not ptr:
                raise RuntimeError(f"msk_buffer({buf_id}) returned NULL")
            return CudaArrayHandle(ptr, (shape[0], shape[1]), self.device, host=self.host_memory)

        self.cuda_rigid_body_data = handle(N.BUF_RIGID_BODY_DATA)
        self.cuda_articulation_qpos = handle(N.BUF_ART_QPOS)
        self.cuda_articulation_qvel = handle(N.BUF_ART_QVEL)
        self.cuda_articulation_qacc = handle(N.BUF_ART_QACC)
        self.cuda_articulation_qf = handle(N.BUF_ART_QF)
        self.cuda_articulation_target_qpos = handle(N.BUF_ART_TARGET_QPOS)
        self.cuda_articulation_target_qvel = handle(N.BUF_ART_TARGET_QVEL)
        self.cuda_rigid_body_force = handle(N.BUF_RIGID_BODY_FORCE)      # (rows, 4): Actor.apply_force (structs/actor.py:316-322)
        self.cuda_rigid_body_torque = handle(N.BUF_RIGID_BODY_TORQUE)
        # (N * arts, max_links, 6) once viewed; Articulation.get_link_incoming_joint_forces (structs/articulation.py:596-620)
        self.cuda_articulation_link_incoming_joint_forces = handle(N.BUF_ART_LINK_JOINT_FORCES)
        self._initialized = True
        # what the library accepted without modelling it (include/msk_physx.h msk_warnings): said once per process, kept on the system
        self.backend_warnings = [w for w in (L.warnings(ctx) or b"").decode().splitlines() if w]
        for w in self.backend_warnings:
            if w not in _WARNED:
                _WARNED.add(w)
                import warnings
                warnings.warn(f"maniskill_amd backend: {w}", stacklevel=2)
        # publish the initial state so that the torch-visible buffers are valid
        self.gpu_update_articulation_kinematics()
        self._fetch(N.FETCH_RIGID_DATA | N.FETCH_ART_QPOS | N.FETCH_ART_QVEL | N.FETCH_ART_QACC | N.FETCH_ART_TARGETS)

    def bind_buffers(self, tensors, row0: int, art_row0: int):
        """``msk_bind_buffers``: this context's nine apply / fetch buffers become row ranges of caller-owned tensors (``tensors``: name ->
        2-D float32 tensor shared by several contexts; body rows start at ``row0``, articulation rows at ``art_row0``).  The handles
        ``cuda_*`` are rebuilt over the bound memory."""
        names = ["rigid_body_data", "articulation_qpos", "articulation_qvel", "articulation_qacc", "articulation_qf",
                 "articulation_target_qpos", "articulation_target_qvel", "rigid_body_force", "rigid_body_torque"]
        pitch = int(tensors["articulation_qpos"].shape[1])
        ptrs = (C.c_void_p * 9)()
        for i, nm in enumerate(names):
            t = tensors[nm]
            assert t.dtype == torch.float32 and t.is_contiguous()
            r0 = art_row0 if nm.startswith("articulation") else row0
            ptrs[i] = t.data_ptr() + 4 * r0 * int(t.shape[1])
        self.lib.check(self.ctx, self.lib.bind_buffers(self.ctx, ptrs, pitch), "bind_buffers")
        self._bound = tensors     # keeps the storage alive
        nb, na = self.num_envs * self.bodies_per_env, self.num_envs * max(self.arts_per_env, 0)
        for nm in names:
            t = tensors[nm]
            r0, n = (art_row0, max(na, 1)) if nm.startswith("articulation") else (row0, nb)
            setattr(self, "cuda_" + nm, _TensorHandle(t[r0:r0 + n]))

    def set_scene_offsets(self, offsets):
        """``px.set_scene_offset`` for every sub-scene at once (sapien_env.py:1202)."""
        off = np.ascontiguousarray(offsets, dtype=np.float32).reshape(self.num_envs, 3)
        self.lib.check(self.ctx, self.lib.set_scene_offsets(self.ctx, off.ctypes.data_as(C.POINTER(C.c_float))),
                       "set_scene_offsets")
        self.scene_offsets = torch.from_numpy(off.copy()).to(self.device)

    # -- apply / fetch / step ------------------------------------------------------------
    def _apply(self, mask):
        self.lib.check(self.ctx, self.lib.apply(self.ctx, mask, self._stream()), "apply")

    def _fetch(self, mask):
        self.lib.check(self.ctx, self.lib.fetch(self.ctx, mask, self._stream()), "fetch")

    def gpu_apply_rigid_dynamic_data(self): self._apply(N.APPLY_RIGID_DATA)
    def gpu_apply_rigid_dynamic_force(self): self._apply(N.APPLY_RIGID_FORCE)     # acts during the next step() only
    def gpu_apply_rigid_dynamic_torque(self): self._apply(N.APPLY_RIGID_TORQUE)
    def gpu_apply_articulation_root_pose(self): self._apply(N.APPLY_ART_ROOT_POSE)
    def gpu_apply_articulation_root_velocity(self): self._apply(N.APPLY_ART_ROOT_VELOCITY)   # floating roots; ignored by fixed ones
    def gpu_apply_articulation_qpos(self): self._apply(N.APPLY_ART_QPOS)
    def gpu_apply_articulation_qvel(self): self._apply(N.APPLY_ART_QVEL)
    def gpu_apply_articulation_qf(self): self._apply(N.APPLY_ART_QF)
    def gpu_apply_articulation_target_position(self): self._apply(N.APPLY_ART_TARGET_QPOS)
    def gpu_apply_articulation_target_velocity(self): self._apply(N.APPLY_ART_TARGET_QVEL)

    def apply_force(self, body: int, force):
        """Actor.apply_force (structs/actor.py:316-322): world-frame force at the centre of mass of dynamic actor ``body`` in
        every env, (N, 3) or (3,); committed at once, acts during the next step() only."""
        F = self.cuda_rigid_body_force.torch().view(self.num_envs, self.bodies_per_env, 4)
        F[:, body, :3] = torch.as_tensor(force, dtype=torch.float32, device=F.device)
        self.gpu_apply_rigid_dynamic_force()

    def gpu_apply_all(self):
        """The calls of ManiSkillScene._gpu_apply_all (scene.py:950-966) in one launch -- all but gpu_apply_articulation_root_velocity, which
        only means something for floating roots: the shim issues it as the reference does, the fused envs of this package have fixed bases."""
        self._apply(N.APPLY_RIGID_DATA | N.APPLY_ART_QPOS | N.APPLY_ART_QVEL | N.APPLY_ART_QF |
                    N.APPLY_ART_TARGET_QPOS | N.APPLY_ART_TARGET_QVEL | N.APPLY_ART_ROOT_POSE)

    def gpu_fetch_rigid_dynamic_data(self): self._fetch(N.FETCH_RIGID_DATA)
    def gpu_fetch_articulation_link_pose(self): self._fetch(N.FETCH_RIGID_DATA)
    def gpu_fetch_articulation_link_velocity(self): self._fetch(N.FETCH_RIGID_DATA)
    def gpu_fetch_articulation_qpos(self): self._fetch(N.FETCH_ART_QPOS)
    def gpu_fetch_articulation_qvel(self): self._fetch(N.FETCH_ART_QVEL)
    def gpu_fetch_articulation_qacc(self): self._fetch(N.FETCH_ART_QACC)
    def gpu_fetch_articulation_target_qpos(self): self._fetch(N.FETCH_ART_TARGETS)
    def gpu_fetch_articulation_target_qvel(self): self._fetch(N.FETCH_ART_TARGETS)
    def gpu_fetch_articulation_link_incoming_joint_forces(self): self._fetch(N.FETCH_ART_LINK_FORCES)

    def get_link_incoming_joint_forces(self) -> torch.Tensor:
        """(num_envs * arts_per_env, max_links, 6) [fx fy fz tx ty tz] per link, fetched live like the reference does."""
        self.gpu_fetch_articulation_link_incoming_joint_forces()
        t = self.cuda_articulation_link_incoming_joint_forces.torch()
        return t.view(self.num_envs * max(self.arts_per_env, 1), -1, 6)

    def gpu_fetch_all(self):
        """The eight calls of ManiSkillScene._gpu_fetch_all (scene.py:968-986) in one launch."""
        self._fetch(N.FETCH_RIGID_DATA | N.FETCH_ART_QPOS | N.FETCH_ART_QVEL | N.FETCH_ART_QACC | N.FETCH_ART_TARGETS)

    def gpu_update_articulation_kinematics(self):
        self.lib.check(self.ctx, self.lib.update_kinematics(self.ctx, self._stream()), "update_kinematics")

    def step(self):
        """``px.step()`` (scene.py:379-380): one substep of ``timestep`` for every sub-scene."""
        self.lib.check(self.ctx, self.lib.step(self.ctx, self._stream()), "step")

    def step_n(self, count: int):
        """``count`` consecutive ``px.step()`` calls with nothing in between (include/msk_physx.h msk_step_n): the reference's substep loop of
        one control step (scene.py:379-380) when no controller acts between the substeps.  The library may run contiguous env partitions
        as independent kernel chains (``step_parts``)."""
        self.lib.check(self.ctx, self.lib.step_n(self.ctx, int(count), self._stream()), "step_n")

    @property
    def step_parts(self) -> int:
        """env partitions msk_step runs side by side (1 = none)"""
        return int(self.lib.get_step_parts(self.ctx))

    def set_step_parts(self, parts: int) -> int:
        """re-partition (tuning, tests; synchronises): -> the partition count in force"""
        return int(self.lib.check(self.ctx, self.lib.set_step_parts(self.ctx, int(parts)), "set_step_parts"))

    # -- contact queries -----------------------------------------------------------------
    def gpu_create_contact_pair_impulse_query(self, body_pairs) -> ContactPairImpulseQuery:
        """body_pairs: list of (body_id_a, body_id_b) template body ids (same pair in every env)."""
        flat = np.ascontiguousarray(np.asarray(body_pairs, dtype=np.int32).reshape(-1, 2))
        qid = self.lib.check(self.ctx, self.lib.query_create_pairs(
            self.ctx, flat.ctypes.data_as(C.POINTER(C.c_int32)), flat.shape[0]), "query_create_pairs")
        shape = (C.c_int64 * 2)()
        ptr = self.lib.query_buffer(self.ctx, qid, shape)
        q = ContactPairImpulseQuery(qid, CudaArrayHandle(ptr, (shape[0], shape[1]), self.device, host=self.host_memory))
        self._queries.append(q)
        return q

    def gpu_create_contact_body_impulse_query(self, bodies) -> ContactPairImpulseQuery:
        """bodies: template body ids; the query returns the net contact impulse on each of them (structs/base.py:116-136)."""
        flat = np.ascontiguousarray(np.asarray(bodies, dtype=np.int32).reshape(-1))
        qid = self.lib.check(self.ctx, self.lib.query_create_bodies(
            self.ctx, flat.ctypes.data_as(C.POINTER(C.c_int32)), flat.shape[0]), "query_create_bodies")
        shape = (C.c_int64 * 2)()
        ptr = self.lib.query_buffer(self.ctx, qid, shape)
        q = ContactPairImpulseQuery(qid, CudaArrayHandle(ptr, (shape[0], shape[1]), self.device, host=self.host_memory))
        self._queries.append(q)
        return q

    def gpu_query_contact_body_impulses(self, query: ContactPairImpulseQuery):
        self.lib.check(self.ctx, self.lib.query_run(self.ctx, query.id, self._stream()), "query_run")

    def gpu_query_contact_pair_impulses(self, query: ContactPairImpulseQuery):
        self.lib.check(self.ctx, self.lib.query_run(self.ctx, query.id, self._stream()), "query_run")

    # -- measurement -------------------------------------------------------------------------
    def timing_enable(self, max_steps: int):
        """Arm per-kernel HIP-event timing for the next ``max_steps`` calls of ``step()``."""
        self.lib.check(self.ctx, self.lib.timing_enable(self.ctx, int(max_steps)), "timing_enable")

    def timing_read(self):
        """{kernel name: (total_ms, launches)} of the armed steps (waits for the events)."""
        out = {}
        for name, slot in N.KERNEL_SLOTS.items():
            ms, n = C.c_double(), C.c_int32()
            self.lib.check(self.ctx, self.lib.timing_read(self.ctx, slot, C.byref(ms), C.byref(n)), "timing_read")
            out[name] = (ms.value, n.value)
        return out

    # -- inspection (parity tests) -------------------------------------------------------
    def get_contacts(self, env: int, max_points: int = 128):
        ids = (C.c_int32 * (3 * max_points))()
        vals = (C.c_float * (8 * max_points))()
        n = self.lib.check(self.ctx, self.lib.get_contacts(self.ctx, env, ids, vals, max_points), "get_contacts")
        n = min(n, max_points)
        return (np.array(ids[: 3 * n], dtype=np.int32).reshape(n, 3),
                np.array(vals[: 8 * n], dtype=np.float32).reshape(n, 8))

    def set_env_boxes(self, shape, half_sizes, local_pos=None):
        """(num_envs, 3) half sizes and optionally (num_envs, 3) local positions of a declared box shape."""
        hs = np.ascontiguousarray(half_sizes, dtype=np.float32).reshape(self.num_envs, 3)
        lp = None if local_pos is None else np.ascontiguousarray(local_pos, dtype=np.float32).reshape(self.num_envs, 3)
        fp = C.POINTER(C.c_float)
        self.lib.check(self.ctx, self.lib.set_env_boxes(self.ctx, int(shape), hs.ctypes.data_as(fp), None if lp is None else lp.ctypes.data_as(fp)),
                       "set_env_boxes")

    def set_env_masses(self, body, mass, principal_inertia):
        m = np.ascontiguousarray(mass, dtype=np.float32).reshape(self.num_envs)
        I = np.ascontiguousarray(principal_inertia, dtype=np.float32).reshape(self.num_envs, 3)
        fp = C.POINTER(C.c_float)
        self.lib.check(self.ctx, self.lib.set_env_masses(self.ctx, int(body), m.ctypes.data_as(fp), I.ctypes.data_as(fp)), "set_env_masses")

    def set_solver_classes(self, caps):
        """Scheduling only (include/msk_physx.h): largest block counts of solver classes 0..2; negative empties a class."""
        arr = (C.c_int32 * 3)(*[int(x) for x in caps])
        self.lib.check(self.ctx, self.lib.set_solver_classes(self.ctx, arr), "set_solver_classes")

    def get_solver_class_counts(self) -> np.ndarray:
        out = (C.c_int32 * 5)()
        self.lib.check(self.ctx, self.lib.get_solver_class_counts(self.ctx, out), "get_solver_class_counts")
        return np.array(list(out), dtype=np.int32)

    def get_env_contact_counts(self) -> np.ndarray:
        """(num_envs,) int32: contact points solved per env in the last step (synchronises)."""
        out = np.zeros(self.num_envs, dtype=np.int32)
        self.lib.check(self.ctx, self.lib.get_env_contact_counts(self.ctx, out.ctypes.data_as(C.POINTER(C.c_int32))),
                       "get_env_contact_counts")
        return out

    def compute_ik_delta(self, ee_body: int, root_body: int, joint_links, delta_pose: torch.Tensor, damping: float = 1e-4, alpha: float = 1.0,
                         commit_targets: bool = False) -> torch.Tensor:
        """``Kinematics.compute_ik(delta, qpos, is_delta_pose=True)`` on the device (agents/controllers/utils/kinematics.py:185-245) for the
        chain of the joints whose child links are ``joint_links`` (body ids, base to tip) between ``root_body`` and ``ee_body``:
        (num_envs, len(joint_links)) joint targets.
        ``delta_pose`` (num_envs, 6) float32 on the engine's device, expressed in the root link's frame."""
        d = N.MskIkDesc()
        dofs = list(joint_links)
        d.ee_body, d.root_body, d.njoints = int(ee_body), int(root_body), len(dofs)
        if not 1 <= len(dofs) <= N.IK_MAX_JOINTS:
            raise ValueError(f"compute_ik_delta: 1 .. {N.IK_MAX_JOINTS} controlled joints")
        for k, q in enumerate(dofs):
            d.joint_links[k] = int(q)
        d.damping, d.alpha = float(damping), float(alpha)
        delta_pose = delta_pose.contiguous()
        assert delta_pose.shape == (self.num_envs, 6) and delta_pose.dtype == torch.float32
        out = torch.empty(self.num_envs, len(dofs), dtype=torch.float32, device=delta_pose.device)
        self.lib.check(self.ctx, self.lib.compute_ik_delta(self.ctx, C.byref(d), C.c_void_p(delta_pose.data_ptr()), C.c_void_p(out.data_ptr()),
                                                           1 if commit_targets else 0, self._stream()), "compute_ik_delta")
        return out

    def get_overflow(self) -> int:
        """1 if any env exceeded its contact capacity since gpu_init (synchronises)."""
        sizes = (C.c_int32 * 8)()
        self.lib.check(self.ctx, self.lib.get_sizes(self.ctx, sizes), "get_sizes")
        return int(sizes[7])

    def close(self):
        if getattr(self, "ctx", None):
            self.lib.destroy(self.ctx)
            self.ctx = None

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass
