"""PhysxGpuSystem / PhysxCpuSystem of the shim and the scene compiler behind ``gpu_init()``.

ManiSkill builds every sub-scene separately in Python (``for scene_idx in scene_idxs: ... sub_scene.add_entity(entity)``,
mani_skill/utils/building/actor_builder.py:234-245, articulation_builder.py:143-205).  The C-ABI library wants ONE scene template
plus per-sub-scene instance records (include/msk_physx.h), so ``gpu_init()`` (envs/scene.py:902-948) compiles the object graph:
sub-scene 0 becomes the template, every other sub-scene is checked against it (same bodies, joints, shapes, materials,
collision groups); boxes whose sizes and bodies whose masses differ between sub-scenes become ``msk_declare_env_box`` /
``msk_declare_env_mass`` instances (PegInsertionSide-v1, envs/tasks/tabletop/peg_insertion_side.py:133-187).

Buffers: rows of ``cuda_rigid_body_data`` are sub-scene-major (row = env * bodies_per_env + template body id), positions are
relative to the sub-scene (docs/source/user_guide/concepts/gpu_simulation.md:9); ``gpu_pose_index`` / ``gpu_index`` hand those
rows out, which is all the reference relies on (utils/structs/base.py:103-109, articulation.py:266-270).
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np
import os

import torch

from ._pose import Pose

_JOINT = {"fixed": 0, "revolute": 1, "revolute_unwrapped": 1, "prismatic": 2}
_SYNC_FETCH = 1 | 2 | 4 | 8 | 16


def _pose7(pose: Pose):
    return tuple(float(x) for x in pose._p) + tuple(float(x) for x in pose._q)


# this process's shard of a larger env set (maniskill_amd.dist): global index of its first sub-scene, its sub-scene count, the total
_SHARD = {"offset": 0, "local": None, "total": None}


_GC_RELAXED = {"owner": None, "threshold": None}     # id() of the system that raised the collector's threshold, and what it was


def _gc_relax(system):
    import gc
    if os.environ.get("MSK_SHIM_KEEP_GC") is not None or not gc.isenabled():
        return
    if _GC_RELAXED["owner"] is None:              # (a stale owner -- a build that failed -- is simply taken over: the saved threshold stays)
        _GC_RELAXED["threshold"] = gc.get_threshold()
    _GC_RELAXED["owner"] = id(system)
    gc.set_threshold(1_000_000, 50, 50)
    system._gc_paused = True


def _gc_restore(system):
    if getattr(system, "_gc_paused", False):
        system._gc_paused = False
        if _GC_RELAXED["owner"] == id(system):
            import gc
            gc.set_threshold(*_GC_RELAXED["threshold"])
            _GC_RELAXED["owner"] = None


def set_shard(env_index_offset, local_envs, total_envs):
    """None for total_envs switches the global layout off again."""
    _SHARD.update(offset=int(env_index_offset), local=None if local_envs is None else int(local_envs), total=None if total_envs is None else int(total_envs))
    _SHARD.pop("spacing", None)


class _Handle3D:
    """cuda_articulation_link_incoming_joint_forces: (num_articulations, max_links, 6) view of the library's 2-D buffer."""

    def __init__(self, handle, shape):
        self._h, self._shape = handle, shape

    def torch(self):
        return self._h.torch().view(*self._shape)


class _GatherQuery:
    """A contact-impulse query: one engine query per structural group, the listed rows gathered on demand."""

    def __init__(self, parts, direct_handle, nrows, device):
        self._parts, self._direct = parts, direct_handle
        self._out = None if direct_handle is not None else torch.zeros(nrows, 3, dtype=torch.float32, device=device)
        self.cuda_impulses = direct_handle if direct_handle is not None else self

    def _run(self):
        for g, q, rows, idx in self._parts:
            g.engine.gpu_query_contact_pair_impulses(q)
            if rows is not None:
                self._out[rows] = q.cuda_impulses.torch().index_select(0, idx)

    def torch(self):
        return self._out


class _Group:
    """Sub-scenes that share one scene template: one context of the C-ABI library."""

    def __init__(self, envs, engine, template):
        self.envs, self.engine, self.template = list(envs), engine, template
        self.n, self.nb, self.na, self.max_dof = len(envs), engine.bodies_per_env, engine.arts_per_env, engine.max_dof
        self.base = self.abase = 0        # first row of the group in the unified body / articulation buffers (_MultiBuffers)
        self.npassive, self.pbase = 0, 0  # pose-only actors per sub-scene and their first row (behind every engine row)


class _Tensor:
    def __init__(self, t):
        self._t = t

    def torch(self):
        return self._t


class _MultiBuffers:
    """Structurally different sub-scenes (other articulations, meshes, body counts per sub-scene: e.g. a different cabinet in every
    sub-scene of OpenCabinetDrawer-v1, envs/tasks/mobile_manipulation/open_cabinet_drawer.py:128-177) run as one library context
    per structural group.  ManiSkill still sees ONE ``cuda_rigid_body_data`` / ``cuda_articulation_*`` set: unified torch buffers in
    which every group owns a contiguous row range (articulation rows padded to the largest dof count, as SAPIEN pads them);
    ``gpu_fetch_*`` copies the groups' buffers in, ``gpu_apply_*`` copies them out before the group's own apply."""

    BODY = ("rigid_body_data", "rigid_body_force", "rigid_body_torque")
    ART = ("qpos", "qvel", "qacc", "qf", "target_qpos", "target_qvel")

    def __init__(self, px, device):
        self.px = px
        gs = px._groups
        rows = arows = 0
        for g in gs:
            g.base, g.abase = rows, arows
            rows += g.n * g.nb
            arows += g.n * max(g.na, 0)
        for g in gs:                                   # pose-only actors: rows behind the engines' ranges
            g.pbase = rows
            rows += g.n * g.npassive
        self.max_dof = max(g.max_dof for g in gs)
        self.max_links = 1
        for g in gs:
            lf = g.engine.cuda_articulation_link_incoming_joint_forces
            self.max_links = max(self.max_links, lf.shape[0] // max(g.n * max(g.na, 1), 1))
        z = lambda *shape: torch.zeros(*shape, dtype=torch.float32, device=device)   # noqa: E731
        self.t = dict(rigid_body_data=z(rows, 13), rigid_body_force=z(rows, 4), rigid_body_torque=z(rows, 4),
                      link_forces=z(max(arows, 1), self.max_links, 6))
        for name in self.ART:
            self.t[name] = z(max(arows, 1), max(self.max_dof, 1))
        px.cuda_rigid_body_data = _Tensor(self.t["rigid_body_data"])
        px.cuda_rigid_body_force = _Tensor(self.t["rigid_body_force"])
        px.cuda_rigid_body_torque = _Tensor(self.t["rigid_body_torque"])
        for name in self.ART:
            setattr(px, "cuda_articulation_" + name, _Tensor(self.t[name]))
        px.cuda_articulation_link_incoming_joint_forces = _Tensor(self.t["link_forces"])
        # every group's context reads and writes ITS ROWS of these tensors directly (msk_bind_buffers): no copies on apply / fetch
        shared = {("" if name in self.BODY else "articulation_") + name: self.t[name] for name in self.BODY + self.ART}
        for g in gs:
            g.engine.bind_buffers(shared, g.base, g.abase)
        self.fetch_all()

    def fetch_all(self):
        self.px._batch("gpu_fetch_all")

    def pull_link_forces(self):
        for g in self.px._groups:
            if g.na == 0:
                continue
            g.engine.gpu_fetch_articulation_link_incoming_joint_forces()
            lf = g.engine.cuda_articulation_link_incoming_joint_forces.torch().view(g.n * g.na, -1, 6)
            self.t["link_forces"][g.abase:g.abase + g.n * g.na, :lf.shape[1]] = lf


class PhysxSystem:
    def __init__(self):
        from . import physx as P
        self._P = P
        self._scenes = []
        self._scene_idx = {}
        self._offsets = {}
        self._initialized = False
        self._components = []           # every PhysxRigidBaseComponent in registration order
        self._timestep = 0.01
        # snapshot of the module-level configuration, as SAPIEN takes it when the system is created
        self._cfg = dict(scene=dict(P._config["scene"]), body=dict(P._config["body"]), shape=dict(P._config["shape"]))
        self._engine = None
        self._body_rows = None
        self._shard = dict(_SHARD)      # the shard this system was created under (maniskill_amd.dist scopes set_shard to its own gym.make)
        # Between here and the engine start, the caller builds one entity tree per sub-scene: millions of long-lived objects at 16k
        # sub-scenes, which the cyclic collector re-traverses on every generation-2 pass (a quarter of the build time).  The collector
        # is NOT switched off for that (a build that raises, or an engine that is never started, would leave it off for the life of the
        # process -- the system then sits in reference cycles, so __del__ could not even run): its generation-0 threshold is raised
        # 1500-fold while the scene is built, so a 16k build sees a handful of passes instead of thousands, and put back when the engine
        # starts, when this object dies, or when the next system is created.  The survivors are frozen out of later passes (_start_engine).
        self._gc_paused = False
        _gc_relax(self)

    def __del__(self):
        _gc_restore(self)

    # -- scenes ---------------------------------------------------------------------------------------------------
    def _register_scene(self, scene):
        self._scene_idx[id(scene)] = len(self._scenes)
        self._scenes.append(scene)

    def _scene_index(self, scene) -> int:
        return self._scene_idx[id(scene)]

    def set_scene_offset(self, scene, offset):
        """ManiSkill lays its sub-scenes out on a square grid from the LOCAL sub-scene index (envs/sapien_env.py:1190-1208).  When this
        process holds one shard of a larger env set (maniskill_amd.dist.make_sharded_gym_env -> set_shard), the k-th call gets the grid cell
        of the GLOBAL index instead -- same formula, same spacing --, so that every sub-scene has the same offset (and the same fp32 rounding
        of position + offset) whatever the partition: results do not depend on the number of ranks."""
        off = np.array(offset, dtype=np.float32).reshape(3)
        sh = self._shard
        k = self._scene_idx.get(id(scene), len(self._offsets))     # the scene's own index (sub-scenes register in index order)
        if sh["total"] is not None and k < sh["local"]:
            L_loc = int(np.ceil(np.sqrt(sh["local"])))
            x_loc, y_loc = k % L_loc - L_loc // 2, k // L_loc - L_loc // 2
            if x_loc != 0:                           # the spacing ManiSkill used (sim_config.spacing), read off its own offset
                sh["spacing"] = float(off[0]) / x_loc
            elif y_loc != 0:
                sh["spacing"] = float(off[1]) / y_loc
            spacing = sh.get("spacing", 5.0)         # (a one-env shard never shows it: SimConfig's default)
            g = sh["offset"] + k
            L = int(np.ceil(np.sqrt(sh["total"])))
            off = np.array([(g % L - L // 2) * spacing, (g // L - L // 2) * spacing, off[2]], dtype=np.float32)
        self._offsets[id(scene)] = off

    def get_scene_offset(self, scene):
        return self._offsets.get(id(scene), np.zeros(3, dtype=np.float32)).copy()

    def _register_component(self, comp):
        if self._initialized:
            raise RuntimeError("cannot add physical components after the simulation was initialised")
        self._components.append(comp)

    def _unregister_component(self, comp):
        if self._initialized:
            raise RuntimeError("cannot remove physical components after the simulation was initialised")
        self._components.remove(comp)

    @property
    def rigid_dynamic_components(self):
        return [c for c in self._components if isinstance(c, self._P.PhysxRigidDynamicComponent)]

    @property
    def rigid_static_components(self):
        return [c for c in self._components if isinstance(c, self._P.PhysxRigidStaticComponent)]

    @property
    def articulation_link_components(self):
        return [c for c in self._components if isinstance(c, self._P.PhysxArticulationLinkComponent)]

    def get_rigid_dynamic_components(self):
        return self.rigid_dynamic_components

    def get_rigid_static_components(self):
        return self.rigid_static_components

    def get_articulation_link_components(self):
        return self.articulation_link_components

    @property
    def timestep(self):
        return self._timestep

    @timestep.setter
    def timestep(self, dt):
        if self._initialized and abs(float(dt) - self._timestep) > 1e-12:
            raise RuntimeError("timestep cannot be changed after the simulation was initialised")
        self._timestep = float(dt)

    def get_timestep(self):
        return self._timestep

    def set_timestep(self, dt):
        self.timestep = dt

    def get_config(self):
        return self._cfg

    def _live(self) -> bool:
        """True once component accessors must go to the simulator's state (the CPU-style system starts lazily, see there)."""
        return self._initialized

    # =============================================================================================================
    # compiler
    # =============================================================================================================
    # Signatures are compared, never decoded: poses enter as the bytes of their float64 arrays (seven float() calls per pose were the bulk
    # of a 16k-sub-scene compile).  One pass yields both forms: `relaxed` (what must agree for two sub-scenes to share a compiled
    # template: box sizes, box positions and masses may differ inside a group) and `strict` (everything).
    @staticmethod
    def _pose_key(pose: Pose):
        return pose._p.tobytes() + pose._q.tobytes()

    def _comp_sig(self, c):
        """-> (relaxed, strict) signature of a component"""
        P = self._P
        if isinstance(c, P.PhysxRigidStaticComponent):
            fold = c.entity._pose
            sh = [s._sig(fold) for s in c.collision_shapes if not isinstance(s, P.PhysxCollisionShapePlane)]   # planes are global, see _compile
            return ("static", tuple(x[0] for x in sh)), ("static", tuple(x[1] for x in sh))
        sh = [s._sig() for s in c.collision_shapes]
        m, com, I6 = c._mass_tensor()
        massp = (m, com.tobytes(), tuple(I6))
        head = (c.linear_damping, c.angular_damping, bool(c.disable_gravity))
        if isinstance(c, P.PhysxArticulationLinkComponent):
            j = c.joint
            head = ("link", c.index, -1 if c.parent is None else c.parent.index, j._type, self._pose_key(j.pose_in_parent),
                    self._pose_key(j.pose_in_child), j._limits.tobytes(), j.stiffness, j.damping, j.force_limit,
                    j.drive_mode, j.friction, j._armature) + head
        else:
            head = ("kinematic" if c.kinematic else "dynamic", tuple(c.locked_motion_axes)) + head
        return head + (tuple(x[0] for x in sh),), head + (massp, tuple(x[1] for x in sh))

    def _compile(self):
        """Partition the sub-scenes into structural groups (same bodies, joints, meshes, materials, collision groups; box sizes and
        masses may differ inside a group) and compile one scene template per group.  -> list of (envs, SceneTemplate, instances)."""
        n_env = len(self._scenes)
        per_env = [[] for _ in range(n_env)]
        for c in self._components:
            per_env[c._env].append(c)
        self._per_env, self._n_env = per_env, n_env
        groups, order = {}, []
        self._strict_sig = []
        for e in range(n_env):
            sigs = [self._comp_sig(c) for c in per_env[e]]
            key = tuple(x[0] for x in sigs)
            self._strict_sig.append(tuple(x[1] for x in sigs))
            if key not in groups:
                groups[key] = []
                order.append(key)
            groups[key].append(e)
        self._shape_owner_of_group = []
        out = []
        for gi, key in enumerate(order):
            envs = groups[key]
            tpl, inst = self._compile_group(envs, gi)
            out.append((envs, tpl, inst))
        return out

    def _compile_group(self, envs, gi):
        """-> (SceneTemplate, per-env instance data) of the sub-scenes `envs`; sets _body_id / _art_id / _group / _lenv."""
        from maniskill_amd import _native as N
        from maniskill_amd.physx import SceneTemplate
        P = self._P
        per_env = [self._per_env[e] for e in envs]
        n_env = len(envs)
        env0 = per_env[0]
        sig0 = self._strict_sig[envs[0]]
        hetero = [e for e in range(1, n_env) if self._strict_sig[envs[e]] != sig0]

        # ---- global planes (a plane on a static actor of ANY sub-scene is one infinite plane of the whole PhysX scene;
        #      the reference therefore attaches it in the first sub-scene only, actor_builder.py:76-90) ---------------------
        planes = {}
        for c in self._components:
            if isinstance(c, P.PhysxRigidStaticComponent):
                off = self._offsets.get(id(self._scenes[c._env]), np.zeros(3))
                for s in c.collision_shapes:
                    if isinstance(s, P.PhysxCollisionShapePlane):
                        gp = c.entity._pose * s._local_pose
                        n = gp.to_transformation_matrix()[:3, 0]
                        d = float(np.dot(n, gp.p + off))          # plane: n . x = d in world coordinates
                        # in a sub-scene frame x_local = x - offset: d_local = d - n . offset; identical for all sub-scenes only
                        # if the offsets lie in the plane (ManiSkill's grid is horizontal, the ground normal is +z)
                        key = (tuple(np.round(n, 4)), round(d, 4))
                        planes.setdefault(key, (gp, s, n, d, off))
        tpl = SceneTemplate()
        for key, (gp, s, n, d, off) in planes.items():
            for sc in self._scenes:
                o = self._offsets.get(id(sc), np.zeros(3))
                # the normal comes out of a float32 quaternion (|n . grid direction| ~ 1e-6): over a few hundred metres of grid that is a
                # fraction of a millimetre, not a tilted plane -- the criterion is the angle, not the absolute distance
                if abs(float(np.dot(n, o - off))) > 1e-4 + 2e-5 * float(np.linalg.norm(o - off)):
                    raise RuntimeError("a static plane is not parallel to the sub-scene grid: cannot be shared by all sub-scenes")
        self._shape_owner = []       # template shape index -> the PhysxCollisionShape of the group's first sub-scene (or the global plane)
        self._shape_owner_of_group.append(self._shape_owner)
        placed_planes = set()

        def add_plane(key):
            gp, s, n, d, off = planes[key]
            mat = s.physical_material     # offsets are in-plane: the plane is the same in every sub-scene frame
            tpl.add_shape(-1, N.SHAPE_PLANE, gp._p, gp._q, (0, 0, 0), None, mat.static_friction, mat.dynamic_friction, mat.restitution,
                          s._groups, s.patch_radius, s.min_patch_radius)
            self._shape_owner.append(s)
            placed_planes.add(key)

        def plane_key(comp, s):
            off = self._offsets.get(id(self._scenes[comp._env]), np.zeros(3))
            gp = comp.entity._pose * s._local_pose
            n = gp.to_transformation_matrix()[:3, 0]
            return (tuple(np.round(n, 4)), round(float(np.dot(n, gp.p + off)), 4))

        # ---- template from sub-scene 0 ----------------------------------------------------------------------------------
        arts0, art_ids = [], {}
        body_ids = {}
        shape_ids = {}       # id(shape of env 0) -> template shape index

        def add_shapes(comp, body, fold):
            for s in comp.collision_shapes:
                if isinstance(s, P.PhysxCollisionShapePlane):
                    if body >= 0:
                        raise RuntimeError("plane collision shapes must belong to static actors")
                    k = plane_key(comp, s)
                    if k not in placed_planes:
                        add_plane(k)
                    continue
                lp = s._local_pose if fold is None else fold * s._local_pose
                mat = s.physical_material
                kw = dict(static_friction=mat.static_friction, dynamic_friction=mat.dynamic_friction, restitution=mat.restitution,
                          groups=s._groups, patch_radius=s.patch_radius, min_patch_radius=s.min_patch_radius)
                subs = [s]
                if isinstance(s, P.PhysxCollisionShapeTriangleMesh):
                    subs = s._hulls
                for sub in subs:
                    if isinstance(sub, P.PhysxCollisionShapeBox):
                        sid = tpl.add_shape(body, N.SHAPE_BOX, lp._p, lp._q, sub._half, None, **kw)
                    elif isinstance(sub, P.PhysxCollisionShapeConvexMesh):
                        sid = tpl.add_shape(body, N.SHAPE_CONVEX, lp._p, lp._q, (0, 0, 0), sub._scaled_vertices, **kw)
                    elif isinstance(sub, P.PhysxCollisionShapeSphere):
                        sid = tpl.add_shape(body, N.SHAPE_SPHERE, lp._p, lp._q, (sub.radius, 0, 0), None, **kw)
                    elif isinstance(sub, P.PhysxCollisionShapeCapsule):
                        sid = tpl.add_shape(body, N.SHAPE_CAPSULE, lp._p, lp._q, (sub.radius, sub.half_length, 0), None, **kw)
                    elif isinstance(sub, P.PhysxCollisionShapeCylinder):
                        sid = tpl.add_shape(body, N.SHAPE_CYLINDER, lp._p, lp._q, (sub.radius, sub.half_length, 0), None, **kw)
                    else:
                        raise RuntimeError(f"collision shape {type(sub).__name__} is not supported")
                    shape_ids.setdefault(id(s), sid)
                    self._shape_owner.append(s)

        # Pose-only actors.  The engine maps a lane to a body (63 per sub-scene).  Tasks that keep hundreds of kinematic actors without any
        # collision shape around only to move them (the 1010 "dots" of envs/tasks/drawing/draw.py:119-149) do not need the engine for
        # those: when the sub-scene does not fit otherwise, such actors -- the last ones first -- get a row of cuda_rigid_body_data that
        # set_pose / pose read and write, and nothing else.  (They are not drawn at their moved poses: the rasteriser takes body poses
        # from the engine)
        passive = {}
        movable = [c for c in env0 if not isinstance(c, P.PhysxRigidStaticComponent)]
        cap = 63
        if len(movable) > cap:
            cand = [c for c in movable if isinstance(c, P.PhysxRigidDynamicComponent) and not isinstance(c, P.PhysxArticulationLinkComponent)
                    and c.kinematic and not c.collision_shapes]
            for c in reversed(cand):
                if len(movable) - len(passive) <= cap:
                    break
                passive[id(c)] = 0
            for k, c in enumerate([c for c in env0 if id(c) in passive]):
                passive[id(c)] = k
        self._npassive_of_group = getattr(self, "_npassive_of_group", {})
        self._npassive_of_group[gi] = len(passive)
        # non-convex scenery is cut into convex pieces (physx.PhysxCollisionShapeTriangleMesh): as many as the template's 64 shapes allow
        tms = [s for c in env0 for s in c.collision_shapes if isinstance(s, P.PhysxCollisionShapeTriangleMesh)]
        others = sum(1 for c in env0 for s in c.collision_shapes if not isinstance(s, (P.PhysxCollisionShapeTriangleMesh, P.PhysxCollisionShapePlane)))
        parts = 16
        while tms and parts > 1 and others + len(planes) + sum(len(s._hulls) for s in tms) > 64:
            parts //= 2
            for s in tms:
                if s._max_parts > parts:
                    s._cook(parts)
        shapes_of = ([], [], [])
        for c in env0:
            if isinstance(c, P.PhysxRigidStaticComponent):
                shapes_of[0].append((c, -1, c.entity._pose))
                body_ids[id(c)] = -1
            elif isinstance(c, P.PhysxArticulationLinkComponent):
                art = c.articulation
                if id(art) not in art_ids:
                    rp = art.root.entity._pose
                    # fix_root_link = False (articulation_builder.py:212): the root link gets six coordinates of its own (msk_set_articulation_floating)
                    art_ids[id(art)] = tpl.add_articulation(art.name, rp._p, rp._q, floating=art.root.joint._type != "fixed")
                    arts0.append(art)
                a = art_ids[id(art)]
                j = c.joint
                if c.parent is not None and j._type not in _JOINT:
                    raise RuntimeError(f"joint {j.name!r}: type {j._type!r} is not supported")
                m, com, I6 = c._mass_tensor()
                lim = j._limits.reshape(-1) if j.dof else (-np.inf, np.inf)
                if j._type == "revolute_unwrapped":
                    lim = (-np.inf, np.inf)
                bid = tpl.add_link(a, c.name, -1 if c.parent is None else body_ids[id(c.parent)],
                                   _JOINT.get(j._type, 0) if c.parent is not None else 0, j.name, list(_pose7(j.pose_in_parent)),
                                   list(_pose7(j.pose_in_child)), (lim[0], lim[1]), m, com, I6, c.disable_gravity, j._armature, j.friction)
                body_ids[id(c)] = bid
                if j.dof:
                    tpl.set_drive(bid, j.stiffness, j.damping, j.force_limit, j.drive_mode)
                shapes_of[2].append((c, bid, None))
            elif id(c) in passive:
                body_ids[id(c)] = -2 - passive[id(c)]      # pose-only: a row of cuda_rigid_body_data outside the engine's range
            else:
                m, com, I6 = c._mass_tensor()
                ep = c.entity._pose
                kind = N.BODY_KINEMATIC if c.kinematic else N.BODY_DYNAMIC
                bid = tpl.add_actor(c.entity.name, kind, ep._p, ep._q, 0.0 if c.kinematic else m, com, I6, c.linear_damping,
                                    c.angular_damping, c.disable_gravity)
                if any(c.locked_motion_axes) and not c.kinematic:
                    tpl.set_locked_axes(bid, c.locked_motion_axes)
                body_ids[id(c)] = bid
                shapes_of[1].append((c, bid, None))
        # Shape order = the order of the candidate-pair table (pairs are enumerated sa < sb) = the order in which an env's contact
        # capacity is handed out: scenery first, then free actors, then articulation links.  What an env over its capacity loses are
        # then the link-against-link pairs of its own robot (TriFinger: a dozen speculative contacts between the three fingers), not
        # the contacts that hold the task's object on the ground.
        for group in shapes_of:
            for comp, bid, fold in group:
                add_shapes(comp, bid, fold)
        for key in planes:
            if key not in placed_planes:
                add_plane(key)
        for art in arts0:
            for t in art._tendons:
                chain, coef = t["chain"], t["coef"]
                act = [(l, k) for l, k in zip(chain, coef) if k != 0.0]
                if len(act) != 2:
                    raise RuntimeError("fixed tendons are supported over exactly two driven joints (URDF mimic joints)")
                (la, ka), (lb, kb) = act
                tpl.add_tendon(body_ids[id(la)], body_ids[id(lb)], ka, kb, t["rest_length"], t["stiffness"], t["damping"])
            by_name = {l.entity.name: l for l in art.links if l.entity is not None}
            for l in art.links:
                for other in sorted(getattr(l, "_srdf_disabled", ())):
                    if other in by_name:
                        tpl.disable_collision(body_ids[id(l)], body_ids[id(by_name[other])])

        # ---- per-env instances ---------------------------------------------------------------------------------------------
        inst = dict(boxes={}, masses={})
        if hetero:
            for k, c0 in enumerate(env0):
                if isinstance(c0, P.PhysxRigidStaticComponent):
                    fold = lambda comp: comp.entity._pose          # noqa: E731
                else:
                    fold = lambda comp: None                       # noqa: E731
                for si, s0 in enumerate(c0.collision_shapes):
                    if not isinstance(s0, P.PhysxCollisionShapeBox):
                        continue
                    hs = np.stack([per_env[e][k].collision_shapes[si]._half for e in range(n_env)])
                    lps = []
                    for e in range(n_env):
                        comp = per_env[e][k]
                        f = fold(comp)
                        lp = comp.collision_shapes[si]._local_pose
                        lps.append((lp if f is None else f * lp)._p)
                    lps = np.stack(lps)
                    if np.ptp(hs, axis=0).max() > 0 or np.ptp(lps, axis=0).max() > 0:
                        sid = shape_ids[id(s0)]
                        tpl.declare_env_box(sid)
                        inst["boxes"][sid] = (hs.astype(np.float32), lps.astype(np.float32))
                if isinstance(c0, P.PhysxRigidDynamicComponent) and not c0.kinematic:
                    mt = [per_env[e][k]._mass_tensor() for e in range(n_env)]
                    ms = np.array([t[0] for t in mt], dtype=np.float32)
                    I6 = np.array([t[2] for t in mt], dtype=np.float32)
                    coms = np.array([t[1] for t in mt], dtype=np.float32)
                    if np.ptp(ms) > 0 or np.ptp(I6, axis=0).max() > 0:
                        if np.abs(coms).max() > 1e-7 or np.abs(I6[:, 3:]).max() > 1e-9 * max(float(np.abs(I6[:, :3]).max()), 1e-30):
                            raise RuntimeError("per-sub-scene masses need the centre of mass at the body origin and a diagonal inertia")
                        bid = body_ids[id(c0)]
                        tpl.declare_env_mass(bid)
                        inst["masses"][bid] = (ms, I6[:, :3].copy())

        # ---- ids on every component -------------------------------------------------------------------------------------------
        for e in range(n_env):
            for k, c in enumerate(per_env[e]):
                c._body_id = body_ids[id(env0[k])]
                c._group, c._lenv = gi, e
                if isinstance(c, P.PhysxArticulationLinkComponent):
                    art = c.articulation
                    art._art_id = art_ids[id(env0[k].articulation)]
                    art._group, art._lenv = gi, e
        self._env_box_shapes_of_group = getattr(self, "_env_box_shapes_of_group", {})
        boxes = {}
        for c0 in env0:
            for s0 in c0.collision_shapes:
                sid = shape_ids.get(id(s0))
                if sid is not None and sid in inst["boxes"]:
                    boxes.setdefault(body_ids[id(c0)], []).append((sid, np.asarray(s0._half, dtype=np.float32)))
        self._env_box_shapes_of_group[gi] = boxes
        return tpl, inst

    # =============================================================================================================
    # engine
    # =============================================================================================================
    def _start_engine(self, torch_device, lib, host_memory):
        from maniskill_amd import physx as E
        compiled = self._compile()
        # body parameters SAPIEN takes and this backend has no counterpart for: said, not dropped silently
        odd = [c for c in self._components if hasattr(c, "max_depenetration_velocity")
               and (c.max_depenetration_velocity != 5.0 or c.max_contact_impulse < 3.0e38)]
        if odd:
            import warnings
            warnings.warn(f"maniskill_amd backend: max_depenetration_velocity / max_contact_impulse of {len(odd)} bodies (e.g. {odd[0].name!r}) "
                          "accepted, not modelled: penetration recovery is capped by the engine's own constant", stacklevel=3)
        sc, bc, shc = self._cfg["scene"], self._cfg["body"], self._cfg["shape"]
        cfg = E.SimConfig(sim_freq=1.0 / self._timestep, control_freq=1.0 / self._timestep, scene_config=E.SceneConfig(
            gravity=[float(g) for g in sc["gravity"]], bounce_threshold=sc["bounce_threshold"], sleep_threshold=bc["sleep_threshold"],
            contact_offset=shc["contact_offset"], rest_offset=shc["rest_offset"],
            solver_position_iterations=int(bc["solver_position_iterations"]), solver_velocity_iterations=int(bc["solver_velocity_iterations"]),
            enable_pcm=bool(sc["enable_pcm"]), enable_tgs=bool(sc["enable_tgs"]),
            # PhysX sizes its contact buffers per scene (GPUMemoryConfig.max_rigid_contact_count, structs/types.py:18-23); this backend per
            # sub-scene: 128 points / 128 solver blocks (include/msk_physx.h: msk_config.contact_capacity = 1; MSK_CONTACT_CAPACITY=0 selects 48 / 64)
            contact_capacity=int(os.environ.get("MSK_CONTACT_CAPACITY", "1"))))
        cls = E.PhysxGpuSystem
        if host_memory:
            cls = type("HostMemorySystem", (E.PhysxGpuSystem,), dict(host_memory=True))
        self._groups = []
        for envs, tpl, inst in compiled:
            eng = cls(torch_device, tpl, len(envs), cfg, lib=lib)
            eng.gpu_init()
            for sid, (hs, lp) in inst["boxes"].items():
                eng.set_env_boxes(sid, hs, lp)
            for bid, (m, I) in inst["masses"].items():
                eng.set_env_masses(bid, m, I)
            g = _Group(envs, eng, tpl)
            g.npassive = self._npassive_of_group.get(len(self._groups), 0)
            self._groups.append(g)
        g0 = self._groups[0]
        self._engine, self._template = g0.engine, g0.template      # single-group scenes: the zero-copy fast path
        self._nb, self._na, self._max_dof = g0.nb, g0.na, g0.max_dof
        self._env_box_shapes = self._env_box_shapes_of_group.get(0, {})
        if len(self._groups) == 1 and g0.npassive == 0:
            eng = g0.engine
            self._multi = None
            self.cuda_rigid_body_data = eng.cuda_rigid_body_data
            self.cuda_rigid_body_force = eng.cuda_rigid_body_force
            self.cuda_rigid_body_torque = eng.cuda_rigid_body_torque
            self.cuda_articulation_qpos = eng.cuda_articulation_qpos
            self.cuda_articulation_qvel = eng.cuda_articulation_qvel
            self.cuda_articulation_qacc = eng.cuda_articulation_qacc
            self.cuda_articulation_qf = eng.cuda_articulation_qf
            self.cuda_articulation_target_qpos = eng.cuda_articulation_target_qpos
            self.cuda_articulation_target_qvel = eng.cuda_articulation_target_qvel
            lf = eng.cuda_articulation_link_incoming_joint_forces
            rows = g0.n * max(g0.na, 1)
            self.cuda_articulation_link_incoming_joint_forces = _Handle3D(lf, (rows, max(lf.shape[0] // max(rows, 1), 1), 6))
        else:
            self._engines = [g.engine for g in self._groups]
            self._multi = _MultiBuffers(self, torch_device)
            rows = self.cuda_rigid_body_data.torch()
            for c in self._components:                    # pose-only actors start where their entities were placed
                if getattr(c, "_body_id", -1) <= -2 and c.entity is not None:
                    ep = c.entity._pose
                    rows[self._pose_index(c), :7] = torch.as_tensor(np.concatenate([ep._p, ep._q]), dtype=torch.float32, device=rows.device)
        self._initialized = True
        if self._gc_paused:
            import gc
            _gc_restore(self)
            gc.unfreeze()     # (what an earlier scene froze is garbage by now if the caller reconfigured: let this collection see it)
            gc.collect()
            gc.freeze()       # the entity trees live as long as the scene: later collections need not walk them

    # indices ------------------------------------------------------------------------------------------------------------
    def _pose_index(self, comp) -> int:
        if comp._body_id == -1:
            raise RuntimeError("static bodies have no row in cuda_rigid_body_data")
        g = self._groups[comp._group]
        if comp._body_id <= -2:
            return g.pbase + comp._lenv * g.npassive + (-2 - comp._body_id)
        return g.base + comp._lenv * g.nb + comp._body_id

    def _art_index(self, art) -> int:
        g = self._groups[art._group]
        return g.abase + art._lenv * g.na + art._art_id

    # CPU-style accessors (synchronous; off the hot path) ---------------------------------------------------------------
    def _sync_in(self):
        if self._multi is not None:
            self._multi.fetch_all()
        else:
            self._engine._fetch(_SYNC_FETCH)

    def _read_body_row(self, comp):
        self._sync_in()
        return self.cuda_rigid_body_data.torch()[self._pose_index(comp)].detach().cpu().numpy().copy()

    def _read_body_pose(self, comp) -> Pose:
        if comp._body_id == -1:
            return comp.entity._pose
        r = self._read_body_row(comp)
        return Pose(r[:3], r[3:7])

    def _write_body_pose(self, comp, pose: Pose):
        if comp._body_id == -1:
            raise RuntimeError("static bodies cannot be moved after the simulation was initialised")
        self._sync_in()
        row = self.cuda_rigid_body_data.torch()[self._pose_index(comp)]
        row[:7] = torch.as_tensor(np.concatenate([pose._p, pose._q]), device=row.device)
        P = self._P
        if isinstance(comp, P.PhysxArticulationLinkComponent):
            if not comp.is_root:
                raise RuntimeError("only the root link of an articulation can be moved")
            self._do("apply", "gpu_apply_articulation_root_pose", ("rigid_body_data",))
            self._do("call", "gpu_update_articulation_kinematics")
        else:
            self._do("apply", "gpu_apply_rigid_dynamic_data", ("rigid_body_data",))

    def _write_body_cols(self, comp, col0, vals):
        self._sync_in()
        row = self.cuda_rigid_body_data.torch()[self._pose_index(comp)]
        row[col0:col0 + 3] = torch.as_tensor(vals.reshape(3), device=row.device)
        self._do("apply", "gpu_apply_rigid_dynamic_data", ("rigid_body_data",))

    def _write_root_velocity(self, comp, col0, vals):
        self._sync_in()
        row = self.cuda_rigid_body_data.torch()[self._pose_index(comp)]
        row[col0:col0 + 3] = torch.as_tensor(vals.reshape(3), device=row.device)
        self._do("apply", "gpu_apply_articulation_root_velocity", ())

    def _art_buf(self, name):
        return getattr(self, "cuda_articulation_" + name).torch()

    def _read_art_vec(self, art, name):
        self._sync_in()
        return self._art_buf(name)[self._art_index(art), :art.dof].detach().cpu().numpy().copy()

    def _write_art_vec(self, art, name, v):
        self._sync_in()
        buf = self._art_buf(name)
        buf[self._art_index(art), :art.dof] = torch.as_tensor(v, device=buf.device)
        self._do("apply", {"qpos": "gpu_apply_articulation_qpos", "qvel": "gpu_apply_articulation_qvel",
                           "qf": "gpu_apply_articulation_qf", "target_qpos": "gpu_apply_articulation_target_position",
                           "target_qvel": "gpu_apply_articulation_target_velocity"}[name], (name,))
        if name == "qpos":
            self._do("call", "gpu_update_articulation_kinematics")

    def _dof_index(self, art, joint):
        return art.active_joints.index(joint)

    def _read_dof(self, art, name, joint):
        return float(self._read_art_vec(art, name)[self._dof_index(art, joint)])

    def _write_dof(self, art, name, joint, v):
        vec = self._read_art_vec(art, name)
        vec[self._dof_index(art, joint)] = v
        self._write_art_vec(art, name, vec)

    def _read_link_joint_forces(self, art):
        if self._multi is not None:
            self._multi.pull_link_forces()
        else:
            self._engine.gpu_fetch_articulation_link_incoming_joint_forces()
        t = self.cuda_articulation_link_incoming_joint_forces.torch()
        return t[self._art_index(art), :len(art.links)].detach().cpu().numpy().copy()

    def _add_force_torque(self, comp, force, torque):
        F = self.cuda_rigid_body_force.torch()
        T = self.cuda_rigid_body_torque.torch()
        i = self._pose_index(comp)
        F[i, :3] += torch.as_tensor(force, device=F.device)
        T[i, :3] += torch.as_tensor(torque, device=T.device)
        self._do("apply", "gpu_apply_rigid_dynamic_force", ("rigid_body_force",))
        self._do("apply", "gpu_apply_rigid_dynamic_torque", ("rigid_body_torque",))

    def _add_force_at_point(self, comp, force, point):
        pose = self._read_body_pose(comp)
        com = (pose * comp.cmass_local_pose).p
        self._add_force_torque(comp, force, np.cross(point - com, force))

    def _drive_changed(self, joint):
        """joint.set_drive_properties after gpu_init (agent.set_control_mode -> controller.set_drive_property): the drive is part of the
        group's template, so the first sub-scene's call moves every sub-scene of its group; the others find it done (msk_set_drive
        returns at once when nothing changes)"""
        link = joint.child_link
        g = self._groups[link._group]
        g.engine.lib.check(g.engine.ctx, g.engine.lib.set_drive(g.engine.ctx, int(link._body_id), float(joint.stiffness), float(joint.damping),
                                                                float(min(joint.force_limit, 3.0e38)), 1 if joint.drive_mode == "acceleration" else 0),
                           "set_drive")

    # sapien call -> (msk_batch op, mask) for a scene of several groups
    _BATCH = None

    @classmethod
    def _batch_table(cls):
        if cls._BATCH is None:
            from maniskill_amd import _native as N
            A, F = N.BATCH_APPLY, N.BATCH_FETCH
            cls._BATCH = {
                "gpu_apply_rigid_dynamic_data": (A, N.APPLY_RIGID_DATA), "gpu_apply_rigid_dynamic_force": (A, N.APPLY_RIGID_FORCE),
                "gpu_apply_rigid_dynamic_torque": (A, N.APPLY_RIGID_TORQUE), "gpu_apply_articulation_root_pose": (A, N.APPLY_ART_ROOT_POSE),
                "gpu_apply_articulation_root_velocity": (A, N.APPLY_ART_ROOT_VELOCITY), "gpu_apply_articulation_qpos": (A, N.APPLY_ART_QPOS),
                "gpu_apply_articulation_qvel": (A, N.APPLY_ART_QVEL), "gpu_apply_articulation_qf": (A, N.APPLY_ART_QF),
                "gpu_apply_articulation_target_position": (A, N.APPLY_ART_TARGET_QPOS),
                "gpu_apply_articulation_target_velocity": (A, N.APPLY_ART_TARGET_QVEL),
                "gpu_fetch_rigid_dynamic_data": (F, N.FETCH_RIGID_DATA), "gpu_fetch_articulation_link_pose": (F, N.FETCH_RIGID_DATA),
                "gpu_fetch_articulation_qpos": (F, N.FETCH_ART_QPOS), "gpu_fetch_articulation_qvel": (F, N.FETCH_ART_QVEL),
                "gpu_fetch_articulation_qacc": (F, N.FETCH_ART_QACC), "gpu_fetch_articulation_target_qpos": (F, N.FETCH_ART_TARGETS),
                "gpu_fetch_all": (F, N.FETCH_RIGID_DATA | N.FETCH_ART_QPOS | N.FETCH_ART_QVEL | N.FETCH_ART_QACC | N.FETCH_ART_TARGETS),
                "gpu_update_articulation_kinematics": (N.BATCH_UPDATE_KINEMATICS, 0), "step": (N.BATCH_STEP, 0),
            }
        return cls._BATCH

    def _batch(self, method):
        """One boundary call on every group's context, in one native call (msk_batch: the groups' kernels are independent and small, so
        the library issues them on a handful of side streams forked from / joined into the current one)."""
        from maniskill_amd.physx import batch_call
        ent = self._batch_table()[method]
        if ent is not None:
            batch_call(self._engines, ent[0], ent[1])

    def _do(self, kind, method, names=()):
        """One boundary call.  With several groups the contexts are bound to their rows of the unified tensors (msk_bind_buffers), so
        'apply' / 'fetch' move nothing on the host side."""
        if self._multi is None:
            getattr(self._engine, method)()
        else:
            self._batch(method)

    def _each_group(self, method):
        for g in self._groups:
            getattr(g.engine, method)()

    def step(self):
        if self._multi is None:
            self._engine.step()
        else:
            self._batch("step")


class PhysxGpuSystem(PhysxSystem):
    """``physx.PhysxGpuSystem(device)`` (sapien_env.py:1187): every sub-scene of this process on one GPU."""

    def __init__(self, device=None):
        super().__init__()
        from . import physx as P
        from ._core import Device
        self.device = device if (isinstance(device, Device) or hasattr(device, "is_cuda")) else Device(device if device is not None else "cuda")
        self._backend = P._backend

    def _torch_device(self):
        if self._backend is not None and self._backend[1]:
            return torch.device("cpu")
        return torch.device("cuda", max(int(getattr(self.device, "cuda_id", 0)), 0))

    def gpu_init(self):
        if self._initialized:
            raise RuntimeError("gpu_init() was already called")
        if not self._scenes:
            raise RuntimeError("gpu_init() without sub-scenes")
        if self._backend is not None:
            lib, host = self._backend
        else:
            from maniskill_amd import _native as N
            lib, host = N.default_lib(), False          # raises if libmsk_physx.so is missing: no CPU fallback
        self._start_engine(self._torch_device(), lib, host)

    # apply / fetch ------------------------------------------------------------------------------------------------------
    def gpu_apply_rigid_dynamic_data(self): self._do("apply", "gpu_apply_rigid_dynamic_data", ("rigid_body_data",))
    def gpu_apply_rigid_dynamic_force(self): self._do("apply", "gpu_apply_rigid_dynamic_force", ("rigid_body_force",))
    def gpu_apply_rigid_dynamic_torque(self): self._do("apply", "gpu_apply_rigid_dynamic_torque", ("rigid_body_torque",))
    def gpu_apply_articulation_root_pose(self): self._do("apply", "gpu_apply_articulation_root_pose", ("rigid_body_data",))
    def gpu_apply_articulation_root_velocity(self): self._do("apply", "gpu_apply_articulation_root_velocity", ())
    def gpu_apply_articulation_qpos(self): self._do("apply", "gpu_apply_articulation_qpos", ("qpos",))
    def gpu_apply_articulation_qvel(self): self._do("apply", "gpu_apply_articulation_qvel", ("qvel",))
    def gpu_apply_articulation_qf(self): self._do("apply", "gpu_apply_articulation_qf", ("qf",))
    def gpu_apply_articulation_target_position(self): self._do("apply", "gpu_apply_articulation_target_position", ("target_qpos",))
    def gpu_apply_articulation_target_velocity(self): self._do("apply", "gpu_apply_articulation_target_velocity", ("target_qvel",))
    def gpu_fetch_rigid_dynamic_data(self): self._do("fetch", "gpu_fetch_rigid_dynamic_data", ("rigid_body_data",))
    def gpu_fetch_articulation_link_pose(self): self._do("fetch", "gpu_fetch_articulation_link_pose", ("rigid_body_data",))
    def gpu_fetch_articulation_link_velocity(self): pass   # same rows as link_pose: fetched together
    def gpu_fetch_articulation_qpos(self): self._do("fetch", "gpu_fetch_articulation_qpos", ("qpos",))
    def gpu_fetch_articulation_qvel(self): self._do("fetch", "gpu_fetch_articulation_qvel", ("qvel",))
    def gpu_fetch_articulation_qacc(self): self._do("fetch", "gpu_fetch_articulation_qacc", ("qacc",))
    def gpu_fetch_articulation_target_qpos(self): self._do("fetch", "gpu_fetch_articulation_target_qpos", ("target_qpos", "target_qvel"))
    def gpu_fetch_articulation_target_qvel(self): pass     # written together with target_qpos
    def gpu_update_articulation_kinematics(self): self._do("call", "gpu_update_articulation_kinematics")

    # extension of this backend (not in SAPIEN): several of the apply / fetch calls above as ONE boundary call -- the library takes a bit mask
    # (include/msk_physx.h: msk_apply / msk_fetch), SAPIEN's API spends a call per buffer.  maniskill_amd/fused_step.py uses it when present.
    APPLY_ALL_MASK = 1 | 2 | 4 | 8 | 16 | 32 | 64 | 512      # what scene._gpu_apply_all() applies (envs/scene.py:950-966)
    FETCH_ALL_MASK = 1 | 2 | 4 | 8 | 16                       # what scene._gpu_fetch_all() fetches (envs/scene.py:968-986)

    def gpu_apply_masked(self, mask: int):
        if self._multi is None:
            self._engine._apply(int(mask))
        else:
            from maniskill_amd.physx import batch_call
            from maniskill_amd import _native as N
            batch_call(self._engines, N.BATCH_APPLY, int(mask))

    def gpu_fetch_masked(self, mask: int):
        if self._multi is None:
            self._engine._fetch(int(mask))
        else:
            from maniskill_amd.physx import batch_call
            from maniskill_amd import _native as N
            batch_call(self._engines, N.BATCH_FETCH, int(mask))

    def gpu_fetch_articulation_link_incoming_joint_forces(self):
        if self._multi is not None:
            self._multi.pull_link_forces()
        else:
            self._engine.gpu_fetch_articulation_link_incoming_joint_forces()

    def sync_poses_gpu_to_cpu(self):
        self._sync_in()
        rows = self.cuda_rigid_body_data.torch().detach().cpu().numpy()
        for c in self._components:
            if c._body_id >= 0 and c.entity is not None:
                r = rows[self._pose_index(c)]
                c.entity._pose = Pose(r[:3], r[3:7])

    # contact queries (envs/scene.py:741-801; utils/structs/base.py:116-136; articulation.py:447-462) ---------------------------
    def _make_query(self, keys, groups, lenvs, create):
        """Listed row j asks for engine key keys[j] of local sub-scene lenvs[j] of group groups[j].  Per group: one engine query over
        the group's unique keys; the listed rows are gathered from the engines' results (identity for the usual one-pair-per-env
        list of a single-group scene: then the engine's own buffer is handed out)."""
        parts = []
        for gi, g in enumerate(self._groups):
            rows = [j for j in range(len(keys)) if groups[j] == gi]
            if not rows:
                continue
            uniq, pos = [], {}
            for j in rows:
                if keys[j] not in pos:
                    pos[keys[j]] = len(uniq)
                    uniq.append(keys[j])
            U = len(uniq)
            idx = np.array([lenvs[j] * U + pos[keys[j]] for j in rows], dtype=np.int64)
            q = create(g.engine, uniq)
            parts.append((g, q, np.asarray(rows, dtype=np.int64), idx))
        dev = self.cuda_rigid_body_data.torch().device
        if len(parts) == 1 and len(self._groups) == 1:
            g, q, rows, idx = parts[0]
            if len(idx) == g.n * (len(idx) // max(g.n, 1)) and np.array_equal(idx, np.arange(len(idx))) and np.array_equal(rows, np.arange(len(rows))):
                return _GatherQuery([(g, q, None, None)], q.cuda_impulses, len(keys), dev)
        return _GatherQuery([(g, q, torch.as_tensor(rows, device=dev), torch.as_tensor(idx, device=dev)) for g, q, rows, idx in parts],
                            None, len(keys), dev)

    def gpu_create_contact_pair_impulse_query(self, body_pairs):
        keys, groups, lenvs = [], [], []
        for a, b in body_pairs:
            if a._env != b._env:
                raise RuntimeError("a contact pair must live in one sub-scene")
            keys.append((a._body_id, b._body_id))
            groups.append(a._group)
            lenvs.append(a._lenv)
        return self._make_query(keys, groups, lenvs, lambda eng, uniq: eng.gpu_create_contact_pair_impulse_query(uniq))

    def gpu_create_contact_body_impulse_query(self, bodies):
        return self._make_query([b._body_id for b in bodies], [b._group for b in bodies], [b._lenv for b in bodies],
                                lambda eng, uniq: eng.gpu_create_contact_body_impulse_query(uniq))

    def gpu_query_contact_pair_impulses(self, query):
        query._run()

    def gpu_query_contact_body_impulses(self, query):
        query._run()


class PhysxCpuSystem(PhysxSystem):
    """``physx.PhysxCpuSystem()`` (sapien_env.py:1212): one sub-scene with SAPIEN's per-object API.  The same C-ABI library
    simulates it (a single-env context on the GPU, or the test-suite's injected checker); component getters / setters
    read and write the state buffers synchronously."""

    def __init__(self):
        super().__init__()
        from . import physx as P
        self._backend = P._backend
        self._starting = False

    def _live(self) -> bool:
        # SAPIEN's CPU system is live from the start; here the template is frozen at the first state access after building
        if not self._initialized and not self._starting:
            self._ensure()
        return self._initialized

    def _ensure(self):
        if self._initialized or self._starting:
            return
        self._starting = True
        try:
            self._ensure_inner()
        finally:
            self._starting = False

    def _ensure_inner(self):
        if len(self._scenes) != 1:
            raise RuntimeError("PhysxCpuSystem simulates exactly one scene")
        if self._backend is not None:
            lib, host = self._backend
            dev = torch.device("cpu") if host else torch.device("cuda", 0)
        else:
            from maniskill_amd import _native as N
            lib, host, dev = N.default_lib(), False, torch.device("cuda", 0)
        self._start_engine(dev, lib, host)
        # host-side initial state recorded before the first step
        for c in self._components:
            if c._body_id == -1:
                continue
            if isinstance(c, self._P.PhysxArticulationLinkComponent):
                continue
            self._write_body_pose(c, c.entity._pose)
        seen = set()
        for c in self._components:
            if isinstance(c, self._P.PhysxArticulationLinkComponent) and id(c.articulation) not in seen:
                art = c.articulation
                seen.add(id(art))
                self._write_body_pose(art.root, art.root.entity._pose)
                if art._qpos0 is not None:
                    self._write_art_vec(art, "qpos", art._qpos0)
                tq = np.array([j._drive_target for j in art.active_joints], dtype=np.float32)
                tv = np.array([j._drive_velocity_target for j in art.active_joints], dtype=np.float32)
                self._write_art_vec(art, "target_qpos", tq)
                self._write_art_vec(art, "target_qvel", tv)

    def step(self):
        self._ensure()
        self._engine.step()

    def get_contacts(self):
        self._ensure()
        P = self._P
        ids, vals = self._engine.get_contacts(0, 256)
        if len(ids) == 0:
            return []
        owners = self._shape_owner_of_group[0]
        by_pair = {}
        for (sa, sb, _), v in zip(ids, vals):
            by_pair.setdefault((int(sa), int(sb)), []).append(v)
        out = []
        for (sa, sb), pts in by_pair.items():
            A, B = owners[sa], owners[sb]
            points = []
            for v in pts:
                n = v[3:6]
                points.append(P.PhysxContactPoint(position=v[:3].copy(), normal=n.copy(), impulse=(n * v[7]).astype(np.float32),
                                                  separation=float(v[6])))
            out.append(P.PhysxContact([A._body, B._body], [A, B], points))
        return out
