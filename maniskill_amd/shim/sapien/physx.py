"""``sapien.physx``: the physics half of the shim.  Host-side Python objects record what ManiSkill's builders create; at
``PhysxGpuSystem.gpu_init()`` the sub-scenes are compiled (``_compile.py``) into the C-ABI library's scene template
(include/msk_physx.h; hand-written HIP kernels for gfx950) and every ``cuda_*`` buffer becomes a zero-copy torch view of the
library's device memory.

Reference call sites (names and argument meaning kept):
  module config      mani_skill/envs/sapien_env.py:244-245,271-275,1173-1180
  systems            sapien_env.py:1187,1202,1212,1227; envs/scene.py:379-380,902-986,741-801
  components         utils/building/actor_builder.py:57-164; utils/building/articulation_builder.py:65-205;
                     utils/structs/{base,actor,link,articulation,articulation_joint,drive}.py
There is no CPU fallback here: the product binds libmsk_physx.so and raises if it is missing; the test-suite injects its CPU
checker through ``_set_backend`` (same C ABI, host memory).
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence

import numpy as np

from ._core import Component, Entity, Scene
from ._pose import Pose
from . import _mesh

# ----------------------------------------------------------------------------------------------------------- module config
_gpu_enabled = False
_config = dict(
    shape=dict(contact_offset=0.01, rest_offset=0.0),
    body=dict(solver_position_iterations=10, solver_velocity_iterations=1, sleep_threshold=0.005),
    scene=dict(gravity=np.array([0, 0, -9.81]), bounce_threshold=2.0, enable_pcm=True, enable_tgs=True, enable_ccd=False,
               enable_enhanced_determinism=False, enable_friction_every_iteration=True, cpu_workers=0),
    material=dict(static_friction=0.3, dynamic_friction=0.3, restitution=0.1),
    gpu_memory=dict(),
)
_GPU_MEMORY_KEYS = ("temp_buffer_capacity", "max_rigid_contact_count", "max_rigid_patch_count", "heap_capacity",
                    "found_lost_pairs_capacity", "found_lost_aggregate_pairs_capacity", "total_aggregate_pairs_capacity",
                    "collision_stack_size")
_default_material = None
_backend = None     # (NativeLib, host_memory): set by tests; None = the HIP library


def _set_backend(lib, host_memory: bool):
    """Test hook: run the systems created from now on against another library exporting the same C ABI."""
    global _backend
    _backend = None if lib is None else (lib, bool(host_memory))


def is_gpu_enabled() -> bool:
    return _gpu_enabled


def enable_gpu():
    global _gpu_enabled
    _gpu_enabled = True


def set_gpu_memory_config(**kw):
    for k in kw:
        if k not in _GPU_MEMORY_KEYS:
            raise TypeError(f"set_gpu_memory_config() got an unexpected keyword argument '{k}'")
    _config["gpu_memory"].update(kw)


def set_shape_config(contact_offset=None, rest_offset=None):
    if contact_offset is not None:
        _config["shape"]["contact_offset"] = float(contact_offset)
    if rest_offset is not None:
        _config["shape"]["rest_offset"] = float(rest_offset)


def set_body_config(solver_position_iterations=None, solver_velocity_iterations=None, sleep_threshold=None):
    for k, v in (("solver_position_iterations", solver_position_iterations), ("solver_velocity_iterations", solver_velocity_iterations),
                 ("sleep_threshold", sleep_threshold)):
        if v is not None:
            _config["body"][k] = v


def set_scene_config(gravity=None, bounce_threshold=None, enable_pcm=None, enable_tgs=None, enable_ccd=None,
                     enable_enhanced_determinism=None, enable_friction_every_iteration=None, cpu_workers=None):
    for k, v in (("gravity", gravity), ("bounce_threshold", bounce_threshold), ("enable_pcm", enable_pcm), ("enable_tgs", enable_tgs),
                 ("enable_ccd", enable_ccd), ("enable_enhanced_determinism", enable_enhanced_determinism),
                 ("enable_friction_every_iteration", enable_friction_every_iteration), ("cpu_workers", cpu_workers)):
        if v is not None:
            _config["scene"][k] = v


def set_default_material(static_friction, dynamic_friction, restitution):
    global _default_material
    _config["material"].update(static_friction=float(static_friction), dynamic_friction=float(dynamic_friction),
                               restitution=float(restitution))
    _default_material = None


def get_default_material():
    global _default_material
    if _default_material is None:
        _default_material = PhysxMaterial(**_config["material"])
    return _default_material


def get_shape_config():
    return dict(_config["shape"])


def get_body_config():
    return dict(_config["body"])


def get_scene_config():
    return dict(_config["scene"])


# ----------------------------------------------------------------------------------------------------------- materials, shapes
_MATERIAL_EPOCH = [0]   # bumped by every write to a material: cached shape signatures (PhysxCollisionShape._sig) carry the epoch they were made in


class PhysxMaterial:
    def __init__(self, static_friction, dynamic_friction, restitution):
        self.static_friction, self.dynamic_friction, self.restitution = float(static_friction), float(dynamic_friction), float(restitution)

    def __setattr__(self, k, v):
        object.__setattr__(self, k, v)
        _MATERIAL_EPOCH[0] += 1

    def get_static_friction(self):
        return self.static_friction

    def get_dynamic_friction(self):
        return self.dynamic_friction

    def get_restitution(self):
        return self.restitution

    def set_static_friction(self, v):
        self.static_friction = float(v)

    def set_dynamic_friction(self, v):
        self.dynamic_friction = float(v)

    def set_restitution(self, v):
        self.restitution = float(v)


_ZERO3F = np.zeros(3, dtype=np.float32)
_ZERO3F.setflags(write=False)
_FREE_LIMITS = np.array([[-np.inf, np.inf]], dtype=np.float32)
_FREE_LIMITS.setflags(write=False)


class PhysxCollisionShape:
    _kind = "shape"

    def __init__(self, material: Optional[PhysxMaterial] = None):
        self.physical_material = material if material is not None else get_default_material()
        self._local_pose = Pose()
        self._groups = [1, 1, 0, 0]
        self.density = 1000.0
        self.patch_radius = 0.0
        self.min_patch_radius = 0.0
        self.contact_offset = _config["shape"]["contact_offset"]
        self.rest_offset = _config["shape"]["rest_offset"]
        self.is_trigger = False
        self._body = None

    @property
    def material(self):
        return self.physical_material

    @property
    def local_pose(self):
        return self._local_pose

    @local_pose.setter
    def local_pose(self, pose):
        self._frozen_check()
        self._local_pose = Pose(pose.p, pose.q)

    def get_local_pose(self):
        return self._local_pose

    def set_local_pose(self, pose):
        self.local_pose = pose

    def _frozen_check(self):
        b = self._body
        if b is not None and b._system is not None and b._system._initialized:
            raise RuntimeError("collision shapes cannot be changed after the simulation was initialised")

    def __setattr__(self, k, v):
        # the compiler's signature of the shape (_sig) is cached on it, the mass properties of the body it is attached to on the body:
        # setting anything drops both
        object.__setattr__(self, k, v)
        d = self.__dict__
        d.pop("_sig_cache", None)
        b = d.get("_body")
        if b is not None:
            b.__dict__.pop("_mt_cache", None)

    def _sig(self, fold=None):
        """-> (relaxed, strict) signature for the scene compiler (_system._compile): what must agree for two sub-scenes to share a compiled
        template (box sizes and positions may differ inside a group), and everything.  Compared, never decoded: poses enter as the bytes of
        their float64 arrays.  `fold`: pose of the static entity the shape sits on (then nothing is cached)."""
        if fold is None:
            cached = self.__dict__.get("_sig_cache")
            if cached is not None and cached[0] == _MATERIAL_EPOCH[0]:     # (materials are shared objects: a write to any of them re-reads them)
                return cached[1]
        lp = self._local_pose if fold is None else fold * self._local_pose
        mat = self.physical_material
        base = (self._kind, tuple(self._groups), mat.static_friction, mat.dynamic_friction, mat.restitution, self.patch_radius,
                self.min_patch_radius, self.contact_offset, self.rest_offset)
        pk = lp._p.tobytes() + lp._q.tobytes()
        kind = self._kind
        if kind == "box":
            out = (base + (lp._q.tobytes(),), base + (pk, self._half.tobytes()))
        elif kind == "convex":
            h = self.__dict__.get("_vhash")
            if h is None:
                h = hash(self._scaled_vertices.tobytes())
                self.__dict__["_vhash"] = h
            out = (base + (pk, h),) * 2
        elif kind == "sphere":
            out = (base + (pk, self.radius),) * 2
        elif kind in ("capsule", "cylinder"):
            out = (base + (pk, self.radius, self.half_length),) * 2
        elif kind == "plane":
            out = (base + (pk,),) * 2
        elif kind == "trimesh":
            out = (base + (pk, self.filename, np.asarray(self.scale, dtype=np.float64).tobytes()),) * 2
        else:
            raise TypeError(type(self))
        if fold is None:
            self.__dict__["_sig_cache"] = (_MATERIAL_EPOCH[0], out)
        return out

    def _clone(self):
        """A shape equal to this one that shares its immutable parts (cooked vertices, faces, material): what a builder makes for sub-scene
        k + 1 after it made this one for sub-scene k (building 16k sub-scenes spends its time in constructors otherwise)."""
        new = object.__new__(type(self))
        d = new.__dict__
        d.update(self.__dict__)
        d["_groups"] = list(self._groups)
        d["_local_pose"] = Pose._like(self._local_pose)
        d["_body"] = None
        return new

    def get_collision_groups(self):
        return list(self._groups)

    def set_collision_groups(self, groups):
        self._frozen_check()
        self._groups = [int(g) & 0xFFFFFFFF for g in groups]

    @property
    def collision_groups(self):
        return list(self._groups)

    def set_density(self, density):
        self.density = float(density)

    def get_density(self):
        return self.density

    def set_patch_radius(self, r):
        self.patch_radius = float(r)

    def get_patch_radius(self):
        return self.patch_radius

    def set_min_patch_radius(self, r):
        self.min_patch_radius = float(r)

    def get_min_patch_radius(self):
        return self.min_patch_radius

    def get_physical_material(self):
        return self.physical_material

    def set_physical_material(self, m):
        self.physical_material = m

    def set_contact_offset(self, v):
        self.contact_offset = float(v)

    def set_rest_offset(self, v):
        self.rest_offset = float(v)

    # (mass, com, inertia about com) in the shape frame
    def _mass_props(self):
        raise NotImplementedError


class PhysxCollisionShapePlane(PhysxCollisionShape):
    _kind = "plane"

    def _mass_props(self):
        return 0.0, np.zeros(3), np.zeros((3, 3))


class PhysxCollisionShapeBox(PhysxCollisionShape):
    _kind = "box"

    def __init__(self, half_size, material=None):
        super().__init__(material)
        self._half = np.array(half_size, dtype=np.float64).reshape(3)

    @property
    def half_size(self):
        return self._half.astype(np.float32)

    def get_half_size(self):
        return self.half_size

    def _mass_props(self):
        return _mesh.box_mass(self._half, self.density)


class PhysxCollisionShapeSphere(PhysxCollisionShape):
    _kind = "sphere"

    def __init__(self, radius, material=None):
        super().__init__(material)
        self.radius = float(radius)

    def get_radius(self):
        return self.radius

    def _mass_props(self):
        return _mesh.sphere_mass(self.radius, self.density)


class PhysxCollisionShapeCapsule(PhysxCollisionShape):
    _kind = "capsule"

    def __init__(self, radius, half_length, material=None):
        super().__init__(material)
        self.radius, self.half_length = float(radius), float(half_length)

    def get_radius(self):
        return self.radius

    def get_half_length(self):
        return self.half_length

    def _mass_props(self):
        return _mesh.capsule_mass(self.radius, self.half_length, self.density)


class PhysxCollisionShapeCylinder(PhysxCollisionShape):
    _kind = "cylinder"

    def __init__(self, radius, half_length, material=None):
        super().__init__(material)
        self.radius, self.half_length = float(radius), float(half_length)

    def get_radius(self):
        return self.radius

    def get_half_length(self):
        return self.half_length

    def _mass_props(self):
        return _mesh.cylinder_mass(self.radius, self.half_length, self.density)


class PhysxCollisionShapeConvexMesh(PhysxCollisionShape):
    """``vertices`` are unscaled, ``scale`` separate (mani_skill/utils/geometry/trimesh_utils.py:29-33 multiplies them)."""
    _kind = "convex"

    def __init__(self, filename=None, scale=(1, 1, 1), material=None, vertices=None):
        super().__init__(material)
        self.scale = np.array(scale, dtype=np.float32).reshape(-1)
        if self.scale.size == 1:
            self.scale = np.full(3, float(self.scale[0]), dtype=np.float32)
        self.filename = filename
        try:
            if vertices is None:
                v, _ = _mesh.cook_convex(filename, (1, 1, 1))
            else:
                vin = np.ascontiguousarray(vertices)
                key = ("reduced", hash(vin.tobytes()), vin.shape)          # every sub-scene cooks the same pieces again
                if key not in _mesh._cache:
                    _mesh._cache[key] = np.asarray(_mesh.reduce_hull(vin), dtype=np.float32)
                v = _mesh._cache[key]
            self.vertices = np.ascontiguousarray(v, dtype=np.float32)
            key = ("hull_faces", hash(self.vertices.tobytes()), self.vertices.shape, tuple(float(x) for x in self.scale))
            if key not in _mesh._cache:
                _mesh._cache[key] = _mesh.hull_faces(self.vertices * self.scale)
            self._faces = _mesh._cache[key]
        except RuntimeError:
            raise
        except Exception as e:          # degenerate input, unreadable file: SAPIEN raises RuntimeError ("failed to cook")
            raise RuntimeError(f"failed to cook convex mesh from {filename}: {e}") from e

    @staticmethod
    def load_multiple(filename, scale=(1, 1, 1), material=None):
        out = []
        for p in _mesh.load_mesh_parts(filename):
            try:
                out.append(PhysxCollisionShapeConvexMesh(filename, scale, material, vertices=p["vertices"]))
            except RuntimeError:
                continue
        if not out:
            raise RuntimeError(f"failed to cook any convex mesh from {filename}")
        return out

    def get_vertices(self):
        return self.vertices

    def get_triangles(self):
        return self._faces.astype(np.uint32)

    def get_scale(self):
        return self.scale

    @property
    def _scaled_vertices(self):
        return np.ascontiguousarray(self.vertices * self.scale, dtype=np.float32)

    def _mass_props(self):
        key = ("mesh_mass", hash(self.vertices.tobytes()), self.vertices.shape, tuple(float(x) for x in self.scale), float(self.density))
        if key not in _mesh._cache:       # the same hull in every sub-scene
            _mesh._cache[key] = _mesh.mesh_mass(self._scaled_vertices, self._faces, self.density)
        return _mesh._cache[key]


class PhysxCollisionShapeTriangleMesh(PhysxCollisionShape):
    """Non-convex triangle meshes collide (in PhysX) only as static / kinematic geometry, triangle by triangle.  This engine collides
    convex shapes: a part of the file that is convex (to 1 % of its size) becomes its hull, any other part is cut into at most 16
    convex pieces (_mesh.convex_decompose; `decomposition_error` = how far a piece's hull still reaches off the surface, as a
    fraction of the mesh size)."""
    _kind = "trimesh"

    def __init__(self, filename, scale=(1, 1, 1), material=None):
        super().__init__(material)
        self.scale = np.array(scale, dtype=np.float32).reshape(-1)
        if self.scale.size == 1:
            self.scale = np.full(3, float(self.scale[0]), dtype=np.float32)
        self.filename = filename
        parts = _mesh.load_mesh_parts(filename)
        self.vertices = np.concatenate([p["vertices"] for p in parts]).astype(np.float32)
        faces, o = [], 0
        for p in parts:
            faces.append(p["faces"] + o)
            o += len(p["vertices"])
        self._faces = np.concatenate(faces)
        self._cook(16)

    def _cook(self, max_parts):
        """(re)build the convex pieces with at most `max_parts` pieces per part of the file (the scene compiler asks for fewer when
        the sub-scene's shapes would not fit the engine's template otherwise)"""
        hulls, self.decomposition_error = _mesh.cook_nonconvex(self.filename, max_parts)
        self._max_parts = max_parts
        self._hulls = [PhysxCollisionShapeConvexMesh(self.filename, self.scale, self.physical_material, vertices=np.asarray(h, dtype=np.float32))
                       for h in hulls]

    def get_vertices(self):
        return self.vertices

    def get_triangles(self):
        return self._faces.astype(np.uint32)

    def _mass_props(self):
        return 0.0, np.zeros(3), np.zeros((3, 3))


# ----------------------------------------------------------------------------------------------------------- components
class PhysxBaseComponent(Component):
    pass


class PhysxRigidBaseComponent(PhysxBaseComponent):
    _is_physx_body = True

    def __init__(self):
        super().__init__()
        self.collision_shapes: list[PhysxCollisionShape] = []
        self._system: Optional["PhysxSystem"] = None
        self._env = -1          # sub-scene index, set when the entity enters a scene
        self._body_id = -1      # template body id, set by the compiler

    def attach(self, shape: PhysxCollisionShape):
        if self._system is not None and self._system._initialized:
            raise RuntimeError("cannot attach collision shapes after the simulation was initialised")
        shape.__dict__["_body"] = self          # (not through the shape's __setattr__: its cached signature stays valid)
        self.collision_shapes.append(shape)
        self.__dict__.pop("_mt_cache", None)
        return self

    def get_collision_shapes(self):
        return self.collision_shapes

    def _on_add_to_scene(self, scene: Scene):
        self._system = scene.physx_system
        if self._system is not None:
            self._env = self._system._scene_index(scene)
            self._system._register_component(self)

    def _on_remove_from_scene(self, scene: Scene):
        if self._system is not None:
            self._system._unregister_component(self)
        self._system = None

    def compute_global_aabb_tight(self):
        return self.get_global_aabb_fast()

    def get_global_aabb_fast(self):
        pts = []
        T = self.pose
        for s in self.collision_shapes:
            if isinstance(s, PhysxCollisionShapeBox):
                c = np.array([[x, y, z] for x in (-1, 1) for y in (-1, 1) for z in (-1, 1)]) * s.half_size
            elif isinstance(s, PhysxCollisionShapeConvexMesh):
                c = s._scaled_vertices
            elif isinstance(s, PhysxCollisionShapeSphere):
                c = np.array([[x, y, z] for x in (-1, 1) for y in (-1, 1) for z in (-1, 1)]) * s.radius
            elif isinstance(s, (PhysxCollisionShapeCapsule, PhysxCollisionShapeCylinder)):
                c = np.array([[x, y, z] for x in (-1, 1) for y in (-1, 1) for z in (-1, 1)]) * [s.half_length + s.radius, s.radius, s.radius]
            else:
                continue
            M = (T * s.local_pose).to_transformation_matrix()
            pts.append(c @ M[:3, :3].T + M[:3, 3])
        if not pts:
            p = T.p
            return np.stack([p, p])
        pts = np.concatenate(pts)
        return np.stack([pts.min(0), pts.max(0)]).astype(np.float32)


class PhysxRigidStaticComponent(PhysxRigidBaseComponent):
    pass


class PhysxRigidBodyComponent(PhysxRigidBaseComponent):
    def __init__(self):
        super().__init__()
        self._mass = None                 # None: computed from the shapes' densities
        self._cmass_local_pose = None
        self._inertia = None
        self.linear_damping = 0.0
        self.angular_damping = 0.05
        self.disable_gravity = False
        self.max_depenetration_velocity = 5.0   # accepted
        self.max_contact_impulse = 3.0e38       # accepted
        self._lin_vel = _ZERO3F      # replaced, never written in place
        self._ang_vel = _ZERO3F

    # -- mass properties --------------------------------------------------------------------------------------------
    @property
    def auto_compute_mass(self):
        return self._mass is None

    def _auto(self):
        """(mass, com [3], inertia about the com in body axes [3,3]) from the attached shapes' densities."""
        parts = []
        for s in self.collision_shapes:
            m, c, I = s._mass_props()
            if m <= 0:
                continue
            M = s.local_pose._matrix64()
            R = M[:3, :3]
            parts.append((m, R @ c + M[:3, 3], R @ I @ R.T))
        if not parts:
            if isinstance(self, PhysxArticulationLinkComponent):
                return 0.0, np.zeros(3), np.zeros((3, 3))   # frame-only links (tool centre points, camera mounts) carry no mass
            return 1.0, np.zeros(3), np.eye(3)     # PhysX's default for a body without shapes
        return _mesh.combine(parts)

    def _mass_tensor(self):
        """What the engine takes: (mass, com [3], inertia6 [ixx iyy izz ixy ixz iyz] about the com, body axes).  Cached (the scene compiler
        asks twice per body of every sub-scene); the setters of the mass properties, attach() and any attribute set on an attached shape
        drop the cache (code that writes _mass / _cmass_local_pose / _inertia / _exact_inertial directly pops "_mt_cache" itself)."""
        mt = self.__dict__.get("_mt_cache")
        if mt is None:
            mt = self._mass_tensor_now()
            self.__dict__["_mt_cache"] = mt
        return mt

    def _mass_tensor_now(self):
        ex = getattr(self, "_exact_inertial", None)
        if ex is not None and self._mass is not None and float(self._mass) == float(ex[0]):
            m, c, I = ex
        elif self._mass is None:
            m, c, I = self._auto()
        else:
            m = float(self._mass)
            cp = self._cmass_local_pose if self._cmass_local_pose is not None else Pose()
            R = cp._matrix64()[:3, :3]
            I = R @ np.diag(np.asarray(self._inertia if self._inertia is not None else [1, 1, 1], dtype=np.float64)) @ R.T
            c = cp._p.copy()
        return float(m), np.asarray(c, dtype=np.float64), [I[0, 0], I[1, 1], I[2, 2], I[0, 1], I[0, 2], I[1, 2]]

    @property
    def mass(self):
        return float(self._mass) if self._mass is not None else float(self._auto()[0])

    @mass.setter
    def mass(self, m):
        self.set_mass(m)

    def get_mass(self):
        return self.mass

    def set_mass(self, m):
        self._frozen_check()
        if self._mass is None:          # freeze the automatic frame / inertia, scaled to the new mass
            m0, c, I = self._auto()
            w, q = _mesh.principal(I)
            self._cmass_local_pose = Pose(c, q)
            self._inertia = np.asarray(w, dtype=np.float32) * (float(m) / m0 if m0 > 0 else 1.0)
        elif self._inertia is not None and self._mass > 0:
            self._inertia = np.asarray(self._inertia, dtype=np.float32) * (float(m) / float(self._mass))
        self._mass = float(m)
        self.__dict__.pop("_mt_cache", None)

    @property
    def cmass_local_pose(self):
        if self._cmass_local_pose is not None:
            return self._cmass_local_pose
        m, c, I = self._auto()
        return Pose(c, _mesh.principal(I)[1])

    @cmass_local_pose.setter
    def cmass_local_pose(self, pose):
        self._frozen_check()
        self._cmass_local_pose = Pose(pose.p, pose.q)
        self.__dict__.pop("_mt_cache", None)

    def get_cmass_local_pose(self):
        return self.cmass_local_pose

    def set_cmass_local_pose(self, pose):
        self.cmass_local_pose = pose

    @property
    def inertia(self):
        if self._inertia is not None:
            return np.asarray(self._inertia, dtype=np.float32)
        return np.asarray(_mesh.principal(self._auto()[2])[0], dtype=np.float32)

    @inertia.setter
    def inertia(self, v):
        self._frozen_check()
        self._inertia = np.array(v, dtype=np.float32).reshape(3)
        self.__dict__.pop("_mt_cache", None)

    def get_inertia(self):
        return self.inertia

    def set_inertia(self, v):
        self.inertia = v

    def _frozen_check(self):
        if self._system is not None and self._system._initialized:
            raise RuntimeError("body properties cannot be changed after the simulation was initialised")

    # -- simple accessors -------------------------------------------------------------------------------------------
    def get_linear_damping(self):
        return self.linear_damping

    def set_linear_damping(self, v):
        self.linear_damping = float(v)

    def get_angular_damping(self):
        return self.angular_damping

    def set_angular_damping(self, v):
        self.angular_damping = float(v)

    def get_disable_gravity(self):
        return self.disable_gravity

    def set_disable_gravity(self, v):
        self.disable_gravity = bool(v)

    # -- velocities (CPU-style access; on the batched system they read / write the state buffers) ---------------------
    @property
    def linear_velocity(self):
        if self._system is not None and self._system._live():
            return self._system._read_body_row(self)[7:10]
        return self._lin_vel.copy()

    @property
    def angular_velocity(self):
        if self._system is not None and self._system._live():
            return self._system._read_body_row(self)[10:13]
        return self._ang_vel.copy()

    def get_linear_velocity(self):
        return self.linear_velocity

    def get_angular_velocity(self):
        return self.angular_velocity

    def add_force_at_point(self, force, point, mode="force"):
        if self._system is None or not self._system._live():
            raise RuntimeError("add_force_at_point before the simulation was initialised")
        self._system._add_force_at_point(self, np.asarray(force, dtype=np.float32), np.asarray(point, dtype=np.float32))

    def add_force_torque(self, force, torque, mode="force"):
        self._system._add_force_torque(self, np.asarray(force, dtype=np.float32), np.asarray(torque, dtype=np.float32))


class PhysxRigidDynamicComponent(PhysxRigidBodyComponent):
    def __init__(self):
        super().__init__()
        self.kinematic = False
        self.locked_motion_axes = [False] * 6
        self._kinematic_target = None

    @property
    def gpu_index(self):
        return self._gpu_pose_index()

    @property
    def gpu_pose_index(self):
        return self._gpu_pose_index()

    def _gpu_pose_index(self):
        if self._system is None or not self._system._initialized:
            return -1       # as SAPIEN: meaningful after gpu_init() only
        return self._system._pose_index(self)

    def get_gpu_index(self):
        return self.gpu_index

    def get_gpu_pose_index(self):
        return self.gpu_pose_index

    def get_kinematic(self):
        return self.kinematic

    def set_kinematic(self, v):
        self._frozen_check()
        self.kinematic = bool(v)

    def set_locked_motion_axes(self, axes):
        self._frozen_check()
        self.locked_motion_axes = [bool(a) for a in axes]

    def get_locked_motion_axes(self):
        return list(self.locked_motion_axes)

    @property
    def is_sleeping(self):
        return False      # bodies never sleep in this backend (always awake)

    def wake_up(self):
        pass

    def put_to_sleep(self):
        pass

    # velocity setters (dynamic bodies only)
    @PhysxRigidBodyComponent.linear_velocity.setter
    def linear_velocity(self, v):
        if self._system is not None and self._system._live():
            self._system._write_body_cols(self, 7, np.asarray(v, dtype=np.float32))
        else:
            self._lin_vel = np.array(v, dtype=np.float32).reshape(3)

    @PhysxRigidBodyComponent.angular_velocity.setter
    def angular_velocity(self, v):
        if self._system is not None and self._system._live():
            self._system._write_body_cols(self, 10, np.asarray(v, dtype=np.float32))
        else:
            self._ang_vel = np.array(v, dtype=np.float32).reshape(3)

    def set_linear_velocity(self, v):
        self.linear_velocity = v

    def set_angular_velocity(self, v):
        self.angular_velocity = v

    @property
    def kinematic_target(self):
        return self._kinematic_target

    @kinematic_target.setter
    def kinematic_target(self, pose):
        self._kinematic_target = pose
        self.entity.pose = pose

    def set_kinematic_target(self, pose):
        self.kinematic_target = pose


class PhysxArticulationJoint:
    def __init__(self, child_link: "PhysxArticulationLinkComponent", parent_link: Optional["PhysxArticulationLinkComponent"]):
        self.name = ""
        self.child_link = child_link
        self.parent_link = parent_link
        self._type = "fixed" if parent_link is not None else "undefined"
        self.pose_in_child = Pose()
        self.pose_in_parent = Pose()
        self._limits = _FREE_LIMITS  # replaced, never written in place
        self.stiffness, self.damping, self.force_limit, self.drive_mode = 0.0, 0.0, 3.4028234663852886e38, "force"
        self.friction = 0.0
        self._armature = 0.0
        self._drive_target = 0.0
        self._drive_velocity_target = 0.0

    def _frozen_check(self, what):
        art = self.child_link.articulation
        if art._system is not None and art._system._initialized:
            raise RuntimeError(f"{what} cannot be changed after the simulation was initialised")

    @property
    def type(self):
        return self._type

    @type.setter
    def type(self, t):
        if t == "continuous":
            t = "revolute_unwrapped"
        if t not in ("fixed", "revolute", "revolute_unwrapped", "prismatic", "free", "undefined"):
            raise ValueError(f"invalid joint type {t}")
        self._frozen_check("joint type")
        self._type = t
        if self.dof == 0:
            self._limits = np.zeros((0, 2), dtype=np.float32)
        elif self._limits.shape[0] != self.dof:
            self._limits = np.array([[-np.inf, np.inf]] * self.dof, dtype=np.float32)

    def get_type(self):
        return self._type

    def set_type(self, t):
        self.type = t

    @property
    def dof(self) -> int:
        return 1 if self._type in ("revolute", "revolute_unwrapped", "prismatic") else 0

    def get_dof(self):
        return self.dof

    @property
    def limits(self):
        return self._limits.copy() if self.dof else np.zeros((0, 2), dtype=np.float32)

    @limits.setter
    def limits(self, v):
        self._frozen_check("joint limits")
        self._limits = np.array(v, dtype=np.float32).reshape(-1, 2)

    limit = limits

    def get_limits(self):
        return self.limits

    def set_limits(self, v):
        self.limits = v

    get_limit, set_limit = get_limits, set_limits

    def set_drive_properties(self, stiffness, damping, force_limit=3.4028234663852886e38, mode="force"):
        self.stiffness, self.damping, self.force_limit, self.drive_mode = float(stiffness), float(damping), float(force_limit), mode
        art = self.child_link.articulation
        if art._system is not None and art._system._initialized:
            art._system._drive_changed(self)

    set_drive_property = set_drive_properties

    def get_stiffness(self):
        return self.stiffness

    def get_damping(self):
        return self.damping

    def get_force_limit(self):
        return self.force_limit

    def get_drive_mode(self):
        return self.drive_mode

    def get_friction(self):
        return self.friction

    def set_friction(self, v):
        self._frozen_check("joint friction")
        self.friction = float(v)

    @property
    def armature(self):
        return np.full(self.dof, self._armature, dtype=np.float32)

    @armature.setter
    def armature(self, v):
        self._frozen_check("joint armature")
        a = np.asarray(v, dtype=np.float32).reshape(-1)
        self._armature = float(a[0]) if a.size else 0.0

    def get_armature(self):
        return self.armature

    def set_armature(self, v):
        self.armature = v

    # drive targets: host-side values; the batched system keeps them in cuda_articulation_target_qpos / _qvel
    @property
    def drive_target(self):
        art = self.child_link.articulation
        if art._system is not None and art._system._live() and self.dof:
            return np.array([art._system._read_dof(art, "target_qpos", self)], dtype=np.float32)
        return np.array([self._drive_target] * self.dof, dtype=np.float32)

    @drive_target.setter
    def drive_target(self, v):
        v = float(np.asarray(v, dtype=np.float32).reshape(-1)[0])
        self._drive_target = v
        art = self.child_link.articulation
        if art._system is not None and art._system._live() and self.dof:
            art._system._write_dof(art, "target_qpos", self, v)

    @property
    def drive_velocity_target(self):
        art = self.child_link.articulation
        if art._system is not None and art._system._live() and self.dof:
            return np.array([art._system._read_dof(art, "target_qvel", self)], dtype=np.float32)
        return np.array([self._drive_velocity_target] * self.dof, dtype=np.float32)

    @drive_velocity_target.setter
    def drive_velocity_target(self, v):
        v = float(np.asarray(v, dtype=np.float32).reshape(-1)[0])
        self._drive_velocity_target = v
        art = self.child_link.articulation
        if art._system is not None and art._system._live() and self.dof:
            art._system._write_dof(art, "target_qvel", self, v)

    def get_drive_target(self):
        return self.drive_target

    def set_drive_target(self, v):
        self.drive_target = v

    def get_drive_velocity_target(self):
        return self.drive_velocity_target

    def set_drive_velocity_target(self, v):
        self.drive_velocity_target = v

    def get_name(self):
        return self.name

    def set_name(self, n):
        self.name = n

    def get_child_link(self):
        return self.child_link

    def get_parent_link(self):
        return self.parent_link

    def get_pose_in_child(self):
        return self.pose_in_child

    def get_pose_in_parent(self):
        return self.pose_in_parent

    def set_pose_in_child(self, p):
        self.pose_in_child = p

    def set_pose_in_parent(self, p):
        self.pose_in_parent = p

    @property
    def global_pose(self):
        return self.child_link.pose * self.pose_in_child

    def get_global_pose(self):
        return self.global_pose


class PhysxArticulation:
    """Created implicitly by the root ``PhysxArticulationLinkComponent(None)``; children join their parent's."""

    def __init__(self):
        self.name = ""
        self.links: list[PhysxArticulationLinkComponent] = []
        self._tendons = []          # (link chain, coefficients, recip coefficients, rest_length, stiffness, damping, ...)
        self._system: Optional["PhysxSystem"] = None
        self._env = -1
        self._art_id = -1
        self._root_pose = Pose()
        self._qpos0 = None

    # -- structure ----------------------------------------------------------------------------------------------
    @property
    def root(self):
        return self.links[0]

    def get_root(self):
        return self.root

    @property
    def joints(self):
        return [l.joint for l in self.links]

    def get_joints(self):
        return self.joints

    def get_links(self):
        return self.links

    @property
    def active_joints(self):
        return [j for j in self.joints if j.dof > 0]

    def get_active_joints(self):
        return self.active_joints

    @property
    def dof(self) -> int:
        return sum(j.dof for j in self.joints)

    def get_dof(self):
        return self.dof

    def find_joint_by_name(self, name):
        for j in self.joints:
            if j.name == name:
                return j
        return None

    def find_link_by_name(self, name):
        for l in self.links:
            if l.name == name:
                return l
        return None

    def get_name(self):
        return self.name

    def set_name(self, n):
        self.name = n

    @property
    def qlimits(self):
        lim = [j.limits for j in self.active_joints]
        return np.concatenate(lim).astype(np.float32) if lim else np.zeros((0, 2), dtype=np.float32)

    def get_qlimits(self):
        return self.qlimits

    get_qlimit = get_qlimits
    qlimit = qlimits

    def create_fixed_tendon(self, link_chain, coefficients, recip_coefficients, rest_length=0.0, offset=0.0, stiffness=0.0,
                            damping=0.0, low=-3.4028234663852886e38, high=3.4028234663852886e38, limit_stiffness=0.0):
        if self._system is not None and self._system._initialized:
            raise RuntimeError("tendons cannot be created after the simulation was initialised")
        self._tendons.append(dict(chain=list(link_chain), coef=[float(c) for c in coefficients],
                                  recip=[float(c) for c in recip_coefficients], rest_length=float(rest_length), offset=float(offset),
                                  stiffness=float(stiffness), damping=float(damping), low=float(low), high=float(high),
                                  limit_stiffness=float(limit_stiffness)))

    # -- indices ----------------------------------------------------------------------------------------------------
    @property
    def gpu_index(self):
        if self._system is None or not self._system._initialized:
            return -1
        return self._system._art_index(self)

    def get_gpu_index(self):
        return self.gpu_index

    # -- pose / state (host-side before init; the batched buffers afterwards) ------------------------------------------
    @property
    def pose(self):
        return self.root.entity.pose if self.root.entity is not None else self._root_pose

    @pose.setter
    def pose(self, pose):
        self._root_pose = Pose(pose.p, pose.q)
        if self.root.entity is not None:
            self.root.entity.pose = pose

    root_pose = pose

    def get_pose(self):
        return self.pose

    def set_pose(self, p):
        self.pose = p

    get_root_pose, set_root_pose = get_pose, set_pose

    def _vec(self, name):
        if self._system is not None and self._system._live():
            return self._system._read_art_vec(self, name)
        if name == "qpos" and self._qpos0 is not None:
            return np.asarray(self._qpos0, dtype=np.float32)
        return np.zeros(self.dof, dtype=np.float32)

    def _set_vec(self, name, v):
        v = np.asarray(v, dtype=np.float32).reshape(-1)
        if self._system is not None and self._system._live():
            self._system._write_art_vec(self, name, v)
        elif name == "qpos":
            self._qpos0 = v.copy()

    qpos = property(lambda s: s._vec("qpos"), lambda s, v: s._set_vec("qpos", v))
    qvel = property(lambda s: s._vec("qvel"), lambda s, v: s._set_vec("qvel", v))
    qf = property(lambda s: s._vec("qf"), lambda s, v: s._set_vec("qf", v))
    qacc = property(lambda s: s._vec("qacc"), lambda s, v: None)

    def get_qpos(self):
        return self.qpos

    def set_qpos(self, v):
        self.qpos = v

    def get_qvel(self):
        return self.qvel

    def set_qvel(self, v):
        self.qvel = v

    def get_qf(self):
        return self.qf

    def set_qf(self, v):
        self.qf = v

    def get_qacc(self):
        return self.qacc

    def _root_floating(self):
        return self.root.joint._type != "fixed"

    def set_root_linear_velocity(self, v):
        if self._root_floating() and self._system is not None and self._system._initialized:
            self._system._write_root_velocity(self.root, 7, np.asarray(v, dtype=np.float32))

    def set_root_angular_velocity(self, v):
        if self._root_floating() and self._system is not None and self._system._initialized:
            self._system._write_root_velocity(self.root, 10, np.asarray(v, dtype=np.float32))

    def get_root_linear_velocity(self):
        if self._root_floating() and self._system is not None and self._system._initialized:
            return self._system._read_body_row(self.root)[7:10]
        return np.zeros(3, dtype=np.float32)

    def get_root_angular_velocity(self):
        if self._root_floating() and self._system is not None and self._system._initialized:
            return self._system._read_body_row(self.root)[10:13]
        return np.zeros(3, dtype=np.float32)

    def get_link_incoming_joint_forces(self):
        return self._system._read_link_joint_forces(self)

    def compute_passive_force(self, gravity=True, coriolis_and_centrifugal=True):
        raise NotImplementedError("compute_passive_force is not provided by this backend")

    def create_pinocchio_model(self):
        from .wrapper.pinocchio_model import PinocchioModel
        return PinocchioModel._from_articulation(self)


class PhysxArticulationLinkComponent(PhysxRigidBodyComponent):
    def __init__(self, parent: Optional["PhysxArticulationLinkComponent"] = None):
        super().__init__()
        self.parent = parent
        self.children: list[PhysxArticulationLinkComponent] = []
        if parent is None:
            self.articulation = PhysxArticulation()
        else:
            self.articulation = parent.articulation
            parent.children.append(self)
        self.index = len(self.articulation.links)
        self.articulation.links.append(self)
        self.joint = PhysxArticulationJoint(self, parent)
        self.angular_damping = 0.0

    @property
    def is_root(self):
        return self.parent is None

    def get_index(self):
        return self.index

    def get_parent(self):
        return self.parent

    def get_children(self):
        return self.children

    def get_joint(self):
        return self.joint

    def get_articulation(self):
        return self.articulation

    @property
    def gpu_pose_index(self):
        if self._system is None or not self._system._initialized:
            return -1
        return self._system._pose_index(self)

    def get_gpu_pose_index(self):
        return self.gpu_pose_index

    @property
    def sleeping(self):
        return False

    def _on_add_to_scene(self, scene):
        super()._on_add_to_scene(scene)
        art = self.articulation
        art._system, art._env = self._system, self._env
        if self.is_root:
            # entity.pose of the root was set before / is the articulation pose
            art._root_pose = self.entity._pose

    def put_to_sleep(self):
        pass

    def wake_up(self):
        pass


PhysxArticulationLink = PhysxArticulationLinkComponent


class PhysxJointComponent(PhysxBaseComponent):
    def __init__(self, body):
        super().__init__()
        self.child = body
        self.parent = None
        self.pose_in_parent, self.pose_in_child = Pose(), Pose()

    def _on_add_to_scene(self, scene):
        raise RuntimeError(f"{type(self).__name__}: body-to-body joints (drives, gears, distance joints) are not provided by this backend")


class PhysxDriveComponent(PhysxJointComponent):
    pass


class PhysxGearComponent(PhysxJointComponent):
    pass


class PhysxDistanceJointComponent(PhysxJointComponent):
    pass


# contact records of the CPU-style API (PhysxCpuSystem.get_contacts)
class PhysxContactPoint:
    def __init__(self, position, normal, impulse, separation):
        self.position, self.normal, self.impulse, self.separation = position, normal, impulse, separation


class PhysxContact:
    def __init__(self, bodies, shapes, points):
        self.bodies, self.shapes, self.points = bodies, shapes, points
        self.components = bodies


class PhysxGpuContactPairImpulseQuery:
    def __init__(self, qid, handle):
        self.id = qid
        self.cuda_impulses = handle


class PhysxGpuContactBodyImpulseQuery(PhysxGpuContactPairImpulseQuery):
    pass


from ._system import PhysxCpuSystem, PhysxGpuSystem, PhysxSystem  # noqa: E402,F401
