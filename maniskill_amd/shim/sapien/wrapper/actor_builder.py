"""sapien.ActorBuilder: the record-keeping base class ManiSkill's ActorBuilder subclasses
(mani_skill/utils/building/actor_builder.py:21-164 reads ``collision_records[i].{type,pose,scale,radius,length,filename,
material,density,patch_radius,min_patch_radius,decomposition,decomposition_params}``, ``visual_records``,
``collision_groups``, ``physx_body_type``, ``_mass/_cmass_local_pose/_inertia/_auto_inertial``, ``name``, ``scene``)."""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Optional

import numpy as np

from .._core import Entity
from .._pose import Pose
from .. import physx, render


@dataclass
class CollisionShapeRecord:
    type: str                      # plane | box | capsule | cylinder | sphere | convex_mesh | nonconvex_mesh | multiple_convex_meshes
    filename: str = ""
    scale: tuple = (1, 1, 1)       # box: half sizes; meshes: scale
    radius: float = 1.0
    length: float = 1.0            # capsule / cylinder: HALF length
    material: Optional[physx.PhysxMaterial] = None
    pose: Pose = field(default_factory=Pose)
    density: float = 1000.0
    patch_radius: float = 0.0
    min_patch_radius: float = 0.0
    is_trigger: bool = False
    decomposition: str = "none"
    decomposition_params: Optional[dict] = None


@dataclass
class VisualShapeRecord:
    type: str                      # file | plane | box | capsule | cylinder | sphere
    filename: str = ""
    scale: tuple = (1, 1, 1)
    radius: float = 1.0
    length: float = 1.0
    material: Optional[render.RenderMaterial] = None
    pose: Pose = field(default_factory=Pose)
    name: str = ""


def _material(m):
    if m is None:
        return render.RenderMaterial()
    if isinstance(m, render.RenderMaterial):
        return m
    arr = list(np.asarray(m, dtype=np.float32).reshape(-1))   # a colour
    if len(arr) == 3:
        arr.append(1.0)
    return render.RenderMaterial(base_color=arr)


class ActorBuilder:
    def __init__(self):
        self.scene = None
        self.name = ""
        self.physx_body_type = "dynamic"
        self.collision_groups = [1, 1, 0, 0]
        self.collision_records: list[CollisionShapeRecord] = []
        self.visual_records: list[VisualShapeRecord] = []
        self._mass = 1.0
        self._cmass_local_pose = Pose()
        self._inertia = np.ones(3, dtype=np.float32)
        self._auto_inertial = True
        self.initial_pose = Pose()

    # -- configuration ------------------------------------------------------------------------------------------------
    def set_scene(self, scene):
        self.scene = scene
        return self

    def set_name(self, name):
        self.name = name
        return self

    def set_initial_pose(self, pose):
        self.initial_pose = pose
        return self

    def set_physx_body_type(self, t):
        assert t in ("dynamic", "kinematic", "static", "link"), t
        self.physx_body_type = t
        return self

    def set_mass_and_inertia(self, mass, cmass_local_pose, inertia):
        self._mass, self._cmass_local_pose, self._inertia = float(mass), cmass_local_pose, np.array(inertia, dtype=np.float32)
        self._auto_inertial = False
        return self

    def set_collision_groups(self, *groups):
        if len(groups) == 1:
            groups = groups[0]
        self.collision_groups = [int(g) for g in groups]
        return self

    def reset_collision_groups(self):
        self.collision_groups = [1, 1, 0, 0]
        return self

    # -- collision records ------------------------------------------------------------------------------------------------
    def _mat(self, material):
        return material if material is not None else physx.get_default_material()

    def add_plane_collision(self, pose=None, material=None, patch_radius=0, min_patch_radius=0, is_trigger=False):
        self.collision_records.append(CollisionShapeRecord("plane", pose=pose or Pose(), material=self._mat(material), density=0,
                                                           patch_radius=patch_radius, min_patch_radius=min_patch_radius, is_trigger=is_trigger))
        return self

    def add_box_collision(self, pose=None, half_size=(1, 1, 1), material=None, density=1000, patch_radius=0, min_patch_radius=0,
                          is_trigger=False):
        self.collision_records.append(CollisionShapeRecord("box", pose=pose or Pose(), scale=tuple(float(x) for x in half_size),
                                                           material=self._mat(material), density=density, patch_radius=patch_radius,
                                                           min_patch_radius=min_patch_radius, is_trigger=is_trigger))
        return self

    def add_capsule_collision(self, pose=None, radius=1, half_length=1, material=None, density=1000, patch_radius=0,
                              min_patch_radius=0, is_trigger=False):
        self.collision_records.append(CollisionShapeRecord("capsule", pose=pose or Pose(), radius=radius, length=half_length,
                                                           material=self._mat(material), density=density, patch_radius=patch_radius,
                                                           min_patch_radius=min_patch_radius, is_trigger=is_trigger))
        return self

    def add_cylinder_collision(self, pose=None, radius=1, half_length=1, material=None, density=1000, patch_radius=0,
                               min_patch_radius=0, is_trigger=False):
        self.collision_records.append(CollisionShapeRecord("cylinder", pose=pose or Pose(), radius=radius, length=half_length,
                                                           material=self._mat(material), density=density, patch_radius=patch_radius,
                                                           min_patch_radius=min_patch_radius, is_trigger=is_trigger))
        return self

    def add_sphere_collision(self, pose=None, radius=1, material=None, density=1000, patch_radius=0, min_patch_radius=0,
                             is_trigger=False):
        self.collision_records.append(CollisionShapeRecord("sphere", pose=pose or Pose(), radius=radius, material=self._mat(material),
                                                           density=density, patch_radius=patch_radius,
                                                           min_patch_radius=min_patch_radius, is_trigger=is_trigger))
        return self

    def _scale3(self, scale):
        s = np.asarray(scale, dtype=np.float32).reshape(-1)
        return tuple(float(x) for x in (np.full(3, s[0]) if s.size == 1 else s))

    def add_convex_collision_from_file(self, filename, pose=None, scale=(1, 1, 1), material=None, density=1000, patch_radius=0,
                                       min_patch_radius=0, is_trigger=False):
        self.collision_records.append(CollisionShapeRecord("convex_mesh", filename=str(filename), pose=pose or Pose(),
                                                           scale=self._scale3(scale), material=self._mat(material), density=density,
                                                           patch_radius=patch_radius, min_patch_radius=min_patch_radius,
                                                           is_trigger=is_trigger))
        return self

    def add_multiple_convex_collisions_from_file(self, filename, pose=None, scale=(1, 1, 1), material=None, density=1000,
                                                 patch_radius=0, min_patch_radius=0, is_trigger=False, decomposition="none",
                                                 decomposition_params=None):
        self.collision_records.append(CollisionShapeRecord("multiple_convex_meshes", filename=str(filename), pose=pose or Pose(),
                                                           scale=self._scale3(scale), material=self._mat(material), density=density,
                                                           patch_radius=patch_radius, min_patch_radius=min_patch_radius,
                                                           is_trigger=is_trigger, decomposition=decomposition,
                                                           decomposition_params=decomposition_params))
        return self

    add_multiple_convex_collisions_from_file.__doc__ = "one convex shape per connected part of the file"

    def add_nonconvex_collision_from_file(self, filename, pose=None, scale=(1, 1, 1), material=None, patch_radius=0,
                                          min_patch_radius=0, is_trigger=False):
        self.collision_records.append(CollisionShapeRecord("nonconvex_mesh", filename=str(filename), pose=pose or Pose(),
                                                           scale=self._scale3(scale), material=self._mat(material), density=0,
                                                           patch_radius=patch_radius, min_patch_radius=min_patch_radius,
                                                           is_trigger=is_trigger))
        return self

    # -- visual records ------------------------------------------------------------------------------------------------------
    def add_plane_visual(self, pose=None, scale=(1, 1, 1), material=None, name=""):
        self.visual_records.append(VisualShapeRecord("plane", pose=pose or Pose(), scale=self._scale3(scale), material=_material(material), name=name))
        return self

    def add_box_visual(self, pose=None, half_size=(1, 1, 1), material=None, name=""):
        self.visual_records.append(VisualShapeRecord("box", pose=pose or Pose(), scale=tuple(float(x) for x in half_size),
                                                     material=_material(material), name=name))
        return self

    def add_capsule_visual(self, pose=None, radius=1, half_length=1, material=None, name=""):
        self.visual_records.append(VisualShapeRecord("capsule", pose=pose or Pose(), radius=radius, length=half_length,
                                                     material=_material(material), name=name))
        return self

    def add_cylinder_visual(self, pose=None, radius=1, half_length=1, material=None, name=""):
        self.visual_records.append(VisualShapeRecord("cylinder", pose=pose or Pose(), radius=radius, length=half_length,
                                                     material=_material(material), name=name))
        return self

    def add_sphere_visual(self, pose=None, radius=1, material=None, name=""):
        self.visual_records.append(VisualShapeRecord("sphere", pose=pose or Pose(), radius=radius, material=_material(material), name=name))
        return self

    def add_visual_from_file(self, filename, pose=None, scale=(1, 1, 1), material=None, name=""):
        self.visual_records.append(VisualShapeRecord("file", filename=str(filename), pose=pose or Pose(), scale=self._scale3(scale),
                                                     material=None if material is None else _material(material), name=name))
        return self

    # -- building ------------------------------------------------------------------------------------------------------
    def build_render_component(self):
        component = render.RenderBodyComponent()
        for r in self.visual_records:
            if r.type == "plane":
                shape = render.RenderShapePlane(r.scale, r.material)
            elif r.type == "box":
                shape = render.RenderShapeBox(r.scale, r.material)
            elif r.type == "sphere":
                shape = render.RenderShapeSphere(r.radius, r.material)
            elif r.type == "capsule":
                shape = render.RenderShapeCapsule(r.radius, r.length, r.material)
            elif r.type == "cylinder":
                shape = render.RenderShapeCylinder(r.radius, r.length, r.material)
            elif r.type == "file":
                shape = render.RenderShapeTriangleMesh(r.filename, r.scale, r.material)
            else:
                raise RuntimeError(f"invalid visual shape type [{r.type}]")
            shape.local_pose = r.pose
            shape.name = r.name
            component.attach(shape)
        return component

    def _collision_prototypes(self):
        """The shapes the collision records describe, constructed once per (records, collision groups) and cloned per sub-scene: ManiSkill
        calls build_physx_component once per sub-scene on the same builder (utils/building/articulation_builder.py:65-113)."""
        fp = (tuple(self.collision_groups), id(physx.get_default_material()),
              tuple((r.type, r.filename, np.asarray(r.scale, dtype=np.float64).tobytes(), r.radius, r.length, r.density, r.patch_radius,
                     r.min_patch_radius, id(r.material), r.pose._p.tobytes(), r.pose._q.tobytes()) for r in self.collision_records))
        cached = self.__dict__.get("_proto_cache")
        if cached is not None and cached[0] == fp:
            return cached[1]
        protos = []
        for r in self.collision_records:
            if r.type == "plane":
                shapes = [physx.PhysxCollisionShapePlane(material=r.material)]
            elif r.type == "box":
                shapes = [physx.PhysxCollisionShapeBox(half_size=r.scale, material=r.material)]
            elif r.type == "capsule":
                shapes = [physx.PhysxCollisionShapeCapsule(radius=r.radius, half_length=r.length, material=r.material)]
            elif r.type == "cylinder":
                shapes = [physx.PhysxCollisionShapeCylinder(radius=r.radius, half_length=r.length, material=r.material)]
            elif r.type == "sphere":
                shapes = [physx.PhysxCollisionShapeSphere(radius=r.radius, material=r.material)]
            elif r.type == "convex_mesh":
                shapes = [physx.PhysxCollisionShapeConvexMesh(filename=r.filename, scale=r.scale, material=r.material)]
            elif r.type == "nonconvex_mesh":
                shapes = [physx.PhysxCollisionShapeTriangleMesh(filename=r.filename, scale=r.scale, material=r.material)]
            elif r.type == "multiple_convex_meshes":
                shapes = physx.PhysxCollisionShapeConvexMesh.load_multiple(filename=r.filename, scale=r.scale, material=r.material)
            else:
                raise RuntimeError(f"invalid collision shape type [{r.type}]")
            for shape in shapes:
                shape.local_pose = r.pose
                shape.set_collision_groups(self.collision_groups)
                shape.set_density(r.density)
                shape.set_patch_radius(r.patch_radius)
                shape.set_min_patch_radius(r.min_patch_radius)
                shape._sig()             # cached on the prototype, inherited by its clones
                protos.append(shape)
        self.__dict__["_proto_cache"] = (fp, protos)
        return protos

    def build_physx_component(self, link_parent=None):
        """Plain-SAPIEN version (ManiSkill overrides it, actor_builder.py:57-164)."""
        if self.physx_body_type == "dynamic":
            component = physx.PhysxRigidDynamicComponent()
        elif self.physx_body_type == "kinematic":
            component = physx.PhysxRigidDynamicComponent()
            component.kinematic = True
        elif self.physx_body_type == "static":
            component = physx.PhysxRigidStaticComponent()
        elif self.physx_body_type == "link":
            component = physx.PhysxArticulationLinkComponent(link_parent)
        else:
            raise RuntimeError(f"invalid physx body type [{self.physx_body_type}]")
        for proto in self._collision_prototypes():
            component.attach(proto._clone())
        if not self._auto_inertial and self.physx_body_type != "kinematic":
            # all three at once: set one by one, `mass` alone would first freeze the shapes' own centre of mass and principal axes
            # (an eigen-decomposition per link per sub-scene) only for the next two lines to overwrite them
            component._mass = float(self._mass)
            component._cmass_local_pose = Pose(self._cmass_local_pose.p, self._cmass_local_pose.q)
            component._inertia = np.array(self._inertia, dtype=np.float32).reshape(3)
        component.name = self.name
        # loader-side data without a SAPIEN counterpart: the URDF's exact inertia tensor, SRDF-disabled partner links
        if hasattr(self, "_exact_inertial") and not self._auto_inertial:
            component._exact_inertial = self._exact_inertial
        component.__dict__.pop("_mt_cache", None)
        if not self._auto_inertial and self.physx_body_type != "kinematic":
            # what the engine takes for these three is the same for every sub-scene this builder is built into: worked out once
            key = (component._mass, component._cmass_local_pose._p.tobytes(), component._cmass_local_pose._q.tobytes(),
                   component._inertia.tobytes(), id(getattr(self, "_exact_inertial", None)))
            mc = self.__dict__.get("_mt_proto")
            if mc is None or mc[0] != key:
                mc = self.__dict__["_mt_proto"] = (key, component._mass_tensor())
            component.__dict__["_mt_cache"] = mc[1]
        elif self.physx_body_type != "kinematic":
            # mass properties from the shapes' densities: the shapes are clones of this builder's prototypes, one evaluation serves them all
            key = self.__dict__["_proto_cache"][0]
            mc = self.__dict__.get("_mt_proto")
            if mc is None or mc[0] != key:
                mc = self.__dict__["_mt_proto"] = (key, component._mass_tensor())
            component.__dict__["_mt_cache"] = mc[1]
        if hasattr(self, "_srdf_disabled"):
            component._srdf_disabled = set(self._srdf_disabled)
        return component

    def build_entity(self):
        entity = Entity()
        if self.visual_records:
            entity.add_component(self.build_render_component())
        entity.add_component(self.build_physx_component())
        entity.name = self.name
        return entity

    def build(self, name=""):
        if name:
            self.set_name(name)
        entity = self.build_entity()
        entity.pose = self.initial_pose
        self.scene.add_entity(entity)
        return entity

    def build_kinematic(self, name=""):
        self.set_physx_body_type("kinematic")
        return self.build(name=name)

    def build_static(self, name=""):
        self.set_physx_body_type("static")
        return self.build(name=name)
