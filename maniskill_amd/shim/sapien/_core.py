"""sapien.Entity / Component / Scene / Device: the object graph ManiSkill's builders populate.

Reference call sites: mani_skill/envs/sapien_env.py:1182-1230 (``sapien.Scene(systems=[...])`` per sub-scene),
utils/building/actor_builder.py:193-261 (``Entity()``, ``add_component``, ``entity.pose = ...``, ``sub_scene.add_entity``),
utils/structs/actor.py:62-95 (``find_component_by_type``), envs/sapien_env.py:1254-1265 (``per_scene_id``).
Host-only, cold path: these objects just record what was built; the physics system compiles them at ``gpu_init``.
"""
from __future__ import annotations

from typing import Optional

from ._pose import Pose


class Device:
    """sapien.Device("cuda" | "cuda:k" | "cpu" | "pci:...") (mani_skill/envs/utils/system/backend.py:60-91)."""

    def __init__(self, name: str):
        self.name = str(name)
        kind = self.name.split(":")[0]
        if kind not in ("cuda", "cpu", "pci"):
            raise RuntimeError(f'failed to find device "{name}"')
        self._kind = kind
        self.cuda_id = int(self.name.split(":")[1]) if (kind == "cuda" and ":" in self.name) else (0 if kind == "cuda" else -1)
        if kind == "cuda":
            import torch
            if not torch.cuda.is_available():
                raise RuntimeError('failed to find device "cuda"')

    def is_cuda(self) -> bool:
        return self._kind == "cuda"

    def is_cpu(self) -> bool:
        return self._kind == "cpu"

    def can_render(self) -> bool:
        return True

    def can_present(self) -> bool:
        return False

    @property
    def pci_string(self):
        return self.name

    def __repr__(self):
        return f"Device({self.name!r})"


class Component:
    def __init__(self):
        self.entity: Optional["Entity"] = None
        self.name = ""
        self._enabled = True

    # pose of a component == pose of its entity (sapien.Component.pose)
    @property
    def pose(self) -> Pose:
        return self.entity.pose if self.entity is not None else Pose()

    @pose.setter
    def pose(self, pose: Pose):
        if self.entity is not None:
            self.entity.pose = pose

    @property
    def entity_pose(self) -> Pose:
        return self.pose

    def get_pose(self):
        return self.pose

    def set_pose(self, pose):
        self.pose = pose

    def get_entity_pose(self):
        return self.pose

    def set_entity_pose(self, pose):
        self.pose = pose

    def get_name(self):
        return self.name

    def set_name(self, name):
        self.name = name
        return self

    def get_entity(self):
        return self.entity

    def enable(self):
        self._enabled = True

    def disable(self):
        self._enabled = False

    @property
    def is_enabled(self):
        return self._enabled

    def _on_add_to_scene(self, scene: "Scene"):
        pass

    def _on_remove_from_scene(self, scene: "Scene"):
        pass


class Entity:
    def __init__(self):
        self.name = ""
        self.components: list[Component] = []
        self.scene: Optional["Scene"] = None
        self.per_scene_id = 0
        self._pose = Pose()

    # -- pose ---------------------------------------------------------------------------------------------------
    @property
    def pose(self) -> Pose:
        px = self._physx_body()
        if px is not None and px._system is not None and px._system._live():
            return px._system._read_body_pose(px)
        return self._pose

    @pose.setter
    def pose(self, pose: Pose):
        self._pose = Pose(pose.p, pose.q)
        px = self._physx_body()
        if px is not None and px._system is not None and px._system._live():
            px._system._write_body_pose(px, self._pose)

    def get_pose(self):
        return self.pose

    def set_pose(self, pose):
        self.pose = pose
        return self

    def _physx_body(self):
        for c in self.components:
            if getattr(c, "_is_physx_body", False):
                return c
        return None

    # -- components ---------------------------------------------------------------------------------------------
    def add_component(self, component: Component):
        if component.entity is not None and component.entity is not self:
            raise RuntimeError("component already belongs to another entity")
        component.entity = self
        self.components.append(component)
        if self.scene is not None:
            component._on_add_to_scene(self.scene)
        return self

    def remove_component(self, component: Component):
        self.components.remove(component)
        if self.scene is not None:
            component._on_remove_from_scene(self.scene)
        component.entity = None

    def get_components(self):
        return self.components

    def find_component_by_type(self, cls):
        for c in self.components:
            if isinstance(c, cls):
                return c
        return None

    # -- names / scene ------------------------------------------------------------------------------------------
    def get_name(self):
        return self.name

    def set_name(self, name):
        self.name = name
        return self

    def get_scene(self):
        return self.scene

    def get_per_scene_id(self):
        return self.per_scene_id

    def add_to_scene(self, scene: "Scene"):
        scene.add_entity(self)
        return self

    def remove_from_scene(self):
        if self.scene is not None:
            self.scene.remove_entity(self)

    def __repr__(self):
        return f"<sapien.Entity {self.name!r}>"


class Scene:
    """One sub-scene: an ordered list of entities attached to a (shared) physics system and an own render system."""

    def __init__(self, systems=None):
        from . import physx as _physx
        from . import render as _render
        if systems is None:
            systems = [_physx.PhysxCpuSystem(), _render.RenderSystem()]
        self.systems = list(systems)
        self.physx_system = next((s for s in self.systems if isinstance(s, _physx.PhysxSystem)), None)
        self.render_system = next((s for s in self.systems if isinstance(s, _render.RenderSystem)), None)
        self.entities: list[Entity] = []
        self._next_per_scene_id = 1        # 0 is the background of the segmentation texture
        self.name = ""
        if self.physx_system is not None:
            self.physx_system._register_scene(self)
        if self.render_system is not None:
            self.render_system._scene = self

    def get_physx_system(self):
        return self.physx_system

    def get_render_system(self):
        return self.render_system

    def add_entity(self, entity: Entity):
        if entity.scene is not None:
            raise RuntimeError("entity already added to a scene")
        if self.physx_system is not None and getattr(self.physx_system, "_initialized", False) and entity._physx_body() is not None:
            raise RuntimeError("cannot add physical entities after gpu_init()")
        entity.scene = self
        entity.per_scene_id = self._next_per_scene_id
        self._next_per_scene_id += 1
        self.entities.append(entity)
        for c in entity.components:
            c._on_add_to_scene(self)
        return self

    def remove_entity(self, entity: Entity):
        if self.physx_system is not None and getattr(self.physx_system, "_initialized", False) and entity._physx_body() is not None:
            raise RuntimeError("cannot remove physical entities after the simulation was initialised")
        self.entities.remove(entity)
        for c in entity.components:
            c._on_remove_from_scene(self)
        entity.scene = None

    def get_entities(self):
        return self.entities

    # rendering hooks (CPU path of ManiSkillScene.update_render, envs/scene.py:410-427)
    def update_render(self):
        if self.render_system is not None:
            self.render_system._update_render_single()

    # lighting / environment: kept as data for the render system
    @property
    def ambient_light(self):
        return self.render_system.ambient_light if self.render_system is not None else [0, 0, 0]

    @ambient_light.setter
    def ambient_light(self, color):
        if self.render_system is not None:
            self.render_system.ambient_light = color

    def set_ambient_light(self, color):
        self.ambient_light = color

    def set_environment_map(self, *a, **k):
        pass

    def create_drive(self, body0, pose0, body1, pose1):
        from . import physx as _physx
        drive = _physx.PhysxDriveComponent(body1)
        drive.parent = body0
        drive.pose_in_parent, drive.pose_in_child = pose0, pose1
        body1.entity.add_component(drive)
        return drive

    def step(self):
        self.physx_system.step()

    @property
    def timestep(self):
        return self.physx_system.timestep

    @timestep.setter
    def timestep(self, v):
        self.physx_system.timestep = v
