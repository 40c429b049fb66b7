"""sapien.wrapper.articulation_builder: LinkBuilder / ArticulationBuilder / MimicJointRecord, the base classes of ManiSkill's
ArticulationBuilder (mani_skill/utils/building/articulation_builder.py:23-205 reads ``link_builders``, ``mimic_joint_records``,
``joint_record.{name,joint_type,pose_in_child,pose_in_parent,limits,damping,friction}``, ``b.index``, ``b.parent``, ``b._check()``)."""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Optional

import numpy as np

from .._core import Entity
from .._pose import Pose
from .actor_builder import ActorBuilder


@dataclass
class JointRecord:
    joint_type: str = "undefined"       # fixed | revolute | revolute_unwrapped | prismatic | undefined (root)
    limits: tuple = ()
    pose_in_parent: Pose = field(default_factory=Pose)
    pose_in_child: Pose = field(default_factory=Pose)
    friction: float = 0.0
    damping: float = 0.0
    name: str = ""


@dataclass
class MimicJointRecord:
    joint: str
    mimic: str
    multiplier: float
    offset: float


class LinkBuilder(ActorBuilder):
    def __init__(self, index: int, parent: Optional["LinkBuilder"] = None):
        super().__init__()
        self.index = index
        self.parent = parent
        self.joint_record = JointRecord()
        self.physx_body_type = "link"

    def set_joint_name(self, name):
        self.joint_record.name = name
        return self

    def set_joint_properties(self, type, limits, pose_in_parent=None, pose_in_child=None, friction=0, damping=0):
        assert type in ("fixed", "revolute", "revolute_unwrapped", "continuous", "prismatic", "free", "undefined"), type
        if type == "continuous":
            type = "revolute_unwrapped"
        self.joint_record.joint_type = type
        self.joint_record.limits = limits
        self.joint_record.pose_in_parent = pose_in_parent or Pose()
        self.joint_record.pose_in_child = pose_in_child or Pose()
        self.joint_record.friction = friction
        self.joint_record.damping = damping
        return self

    def _check(self):
        t = self.joint_record.joint_type
        if self.parent is None:
            return
        if t in ("revolute", "prismatic"):
            assert np.asarray(self.joint_record.limits).size == 2, f"joint {self.joint_record.name}: {t} joints need [[low, high]]"
        if t == "undefined":
            raise RuntimeError(f"link {self.name}: a non-root link needs a joint type")

    def build_entity(self):
        raise NotImplementedError("links are built by their ArticulationBuilder")

    build = build_entity


class ArticulationBuilder:
    def __init__(self):
        self.scene = None
        self.link_builders: list[LinkBuilder] = []
        self.mimic_joint_records: list[MimicJointRecord] = []
        self.initial_pose = Pose()
        self._disabled_pairs: list[tuple] = []     # (link index, link index): SRDF <disable_collisions>

    def set_scene(self, scene):
        self.scene = scene
        return self

    def set_initial_pose(self, pose):
        self.initial_pose = pose
        return self

    def create_link_builder(self, parent: LinkBuilder = None):
        if self.link_builders:
            assert parent and parent in self.link_builders
        builder = LinkBuilder(len(self.link_builders), parent)
        self.link_builders.append(builder)
        return builder

    def build_entities(self, fix_root_link=None, name_prefix=""):
        from .. import physx
        entities, links = [], []
        for b in self.link_builders:
            b._check()
            b.physx_body_type = "link"
            entity = Entity()
            link = b.build_physx_component(links[b.parent.index] if b.parent else None)
            entity.add_component(link)
            if b.visual_records:
                entity.add_component(b.build_render_component())
            entity.name = b.name
            link.name = f"{name_prefix}{b.name}"
            link.joint.name = f"{name_prefix}{b.joint_record.name}"
            link.joint.type = b.joint_record.joint_type
            link.joint.pose_in_child = b.joint_record.pose_in_child
            link.joint.pose_in_parent = b.joint_record.pose_in_parent
            if link.joint.type in ("revolute", "prismatic", "revolute_unwrapped"):
                link.joint.limit = np.array(b.joint_record.limits).flatten()
                link.joint.set_drive_property(0, b.joint_record.damping)
            links.append(link)
            entities.append(entity)
        if fix_root_link is not None:
            entities[0].components[0].joint.type = "fixed" if fix_root_link else "undefined"
        entities[0].pose = self.initial_pose
        return entities

    def build(self, fix_root_link=None, name_prefix=""):
        entities = self.build_entities(fix_root_link, name_prefix)
        art = entities[0].components[0].articulation
        _finish_articulation(self, art)
        for e in entities:
            self.scene.add_entity(e)
        return art


def _finish_articulation(builder: ArticulationBuilder, art):
    """Carry the builder-level data that has no per-link component over to the PhysxArticulation."""
    art._disabled_pairs = list(builder._disabled_pairs)
