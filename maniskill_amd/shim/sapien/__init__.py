"""``sapien``: drop-in Python package that runs ManiSkill's unmodified host code (mani_skill/envs/sapien_env.py:10-13,
envs/scene.py:6-10, utils/building/*, utils/structs/*, sensors/camera.py) on the MI355X-native backend of this repository.

Layer (i) of the boundary: pure-Python objects with SAPIEN 3.0's names and semantics — ``Pose``, ``Device``,
``Entity``, ``Scene``, ``physx.*``, ``render.*``, ``wrapper.{actor_builder, articulation_builder, urdf_loader}`` — that record
what the task code builds; ``physx.PhysxGpuSystem.gpu_init()`` compiles the recorded sub-scenes into the C-ABI library
(layer (ii): include/msk_physx.h, include/msk_render.h; hand-written HIP kernels for gfx950).  ``sapien.wrapper.scene`` is
deliberately absent: mani_skill/render/version.py:2-8 probes it to choose the 3.1 render API and must fail.

Put on sys.path by ``maniskill_amd.shim.install()``.
"""
__version__ = "3.0.0"

from . import _pose as math  # noqa: F401  (sapien.math.shortest_rotation)
from ._pose import Pose
from ._core import Component, Device, Entity, Scene
from . import physx, render  # noqa: E402,F401
from .wrapper.actor_builder import ActorBuilder  # noqa: E402
from .wrapper.articulation_builder import ArticulationBuilder  # noqa: E402
from .wrapper.urdf_loader import URDFLoader  # noqa: E402
from . import wrapper, utils  # noqa: E402,F401
from . import core, pysapien  # noqa: E402,F401

_log_level = "warn"


def set_log_level(level):
    global _log_level
    _log_level = level


__all__ = ["Pose", "Device", "Entity", "Component", "Scene", "ActorBuilder", "ArticulationBuilder", "URDFLoader", "physx", "render",
           "math", "wrapper", "utils", "set_log_level"]
