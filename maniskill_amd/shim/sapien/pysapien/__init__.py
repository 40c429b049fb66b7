"""``sapien.pysapien``: name of SAPIEN's binary module; ManiSkill mentions ``sapien.pysapien.Pose`` / ``.physx`` in annotations."""
from .._pose import Pose  # noqa: F401
from .._core import Component, Device, Entity, Scene  # noqa: F401
from .. import physx, render  # noqa: F401
