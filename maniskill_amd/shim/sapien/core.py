"""``sapien.core``: legacy alias module (``from sapien.core import Pose``)."""
from ._pose import Pose  # noqa: F401
from ._core import Component, Device, Entity, Scene  # noqa: F401
