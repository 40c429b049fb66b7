"""``sapien.utils``: the GUI viewer is not part of this backend."""
from .viewer import Viewer  # noqa: F401
