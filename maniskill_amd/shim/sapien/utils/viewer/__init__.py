class Viewer:
    """sapien.utils.Viewer: ManiSkill opens it for render_mode="human" only (mani_skill/utils/sapien_utils.py create_viewer)."""

    def __init__(self, *a, **k):
        raise RuntimeError("the interactive viewer is not available in this backend (no display); use rgb_array / sensor rendering")


from . import control_window  # noqa: E402,F401
