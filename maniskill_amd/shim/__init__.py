"""Drop-in layer for ManiSkill's unmodified Python: ``install()`` makes ``import sapien`` resolve to the shim in
``maniskill_amd/shim/sapien`` and provides stand-ins for the pure-Python third-party packages ManiSkill imports that this image
lacks (gymnasium, dacite, transforms3d, trimesh, lxml, ...).  A stand-in is only visible when the real package is not
installed: the stand-in directory is appended to ``sys.path``, the shim directory is prepended.

    import maniskill_amd.shim as shim; shim.install()
    import gymnasium as gym, mani_skill.envs           # the reference's own package, unmodified
    env = gym.make("PickCube-v1", num_envs=4096)       # steps on libmsk_physx.so (HIP, gfx950)
"""
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
SHIM_DIR = _HERE
STANDIN_DIR = os.path.join(_HERE, "standins")


def install(mani_skill_root: str | None = None):
    """mani_skill_root: directory that contains the ``mani_skill`` package (e.g. a ManiSkill checkout), added to sys.path."""
    if "sapien" in sys.modules and not getattr(sys.modules["sapien"], "__file__", "").startswith(SHIM_DIR):
        raise RuntimeError("another `sapien` is already imported")
    if SHIM_DIR not in sys.path:
        sys.path.insert(0, SHIM_DIR)
    if STANDIN_DIR not in sys.path:
        sys.path.append(STANDIN_DIR)
    if mani_skill_root and mani_skill_root not in sys.path:
        sys.path.insert(1, mani_skill_root)
    os.environ.setdefault("MS_SKIP_ASSET_DOWNLOAD_PROMPT", "1")
