"""maniskill_amd — MI355X-native batched rigid-body backend for ManiSkill's hot path.

Scope: the physics substep behind ``PhysxGpuSystem.step()`` and the
buffer/apply/fetch/contact-query contract around it, as hand-written HIP kernels behind a
C ABI (include/msk_physx.h), plus the thin Python host that mirrors the reference's
``sapien.physx`` / ``BaseEnv`` interfaces for that path.  No CPU fallback lives here.
"""
import os

PACKAGE_DIR = os.path.dirname(os.path.abspath(__file__))
PACKAGE_ASSET_DIR = os.path.join(PACKAGE_DIR, "assets")
__version__ = "0.1.0"


def make(env_id: str, **kwargs):
    """gym.make-style constructor of a batched env: ``maniskill_amd.make("PickCube-v1", num_envs=4096, device="cuda:0")``
    (mani_skill/utils/registration.py:176-196).  Auto resets / episode metrics: maniskill_amd.vector.ManiSkillVectorEnv."""
    from .envs import registered
    reg = registered()
    if env_id not in reg:
        raise KeyError(f"{env_id!r} is not built on this backend; available: {sorted(reg)}")
    return reg[env_id](**kwargs)
