"""Action / observation spaces of the envs (``single_action_space`` / ``action_space`` of
BaseEnv and ManiSkillVectorEnv, envs/sapien_env.py:330-340, vector/wrappers/gymnasium.py:60-80).

gymnasium is not part of this image: ``Box`` below carries the same fields (low, high, shape, dtype, sample, contains) and is
replaced by ``gymnasium.spaces.Box`` when that package is importable, so downstream RL code sees the class it expects.
"""
from __future__ import annotations

import numpy as np

try:   # optional
    import gymnasium as _gym  # type: ignore
    _GymBox = _gym.spaces.Box if getattr(_gym, "__file__", None) else None    # a real installation, not a stand-in module
except Exception:   # pragma: no cover
    _GymBox = None


class _Box:
    def __init__(self, low, high, shape=None, dtype=np.float32):
        low, high = np.asarray(low, dtype=dtype), np.asarray(high, dtype=dtype)
        shape = tuple(shape) if shape is not None else np.broadcast(low, high).shape
        self.low, self.high = np.broadcast_to(low, shape).copy(), np.broadcast_to(high, shape).copy()
        self.shape, self.dtype = shape, np.dtype(dtype)
        self._rng = np.random.default_rng()

    def seed(self, seed=None):
        self._rng = np.random.default_rng(seed)

    def sample(self):
        lo = np.where(np.isfinite(self.low), self.low, -1.0)
        hi = np.where(np.isfinite(self.high), self.high, 1.0)
        return self._rng.uniform(lo, hi).astype(self.dtype)

    def contains(self, x) -> bool:
        x = np.asarray(x)
        return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))

    def __repr__(self):
        return f"Box({self.low.min()}, {self.high.max()}, {self.shape}, {self.dtype})"


Box = _GymBox if _GymBox is not None else _Box


def batch_space(space, n: int):
    """gymnasium.vector.utils.batch_space for a Box."""
    return Box(np.repeat(space.low[None], n, axis=0), np.repeat(space.high[None], n, axis=0), dtype=space.dtype)


def panda_action_space(control_mode: str, arm_qlimits: np.ndarray):
    """Panda._controller_configs (agents/robots/panda/panda.py:77-211): normalised modes live in [-1, 1]; the others in their
    controllers' own units."""
    one = lambda k: (-np.ones(k, np.float32), np.ones(k, np.float32))
    g_lo, g_hi = one(1)                                             # gripper: PDJointPosMimic, normalised
    lo7, hi7 = arm_qlimits[:7, 0].astype(np.float32), arm_qlimits[:7, 1].astype(np.float32)
    if control_mode in ("pd_joint_delta_pos", "pd_joint_target_delta_pos", "pd_joint_vel"):
        lo, hi = one(7)
    elif control_mode == "pd_joint_pos":
        lo, hi = lo7, hi7
    elif control_mode == "pd_joint_pos_vel":
        lo, hi = np.concatenate([lo7, -np.ones(7, np.float32)]), np.concatenate([hi7, np.ones(7, np.float32)])
    elif control_mode == "pd_joint_delta_pos_vel":
        lo, hi = one(14)
    elif control_mode in ("pd_ee_delta_pos", "pd_ee_target_delta_pos"):
        lo, hi = one(3)
    elif control_mode in ("pd_ee_delta_pose", "pd_ee_target_delta_pose"):
        lo, hi = one(6)
    elif control_mode == "pd_ee_pose":                              # pos_lower/upper +-2, rot_lower/upper +-2 pi, not normalised
        b = np.array([2.0] * 3 + [2 * np.pi] * 3, dtype=np.float32)
        lo, hi = -b, b
    else:
        raise NotImplementedError(control_mode)
    return Box(np.concatenate([lo, g_lo]), np.concatenate([hi, g_hi]), dtype=np.float32)
