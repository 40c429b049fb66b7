"""VectorEnv base class (pre-1.0 constructor: num_envs, single observation / action space) and batch_space."""
from enum import Enum

from ..core import Env
from . import utils  # noqa: F401
from .utils import batch_space


class AutoresetMode(Enum):
    NEXT_STEP = "NextStep"
    SAME_STEP = "SameStep"
    DISABLED = "Disabled"


class VectorEnv(Env):
    def __init__(self, num_envs: int, observation_space, action_space):
        self.num_envs = num_envs
        self.is_vector_env = True
        self.observation_space = batch_space(observation_space, n=num_envs)
        self.action_space = batch_space(action_space, n=num_envs)
        self.single_observation_space = observation_space
        self.single_action_space = action_space
        self.closed = False

    def reset(self, *, seed=None, options=None):
        raise NotImplementedError

    def step(self, actions):
        raise NotImplementedError

    def close_extras(self, **kwargs):
        pass

    def close(self, **kwargs):
        if self.closed:
            return
        self.close_extras(**kwargs)
        self.closed = True

    @property
    def unwrapped(self):
        return self


class SyncVectorEnv(VectorEnv):
    """Not provided: the reference only uses it for multi-process CPU PhysX baselines."""

    def __init__(self, *a, **k):
        raise NotImplementedError("gymnasium stand-in: SyncVectorEnv is not provided")


class AsyncVectorEnv(VectorEnv):
    def __init__(self, *a, **k):
        raise NotImplementedError("gymnasium stand-in: AsyncVectorEnv is not provided")
