"""A deterministic stand-in for a batched env, used to compare this package's ManiSkillVectorEnv with the reference's wrapper
logic step by step (tests/golden/make_reference_vectors.py records the reference's outputs, tests/test_reference_vectors.py replays)."""
import torch


class ScriptedEnv:
    """4 envs; observations encode (time, env), successes / failures / truncations follow a fixed script."""

    num_envs, action_dim, max_episode_steps = 4, 2, 5
    device = torch.device("cpu")
    spec = None
    reconfiguration_freq = 0

    def __init__(self):
        self.elapsed_steps = torch.zeros(self.num_envs, dtype=torch.int32)
        self.t = 0
        self.unwrapped = self

    def _obs(self):
        return self.t * 10.0 + torch.arange(self.num_envs, dtype=torch.float32)[:, None] + torch.tensor([[0.0, 0.5]])

    def reset(self, seed=None, options=None):
        idx = torch.arange(self.num_envs) if not options or "env_idx" not in options else torch.as_tensor(options["env_idx"])
        self.elapsed_steps[idx] = 0
        obs = self._obs()
        obs[idx] = -1.0 - idx[:, None].float()          # a fresh episode's first observation
        return obs, dict(reconfigure=False)

    def step(self, actions):
        self.t += 1
        self.elapsed_steps += 1
        e = torch.arange(self.num_envs)
        success = (self.t % 3 == 0) & (e == 1) | (self.t == 7) & (e == 3)
        fail = (self.t == 4) & (e == 2)
        rew = 0.1 * self.t + 0.01 * e.float() + actions[:, 0]
        trunc = self.elapsed_steps >= self.max_episode_steps
        info = dict(elapsed_steps=self.elapsed_steps.clone(), success=success, fail=fail)
        return self._obs(), rew, (success | fail).clone(), trunc, info
