"""ManiSkillVectorEnv on the MI355X-native backend: row A1 of SURVEY.md §8(a).

Mirrors the behaviour of ``mani_skill/vector/wrappers/gymnasium.py:17-190`` (the gymnasium ``VectorEnv`` the RL
baselines drive): episode metrics, ``ignore_terminations``, and the SAME-STEP auto partial reset — envs whose episode
ended are reset inside ``step`` and the pre-reset observation / info travel in ``infos["final_observation"]`` /
``infos["final_info"]`` with the ``_final_*`` masks.  Works on ``PickCubeEnv`` / ``PushTEnv`` (state or dict
observations); gymnasium itself is not required.

Like the reference, ``dones.any()`` is one host synchronisation per step (it decides whether a reset is launched).
"""
from __future__ import annotations

from typing import Optional, Union

import torch

ENVS = {}


def _registry():
    if not ENVS:
        from .envs.pick_cube import PickCubeEnv
        from .envs.peg_insertion_side import PegInsertionSideEnv
        from .envs.push_cube import PushCubeEnv
        from .envs.push_t import PushTEnv
        from .envs.stack_cube import StackCubeEnv
        from .envs.pull_cube import PullCubeEnv
        from .envs.lift_peg_upright import LiftPegUprightEnv
        from .envs.poke_cube import PokeCubeEnv
        from .envs.stack_pyramid import StackPyramidEnv
        from .envs.pull_cube_tool import PullCubeToolEnv
        ENVS.update({"PickCube-v1": PickCubeEnv, "PushCube-v1": PushCubeEnv, "StackCube-v1": StackCubeEnv, "PushT-v1": PushTEnv,
                     "PegInsertionSide-v1": PegInsertionSideEnv, "PullCube-v1": PullCubeEnv, "LiftPegUpright-v1": LiftPegUprightEnv,
                     "PokeCube-v1": PokeCubeEnv, "StackPyramid-v1": StackPyramidEnv, "PullCubeTool-v1": PullCubeToolEnv})
    return ENVS


def torch_clone_dict(x):
    """utils/common.py torch_clone_dict: deep copy of nested dicts of tensors."""
    if isinstance(x, dict):
        return {k: torch_clone_dict(v) for k, v in x.items()}
    return x.clone() if isinstance(x, torch.Tensor) else x


class ManiSkillVectorEnv:
    def __init__(self, env: Union[object, str], num_envs: int = 1, auto_reset: bool = True, ignore_terminations: bool = False,
                 record_metrics: bool = False, step_graph: bool = False, **kwargs):
        if isinstance(env, str):
            env = _registry()[env](num_envs=num_envs, **kwargs)
        if step_graph:   # one HIP graph replay per control step (maniskill_amd/graph.py); resets stay eager
            env.enable_step_graph()
        self._env = env
        self.num_envs = env.num_envs
        for name in ("single_action_space", "action_space", "single_observation_space", "observation_space"):
            if hasattr(env, name):
                setattr(self, name, getattr(env, name))
        self.auto_reset = auto_reset
        self.ignore_terminations = ignore_terminations
        self.record_metrics = record_metrics
        if record_metrics:
            dev = env.device
            self.success_once = torch.zeros(self.num_envs, device=dev, dtype=torch.bool)
            self.fail_once = torch.zeros(self.num_envs, device=dev, dtype=torch.bool)
            self.returns = torch.zeros(self.num_envs, device=dev, dtype=torch.float32)

    @property
    def device(self):
        return self._env.device

    @property
    def base_env(self):
        return self._env

    @property
    def unwrapped(self):
        return self._env

    def reset(self, *, seed: Optional[Union[int, list]] = None, options: Optional[dict] = None):
        obs, info = self._env.reset(seed=seed, options=options)
        if self.record_metrics:
            if options is not None and "env_idx" in options:
                mask = torch.zeros(self.num_envs, dtype=torch.bool, device=self.device)
                mask[options["env_idx"]] = True
                self.success_once[mask] = False
                self.fail_once[mask] = False
                self.returns[mask] = 0
            else:
                self.success_once[:] = False
                self.fail_once[:] = False
                self.returns[:] = 0
        return obs, info

    def step(self, actions):
        obs, rew, terminations, truncations, infos = self._env.step(actions)
        episode_info = None
        if self.record_metrics:
            episode_info = dict()
            self.returns += rew
            if "success" in infos:
                self.success_once = self.success_once | infos["success"]
                episode_info["success_once"] = self.success_once.clone()
            if "fail" in infos:
                self.fail_once = self.fail_once | infos["fail"]
                episode_info["fail_once"] = self.fail_once.clone()
            episode_info["return"] = self.returns.clone()
            episode_info["episode_len"] = infos["elapsed_steps"].clone()
            episode_info["reward"] = episode_info["return"] / episode_info["episode_len"]
        if self.ignore_terminations:
            terminations = torch.zeros_like(terminations)
            if episode_info is not None:
                if "success" in infos:
                    episode_info["success_at_end"] = infos["success"].clone()
                if "fail" in infos:
                    episode_info["fail_at_end"] = infos["fail"].clone()
        if self.record_metrics:
            infos["episode"] = episode_info
        dones = torch.logical_or(terminations, truncations)
        if self.auto_reset and bool(dones.any()):
            final_obs = torch_clone_dict(obs)
            env_idx = torch.arange(0, self.num_envs, device=self.device)[dones]
            final_info = torch_clone_dict(infos)
            obs, infos = self.reset(options=dict(env_idx=env_idx))
            infos["final_observation"] = final_obs
            infos["final_info"] = final_info
            infos["_final_info"] = dones
            infos["_final_observation"] = dones
            infos["_elapsed_steps"] = dones
        return obs, rew, terminations, truncations, infos

    def call(self, name: str, *args, **kwargs):
        fn = getattr(self._env, name)
        return fn(*args, **kwargs) if callable(fn) else fn

    def get_attr(self, name: str):
        return getattr(self._env, name)

    def close(self):
        self._env.close()
