"""Vector-env front of the native batched envs (`maniskill_amd.make`): the behaviour RL code gets from the reference's
`ManiSkillVectorEnv` (mani_skill/vector/wrappers/gymnasium.py:18-184), over `maniskill_amd.envs` instead of a `BaseEnv`.

What a caller of the reference's wrapper relies on, and gets here:
  * SAME-STEP auto reset of exactly the sub-scenes that finished (terminated | truncated): the step returns the first observation of
    the new episode, `infos["final_observation"]` / `infos["final_info"]` hold what the finished episodes ended on and
    `infos["_final_info"]` / `["_final_observation"]` / `["_elapsed_steps"]` the mask of the envs they apply to (gymnasium.py:161-177);
  * `ignore_terminations`: the termination flags are cleared before they decide a reset, success / fail of the step are reported as
    `success_at_end` / `fail_at_end` (gymnasium.py:149-156);
  * `record_metrics`: `infos["episode"]` with return, episode_len, reward (return per step), success_once, fail_once (gymnasium.py:131-147),
    cleared for the envs a reset touches (gymnasium.py:104-125).
The envs underneath are already batched on the device, so there is nothing to vectorise: this class only keeps the episode book-keeping
(three [num_envs] device tensors) and issues the masked reset.  No gymnasium import: the spaces are maniskill_amd.spaces' (gymnasium's
own classes when that package is importable).
"""
from __future__ import annotations

from typing import Optional

import torch


def _copy_tree(x):
    """Deep copy of an observation / info tree of tensors (the reset below overwrites the buffers the step's outputs may alias)."""
    if isinstance(x, dict):
        return {k: _copy_tree(v) for k, v in x.items()}
    return x.clone() if torch.is_tensor(x) else x


class _EpisodeBook:
    """Per-env running metrics of the episode in progress."""

    def __init__(self, n: int, device):
        self.ret = torch.zeros(n, dtype=torch.float32, device=device)
        self.success = torch.zeros(n, dtype=torch.bool, device=device)
        self.fail = torch.zeros(n, dtype=torch.bool, device=device)

    def clear(self, rows=None):
        sel = slice(None) if rows is None else rows
        self.ret[sel] = 0.0
        self.success[sel] = False
        self.fail[sel] = False

    def clear_mask(self, done):
        """the same for the envs a device-side mask names, without an index list"""
        self.ret.masked_fill_(done, 0.0)
        self.success &= ~done
        self.fail &= ~done

    def account(self, reward, infos, elapsed, at_end: bool) -> dict:
        self.ret += reward
        out = {}
        for key, seen in (("success", self.success), ("fail", self.fail)):
            if key in infos:
                seen |= infos[key].to(torch.bool)
                out[key + "_once"] = seen.clone()
                if at_end:
                    out[key + "_at_end"] = infos[key].clone()
        out["return"] = self.ret.clone()
        out["episode_len"] = elapsed.clone()
        out["reward"] = out["return"] / out["episode_len"]
        return out


class ManiSkillVectorEnv:
    """`ManiSkillVectorEnv(env_or_id, num_envs=..., auto_reset=True, ignore_terminations=False, record_metrics=False, **make_kwargs)`."""

    def __init__(self, env, num_envs: int = 1, auto_reset: bool = True, ignore_terminations: bool = False, record_metrics: bool = False, **kwargs):
        if isinstance(env, str):
            from . import make
            env = make(env, num_envs=num_envs, **kwargs)
        self._env = env
        self.num_envs = int(env.num_envs)
        self.auto_reset, self.ignore_terminations, self.record_metrics = bool(auto_reset), bool(ignore_terminations), bool(record_metrics)
        self.single_action_space, self.action_space = env.single_action_space, env.action_space
        self.single_observation_space, self.observation_space = env.single_observation_space, env.observation_space
        self.metadata = dict(getattr(env, "metadata", {}) or {}, autoreset_mode="same_step")
        self.spec = getattr(env, "spec", None)
        self._book = _EpisodeBook(self.num_envs, env.device) if self.record_metrics else None
        self._rows = torch.arange(self.num_envs, device=env.device)
        self.book_kernel = True      # the step's book-keeping through msk_episode_book_step where the env's library has it (False: the torch ops)

    # the reference's accessors
    @property
    def device(self):
        return self._env.device

    @property
    def base_env(self):
        return self._env

    @property
    def unwrapped(self):
        return self._env

    def reset(self, *, seed=None, options: Optional[dict] = None):
        obs, info = self._env.reset(seed=seed, options=options)
        if self._book is not None:
            self._book.clear(None if not options or "env_idx" not in options else torch.as_tensor(options["env_idx"], device=self.device, dtype=torch.long))
        return obs, info

    # ------------------------------------------------------------------------------------------------------------ the book-keeping as one launch
    def _book_in_one_launch(self, rew, terminated, truncated, infos):
        """include/msk_physx.h msk_episode_book_step: account, report, `terminated | truncated`, `dones.any()` and the clearing of the envs about to be reset in ONE
        launch (they are a dozen elementwise launches of 4096 elements each otherwise: ~35 us of a 0.8 ms step).  -> (episode dict or None, terminated, done, flag)
        or None when the env's library or the tensors at hand do not fit (then the torch ops below do the same)"""
        env = self._env
        px = getattr(env, "px", None)
        lib = getattr(px, "lib", None)
        if lib is None or not hasattr(lib, "episode_book_step"):
            return None
        n, dev = self.num_envs, rew.device
        elapsed = infos["elapsed_steps"] if "elapsed_steps" in infos else env._elapsed_steps

        def flag(t):      # a bool / uint8 [n] tensor with any stride -> (pointer, bytes between envs)
            if not torch.is_tensor(t) or t.dtype not in (torch.bool, torch.uint8) or t.ndim != 1 or t.shape[0] != n or t.device != dev:
                raise TypeError
            return t.data_ptr(), int(t.stride(0))
        try:
            te, tr = flag(terminated), flag(truncated)
            su = flag(infos["success"]) if self._book is not None and "success" in infos else (None, 1)
            fa = flag(infos["fail"]) if self._book is not None and "fail" in infos else (None, 1)
        except TypeError:
            return None
        if self._book is not None and not (rew.dtype == torch.float32 and rew.is_contiguous() and rew.shape == (n,) and elapsed.dtype == torch.int32
                                           and elapsed.is_contiguous() and elapsed.device == dev):
            return None
        from ._native import MskEpisodeBook
        import ctypes as C
        f32 = torch.empty(2, n, dtype=torch.float32, device=dev)
        i32 = torch.empty(n + 1, dtype=torch.int32, device=dev)
        u8 = torch.empty(6, n, dtype=torch.bool, device=dev)
        b = MskEpisodeBook()
        b.terminated, b.terminated_stride = te
        b.truncated, b.truncated_stride = tr
        b.success, b.success_stride = su
        b.fail, b.fail_stride = fa
        b.record_metrics, b.ignore_terminations, b.clear_done = int(self._book is not None), int(self.ignore_terminations), int(self.auto_reset)
        b.out_terminated, b.out_done, b.any_done = u8[4].data_ptr(), u8[5].data_ptr(), i32[n:].data_ptr()
        episode = None
        if self._book is not None:
            bk = self._book
            b.reward, b.elapsed = rew.data_ptr(), elapsed.data_ptr()
            b.returns, b.success_once, b.fail_once = bk.ret.data_ptr(), bk.success.data_ptr(), bk.fail.data_ptr()
            b.out_return, b.out_reward, b.out_episode_len = f32[0].data_ptr(), f32[1].data_ptr(), i32.data_ptr()
            b.out_success_once, b.out_fail_once, b.out_success_at_end, b.out_fail_at_end = (u8[k].data_ptr() for k in range(4))
            episode = {}
            for key, has, once, at_end in (("success", su[0], u8[0], u8[2]), ("fail", fa[0], u8[1], u8[3])):
                if has is not None:
                    episode[key + "_once"] = once
                    if self.ignore_terminations:
                        episode[key + "_at_end"] = at_end
            episode["return"], episode["episode_len"], episode["reward"] = f32[0], i32[:n], f32[1]
        stream = px._stream() if hasattr(px, "_stream") else None
        lib.check(px.ctx, lib.episode_book_step(px.ctx, n, C.byref(b), stream), "episode_book_step")
        return episode, u8[4], u8[5], i32[n:]

    def step(self, actions):
        obs, rew, terminated, truncated, infos = self._env.step(actions)
        fused = self._book_in_one_launch(rew, terminated, truncated, infos) if self.book_kernel else None
        if fused is not None:
            episode, terminated, done, flag = fused
            if episode is not None:
                infos["episode"] = episode
            if not self.auto_reset:
                return obs, rew, terminated, truncated, infos
            env = self._env
            if getattr(env, "reset_mask", None) is not None and getattr(env, "_device_reset_wanted", lambda: False)():
                new_obs, new_infos = env.reset_mask(done)      # (issued before the one wait below, as in the branch further down)
                if bool(flag):
                    last_obs, last_info = (obs, infos) if getattr(env, "fused", False) else (_copy_tree(obs), _copy_tree(infos))
                    obs, infos = new_obs, new_infos
                    infos["final_observation"], infos["final_info"] = last_obs, last_info
                    infos["_final_info"] = infos["_final_observation"] = infos["_elapsed_steps"] = done
                return obs, rew, terminated, truncated, infos
            if bool(flag):
                last_obs, last_info = _copy_tree(obs), _copy_tree(infos)
                obs, infos = self._env.reset(options=dict(env_idx=self._rows[done]))      # (the kernel has cleared the book of these envs already)
                infos["final_observation"], infos["final_info"] = last_obs, last_info
                infos["_final_info"] = infos["_final_observation"] = infos["_elapsed_steps"] = done
            return obs, rew, terminated, truncated, infos
        if self._book is not None:
            infos["episode"] = self._book.account(rew, infos, infos["elapsed_steps"] if "elapsed_steps" in infos else self._env._elapsed_steps,
                                                  at_end=self.ignore_terminations)
        if self.ignore_terminations:
            terminated = torch.zeros_like(terminated)
        done = terminated | truncated
        if not self.auto_reset:
            return obs, rew, terminated, truncated, infos
        env = self._env
        if getattr(env, "reset_mask", None) is not None and getattr(env, "_device_reset_wanted", lambda: False)():
            # the envs of this package reset from the device-side mask itself (envs/_device_reset.py: one kernel, then the task's observe kernel): everything is
            # ISSUED before the one wait the reference's wrapper has (`if dones.any()`, gymnasium.py:164) -- a reset over an empty mask changes nothing, so it needs
            # no verdict first, and the device never idles behind the wait.  The step's own outputs are fresh tensors (nothing below overwrites them).
            new_obs, new_infos = env.reset_mask(done)
            if self._book is not None:
                self._book.clear_mask(done)
            if bool(done.any()):
                last_obs, last_info = (obs, infos) if getattr(env, "fused", False) else (_copy_tree(obs), _copy_tree(infos))
                obs, infos = new_obs, new_infos
                infos["final_observation"], infos["final_info"] = last_obs, last_info
                infos["_final_info"] = infos["_final_observation"] = infos["_elapsed_steps"] = done
            return obs, rew, terminated, truncated, infos
        if bool(done.any()):      # (the reference's own wait: `if dones.any()`, gymnasium.py:164)
            last_obs, last_info = _copy_tree(obs), _copy_tree(infos)
            obs, infos = self.reset(options=dict(env_idx=self._rows[done]))
            infos["final_observation"], infos["final_info"] = last_obs, last_info
            infos["_final_info"] = infos["_final_observation"] = infos["_elapsed_steps"] = done
        return obs, rew, terminated, truncated, infos

    def call(self, name: str, *args, **kwargs):
        return getattr(self._env, name)(*args, **kwargs)

    def get_attr(self, name: str):
        raise RuntimeError("read attributes from .base_env (the reference's wrapper refuses this call as well: gymnasium.py:190-193)")

    def render(self):
        return self._env.render()

    def close(self):
        self._env.close()

    close_extras = close
