"""The Python around the fused task kernels only runs with the HIP library; here its control flow is executed on the CPU with the
library's task entry points replaced by recording stand-ins (no arithmetic is checked -- tests/test_gpu_parity.py and friends do
that on the GPU): argument plumbing, reward-mode conversion, buffer staleness handling."""
import types

import pytest
import torch

from maniskill_amd.envs.peg_insertion_side import PegInsertionSideEnv
from maniskill_amd.envs.pick_cube import PickCubeEnv


def _with_fake_task_kernels(env, calls):
    lib = env.px.lib
    fake = types.SimpleNamespace(check=lambda ctx, code, what: code)
    for name in ("task_pickcube_set_action", "task_pickcube_set_action_ee", "control_step", "task_pickcube_observe", "task_peg_observe",
                 "task_pusht_set_action", "task_pusht_observe"):
        setattr(fake, name, (lambda n: (lambda *a: calls.append((n, len(a))) or 0))(name))
    for name in dir(lib):
        if not name.startswith("__") and not hasattr(fake, name):
            setattr(fake, name, getattr(lib, name))
    env.px.lib = fake
    env.fused = True
    return env


@pytest.mark.parametrize("mode,reward_mode", [("pd_joint_delta_pos", "normalized_dense"), ("pd_ee_delta_pose", "dense"), ("pd_joint_pos", "sparse"),
                                              ("pd_joint_delta_pos", "none")])
def test_pickcube_fused_step_control_flow(oracle_factory, mode, reward_mode):
    calls = []
    env = _with_fake_task_kernels(PickCubeEnv(num_envs=3, px_factory=oracle_factory, control_mode=mode, reward_mode=reward_mode), calls)
    obs, rew, term, trunc, info = env.step(torch.zeros(3, env.action_dim))
    names = [c[0] for c in calls]
    assert names[-2:] == ["control_step", "task_pickcube_observe"]
    assert ("task_pickcube_set_action" in names) == (mode == "pd_joint_delta_pos") and ("task_pickcube_set_action_ee" in names) == (mode == "pd_ee_delta_pose")
    assert obs.shape == (3, 42) and rew.shape == (3,) and term.dtype == torch.bool and set(info) >= {"success", "is_grasped", "elapsed_steps"}
    assert env._buffers_stale                       # the sapien-style buffers are refreshed on demand ...
    env.get_state()
    assert not env._buffers_stale                   # ... by the first host-side read
    if reward_mode == "none":
        assert torch.equal(rew, torch.zeros(3))
    obs2, info2 = env.reset(seed=1)                 # reset on the fused path observes through the kernel as well
    assert calls[-1][0] == "task_pickcube_observe" and obs2.shape == (3, 42)


def test_peg_fused_step_control_flow(oracle_factory):
    calls = []
    env = _with_fake_task_kernels(PegInsertionSideEnv(num_envs=2, px_factory=oracle_factory), calls)
    obs, rew, term, trunc, info = env.step(torch.zeros(2, 8))
    assert [c[0] for c in calls][-3:] == ["task_pickcube_set_action", "control_step", "task_peg_observe"]
    assert obs.shape == (2, 43) and info["peg_head_pos_at_hole"].shape == (2, 3) and "success" in info


def test_pusht_fused_step_control_flow(oracle_factory):
    from maniskill_amd.envs.push_t import PushTEnv

    calls = []
    env = _with_fake_task_kernels(PushTEnv(num_envs=2, px_factory=oracle_factory), calls)
    obs, rew, term, trunc, info = env.step(torch.zeros(2, 7))
    assert [c[0] for c in calls][-3:] == ["task_pusht_set_action", "control_step", "task_pusht_observe"]
    assert obs.shape == (2, 31) and rew.shape == (2,) and "success" in info
    env.get_state()
    assert not env._buffers_stale
