"""ManiSkillVectorEnv: metrics, ignore_terminations, SAME-STEP auto partial reset."""
import numpy as np
import pytest
import torch

from maniskill_amd.envs.pick_cube import PickCubeEnv
from maniskill_amd.vector import ManiSkillVectorEnv


def test_auto_reset_at_the_time_limit_and_metrics(oracle_factory):
    n = 3
    venv = ManiSkillVectorEnv(PickCubeEnv(num_envs=n, px_factory=oracle_factory), record_metrics=True)
    obs, _ = venv.reset(seed=0)
    ret = torch.zeros(n)
    for t in range(49):
        obs, rew, term, trunc, info = venv.step(torch.zeros(n, 8))
        ret += rew
        assert "final_observation" not in info and not trunc.any()
        assert torch.allclose(info["episode"]["return"], ret) and (info["episode"]["episode_len"] == t + 1).all()
    pre = venv.base_env.get_state().clone()
    obs, rew, term, trunc, info = venv.step(torch.zeros(n, 8))      # step 50: TimeLimit -> every env is reset in this step
    ret += rew
    assert trunc.all() and info["_final_observation"].all() and info["_final_info"].all()
    assert torch.allclose(info["final_info"]["episode"]["return"], ret)
    assert (info["final_info"]["elapsed_steps"] == 50).all() and (info["elapsed_steps"] == 0).all()
    assert not torch.equal(info["final_observation"], obs)             # obs is the first observation of the new episode
    assert (venv._book.ret == 0).all() and not venv._book.success.any()
    assert not torch.allclose(venv.base_env.get_state(), pre)          # cube / goal / robot re-randomised
    # the new episode runs on
    obs, rew, term, trunc, info = venv.step(torch.zeros(n, 8))
    assert (info["elapsed_steps"] == 1).all() and "final_observation" not in info


def test_partial_auto_reset_only_touches_finished_envs(oracle_factory):
    n = 4
    env = PickCubeEnv(num_envs=n, px_factory=oracle_factory)
    venv = ManiSkillVectorEnv(env, auto_reset=True)
    venv.reset(seed=1)
    for _ in range(10):
        venv.step(torch.zeros(n, 8))
    env._elapsed_steps[1] = 49                                       # env 1 reaches its time limit at the next step
    before = env.get_state().clone()
    obs, rew, term, trunc, info = venv.step(torch.zeros(n, 8))
    assert trunc.tolist() == [False, True, False, False]
    assert info["_final_observation"].tolist() == [False, True, False, False]
    assert info["elapsed_steps"].tolist() == [11, 0, 11, 11]
    after = env.get_state()
    moved = (after - before).abs().amax(1)
    assert moved[1] > 1e-3 and (moved[[0, 2, 3]] < 1e-2).all()        # the others only took one quiet physics step


def test_ignore_terminations_and_no_auto_reset(oracle_factory):
    n = 2
    venv = ManiSkillVectorEnv(PickCubeEnv(num_envs=n, px_factory=oracle_factory), auto_reset=False, ignore_terminations=True,
                              record_metrics=True)
    venv.reset(seed=2)
    for _ in range(50):
        obs, rew, term, trunc, info = venv.step(None)
    assert trunc.all() and not term.any() and "final_observation" not in info
    assert "success_at_end" in info["episode"] and (info["elapsed_steps"] == 50).all()


@pytest.mark.gpu
@pytest.mark.parametrize("fused", [False, True])
def test_vector_env_on_the_gpu_matches_the_oracle_through_a_reset(oracle_factory, fused):
    """fused=True is the benchmarked form of the native envs (controller / observation / reward kernels of the library)."""
    n = 64
    g = ManiSkillVectorEnv("PickCube-v1", num_envs=n, device="cuda:0", fused=fused, record_metrics=True)
    c = ManiSkillVectorEnv(PickCubeEnv(num_envs=n, px_factory=oracle_factory), record_metrics=True)
    g.reset(seed=9); c.reset(seed=9)
    gen = torch.Generator().manual_seed(4)
    for t in range(55):                                               # crosses the auto reset at step 50
        a = 2 * torch.rand(n, 8, generator=gen) - 1
        og, rg, tg, ug, ig = g.step(a.to("cuda:0"))
        oc, rc, tc, uc, ic = c.step(a)
        assert np.allclose(og.cpu().numpy(), oc.numpy(), rtol=1e-4, atol=1e-5), t
        assert torch.equal(ug.cpu(), uc) and ("final_observation" in ig) == ("final_observation" in ic)
    assert np.allclose(g.base_env.get_state().cpu().numpy(), c.base_env.get_state().numpy(), rtol=1e-4, atol=1e-5)
