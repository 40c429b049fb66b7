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


def _tree_equal(a, b, what=""):
    if isinstance(a, dict):
        assert set(a) == set(b), (what, sorted(a), sorted(b))
        for k in a:
            _tree_equal(a[k], b[k], f"{what}.{k}")
    elif torch.is_tensor(a):
        assert a.dtype == b.dtype and a.shape == b.shape and torch.equal(a, b), what


@pytest.mark.parametrize("record,ignore,auto,device_reset", [(True, False, True, True), (True, True, True, False), (False, False, True, True), (True, True, False, False)])
def test_the_book_keeping_kernel_reports_what_the_torch_ops_report(oracle_factory, record, ignore, auto, device_reset):
    """include/msk_physx.h msk_episode_book_step (here the oracle library's mirror) against the wrapper's torch ops -- gymnasium.py:131-176 restated twice -- over a
    rollout with early time limits: every output of every step, the `episode` metrics, the final_* entries, and the book itself"""
    n = 5
    mk = lambda: PickCubeEnv(num_envs=n, px_factory=oracle_factory, fused=False, device_reset=device_reset)      # noqa: E731
    a = ManiSkillVectorEnv(mk(), record_metrics=record, ignore_terminations=ignore, auto_reset=auto)
    b = ManiSkillVectorEnv(mk(), record_metrics=record, ignore_terminations=ignore, auto_reset=auto)
    b.book_kernel = False
    a.reset(seed=4); b.reset(seed=4)
    for v in (a, b):
        v.base_env._elapsed_steps.copy_(torch.tensor([40, 47, 38, 48, 45], dtype=torch.int32))
    g = torch.Generator().manual_seed(0)
    used = 0
    for t in range(14):
        act = 2 * torch.rand(n, 8, generator=g) - 1
        ra, rb = a.step(act), b.step(act)
        used += ("episode" in ra[4]) or ("final_info" in ra[4] and "episode" in ra[4]["final_info"]) or not record
        for k in range(4):
            assert torch.equal(ra[k], rb[k]) and ra[k].dtype == rb[k].dtype, (t, k)
        _tree_equal(ra[4], rb[4], f"step {t} infos")
        if record:
            assert torch.equal(a._book.ret, b._book.ret) and torch.equal(a._book.success, b._book.success)
    assert used == 14 and (not auto or a.base_env._elapsed_steps.max() < 20)      # (with auto reset every env went through its time limit)


def test_the_hip_book_keeping_kernel_under_emulation_equals_the_oracle_mirror(built):
    """k_episode_book (maniskill_amd/csrc/msk_kernels.h, compiled against tests/hipemu) against orc_episode_book_step on random books: strided flag columns, with
    and without `fail`, ignore_terminations, clear_done -- every output and the book itself"""
    import ctypes as C
    from emu_backend import emu_lib
    from maniskill_amd import _native as N
    import oracle_backend
    emu, orc = emu_lib(), N.NativeLib(oracle_backend.ORACLE_LIB, "orc_")
    n = 2500
    g = torch.Generator().manual_seed(0)
    def run(lib, ignore, clear, with_fail):
        torch.manual_seed(1)
        fl = (torch.rand(n, 6) < 0.3)
        rew = torch.rand(n); el = torch.randint(1, 50, (n,), dtype=torch.int32)
        ret = torch.rand(n); so = torch.rand(n) < 0.2; fo = torch.rand(n) < 0.2
        f32 = torch.zeros(2, n); i32 = torch.zeros(n + 1, dtype=torch.int32); u8 = torch.zeros(6, n, dtype=torch.bool)
        b = N.MskEpisodeBook()
        b.reward, b.elapsed = rew.data_ptr(), el.data_ptr()
        b.success, b.success_stride = fl[:, 0].data_ptr(), 6
        if with_fail: b.fail, b.fail_stride = fl[:, 1].data_ptr(), 6
        b.terminated, b.terminated_stride = fl[:, 4].data_ptr(), 6
        b.truncated, b.truncated_stride = fl[:, 5].data_ptr(), 6
        b.record_metrics, b.ignore_terminations, b.clear_done = 1, ignore, clear
        b.returns, b.success_once, b.fail_once = ret.data_ptr(), so.data_ptr(), fo.data_ptr()
        b.out_return, b.out_reward, b.out_episode_len = f32[0].data_ptr(), f32[1].data_ptr(), i32.data_ptr()
        b.out_success_once, b.out_fail_once, b.out_success_at_end, b.out_fail_at_end = (u8[k].data_ptr() for k in range(4))
        b.out_terminated, b.out_done, b.any_done = u8[4].data_ptr(), u8[5].data_ptr(), i32[n:].data_ptr()
        r = lib.episode_book_step(None, n, C.byref(b), None)
        assert r == 0, r
        return [f32, i32, u8, ret, so, fo]

    for ignore in (0, 1):
        for clear in (0, 1):
            for wf in (0, 1):
                a, b = run(emu, ignore, clear, wf), run(orc, ignore, clear, wf)
                assert all(torch.equal(x, y) for x, y in zip(a, b)), (ignore, clear, wf)
                assert int(a[1][-1]) == 1
