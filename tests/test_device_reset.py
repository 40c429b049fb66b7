"""Partial resets from a device-side mask (include/msk_physx.h msk_reset_masked, maniskill_amd/envs/_device_reset.py) against the host-side reset of the same
env: the same episodes (seeds 2022 + index, episode seed from the episode counter), the same rows, the same apply -- state, observation, reward and flags
bit-equal through many resets, whichever path resets which env; the ring of prepared episodes is kept small here so that it is refilled several times.
CPU suite: the oracle library (orc_reset_masked, torch task code) and the HIP sources under tests/hipemu (k_reset_masked, fused task kernels)."""
import numpy as np
import pytest
import torch

from maniskill_amd.envs.peg_insertion_side import PegInsertionSideEnv
from maniskill_amd.envs.pick_cube import PickCubeEnv
from maniskill_amd.envs.push_t import PushTEnv
from maniskill_amd.vector import ManiSkillVectorEnv


@pytest.fixture()
def emu_factory(built):
    from emu_backend import EmuPhysxSystem
    return lambda tpl, n, cfg: EmuPhysxSystem(tpl, n, cfg)


def _pair(cls, n, factory, fused, slots=4, **kw):
    host = cls(num_envs=n, px_factory=factory, fused=fused, device_reset=False, **kw)
    dev = cls(num_envs=n, px_factory=factory, fused=fused, device_reset=True, **kw)
    dev.device_reset_slots = slots
    return host, dev


def _same(a, b, what):
    if isinstance(a, dict):
        for k in a:
            _same(a[k], b[k], f"{what}.{k}")
    elif torch.is_tensor(a):
        assert torch.equal(a, b), f"{what} differs"


def _rollout(host, dev, adim, steps, seed=0, every=3):
    """both envs through the same actions; every few steps a random subset is reset -- host-side on one env, from the mask on the other"""
    n = host.num_envs
    g = torch.Generator().manual_seed(seed)
    oh, _ = host.reset(seed=2022)
    od, _ = dev.reset(seed=2022)
    _same(oh, od, "first obs")
    if dev.device_reset:      # (where the ring is built by a worker -- on a GPU, or when asked for -- this rollout waits for it: it is about the device path)
        dev._device_reset_wanted()
        if dev._dev_reset is not None:
            dev._dev_reset.wait_ready()
    for t in range(steps):
        a = 2 * torch.rand(n, adim, generator=g) - 1
        rh, rd = host.step(a), dev.step(a)
        for k in range(4):
            _same(rh[k], rd[k], f"step {t} output {k}")
        assert torch.equal(host.get_state(), dev.get_state()), f"state differs at step {t}"
        if t % every == every - 1:
            done = torch.rand(n, generator=g) < 0.4
            if t % (2 * every) == every - 1:
                done[:] = done | (torch.arange(n) == (t % n))          # (never empty)
            rows = torch.nonzero(done).reshape(-1)
            oh, ih = host.reset(options=dict(env_idx=rows)) if len(rows) else (None, None)
            od, idv = dev.reset_mask(done)
            if oh is not None:
                _same(oh, od, f"obs after the reset at step {t}")
                _same(ih, idv, f"info after the reset at step {t}")
            assert torch.equal(host.get_state(), dev.get_state()), f"state differs after the reset at step {t}"
            assert torch.equal(host._elapsed_steps, dev._elapsed_steps)
    return dev._dev_reset


@pytest.mark.parametrize("cls,adim", [(PickCubeEnv, 8), (PegInsertionSideEnv, 8), (PushTEnv, 7)])
def test_reset_from_a_mask_equals_the_host_side_reset_on_the_oracle(oracle_factory, cls, adim):
    host, dev = _pair(cls, 6, oracle_factory, fused=False)
    dr = _rollout(host, dev, adim, 40)
    assert dr is not None and dr.resets >= 12 and dr.refreshes >= 3           # the ring of 4 episodes went round
    counts = dr.pull_counts()
    assert np.array_equal(counts, host._episode_count.astype(np.int64))      # same episode numbering on both paths


def test_refills_by_the_worker_thread_leave_the_same_rows(oracle_factory):
    """the refill of the ring handed to a worker thread (what a GPU run does: envs/_device_reset.py refresh): same episodes, same bits, and an error in the worker
    surfaces on the caller's thread at the next join"""
    host, dev = _pair(PickCubeEnv, 6, oracle_factory, fused=False)
    dev.device_reset_threaded = True
    dr = _rollout(host, dev, 8, 40)
    assert dr.threaded and dr.refreshes >= 3
    assert np.array_equal(dr.pull_counts(), host._episode_count.astype(np.int64))
    dr._refill = lambda ep: (_ for _ in ()).throw(RuntimeError("boom"))
    dr.refresh()
    with pytest.raises(RuntimeError, match="boom"):
        dr.join()


def test_resets_are_the_hosts_while_a_worker_builds_the_ring_and_the_devices_afterwards(oracle_factory):
    """a seeded reset voids the prepared episodes; building 64 per sub-scene again takes seconds at 4096 envs.  Threaded, the caller goes on at once: resets are
    host-side until the worker is done (same episodes, same bits), then the device's counters take the host's over and the mask path resumes"""
    import threading
    host, dev = _pair(PickCubeEnv, 6, oracle_factory, fused=False)
    dev.device_reset_threaded = True
    host.reset(seed=11); dev.reset(seed=11)
    dev._device_reset_wanted()
    dr = dev._dev_reset
    dr.wait_ready()
    gate = threading.Event()
    fill = dr._fill
    dr._fill = lambda *a, **k: (gate.wait(), fill(*a, **k))[1]      # the next ring build waits for the gate
    host.reset(seed=12); dev.reset(seed=12)                          # ... this one's
    assert not dr.ready() and not dev._device_reset_wanted()
    g = torch.Generator().manual_seed(0)

    def round_(k):
        a = 2 * torch.rand(6, 8, generator=g) - 1
        rh, rd = host.step(a), dev.step(a)
        assert torch.equal(rh[0], rd[0])
        done = torch.tensor([k % 2 == 0, True, False, k % 3 == 0, False, True])
        oh, _ = host.reset(options=dict(env_idx=torch.nonzero(done).reshape(-1)))
        od, _ = dev.reset_mask(done)
        assert torch.equal(oh, od) and torch.equal(host.get_state(), dev.get_state())
    before = dr.resets
    for k in range(3):
        round_(k)
    assert dr.resets == before                                        # host-side so far
    gate.set()
    dr.wait_ready()
    assert dev._device_reset_wanted()
    for k in range(3, 9):
        round_(k)
    assert dr.resets == before + 6 and dr.rebuilds >= 2
    assert np.array_equal(dr.pull_counts(), host._episode_count.astype(np.int64))
    host.reset(seed=13); dev.reset(seed=13)                          # a build in flight is cancelled by the next seeded reset
    host.reset(seed=14); dev.reset(seed=14)
    dr.wait_ready()
    for k in range(9, 12):
        round_(k)


def test_close_stops_the_workers_and_leaves_the_host_path_until_the_next_seeded_reset(oracle_factory):
    """what the exit hook does (a daemon worker inside torch at interpreter shutdown = `terminate called without an active exception`): a build in flight is
    cancelled, resets stay correct on the host path, the next seeded reset builds the ring again"""
    import threading
    from maniskill_amd.envs import _device_reset as mod
    host, dev = _pair(PickCubeEnv, 5, oracle_factory, fused=False)
    dev.device_reset_threaded = True
    host.reset(seed=21); dev.reset(seed=21)
    dev._device_reset_wanted()
    dr = dev._dev_reset
    assert dr in mod._LIVE
    dr.wait_ready()
    gate = threading.Event()
    fill = dr._fill
    dr._fill = lambda *a, **k: (gate.wait(), fill(*a, **k))[1]
    host.reset(seed=22); dev.reset(seed=22)
    t = dr._rebuild_job[0]
    gate.set()
    mod._stop_workers()
    assert not t.is_alive() and dr._rebuild_job is None and dr._job is None
    assert not dr.ready() and not dev._device_reset_wanted()
    done = torch.tensor([True, False, True, False, True])
    a = torch.zeros(5, 8)
    assert torch.equal(host.step(a)[0], dev.step(a)[0])
    oh, _ = host.reset(options=dict(env_idx=torch.nonzero(done).reshape(-1)))
    od, _ = dev.reset_mask(done)
    assert torch.equal(oh, od)
    dr._fill = fill
    host.reset(seed=23); dev.reset(seed=23)
    dr.wait_ready()
    assert dr.ready() and dev._device_reset_wanted()
    before = dr.resets
    oh, _ = host.reset(options=dict(env_idx=torch.nonzero(done).reshape(-1)))
    od, _ = dev.reset_mask(done)
    assert torch.equal(oh, od) and dr.resets == before + 1


def test_an_env_built_under_inference_mode_refills_from_the_worker_thread(oracle_factory):
    """bench.py steps under torch.inference_mode(); the refill thread is outside it (the mode is thread-local) and must still be allowed to write the shadow's
    buffers and the ring: round 6's first default bench run died on `Inplace update to inference tensor outside InferenceMode`"""
    with torch.inference_mode():
        env = PickCubeEnv(num_envs=6, px_factory=oracle_factory, fused=False, device_reset=True)
        env.device_reset_slots, env.device_reset_threaded = 4, True
        env.reset(seed=3)
        env._device_reset_wanted()
        env._dev_reset.wait_ready()
        for _ in range(10):
            env.step(torch.zeros(6, 8))
            env.reset_mask(torch.tensor([True, False, True, False, True, True]))
        env._dev_reset.join()
        assert env._dev_reset.threaded and env._dev_reset.refreshes >= 3


@pytest.mark.parametrize("cls,adim", [(PickCubeEnv, 8), (PushTEnv, 7)])
def test_reset_from_a_mask_on_the_emulated_hip_library_with_the_fused_task_kernels(emu_factory, cls, adim):
    host, dev = _pair(cls, 5, emu_factory, fused=True)
    dr = _rollout(host, dev, adim, 14, every=2)
    assert dr.resets >= 7 and dr.refreshes >= 2


def test_full_resets_without_a_seed_take_the_device_path_and_seeded_ones_restart_the_ring(oracle_factory):
    host, dev = _pair(PickCubeEnv, 4, oracle_factory, fused=False)
    g = torch.Generator().manual_seed(3)
    for rnd in range(3):
        kw = dict(seed=7 + rnd) if rnd != 1 else {}
        oh, _ = host.reset(**kw)
        od, _ = dev.reset(**kw)
        assert torch.equal(oh, od)
        for _ in range(3):
            a = 2 * torch.rand(4, 8, generator=g) - 1
            rh, rd = host.step(a), dev.step(a)
            assert torch.equal(rh[0], rd[0]) and torch.equal(host.get_state(), dev.get_state())
        idx = torch.tensor([1, 3])
        oh, _ = host.reset(options=dict(env_idx=idx))
        od, _ = dev.reset(options=dict(env_idx=idx))
        assert torch.equal(oh, od) and torch.equal(host.get_state(), dev.get_state())
    assert dev._dev_reset.resets >= 4


def test_vector_env_auto_resets_from_the_mask(oracle_factory):
    """ManiSkillVectorEnv's same-step auto reset (vector/wrappers/gymnasium.py:164-176) over both kinds of env: identical outputs, final_* entries included"""
    host, dev = _pair(PickCubeEnv, 5, oracle_factory, fused=False)
    host.max_episode_steps = dev.max_episode_steps = 4
    vh, vd = ManiSkillVectorEnv(host, record_metrics=True), ManiSkillVectorEnv(dev, record_metrics=True)
    vh.reset(seed=11); vd.reset(seed=11)
    host._elapsed_steps[:] = torch.tensor([0, 1, 2, 3, 1], dtype=torch.int32)      # episodes out of phase: some env finishes at almost every step
    dev._elapsed_steps[:] = host._elapsed_steps
    g = torch.Generator().manual_seed(5)
    finals = 0
    for t in range(12):
        a = 2 * torch.rand(5, 8, generator=g) - 1
        rh, rd = vh.step(a), vd.step(a)
        for k in range(4):
            _same(rh[k], rd[k], f"step {t} output {k}")
        assert set(rh[4]) == set(rd[4])
        _same(rh[4], rd[4], f"step {t} infos")
        finals += int("final_info" in rd[4])
    assert finals >= 8 and dev._dev_reset is not None and dev._dev_reset.resets >= finals      # (the masked reset is issued at every step, before the wait: an empty mask changes nothing)


# ---------------------------------------------------------------------------------------------------------------- on the GPU
DEV = "cuda:0"


@pytest.mark.gpu
@pytest.mark.parametrize("cls,adim", [(PickCubeEnv, 8), (PegInsertionSideEnv, 8), (PushTEnv, 7)])
def test_hip_reset_from_a_mask_equals_the_oracles_host_side_reset(oracle_factory, cls, adim):
    """k_reset_masked on hardware against the CPU oracle stepping the same envs with HOST-side resets: states bit-equal through 20 partial resets"""
    n = 64
    cpu = cls(num_envs=n, px_factory=oracle_factory, fused=False, device_reset=False)
    gpu = cls(num_envs=n, device=DEV, fused=False, device_reset=True)
    gpu.device_reset_slots = 4
    g = torch.Generator().manual_seed(0)
    cpu.reset(seed=2022); gpu.reset(seed=2022)
    gpu._device_reset_wanted()          # (on a GPU the ring is built by a worker after a seeded reset and resets are the host's meanwhile: this test is about the kernel)
    gpu._dev_reset.wait_ready()
    for t in range(60):
        a = 2 * torch.rand(n, adim, generator=g) - 1
        cpu.step(a); gpu.step(a.to(DEV))
        assert torch.equal(cpu.get_state(), gpu.get_state().cpu()), f"state differs at step {t}"
        if t % 3 == 2:
            done = torch.rand(n, generator=g) < 0.3
            done[t % n] = True
            cpu.reset(options=dict(env_idx=torch.nonzero(done).reshape(-1)))
            gpu.reset_mask(done.to(DEV))
            assert torch.equal(cpu.get_state(), gpu.get_state().cpu()), f"state differs after the reset at step {t}"
    assert gpu._dev_reset.resets == 20 and gpu._dev_reset.refreshes >= 5


@pytest.mark.gpu
def test_hip_vector_env_with_step_graph_and_mask_resets_equals_the_host_side_path():
    """what an RL trainer runs: the fused env, its control step replayed as a graph, same-step auto resets from the mask -- against the same env with eager steps and
    host-side resets: observation, reward, flags and final_* bit-equal over 300 steps with episodes out of phase"""
    n = 256
    a_env = PickCubeEnv(num_envs=n, device=DEV, device_reset=False)
    b_env = PickCubeEnv(num_envs=n, device=DEV, device_reset=True)
    b_env.device_reset_slots = 8
    b_env.enable_step_graph()
    va, vb = ManiSkillVectorEnv(a_env, record_metrics=True), ManiSkillVectorEnv(b_env, record_metrics=True)
    va.reset(seed=5); vb.reset(seed=5)
    b_env._device_reset_wanted()
    b_env._dev_reset.wait_ready()       # (the ring is rebuilt by a worker after the seeded reset: the comparison is about the device path)
    phase = torch.randint(0, 50, (n,), generator=torch.Generator().manual_seed(1), dtype=torch.int32).to(DEV)
    a_env._elapsed_steps.copy_(phase); b_env._elapsed_steps.copy_(phase)
    torch.manual_seed(0)
    finals = 0
    for t in range(300):
        act = 2 * torch.rand(n, 8, device=DEV) - 1
        ra, rb = va.step(act), vb.step(act)
        for k in range(4):
            _same(ra[k], rb[k], f"step {t} output {k}")
        assert set(ra[4]) == set(rb[4])
        _same(ra[4], rb[4], f"step {t} infos")
        finals += int("final_info" in rb[4])
    assert finals >= 250
