"""msk_config.contact_capacity = 1: 128 contact points / 128 solver blocks per sub-scene instead of 48 / 64 -- the wide solver class (two blocks
per lane, csrc/msk_solve_wide.h) behind the others.  PhysX sizes its contact buffers for the whole scene (mani_skill/utils/structs/types.py:18-23:
max_rigid_contact_count = 2**19), so a hand closed around an object or loose parts in a fixture never lose contacts there; here they did
(RotateSingleObjectInHand: 70 points, FMBAssembly1Easy: 90).  Known answers on the oracle; HIP against the oracle under the emulation of
tests/hipemu (CPU suite) and on hardware (-m gpu).  The default capacity is unchanged, bit for bit: every other test runs with it."""
import numpy as np
import pytest
import torch

from maniskill_amd import _native as N
from maniskill_amd.envs import scene_builders as sb
from maniskill_amd.physx import SceneConfig, SceneTemplate, SimConfig

from test_hull_heaps import _heap


def _cfg(capacity):
    return SimConfig(scene_config=SceneConfig(contact_capacity=capacity))


def _comb(nteeth, nfree=0):
    """one body of `nteeth` boxes side by side lying on the table (4 points each against the table) + `nfree` small cubes dropped on it: six
    coordinates (the 16-coordinate kernels) or more, and more contact points than the default capacity holds"""
    tpl = SceneTemplate(); sb.add_table_scene(tpl)
    m = 0.4
    b = tpl.add_actor("comb", N.BODY_DYNAMIC, p=(0.0, 0.0, 0.0105), mass=m, inertia6=(m * 0.02, m * 0.0008, m * 0.02, 0, 0, 0))
    for k in range(nteeth):
        tpl.add_shape(b, N.SHAPE_BOX, params=(0.008, 0.04, 0.01), p=(0.02 * (k - (nteeth - 1) / 2), 0.0, 0.0))
    ids = [b]
    for k in range(nfree):
        c = tpl.add_actor(f"cube{k}", N.BODY_DYNAMIC, p=(0.04 * (k - (nfree - 1) / 2), 0.0, 0.0305 + 0.001), mass=0.05, inertia6=(1e-5,) * 3 + (0, 0, 0))
        tpl.add_shape(c, N.SHAPE_BOX, params=(0.01, 0.01, 0.01))
        ids.append(c)
    return tpl, ids


def _roll(factory, tpl, n, steps, capacity, push=None):
    px = factory(tpl, n, _cfg(capacity)); px.gpu_init()
    rbd = px.cuda_rigid_body_data.torch().view(n, px.bodies_per_env, 13)
    rbd[:, tpl.body_id("table-workspace"), :7] = torch.tensor([-0.12, 0.0, -sb.TABLE_HEIGHT, np.cos(np.pi / 4), 0, 0, np.sin(np.pi / 4)], device=rbd.device)
    if push is not None:
        for e in range(n):
            rbd[e, push, 7:10] = torch.tensor([0.05 * (e % 7), 0.02 * (e % 7), 0.0], device=rbd.device)      # (a comb sent off at 15 m/s ends in a heap of more than 128 points)
    px.gpu_apply_all()
    px.set_scene_offsets(np.zeros((n, 3)))
    out = []
    for _ in range(steps):
        px.step()
        px.gpu_fetch_all(); out.append(rbd.cpu().clone())
    return torch.stack(out), px


def test_a_comb_of_twenty_boxes_keeps_all_its_contacts(oracle_factory):
    """80 points against the table: the default capacity keeps 48 and flags it, the wide one keeps them all; either way the comb rests"""
    tpl, ids = _comb(20)
    for capacity, kept, flag in ((0, 48, 1), (1, 80, 0)):
        traj, px = _roll(oracle_factory, tpl, 2, 60, capacity)
        assert px.get_overflow() == flag, (capacity, px.get_overflow())
        assert len(px.get_contacts(0)[0]) == kept
        last = traj[-1, 0, ids[0]]
        assert abs(last[2].item() - 0.01) < 1e-3 and last[7:13].abs().max() < 1e-2, last
    tpl, ids = _comb(12, nfree=3)      # 24 coordinates (the 32-coordinate kernels): 48 points under the comb + those under the cubes
    traj, px = _roll(oracle_factory, tpl, 2, 80, 1)
    assert px.get_overflow() == 0 and len(px.get_contacts(0)[0]) > 64
    assert (traj[-1, 0, ids[1:], 2] - 0.03).abs().max() < 2e-3 and traj[-1, 0, ids, 7:13].abs().max() < 2e-2


def test_an_overflowing_heap_keeps_all_its_contacts_with_the_wide_capacity(oracle_factory):
    tpl, ids = _heap(10, 5, 0.04)
    _, px0 = _roll(oracle_factory, tpl, 1, 12, 0)
    traj, px1 = _roll(oracle_factory, tpl, 1, 300, 1)
    assert px0.get_overflow() == 1 and px1.get_overflow() == 0
    assert torch.isfinite(traj).all() and traj[-1, 0, ids, 7:13].abs().max() < 0.3 and (traj[-1, 0, ids, 2] > -0.002).all()


@pytest.mark.parametrize("scene", ["comb16", "comb32", "comb64"])
def test_hip_wide_class_matches_the_oracle_under_emulation(oracle_factory, scene):
    from emu_backend import EmuPhysxSystem
    tpl, ids = {"comb16": lambda: _comb(20), "comb32": lambda: _comb(14, nfree=3), "comb64": lambda: _comb(14, nfree=7)}[scene]()
    n, steps = 3, 10
    emu, pa = _roll(lambda t, k, c: EmuPhysxSystem(t, k, c), tpl, n, steps, 1, push=ids[0])
    orc, pb = _roll(oracle_factory, tpl, n, steps, 1, push=ids[0])
    assert torch.equal(emu, orc), (scene, (emu - orc).abs().max().item())
    assert pa.get_overflow() == 0 and pb.get_overflow() == 0
    assert pa.get_solver_class_counts()[4] > 0, pa.get_solver_class_counts()          # the wide class did the work
    for e in range(n):
        ia, va = pa.get_contacts(e); ib, vb = pb.get_contacts(e)
        assert ia.shape == ib.shape and (ia == ib).all() and np.array_equal(va, vb)
    assert len(pa.get_contacts(0)[0]) > 64


@pytest.mark.gpu
@pytest.mark.parametrize("scene", ["comb16", "comb32", "comb64"])
def test_hip_wide_class_matches_the_oracle(oracle_factory, scene):
    from maniskill_amd.physx import PhysxGpuSystem
    tpl, ids = {"comb16": lambda: _comb(20), "comb32": lambda: _comb(14, nfree=3), "comb64": lambda: _comb(14, nfree=7)}[scene]()
    n, steps = 300, 20                                                                  # more envs than wide workers: the workers loop
    hip, pa = _roll(lambda t, k, c: PhysxGpuSystem("cuda:0", t, k, c), tpl, n, steps, 1, push=ids[0])
    orc, pb = _roll(oracle_factory, tpl, n, steps, 1, push=ids[0])
    assert torch.equal(hip, orc), (scene, (hip - orc).abs().max().item())
    assert pa.get_overflow() == pb.get_overflow() == 0
    assert pa.get_solver_class_counts()[4] > 0


def test_the_reference_tasks_that_overflowed_run_without_overflow_with_the_wide_capacity(built):
    """RotateSingleObjectInHand (Allegro hand around an object: ~70 points) and FMBAssembly1Easy-v1 (~90) built by the reference's own code over the
    shim with MSK_CONTACT_CAPACITY=1: no overflow flag, and the emulated HIP library gives the oracle's bits in every buffer the reference reads"""
    import os
    import ref_harness
    import test_reference_conformance as T
    if ref_harness.find_reference() is None:
        pytest.skip("no reference checkout / build")
    ids = ("RotateSingleObjectInHandLevel0-v1", "RotateSingleObjectInHandLevel1-v1", "FMBAssembly1Easy-v1")
    os.environ["ZOO_HASH"] = "1"; os.environ["MSK_CONTACT_CAPACITY"] = "1"
    try:
        a, b = T._run_zoo("emu", "4", ids, "2"), T._run_zoo("oracle", "4", ids, "2")
    finally:
        os.environ.pop("ZOO_HASH", None); os.environ.pop("MSK_CONTACT_CAPACITY", None)
    assert a == b and set(a) == set(ids), (a, b)
    assert all(v.startswith("ok") and v.endswith("overflow=0") for v in a.values()), a
