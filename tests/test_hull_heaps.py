"""Heaps of convex hulls and boxes that start interpenetrating: many hull pairs per env, deep contacts (GJK's EPA branch), several queue items
per narrowphase wavefront -- the regime FMBAssembly1Easy-v1 starts in and the settled scenes of the parity rollouts never reach.
CPU: the oracle brings such a heap to rest, and the HIP sources under the emulation of tests/hipemu give the oracle's bits (envs sharing
narrowphase wavefronts).  -m gpu: every one of many identical envs gives the same bits, and those of the oracle.
(tools/emu_hull_fuzz.py is the pairwise version of this.)"""
import numpy as np
import pytest
import torch

from maniskill_amd import _native as N
from maniskill_amd.envs import scene_builders as sb
from maniskill_amd.physx import SceneTemplate, SimConfig


def _rand_hull(rng, nv, scale):
    v = rng.normal(size=(nv, 3))
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    return (v * scale * rng.uniform(0.6, 1.0, size=(nv, 1))).astype(np.float32)


def _heap(nbody, seed, spread):
    rng = np.random.default_rng(seed)
    tpl = SceneTemplate(); sb.add_table_scene(tpl)
    ids = []
    for k in range(nbody):
        r = rng.uniform(0.03, 0.05)
        q = rng.normal(size=4); q /= np.linalg.norm(q)
        b = tpl.add_actor(f"h{k}", N.BODY_DYNAMIC, p=tuple(float(x) for x in rng.uniform(-spread, spread, size=2)) + (0.05 + 0.02 * k,),
                          q=tuple(float(x) for x in (q if q[0] > 0 else -q)), mass=0.2, inertia6=(2e-4,) * 3 + (0, 0, 0))
        if k % 3 == 0:
            tpl.add_shape(b, N.SHAPE_BOX, params=tuple(float(x) for x in rng.uniform(0.5, 0.8, size=3) * r))
        else:
            tpl.add_shape(b, N.SHAPE_CONVEX, verts=_rand_hull(rng, int(rng.integers(6, 24)), r))
        ids.append(b)
    return tpl, ids


def _roll(factory, tpl, n, steps):
    px = factory(tpl, n, SimConfig()); px.gpu_init()
    rbd = px.cuda_rigid_body_data.torch().view(n, px.bodies_per_env, 13)
    rbd[:, tpl.body_id("table-workspace"), :7] = torch.tensor([-0.12, 0.0, -sb.TABLE_HEIGHT, np.cos(np.pi / 4), 0, 0, np.sin(np.pi / 4)], device=rbd.device)
    px.gpu_apply_all()
    px.set_scene_offsets(np.zeros((n, 3)))
    for _ in range(steps):
        px.step()
    px.gpu_fetch_all()
    return rbd.cpu().clone(), px


def test_an_interpenetrating_heap_comes_apart_and_rests(oracle_factory):
    tpl, ids = _heap(6, 7, 0.03)
    st, px = _roll(oracle_factory, tpl, 2, 400)
    assert torch.isfinite(st).all() and torch.equal(st[0], st[1])
    assert (st[0, ids, 2] > -0.002).all() and st[0, ids, 7:13].abs().max() < 0.2, st[0, ids, 7:13].abs().max()      # on the table (a hull's origin may lie close to one of its faces), at rest


def test_hip_heaps_of_hulls_match_the_oracle_under_emulation(oracle_factory):
    """30, 48 and 60 coordinates (the 32- and 64-coordinate solver forms); the last heap also runs over the contact capacity: same rows dropped"""
    from emu_backend import EmuPhysxSystem
    for nbody, seed, spread, n in ((5, 7, 0.03, 4), (8, 7, 0.04, 2), (10, 5, 0.04, 2)):
        tpl, ids = _heap(nbody, seed, spread)
        emu, pa = _roll(lambda t, k, c: EmuPhysxSystem(t, k, c), tpl, n, 12)
        orc, pb = _roll(oracle_factory, tpl, n, 12)
        assert torch.equal(emu, orc), (nbody, (emu - orc).abs().max().item())
        assert pa.get_overflow() == pb.get_overflow()


@pytest.mark.gpu
def test_hip_heaps_of_hulls_identical_envs_identical_bits_and_the_oracles(oracle_factory):
    from maniskill_amd.physx import PhysxGpuSystem
    for nbody, spread in ((5, 0.03), (8, 0.04)):
        tpl, ids = _heap(nbody, 7, spread)
        hip, _ = _roll(lambda t, n, c: PhysxGpuSystem("cuda:0", t, n, c), tpl, 1024, 30)       # ~8 hull items per narrowphase block: waves are shared
        orc, _ = _roll(oracle_factory, tpl, 2, 30)
        same = (hip == hip[:1]).flatten(1).all(1)
        assert same.all(), f"{int((~same).sum())} of 1024 identical envs differ from env 0 ({nbody} bodies)"
        assert torch.equal(hip[0], orc[0]), (nbody, (hip[0] - orc[0]).abs().max().item())
