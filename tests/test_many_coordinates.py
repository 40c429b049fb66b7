"""More than 32 generalized velocities per sub-scene (MSK_MAX_NV 64: the 64-coordinate forms of the solver and of the joint-space dynamics):
several free bodies next to each other / next to an arm, and articulations of more than 31 joints -- what FMBAssembly1Easy-v1 (Panda + five
loose pieces: 39) and UnitreeG1Stand-v1 (37 joints + a floating root: 43) need (mani_skill/utils/structs/types.py:18-23 sizes such scenes at
run time).  Known answers on the oracle; HIP against the oracle under the emulation of tests/hipemu (CPU suite) and on hardware (-m gpu)."""
import numpy as np
import pytest
import torch

from maniskill_amd import _native as N
from maniskill_amd.envs import scene_builders as sb
from maniskill_amd.physx import SceneTemplate, SimConfig


def _boxes_on_a_table(nbox, seed=0):
    rng = np.random.default_rng(seed)
    tpl = SceneTemplate(); sb.add_table_scene(tpl)
    ids, halves = [], []
    for k in range(nbox):
        hs = rng.uniform(0.015, 0.03, size=3); m = 500 * 8 * hs.prod()
        b = tpl.add_actor(f"b{k}", N.BODY_DYNAMIC, p=(0.07 * (k % 3) - 0.07, 0.07 * (k // 3) - 0.07, 0.04 + 0.07 * (k % 2)), mass=m,
                          inertia6=tuple(m / 3 * np.array([hs[1] ** 2 + hs[2] ** 2, hs[0] ** 2 + hs[2] ** 2, hs[0] ** 2 + hs[1] ** 2])) + (0, 0, 0))
        tpl.add_shape(b, N.SHAPE_BOX, params=tuple(hs))
        ids.append(b); halves.append(hs)
    return tpl, ids, halves


def _roll(factory, tpl, ids, n, steps, every=10):
    px = factory(tpl, n, SimConfig()); px.gpu_init()
    rbd = px.cuda_rigid_body_data.torch().view(n, px.bodies_per_env, 13)
    rbd[:, tpl.body_id("table-workspace"), :7] = torch.tensor([-0.12, 0.0, -sb.TABLE_HEIGHT, np.cos(np.pi / 4), 0, 0, np.sin(np.pi / 4)], device=rbd.device)
    for e in range(n):                                   # every env its own sideways push
        rbd[e, ids, 7:10] = torch.tensor([0.1 * e, -0.05 * e, 0.0], device=rbd.device)
    px.gpu_apply_all()
    out = []
    for t in range(steps):
        px.step()
        if t % every == every - 1:
            px.gpu_fetch_all(); out.append(rbd.cpu().clone())
    return torch.stack(out), px


def test_seven_and_nine_free_boxes_come_to_rest_on_the_table(oracle_factory):
    """42 and 54 coordinates: every box -- also those whose coordinates lie beyond the 32nd -- lands and rests at its half height"""
    for nbox in (7, 9):
        tpl, ids, halves = _boxes_on_a_table(nbox)
        traj, px = _roll(oracle_factory, tpl, ids, 2, 150)
        last = traj[-1]
        assert torch.isfinite(traj).all() and px.get_overflow() == 0
        for b, hs in zip(ids, halves):
            z = last[0, b, 2].item()
            assert min(hs) - 2e-3 < z < max(hs) + 2e-3 + 2 * max(hs), (nbox, b, z)       # on the table (or on another box), not through it
            assert last[0, b, 7:13].abs().max() < 0.05, (nbox, b)
        assert last[0, ids[-1], 2] > 0.01                                                    # the LAST body's coordinates (>= 36) felt the table


def test_hip_many_free_boxes_match_oracle_under_emulation(oracle_factory):
    from emu_backend import EmuPhysxSystem
    for nbox in (6, 9):
        tpl, ids, _ = _boxes_on_a_table(nbox, seed=1)
        a, pa = _roll(lambda t, n, c: EmuPhysxSystem(t, n, c), tpl, ids, 3, 60)
        b, pb = _roll(oracle_factory, tpl, ids, 3, 60)
        assert torch.equal(a, b), (nbox, (a - b).abs().max().item())
        for e in range(3):
            ia, va = pa.get_contacts(e); ib, vb = pb.get_contacts(e)
            assert ia.shape == ib.shape and (ia == ib).all() and np.array_equal(va, vb)
        assert pa.get_overflow() == 0 and pb.get_overflow() == 0


@pytest.mark.gpu
def test_hip_many_free_boxes_match_oracle(oracle_factory):
    from maniskill_amd.physx import PhysxGpuSystem
    for nbox in (6, 9):
        tpl, ids, _ = _boxes_on_a_table(nbox, seed=1)
        a, pa = _roll(lambda t, n, c: PhysxGpuSystem("cuda:0", t, n, c), tpl, ids, 70, 80)
        b, pb = _roll(oracle_factory, tpl, ids, 70, 80)
        assert torch.equal(a, b), (nbox, (a - b).abs().max().item())
        assert pa.get_overflow() == 0


def _tree_with_friction_beyond_the_32nd_coordinate(nbranch=12, nlink=3):
    """A fixed hub carrying `nbranch` short chains of `nlink` hinges (36 coordinates): joint friction on the LAST link of every branch, so
    friction blocks sit on coordinates below and above 32 (DModel::jfric_mask is one bit per coordinate: a 32-bit mask put the friction of
    coordinate 32 + k on coordinate k)."""
    tpl = SceneTemplate()
    art = tpl.add_articulation("tree", root_p=(0, 0, 1.0))
    hub = tpl.add_link(art, "hub", -1, N.JOINT_FIXED, mass=1.0, inertia6=(1e-2, 1e-2, 1e-2, 0, 0, 0))
    tips = []
    for b in range(nbranch):
        c, s = np.cos(2 * np.pi * b / nbranch), np.sin(2 * np.pi * b / nbranch)
        parent = hub
        for k in range(nlink):
            pin = [0.05 * c, 0.05 * s, 0.0, 1, 0, 0, 0] if k == 0 else [0.0, 0.0, -0.1, 1, 0, 0, 0]
            parent = tpl.add_link(art, f"l{b}_{k}", parent, N.JOINT_REVOLUTE, joint_name=f"j{b}_{k}", pose_in_parent=pin, mass=0.2, com=(0.02, 0, -0.05),
                                  inertia6=(2e-4, 2e-4, 1e-4, 0, 0, 0), friction=(0.05 + 0.01 * b) if k == nlink - 1 else 0.0)
        tips.append(parent)
    return tpl, hub, tips


def _roll_tree(factory, n, steps):
    tpl, hub, tips = _tree_with_friction_beyond_the_32nd_coordinate()
    px = factory(tpl, n, SimConfig()); px.gpu_init()
    rbd = px.cuda_rigid_body_data.torch().view(n, px.bodies_per_env, 13)
    rbd[:, hub, :7] = torch.tensor([0.0, 0.0, 1.0, 1, 0, 0, 0], device=rbd.device)
    q = px.cuda_articulation_qpos.torch(); qd = px.cuda_articulation_qvel.torch()
    nq = 36
    for e in range(n):
        q[e, :nq] = torch.linspace(-0.6, 0.7, nq, device=q.device) * (1.0 + 0.1 * e)
        qd[e, :nq] = torch.linspace(1.0, -1.5, nq, device=q.device)
    px.gpu_apply_all()
    out = []
    for t in range(steps):
        px.step()
        px.gpu_fetch_all()
        out.append(torch.cat([q[:, :nq], qd[:, :nq]], 1).cpu().clone())
    return torch.stack(out), px


def test_joint_friction_beyond_the_32nd_coordinate_matches_oracle_under_emulation(oracle_factory):
    from emu_backend import EmuPhysxSystem
    a, _ = _roll_tree(lambda t, n, c: EmuPhysxSystem(t, n, c), 2, 25)
    b, _ = _roll_tree(oracle_factory, 2, 25)
    assert torch.isfinite(b).all()
    assert torch.equal(a, b), (a - b).abs().max().item()
    # the friction is felt where it was put: the tip joints (coordinates 2, 5, ..., 35) lose speed against a frictionless twin's
    assert b[-1, 0, 36 + 35].abs() < 5.0


@pytest.mark.gpu
def test_hip_joint_friction_beyond_the_32nd_coordinate_matches_oracle(oracle_factory):
    from maniskill_amd.physx import PhysxGpuSystem
    a, _ = _roll_tree(lambda t, n, c: PhysxGpuSystem("cuda:0", t, n, c), 70, 40)
    b, _ = _roll_tree(oracle_factory, 70, 40)
    assert torch.equal(a, b), (a - b).abs().max().item()
