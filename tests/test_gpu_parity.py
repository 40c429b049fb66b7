"""Parity tests proper (run with ``-m gpu`` on an MI355X): the HIP backend, called through the C ABI
(include/msk_physx.h), against the CPU oracle on identical seeds/actions.

Bar (BASELINE.json north_star): bit-exact contact-pair indices, fp32 pose/velocity within 1e-4 rel
over 100 steps.  The HIP kernels mirror the oracle's arithmetic order, so most checks below are in
fact bit-exact; the asserted tolerance is the stated one.
"""
import os

import numpy as np
import pytest
import torch

from maniskill_amd.envs import scene_builders as sb
from maniskill_amd.envs.pick_cube import PickCubeEnv
from maniskill_amd.physx import PhysxGpuSystem, SceneTemplate

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
DEV = "cuda:0"
RTOL, ATOL = 1e-4, 1e-5  # stated tolerance: 1e-4 relative; ATOL covers values that pass through zero


def _close(a, b):
    return np.allclose(a, b, rtol=RTOL, atol=ATOL)


def test_native_library_is_what_runs():
    env = PickCubeEnv(num_envs=4, device=DEV)
    env.step(torch.zeros(4, 8, device=DEV))
    with open("/proc/self/maps") as f:
        maps = f.read()
    assert "libmsk_physx.so" in maps
    assert env.px.cuda_rigid_body_data.torch().is_cuda


def test_rollout_matches_oracle_100_steps(oracle_factory):
    n, steps = 256, 100
    gpu = PickCubeEnv(num_envs=n, device=DEV, fused=False)   # same host code on both sides: a pure physics comparison
    cpu = PickCubeEnv(num_envs=n, px_factory=oracle_factory)
    og, _ = gpu.reset(seed=2022)
    oc, _ = cpu.reset(seed=2022)
    assert torch.equal(og.cpu(), oc)
    gen = torch.Generator().manual_seed(0)
    nbit = 0
    for t in range(steps):
        a = 2 * torch.rand(n, 8, generator=gen) - 1
        og, rg, tg, ug, _ = gpu.step(a.to(DEV))
        oc, rc, tc, uc, _ = cpu.step(a)
        assert _close(og.cpu().numpy(), oc.numpy()), f"obs diverged at step {t}"
        assert _close(rg.cpu().numpy(), rc.numpy())
        assert torch.equal(tg.cpu(), tc) and torch.equal(ug.cpu(), uc)
        nbit += int(torch.equal(og.cpu(), oc))
        if t % 10 == 9:  # bit-exact contact-pair indices (shape ids + body pair), and contact data
            for e in range(0, n, 16):
                gi, gv = gpu.px.get_contacts(e)
                ci, cv = cpu.px.get_contacts(e)
                assert gi.shape == ci.shape and (gi == ci).all(), f"contact pairs differ, env {e} step {t}"
                assert _close(gv, cv)
    assert _close(gpu.get_state().cpu().numpy(), cpu.get_state().numpy())
    assert gpu.px.get_overflow() & 6 == 0          # no solver scheduling error (misclassified env, contact total out of step)
    print(f"bit-exact observation steps: {nbit}/{steps}")


def test_matches_committed_golden_rollout():
    """tests/golden/pickcube_oracle_rollout.npz travels to the GPU box (made by tests/golden/make_golden.py)."""
    g = np.load(os.path.join(HERE, "golden", "pickcube_oracle_rollout.npz"))
    n = g["actions"].shape[1]
    env = PickCubeEnv(num_envs=n, device=DEV, fused=False)
    obs, _ = env.reset(seed=2022)
    assert _close(obs.cpu().numpy(), g["obs"][0])
    off = g["contact_ids_offsets"]
    for t in range(g["actions"].shape[0]):
        obs, r, *_ = env.step(torch.from_numpy(g["actions"][t]).to(DEV))
        assert _close(obs.cpu().numpy(), g["obs"][t + 1]), t
        assert _close(r.cpu().numpy(), g["rew"][t])
        ids = [env.px.get_contacts(e)[0][:, :2] for e in range(n)]
        flat = np.array([len(i) for i in ids] + [int(x) for i in ids for x in i.reshape(-1)], dtype=np.int32)
        assert (flat == g["contact_ids"][off[t]:off[t + 1]]).all(), f"contact-pair indices differ at step {t}"
    assert _close(env.get_state().cpu().numpy(), g["state"])


def test_scripted_grasp_matches_oracle(oracle_factory):
    """Contact-rich case: close the gripper on the cube and lift it (same script as the oracle's
    known-answer test, fixture tests/golden/grasp_waypoints.json), HIP vs oracle step by step."""
    import json

    wp = json.load(open(os.path.join(HERE, "golden", "grasp_waypoints.json")))
    n = 8
    outs = []
    for kw in (dict(device=DEV, fused=False), dict(px_factory=oracle_factory)):
        env = PickCubeEnv(num_envs=n, robot_init_qpos_noise=0.0, **kw)
        env.reset(seed=0)
        dev = env.device
        st = env.get_state()
        st[:, 13:16] = torch.tensor([0.0, 0.0, 0.02], device=dev)       # cube at the origin, axis aligned
        st[:, 16:20] = torch.tensor([1.0, 0, 0, 0], device=dev)
        env.set_state(st)
        tq = env._target_qpos_buf
        traj = []

        def go(qa, qb, grip, steps):
            for k in range(steps):
                a = (k + 1) / steps
                tq[:, :7] = torch.tensor(np.asarray(qa) * (1 - a) + np.asarray(qb) * a, dtype=torch.float32, device=dev)
                tq[:, 7:9] = grip
                env.px.gpu_apply_articulation_target_position()
                for _ in range(5):
                    env.px.step()
                env.px.gpu_fetch_all()
                traj.append(env.get_state().cpu().numpy().copy())

        go(wp["q_rest"], wp["q_pre"], 0.04, 20)
        go(wp["q_pre"], wp["q_grasp"], 0.04, 20)
        go(wp["q_grasp"], wp["q_grasp"], -0.01, 20)
        go(wp["q_grasp"], wp["q_lift"], -0.01, 40)
        ids = [env.px.get_contacts(e)[0] for e in range(n)]
        outs.append((np.stack(traj), env.is_grasping().cpu().numpy(), env.cube_pose.cpu().numpy(), ids))
    (tg, gg, cg, ig), (tc, gc, cc, ic) = outs
    assert gc.all() and cc[:, 2].min() > 0.1, "oracle itself failed to grasp and lift (known-answer T2)"
    assert (gg == gc).all()
    assert _close(tg, tc)
    for a, b in zip(ig, ic):
        assert a.shape == b.shape and (a == b).all()


def test_cube_known_answers_on_gpu():
    """T2 known answers on the HIP path at full size: 4096 cubes rest on the table, m g dt of normal impulse."""
    n = 4096
    tpl = SceneTemplate()
    sb.add_table_scene(tpl)
    cube = sb.add_cube(tpl, "cube", 0.02, (0, 0, 0.02))
    px = PhysxGpuSystem(DEV, tpl, n, None)
    px.gpu_init()
    rbd = px.cuda_rigid_body_data.torch().view(n, -1, 13)
    rbd[::2, cube, 2] = 0.1  # every other cube is dropped from 8 cm
    px.gpu_apply_all()
    for _ in range(300):
        px.step()
    px.gpu_fetch_all()
    torch.cuda.synchronize()
    assert torch.isfinite(rbd).all()
    assert (rbd[:, cube, 2] - 0.02).abs().max().item() < 1.5e-3
    assert rbd[:, cube, 7:10].norm(dim=1).max().item() < 1e-2
    ids, vals = px.get_contacts(1)
    assert len(ids) == 4
    assert abs(vals[:, 7].sum() - 1000 * 0.04 ** 3 * 9.81 * px.timestep) < 2e-5


def test_full_size_properties_4096():
    """Size-independent properties at BASELINE.json's size: partition invariance, state round trip,
    partial-reset isolation, finiteness."""
    n = 4096
    torch.manual_seed(0)
    env = PickCubeEnv(num_envs=n, device=DEV)
    half = PickCubeEnv(num_envs=n // 2, device=DEV, env_index_offset=n // 2, total_envs=n)
    o, _ = env.reset(seed=2022)
    oh, _ = half.reset(seed=2022)
    assert torch.equal(o[n // 2:], oh)
    acts = [2 * torch.rand(n, 8, device=DEV) - 1 for _ in range(12)]
    for a in acts[:6]:
        o, *_ = env.step(a)
        oh, *_ = half.step(a[n // 2:])
    assert torch.isfinite(o).all()
    assert torch.equal(o[n // 2:], oh), "env results depend on the batch they run in"
    # state round trip (reference tests/test_envs.py:196-212): replays from a saved state are reproducible
    # (a teleport drops the env's contact warm-start cache, so nothing but the state carries over)
    st = env.get_state().clone()
    cont = [env.step(a)[0].clone() for a in acts[6:9]]

    def replay():
        env.set_state(st)
        env._target_qpos[:] = env.qpos
        return [env.step(a)[0].clone() for a in acts[6:9]]

    r1, r2 = replay(), replay()
    assert all(torch.equal(x, y) for x, y in zip(r1, r2))
    # against the original continuation only the first step is compared: set_state re-derives the internal
    # pose from the offset frame (1 ulp) and the continuation kept its warm-start cache
    dev = (r1[0] - cont[0]).abs().max(dim=1)[0]
    assert (dev < 1e-2).float().mean().item() > 0.95
    # partial reset leaves the other envs bit-identical
    before = env.get_state().clone()
    idx = torch.arange(0, n, 3, device=DEV)
    env.reset(options={"env_idx": idx})
    after = env.get_state()
    keep = torch.ones(n, dtype=torch.bool, device=DEV)
    keep[idx] = False
    assert torch.equal(before[keep], after[keep])
    assert not torch.equal(before[idx], after[idx])
    sizes = env.px.get_overflow()
    assert sizes == 0, "per-env contact capacity exceeded"


def test_fused_task_kernels_match_torch_reference():
    """include/msk_task.h: the fused controller / evaluate / obs / reward kernels against the torch mirror of
    the reference task code (maniskill_amd/envs/pick_cube.py), same physics, 4096 envs, 30 steps."""
    n = 4096
    a = PickCubeEnv(num_envs=n, device=DEV, fused=True)
    b = PickCubeEnv(num_envs=n, device=DEV, fused=False)
    oa, _ = a.reset(seed=2022)
    ob, _ = b.reset(seed=2022)
    # the torch path reports (p + offset) - offset: its positions carry the fp32 spacing of the 160 m scene grid
    assert np.allclose(oa.cpu().numpy(), ob.cpu().numpy(), rtol=1e-4, atol=5e-5)
    torch.manual_seed(3)
    nflag = 0
    for t in range(30):
        act = 3 * torch.rand(n, 8, device=DEV) - 1.5   # exercises the clipping too
        oa, ra, ta, ua, ia = a.step(act)
        ob, rb, tb, ub, ib = b.step(act)
        assert torch.equal(a.get_state(), b.get_state()), f"physics state differs at step {t}"
        A, B = oa.cpu().numpy(), ob.cpu().numpy()
        same_flag = A[:, 18] == B[:, 18]
        nflag += int((~same_flag).sum())
        assert np.allclose(np.delete(A, 18, axis=1), np.delete(B, 18, axis=1), rtol=1e-4, atol=5e-5)
        assert np.allclose(ra.cpu().numpy()[same_flag], rb.cpu().numpy()[same_flag], atol=2e-5)
        assert torch.equal(ua, ub) and (ta == tb).float().mean().item() > 0.999
        for k in ("success", "is_obj_placed", "is_robot_static", "is_grasped"):
            assert (ia[k] == ib[k]).float().mean().item() > 0.999, k
    assert nflag <= 0.001 * 30 * n   # is_grasped may flip only on the 85 degree / 0.5 N boundary


def test_timing_api_reports_kernels():
    env = PickCubeEnv(num_envs=128, device=DEV)
    env.px.timing_enable(5)
    env.step(torch.zeros(128, 8, device=DEV))
    t = env.px.timing_read()
    assert set(t) >= {"k_dynamics", "k_narrowphase", "k_csolve", "substep"} and all(v[1] == 5 for v in t.values())
    own = sum(t[k][0] for k in ("k_dynamics", "k_narrowphase", "k_csolve"))
    assert 0.0 < own <= t["substep"][0] * 1.001      # the kernels' own durations fit inside the span from the first begin to the last end
    assert all(v[0] > 0 for v in t.values())


@pytest.mark.parametrize("caps", [(-1, -1, -1), (-1, -1, 32), (-1, 20, 32), (4, 8, 16)])
def test_every_solver_class_computes_the_same_bits(caps):
    """The solver sorts envs into LDS capacity classes (include/msk_physx.h: msk_set_solver_classes); which class an
    env lands in is scheduling only.  Force every env through class 3 (A image in global memory), class 2, classes
    1-2, and a fine split: each rollout must equal, bit for bit, the default schedule's (which the tests above hold
    against the oracle)."""
    n, steps = 64, 40
    ref = PickCubeEnv(num_envs=n, device=DEV, fused=False)
    alt = PickCubeEnv(num_envs=n, device=DEV, fused=False)
    alt.px.set_solver_classes(caps)
    ref.reset(seed=7)
    alt.reset(seed=7)
    gen = torch.Generator().manual_seed(1)
    seen = None
    for t in range(steps):
        a = (2 * torch.rand(n, 8, generator=gen) - 1).to(DEV)
        ref.step(a)
        alt.step(a)
        counts = alt.px.get_solver_class_counts()      # one count per class of include/msk_physx.h (MSK_SOLVER_CLASSES: 4 + the wide class)
        assert counts.sum() == n
        seen = np.zeros_like(counts) if seen is None else seen
        seen += counts
        assert torch.equal(ref.get_state(), alt.get_state()), f"classes {caps}: state differs at step {t}"
    if caps[0] < 0:
        assert seen[0] == 0
    if caps == (-1, -1, -1):
        assert seen[3] == n * steps
    assert seen[1:4].sum() > 0
    assert seen[4:].sum() == 0                      # the wide class (capacity 1, the default) takes envs of more than 64 blocks only: none in this rollout
    assert alt.px.get_overflow() == 0


@pytest.mark.parametrize("n", [1, 37, 100, 1000])
def test_env_counts_that_are_not_powers_of_two(oracle_factory, n):
    """The launch geometry (env groups, 64-env classification chunks, solver lists) must not care about the env count."""
    gpu = PickCubeEnv(num_envs=n, device=DEV, fused=False)
    cpu = PickCubeEnv(num_envs=n, px_factory=oracle_factory)
    gpu.reset(seed=3)
    cpu.reset(seed=3)
    gen = torch.Generator().manual_seed(5)
    for t in range(12):
        a = 2 * torch.rand(n, 8, generator=gen) - 1
        gpu.step(a.to(DEV))
        cpu.step(a)
        assert gpu.px.get_solver_class_counts().sum() == n
    assert _close(gpu.get_state().cpu().numpy(), cpu.get_state().numpy())
    assert gpu.px.get_overflow() == 0


def test_body_impulse_query_matches_oracle(oracle_factory):
    """gpu_create_contact_body_impulse_query: net contact impulses on the cube, a finger and the table, after a random rollout."""
    n = 128
    gpu = PickCubeEnv(num_envs=n, device=DEV, fused=False)
    cpu = PickCubeEnv(num_envs=n, px_factory=oracle_factory)
    bodies = [gpu._b_cube, gpu._b_f1, gpu._b_table, gpu._b_tcp]
    qg = gpu.px.gpu_create_contact_body_impulse_query(bodies)
    qc = cpu.px.gpu_create_contact_body_impulse_query(bodies)
    gpu.reset(seed=21); cpu.reset(seed=21)
    gen = torch.Generator().manual_seed(8)
    for _ in range(25):
        a = 2 * torch.rand(n, 8, generator=gen) - 1
        gpu.step(a.to(DEV)); cpu.step(a)
    gpu.px.gpu_query_contact_body_impulses(qg)
    cpu.px.gpu_query_contact_body_impulses(qc)
    g, c = qg.cuda_impulses.torch().cpu().numpy(), qc.cuda_impulses.torch().numpy()
    assert g.shape == (n * 4, 3) and np.abs(c).max() > 1e-4
    assert np.allclose(g, c, rtol=1e-4, atol=1e-6)


def test_per_env_box_instances_match_oracle(oracle_factory):
    """msk_declare_env_box / msk_set_env_boxes / msk_set_env_masses: every env drops a cube of its own size and mass next to
    a declared static block with its own size and offset; HIP state and picture against the oracle's."""
    from maniskill_amd.physx import SimConfig
    from maniskill_amd.render import CameraConfig, RenderCameraGroup, attach_template_visuals, look_at
    from maniskill_amd import _native as NN

    n = 64
    rng = np.random.default_rng(5)
    half = rng.uniform(0.01, 0.04, size=(n, 3)).astype(np.float32)
    mass = (8 * half.prod(1) * 1000.0).astype(np.float32)
    inertia = np.stack([mass / 3 * (half[:, 1] ** 2 + half[:, 2] ** 2), mass / 3 * (half[:, 0] ** 2 + half[:, 2] ** 2),
                        mass / 3 * (half[:, 0] ** 2 + half[:, 1] ** 2)], axis=1).astype(np.float32)
    bhalf = rng.uniform(0.02, 0.05, size=(n, 3)).astype(np.float32)
    bpos = np.concatenate([rng.uniform(-0.02, 0.02, size=(n, 2)), bhalf[:, 2:3]], axis=1).astype(np.float32)
    results = []
    for dev, fac in ((DEV, None), (None, oracle_factory)):
        tpl = SceneTemplate()
        table = sb.add_table_scene(tpl)
        cube = sb.add_cube(tpl, "cube", 0.02, (0, 0, 0.1))
        cube_shape = tpl.nshapes - 1
        block = tpl.add_actor("block", NN.BODY_KINEMATIC, p=(0, 0, 0))
        block_shape = tpl.add_shape(block, NN.SHAPE_BOX, params=(0.03, 0.03, 0.03))
        tpl.declare_env_box(cube_shape); tpl.declare_env_box(block_shape); tpl.declare_env_mass(cube)
        px = PhysxGpuSystem(torch.device(dev), tpl, n, SimConfig()) if fac is None else fac(tpl, n, SimConfig())
        px.gpu_init()
        px.set_scene_offsets(np.zeros((n, 3)))
        px.set_env_boxes(cube_shape, half); px.set_env_masses(cube, mass, inertia)
        px.set_env_boxes(block_shape, bhalf, bpos)
        attach_template_visuals(px, tpl)
        p, q = look_at(eye=[0.35, 0.1, 0.4], target=[0, 0, 0.02])
        cam = RenderCameraGroup(px, CameraConfig("cam", p, q, 128, 128, np.pi / 2, 0.01, 100.0))
        cam.enable_color()
        rbd = px.cuda_rigid_body_data.torch().view(n, px.bodies_per_env, 13)
        rbd[:, cube, :3] = torch.tensor([0.01, 0.0, 0.16], device=rbd.device)      # lands on the block, topples off or stays
        rbd[:, cube, 3:7] = torch.tensor([0.9990482, 0.0308436, 0.0308436, 0.0], device=rbd.device)
        rbd[:, table, :7] = torch.tensor([-0.12, 0.0, -sb.TABLE_HEIGHT, np.cos(np.pi / 4), 0, 0, np.sin(np.pi / 4)], device=rbd.device)
        rbd[:, block, 3] = 1.0
        px.gpu_apply_all()
        for _ in range(120):
            px.step()
        px.gpu_fetch_all()
        cam.take_picture()
        results.append((rbd.cpu().clone(), cam.get_picture_cuda().torch().cpu().clone(), cam.get_picture_cuda("Color").torch().cpu().clone()))
    (rg, pg, cg), (rc, pc, cc) = results
    assert _close(rg.numpy(), rc.numpy())
    assert torch.equal(pg, pc) and torch.equal(cg, cc)
    z = rc[:, 1, 2]
    assert (z > 0.005).all() and z.std() > 0.01           # different sizes, different resting heights


def test_external_wrench_matches_oracle(oracle_factory):
    """cuda_rigid_body_force / _torque + gpu_apply_rigid_dynamic_force / _torque: random wrenches on the PickCube cube every
    few substeps (it is kicked around the table top and into the arm), HIP state against the oracle's."""
    n = 64
    gen = torch.Generator().manual_seed(21)
    forces = (torch.rand(25, n, 3, generator=gen) * 2 - 1) * 3.0
    torques = (torch.rand(25, n, 3, generator=gen) * 2 - 1) * 2e-3
    out = []
    for dev, fac in ((DEV, None), (None, oracle_factory)):
        env = PickCubeEnv(num_envs=n, device=dev, fused=False) if fac is None else PickCubeEnv(num_envs=n, px_factory=fac)
        env.reset(seed=4)
        px = env.px
        F = px.cuda_rigid_body_force.torch().view(n, px.bodies_per_env, 4)
        T = px.cuda_rigid_body_torque.torch().view(n, px.bodies_per_env, 4)
        for k in range(25):
            F[:, env._b_cube, :3] = forces[k].to(F.device)
            px.gpu_apply_rigid_dynamic_force()
            if k % 2 == 0:
                T[:, env._b_cube, :3] = torques[k].to(T.device)
                px.gpu_apply_rigid_dynamic_torque()
            for _ in range(3):      # the wrench acts in the first of the three substeps only
                px.step()
        px.gpu_fetch_all()
        out.append(px.cuda_rigid_body_data.torch().cpu().clone())
    assert _close(out[0].numpy(), out[1].numpy())
    moved = (out[1].view(n, -1, 13)[:, env._b_cube, :2] - out[1].view(n, -1, 13)[0, env._b_cube, :2]).abs().max()
    assert moved > 1e-3


def test_link_incoming_joint_forces_match_oracle(oracle_factory):
    """cuda_articulation_link_incoming_joint_forces: the inverse-dynamics kernel against the oracle's statement, on a Panda that
    carries its own weight (known answers: tests/test_oracle_physics.py) and on PickCube rollouts where the arm is driven
    around and pushes on the table and the cube (contact wrenches, accelerations, drive forces all present)."""
    from test_oracle_physics import _panda_under_gravity

    res = []
    for fac in (lambda t, n, c: PhysxGpuSystem(torch.device(DEV), t, n, c), oracle_factory):
        px, tpl = _panda_under_gravity(fac, 8)
        tq = px.cuda_articulation_target_qpos.torch().view(8, -1)
        tq[:, 1] += torch.linspace(-0.3, 0.3, 8).to(tq.device)      # every env swings to its own target
        px.gpu_apply_articulation_target_position()
        for _ in range(40):
            px.step()
        px.gpu_fetch_all()
        res.append(px.get_link_incoming_joint_forces().cpu().clone())
    assert res[0].shape == res[1].shape and res[1][:, 1, :3].norm(dim=-1).median() > 100.0   # (the last env's arm leans on the table)
    assert np.allclose(res[0].numpy(), res[1].numpy(), rtol=1e-3, atol=2e-3)
    n = 64
    gen = torch.Generator().manual_seed(8)
    acts = 2 * torch.rand(30, n, 8, generator=gen) - 1
    out = []
    for dev, fac in ((DEV, None), (None, oracle_factory)):
        env = PickCubeEnv(num_envs=n, device=dev, fused=False) if fac is None else PickCubeEnv(num_envs=n, px_factory=fac)
        env.reset(seed=11)
        for k in range(30):
            env.step(acts[k].to(env.device))
        out.append(env.px.get_link_incoming_joint_forces().cpu().clone())
    assert out[1].abs().max() > 1.0
    assert np.allclose(out[0].numpy(), out[1].numpy(), rtol=1e-3, atol=2e-3)


@pytest.mark.gpu
def test_hip_vs_oracle_at_the_metrics_env_count(oracle_factory):
    """The oracle comparison at BASELINE.json's env count and the north star's length: 4096 PickCube envs x 100 control steps on the HIP
    side; the oracle runs the first 512 of them (the same global seeds and grid cells: results do not depend on the batch an env runs in,
    test_full_size_properties_4096) -- contact-pair sets bit-exact, states within 1e-4 relative, no solver scheduling flags."""
    n, m, steps = 4096, 512, 100
    gpu = PickCubeEnv(num_envs=n, device="cuda:0", fused=False)
    cpu = PickCubeEnv(num_envs=m, px_factory=oracle_factory, env_index_offset=0, total_envs=n)
    og, _ = gpu.reset(seed=2022); oc, _ = cpu.reset(seed=2022)
    assert torch.allclose(og[:m].cpu(), oc, atol=2e-6)
    gen = torch.Generator().manual_seed(5)
    for t in range(steps):
        a = 2 * torch.rand(n, 8, generator=gen) - 1
        og, rg, tg, *_ = gpu.step(a.to("cuda:0"))
        oc, rc, tc, *_ = cpu.step(a[:m])
        assert torch.isfinite(og).all() and torch.isfinite(oc).all()
        assert torch.allclose(og[:m].cpu(), oc, rtol=1e-4, atol=1e-5), (t, float((og[:m].cpu() - oc).abs().max()))
        assert torch.equal(tg[:m].cpu(), tc)
        if t % 20 == 19:
            assert np.array_equal(gpu.px.get_env_contact_counts()[:m], cpu.px.get_env_contact_counts())
            for e in (0, 1, 77, 300, 511):
                ig, _ = gpu.px.get_contacts(e)
                ic, _ = cpu.px.get_contacts(e)
                assert np.array_equal(ig, ic), (t, e)          # shape-pair ids of every contact point, in order
    sg, sc = gpu.get_state()[:m].cpu(), cpu.get_state()
    assert torch.allclose(sg, sc, rtol=1e-4, atol=1e-5)
    assert gpu.px.get_overflow() & 6 == 0


@pytest.mark.gpu
def test_hip_vs_oracle_late_in_the_rollout(oracle_factory):
    """The contact-rich regime of the 1000-step benchmark rollout (arms lying on the table: the large solver classes, EPA, contact
    overflow): 4096 HIP envs are rolled 300 control steps under random actions, then both sides start from the snapshot of that state
    (a teleport: empty warm-start caches on both sides) -- the HIP batch of 4096 and the oracle on envs 2048..2175 -- and are compared
    over steps 300..320 with contact-pair ids."""
    n, m, e0 = 4096, 128, 2048
    gpu = PickCubeEnv(num_envs=n, device="cuda:0", fused=False)
    cpu = PickCubeEnv(num_envs=m, px_factory=oracle_factory, env_index_offset=e0, total_envs=n)
    gpu.reset(seed=2022); cpu.reset(seed=2022)
    gen = torch.Generator().manual_seed(9)
    for t in range(300):
        gpu.step((2 * torch.rand(n, 8, generator=gen) - 1).to("cuda:0"))
    cc = gpu.px.get_env_contact_counts()
    assert cc.max() >= 8 and (cc[e0:e0 + m] > 4).sum() >= 3, (cc.max(), np.bincount(cc[e0:e0 + m]))            # the regime this test is about
    snap = gpu.get_state().clone()
    gpu.reset(seed=1)                   # away from the snapshot, so that writing it back is a teleport for every env: apply leaves rows that
    gpu.set_state(snap)                 # still hold the fetched values alone, and such an env would keep its warm-start cache
    gpu._target_qpos[:] = gpu.qpos
    cpu.set_state(snap[e0:e0 + m].cpu())
    cpu._target_qpos[:] = cpu.qpos
    for t in range(20):
        a = 2 * torch.rand(n, 8, generator=gen) - 1
        og, *_ = gpu.step(a.to("cuda:0"))
        oc, *_ = cpu.step(a[e0:e0 + m])
        assert torch.allclose(og[e0:e0 + m].cpu(), oc, rtol=1e-4, atol=1e-5), (t, float((og[e0:e0 + m].cpu() - oc).abs().max()))
        if t % 5 == 4:
            assert np.array_equal(gpu.px.get_env_contact_counts()[e0:e0 + m], cpu.px.get_env_contact_counts())
            for e in range(0, m, 16):
                ig, _ = gpu.px.get_contacts(e0 + e)
                ic, _ = cpu.px.get_contacts(e)
                assert np.array_equal(ig, ic), (t, e)
    assert torch.allclose(gpu.get_state()[e0:e0 + m].cpu(), cpu.get_state(), rtol=1e-4, atol=1e-5)
    assert gpu.px.get_overflow() & 6 == 0


@pytest.mark.parametrize("n", [1, 3, 5, 17, 63, 65, 130])
def test_ragged_env_counts_match_the_oracle(oracle_factory, n):
    """Env counts that fill no wavefront, no 4-env solver workgroup, no 16-env narrowphase group and no 64-env classification chunk evenly
    (n = 1 is BASELINE config 1's size): every kernel runs with idle lanes / partial groups, and the result is still the oracle's, in the
    eager and in the fused form (contact-pair ids included)."""
    gpu = PickCubeEnv(num_envs=n, device=DEV, fused=False)
    fus = PickCubeEnv(num_envs=n, device=DEV, fused=True)
    cpu = PickCubeEnv(num_envs=n, px_factory=oracle_factory)
    og, _ = gpu.reset(seed=11); of, _ = fus.reset(seed=11); oc, _ = cpu.reset(seed=11)
    assert torch.equal(og.cpu(), oc)
    gen = torch.Generator().manual_seed(n)
    for t in range(30):
        a = 2 * torch.rand(n, 8, generator=gen) - 1
        og, rg, tg, ug, _ = gpu.step(a.to(DEV))
        of, rf, *_ = fus.step(a.to(DEV))
        oc, rc, tc, uc, _ = cpu.step(a)
        assert _close(og.cpu().numpy(), oc.numpy()) and _close(rg.cpu().numpy(), rc.numpy()), (n, t)
        assert _close(of.cpu().numpy(), oc.numpy()) and _close(rf.cpu().numpy(), rc.numpy()), (n, t, "fused")
        assert torch.equal(tg.cpu(), tc) and torch.equal(ug.cpu(), uc)
    for e in sorted({0, n // 2, n - 1}):
        gi, gv = gpu.px.get_contacts(e)
        ci, cv = cpu.px.get_contacts(e)
        assert gi.shape == ci.shape and (gi == ci).all() and _close(gv, cv), (n, e)
    assert gpu.px.get_overflow() & 6 == 0 and fus.px.get_overflow() & 6 == 0


@pytest.mark.parametrize("parts", [2, 4])
def test_env_partitions_of_the_substep_change_no_bit(parts):
    """msk_step_n runs contiguous env partitions as independent kernel chains on streams of their own (include/msk_physx.h): envs never
    interact, so the rollout is the same, bit for bit, whatever the partition count -- here against the unpartitioned run, contact-rich."""
    n = 512
    a, b = PickCubeEnv(num_envs=n, device=DEV), PickCubeEnv(num_envs=n, device=DEV)
    assert a.px.set_step_parts(1) == 1 and b.px.set_step_parts(parts) == parts and b.px.step_parts == parts
    assert b.px.set_step_parts(3) == 2 and b.px.set_step_parts(parts) == parts     # 512 envs are 8 chunks: 3 partitions are lowered to 2
    a.reset(seed=11); b.reset(seed=11)
    gen = torch.Generator().manual_seed(4)
    for t in range(60):
        act = (2 * torch.rand(n, 8, generator=gen) - 1).to(DEV)
        oa, ra, *_ = a.step(act)
        ob, rb, *_ = b.step(act)
        assert torch.equal(oa, ob) and torch.equal(ra, rb), t
    assert torch.equal(a.get_state(), b.get_state())
    assert a.px.get_solver_class_counts().sum() == n == b.px.get_solver_class_counts().sum()
    b.px.timing_enable(5)
    b.step(act)
    t = b.px.timing_read()
    assert all(v[1] == 5 * parts for v in t.values())      # every partition's launch is timed
