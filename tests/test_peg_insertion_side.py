"""PegInsertionSide-v1 (BASELINE.json config 4; mani_skill/envs/tasks/tabletop/peg_insertion_side.py): per-env peg / box sizes,
known answers on the CPU oracle; HIP parity under -m gpu."""
import numpy as np
import pytest
import torch

from maniskill_amd.envs.peg_insertion_side import PegInsertionSideEnv, _pose_inv, _pose_mul


def test_sizes_layout_and_observation(oracle_factory):
    env = PegInsertionSideEnv(num_envs=6, px_factory=oracle_factory)
    obs, info = env.reset(seed=0)
    assert obs.shape == (6, 43) and not info["success"].any()
    hs = env.peg_half_sizes
    assert ((hs[:, 0] >= 0.085) & (hs[:, 0] <= 0.125) & (hs[:, 1] >= 0.015) & (hs[:, 1] <= 0.025)).all() and hs[:, 0].std() > 1e-3
    assert torch.allclose(env.box_hole_radii, hs[:, 1] + 0.003)
    for _ in range(15):
        obs, r, term, trunc, info = env.step(None)
    peg, box = env.peg_pose, env.box_pose
    assert torch.allclose(peg[:, 2], hs[:, 1], atol=1.5e-3)                       # every peg rests on the table at its own radius
    assert torch.allclose(box[:, 2], hs[:, 0], atol=1e-5)                         # the kinematic box stands at its own half length
    assert torch.allclose(obs[:, 25:32], peg) and torch.allclose(obs[:, 32:35], hs) and torch.allclose(obs[:, 42], env.box_hole_radii)
    assert (r > 0).all() and (r < 0.2).all() and env.px.get_overflow() == 0


def test_a_peg_put_into_its_hole_is_a_success_and_stays_there(oracle_factory):
    env = PegInsertionSideEnv(num_envs=4, px_factory=oracle_factory)
    env.reset(seed=2)
    goal = env.goal_pose                                                            # peg pose with its head at the hole's centre
    env._rbd[:, env._b_cube, :3] = goal[:, :3] + env._offsets
    env._rbd[:, env._b_cube, 3:7] = goal[:, 3:7]
    env._rbd[:, env._b_cube, 7:13] = 0.0
    env.px.gpu_apply_all(); env.px.gpu_fetch_all()
    ok, p = env.has_peg_inserted()
    assert ok.all() and p.abs().max() < 1e-5
    for _ in range(10):
        obs, r, term, trunc, info = env.step(None)
    # the four slabs of this env's box hold this env's peg: it drops by the 3 mm clearance at most and stays inserted
    assert info["success"].all() and term.all() and torch.allclose(r, torch.ones(4))
    assert (info["peg_head_pos_at_hole"][:, 1:].abs() <= env.box_hole_radii[:, None] + 1e-4).all()
    assert env.px.get_overflow() == 0
    # pulled back by 3 cm along the hole: not inserted
    back = _pose_mul(env.peg_pose, torch.tensor([[-0.03, 0, 0, 1.0, 0, 0, 0]]).repeat(4, 1))
    env._rbd[:, env._b_cube, :3] = back[:, :3] + env._offsets
    env.px.gpu_apply_all(); env.px.gpu_fetch_all()
    assert not env.has_peg_inserted()[0].any()


def test_the_hand_camera_rides_on_camera_link_and_sees_the_table_where_its_ray_meets_it(oracle_factory):
    """panda_wristcam's hand_camera (agents/robots/panda/panda_wristcam.py:19-32: 128 x 128, fov pi / 2, identity pose on camera_link).
    Known answers: its model matrix is camera_link's pose (OpenGL axes: x = -left, y = up, z = -forward of the link's x-forward frame);
    where the centre pixel shows the table, its depth is the forward distance at which that pixel's ray meets the plane z = 0; after a
    few steps of arm motion the camera has moved with the link."""
    env = PegInsertionSideEnv(num_envs=3, px_factory=oracle_factory, obs_mode="depth+segmentation")
    obs, _ = env.reset(seed=0)
    assert list(obs["sensor_data"]) == ["base_camera", "hand_camera"] and list(obs["sensor_param"]) == ["base_camera", "hand_camera"]
    link = env.px.template.body_id("camera_link")
    table_id = env.px.template.body_id("table-workspace") + 1

    def check(obs):
        pose = env._pose(link)
        M = obs["sensor_param"]["hand_camera"]["cam2world_gl"]
        R = torch.stack([env._qrot(pose[:, 3:7], torch.tensor(v).expand(3, 3)) for v in ([1.0, 0, 0], [0.0, 1, 0], [0.0, 0, 1])], dim=-1)  # columns: forward, left, up
        assert torch.allclose(M[:, :3, 3], pose[:, :3], atol=1e-6)
        assert torch.allclose(M[:, :3, 0], -R[:, :, 1], atol=1e-5) and torch.allclose(M[:, :3, 1], R[:, :, 2], atol=1e-5) and torch.allclose(M[:, :3, 2], -R[:, :, 0], atol=1e-5)
        depth, seg = obs["sensor_data"]["hand_camera"]["depth"], obs["sensor_data"]["hand_camera"]["segmentation"]
        seen = 0
        for e in range(3):
            for (i, j) in ((64, 64), (20, 30), (100, 90)):
                if seg[e, i, j, 0].item() != table_id:
                    continue
                # pixel (row i, column j), centre at (j + 0.5, i + 0.5); focal length 64 pixels at fov pi / 2: ray = forward - left (u - 64) / 64 - up (v - 64) / 64
                d = R[e, :, 0] - R[e, :, 1] * ((j + 0.5 - 64) / 64) - R[e, :, 2] * ((i + 0.5 - 64) / 64)
                t = -pose[e, 2] / d[2]                      # the table top is the plane z = 0 of the sub-scene
                assert abs(depth[e, i, j, 0].item() - 1000.0 * t.item()) <= 1.0, (e, i, j, depth[e, i, j, 0].item(), 1000.0 * t.item())
                seen += 1
        assert seen >= 3
        return pose.clone()

    p0 = check(obs)
    a = torch.zeros(3, 8); a[:, 1] = -1.0; a[:, 3] = 1.0
    for _ in range(5):
        obs = env.step(a)[0]
    p1 = check(obs)
    assert (p1[:, :3] - p0[:, :3]).norm(dim=1).min().item() > 0.01


@pytest.mark.gpu
def test_hip_matches_oracle_rollout(oracle_factory):
    n = 64
    gpu = PegInsertionSideEnv(num_envs=n, device="cuda:0", obs_mode="rgb+depth+segmentation")
    cpu = PegInsertionSideEnv(num_envs=n, px_factory=oracle_factory, obs_mode="rgb+depth+segmentation")
    og, _ = gpu.reset(seed=2022); oc, _ = cpu.reset(seed=2022)
    # the GPU env runs the fused task kernel (env-frame poses), the oracle env the torch task code (world poses minus the scene
    # offset): the same physics bits, observations within fp32 rounding of that round trip
    assert torch.allclose(og["state"].cpu(), oc["state"], atol=2e-6)
    gen = torch.Generator().manual_seed(0)
    for t in range(30):
        a = 2 * torch.rand(n, 8, generator=gen) - 1
        og, rg, tg, ug, _ = gpu.step(a.to("cuda:0"))
        oc, rc, tc, uc, _ = cpu.step(a)
        assert np.allclose(og["state"].cpu().numpy(), oc["state"].numpy(), rtol=1e-4, atol=1e-5), t
        assert np.allclose(rg.cpu().numpy(), rc.numpy(), atol=2e-5) and torch.equal(tg.cpu(), tc)
    assert list(og["sensor_data"]) == ["base_camera", "hand_camera"]
    for uid in ("base_camera", "hand_camera"):          # the fixed camera and the one riding on camera_link, after 30 random steps
        cg, cc = og["sensor_data"][uid], oc["sensor_data"][uid]
        assert torch.equal(cg["rgb"].cpu(), cc["rgb"]) and torch.equal(cg["depth"].cpu(), cc["depth"]), uid
        assert torch.equal(cg["segmentation"].cpu(), cc["segmentation"]), uid
        for k in ("extrinsic_cv", "cam2world_gl", "intrinsic_cv"):
            assert torch.allclose(og["sensor_param"][uid][k].cpu(), oc["sensor_param"][uid][k], rtol=1e-4, atol=1e-5), (uid, k)
    # inserted pegs stay inserted on the GPU too
    for env in (gpu, cpu):
        goal = env.goal_pose
        env._rbd[:, env._b_cube, :3] = goal[:, :3] + env._offsets
        env._rbd[:, env._b_cube, 3:7] = goal[:, 3:7]
        env._rbd[:, env._b_cube, 7:13] = 0.0
        env.px.gpu_apply_all(); env.px.gpu_fetch_all()
    for _ in range(5):
        ig = gpu.step(None)[4]; ic = cpu.step(None)[4]
    # (wherever the random rollout left an arm in the way of the teleported peg it is pushed back out: the same envs on both sides)
    assert torch.equal(ig["success"].cpu(), ic["success"]) and ig["success"].float().mean().item() > 0.9


@pytest.mark.gpu
def test_fused_task_kernel_matches_torch_task_code():
    """msk_task_peg_observe (include/msk_task.h) against the torch statement of evaluate / obs / reward on the same HIP
    simulation: same physics bits, task arithmetic within fp32 rounding (the fused kernel works in the env frame, the torch code
    on world poses minus the scene offset)."""
    n = 128
    fused = PegInsertionSideEnv(num_envs=n, device="cuda:0")
    plain = PegInsertionSideEnv(num_envs=n, device="cuda:0", fused=False)
    assert fused.fused and not plain.fused
    of, _ = fused.reset(seed=7); op, _ = plain.reset(seed=7)
    assert torch.allclose(of, op, atol=2e-6)
    gen = torch.Generator().manual_seed(2)
    for t in range(25):
        a = (0.8 * (2 * torch.rand(n, 8, generator=gen) - 1)).to("cuda:0")
        of, rf, tf, uf, inf_ = fused.step(a)
        op, rp, tp, up, inp = plain.step(a)
        assert torch.allclose(of, op, rtol=1e-5, atol=3e-6), t
        assert torch.allclose(rf, rp, atol=2e-5) and torch.equal(tf, tp) and torch.equal(uf, up)
        assert torch.allclose(inf_["peg_head_pos_at_hole"], inp["peg_head_pos_at_hole"], atol=3e-6)
    assert torch.equal(fused.get_state(), plain.get_state())           # the simulations themselves are bit-identical
    # success, reward 1 and termination for inserted pegs through the fused kernel as well
    for env in (fused, plain):
        goal = env.goal_pose
        env._rbd[:, env._b_cube, :3] = goal[:, :3] + env._offsets
        env._rbd[:, env._b_cube, 3:7] = goal[:, 3:7]
        env._rbd[:, env._b_cube, 7:13] = 0.0
        env.px.gpu_apply_all(); env.px.gpu_fetch_all()
    for _ in range(3):
        sf, sp = fused.step(None), plain.step(None)
    ok = sf[4]["success"]                       # (an arm left in the way by the random rollout may knock a peg out again)
    assert ok.float().mean() > 0.9 and torch.equal(ok, sp[4]["success"]) and torch.equal(sf[2], sp[2])
    assert torch.allclose(sf[1][ok], torch.ones(int(ok.sum()), device="cuda:0")) and torch.allclose(sf[1], sp[1], atol=2e-5)


def test_long_random_rollout_stays_finite(oracle_factory):
    """Regression: at control step 84 of this rollout env 16's arm hits the box so that EPA ends on a polytope of slivers and used to hand
    back a null normal; the contact row without a direction (J = 0) turned the env into NaNs.  Such polytopes now take the degenerate
    fallback (oracle/orc_collide.c and msk_collide.h gjk_epa)."""
    env = PegInsertionSideEnv(num_envs=64, px_factory=oracle_factory)
    env.reset(seed=2022)
    gen = torch.Generator().manual_seed(0)
    for k in range(100):
        obs, rew, term, trunc, info = env.step(2 * torch.rand(64, 8, generator=gen) - 1)
        assert torch.isfinite(obs).all() and torch.isfinite(rew).all(), k
    assert torch.isfinite(env.get_state()).all() and env.qvel.abs().max() < 20.0


@pytest.mark.gpu
def test_epa_null_normal_reproducer_hip_vs_oracle(oracle_factory):
    """the sliver-polytope rollout (seed 2022, 64 envs, 100 control steps; the old failure sat at step 84) on HIP
    against the oracle -- finiteness on BOTH sides first (a shared NaN compares equal to nothing and used to go unnoticed), then
    the states."""
    n = 64
    gpu = PegInsertionSideEnv(num_envs=n, device="cuda:0", fused=False)
    cpu = PegInsertionSideEnv(num_envs=n, px_factory=oracle_factory)
    gpu.reset(seed=2022); cpu.reset(seed=2022)
    gen = torch.Generator().manual_seed(0)
    for k in range(100):
        a = 2 * torch.rand(n, 8, generator=gen) - 1
        og, rg, *_ = gpu.step(a.to("cuda:0"))
        oc, rc, *_ = cpu.step(a)
        assert torch.isfinite(og).all() and torch.isfinite(rg).all(), f"HIP not finite at step {k}"
        assert torch.isfinite(oc).all() and torch.isfinite(rc).all(), f"oracle not finite at step {k}"
        if k % 10 == 9 or 80 <= k <= 90:
            sg, sc = gpu.get_state().cpu(), cpu.get_state()
            assert torch.isfinite(sg).all() and torch.isfinite(sc).all()
            assert torch.allclose(sg, sc, rtol=1e-4, atol=1e-5), (k, float((sg - sc).abs().max()))


@pytest.mark.gpu
def test_hip_vs_oracle_at_config4_scale(oracle_factory):
    """BASELINE config 4's per-GPU share: 2048 PegInsertionSide-v1 envs (per-env peg and hole sizes) x 100 control steps of random actions
    on HIP; the oracle runs the first 256 of them (same global seeds, sizes and grid cells).  States within 1e-4 relative, contact counts
    and contact-pair ids equal, no solver scheduling flags (contact overflow -- arms ploughing through peg and box -- is part of the
    contract and happens on both sides alike)."""
    n, m, steps = 2048, 256, 100
    gpu = PegInsertionSideEnv(num_envs=n, device="cuda:0", fused=False)
    cpu = PegInsertionSideEnv(num_envs=m, px_factory=oracle_factory, env_index_offset=0, total_envs=n)
    og, _ = gpu.reset(seed=2022); oc, _ = cpu.reset(seed=2022)
    assert torch.allclose(og[:m].cpu(), oc, atol=2e-6)
    gen = torch.Generator().manual_seed(4)
    for t in range(steps):
        a = 2 * torch.rand(n, 8, generator=gen) - 1
        og, rg, *_ = gpu.step(a.to("cuda:0"))
        oc, rc, *_ = cpu.step(a[:m])
        assert torch.isfinite(og).all() and torch.isfinite(oc).all(), t
        if t % 10 == 9:
            sg, sc = gpu.get_state()[:m].cpu(), cpu.get_state()
            assert torch.allclose(sg, sc, rtol=1e-4, atol=1e-5), (t, float((sg - sc).abs().max()))
            assert np.array_equal(gpu.px.get_env_contact_counts()[:m], cpu.px.get_env_contact_counts()), t
            for e in (0, 100, 255):
                assert np.array_equal(gpu.px.get_contacts(e)[0], cpu.px.get_contacts(e)[0]), (t, e)
    assert gpu.px.get_overflow() & 6 == 0


@pytest.mark.gpu
@pytest.mark.parametrize("n", [1, 5, 67])
def test_ragged_env_counts_with_per_env_sizes_match_the_oracle(oracle_factory, n):
    """Partial wavefronts / groups with per-env peg and hole sizes (the env records carry them): HIP == oracle for env counts that fill
    nothing evenly."""
    gpu = PegInsertionSideEnv(num_envs=n, device="cuda:0", fused=False)
    cpu = PegInsertionSideEnv(num_envs=n, px_factory=oracle_factory)
    og, _ = gpu.reset(seed=3); oc, _ = cpu.reset(seed=3)
    assert torch.allclose(og.cpu(), oc, atol=2e-6)
    gen = torch.Generator().manual_seed(10 + n)
    for t in range(25):
        a = 2 * torch.rand(n, 8, generator=gen) - 1
        og, rg, *_ = gpu.step(a.to("cuda:0")); oc, rc, *_ = cpu.step(a)
        assert torch.isfinite(og).all() and torch.isfinite(oc).all(), t
        assert torch.allclose(og.cpu(), oc, rtol=1e-4, atol=1e-5), (n, t, float((og.cpu() - oc).abs().max()))
    assert torch.allclose(gpu.get_state().cpu(), cpu.get_state(), rtol=1e-4, atol=1e-5)
    assert gpu.px.get_overflow() & 6 == 0
