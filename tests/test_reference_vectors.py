"""This package's host code against vectors computed by the reference's own code (tests/golden/reference_vectors.npz, written by
tests/golden/make_reference_vectors.py where /root/reference is importable): rotation conversions, pose algebra, controller action
scaling, flatten order, Panda.is_grasping / is_static and, per task, evaluate / observation / normalized dense reward on simulator
states produced by this package.  The physics itself stays unpinned against PhysX; everything around it is pinned
here against the reference's arithmetic."""
import os

import numpy as np
import pytest
import torch

from maniskill_amd.envs.pick_cube import PickCubeEnv
from maniskill_amd.structs import Pose
from maniskill_amd.envs import registered as _registry

HERE = os.path.dirname(os.path.abspath(__file__))
G = np.load(os.path.join(HERE, "golden", "reference_vectors.npz"))
T = lambda k: torch.from_numpy(G[k])   # noqa: E731


def test_rotation_conversions():
    """utils/geometry/rotation_conversions.py: euler XYZ <-> quaternion (sign-free comparison), multiply, apply."""
    q = PickCubeEnv._euler_xyz_to_quat(T("rot/euler"))
    want = T("rot/quat_from_euler")
    assert torch.minimum((q - want).abs().max(dim=1)[0], (q + want).abs().max(dim=1)[0]).max() < 1e-6
    e = PickCubeEnv._quat_to_euler_xyz(T("rot/quat"))
    assert torch.allclose(e, T("rot/euler_from_quat"), atol=2e-5)
    assert torch.allclose(PickCubeEnv._qmul(T("rot/quat"), T("rot/quat2")), T("rot/quat_mul"), atol=1e-6)
    assert torch.allclose(PickCubeEnv._qrot(T("rot/quat"), T("rot/vec")), T("rot/quat_apply"), atol=1e-5)


def test_pose_algebra():
    a, b = Pose(T("pose/a")), Pose(T("pose/b"))
    assert torch.allclose((a * b).raw_pose, T("pose/a_mul_b"), atol=1e-5)
    assert torch.allclose(a.inv().raw_pose, T("pose/a_inv"), atol=1e-5)
    assert torch.allclose(a.to_transformation_matrix(), T("pose/a_matrix"), atol=1e-6)


def test_controller_action_scaling(oracle_factory):
    """gym_utils.clip_and_scale_action through PDJointPos(use_delta) + PDJointPosMimic, and PDEEPoseController's rotation clip."""
    act = T("clip_scale/action")
    env = PickCubeEnv(num_envs=len(act), px_factory=oracle_factory)
    env.reset(seed=0)
    q0 = env.qpos.clone()
    env._set_action(act)
    assert torch.allclose(env._target_qpos[:, :7] - q0[:, :7], T("clip_scale/arm"), atol=1e-6)
    assert torch.allclose(env._target_qpos[:, 7:8], T("clip_scale/gripper"), atol=1e-7) and torch.equal(env._target_qpos[:, 7], env._target_qpos[:, 8])
    ee = PickCubeEnv(num_envs=len(act), px_factory=oracle_factory, control_mode="pd_ee_delta_pose")
    a7 = torch.cat([T("ee_clip_scale/action"), torch.zeros(len(act), 1)], dim=1)
    assert torch.allclose(ee._ee_delta(a7), T("ee_clip_scale/out"), atol=1e-6)


def test_flatten_order():
    """common.flatten_state_dict: agent (qpos, qvel) then extra in insertion order, bools as floats -- PickCube's obs layout."""
    mine = torch.hstack([T("flatten/qpos"), T("flatten/qvel"), T("flatten/is_grasped")[:, None].float(), T("flatten/tcp_pose"), T("flatten/goal_pos")])
    assert torch.equal(mine, T("flatten/out"))


TASKS = ["PickCube-v1", "PushCube-v1", "PullCube-v1", "LiftPegUpright-v1", "StackCube-v1", "PokeCube-v1", "PegInsertionSide-v1", "StackPyramid-v1", "PullCubeTool-v1"]


@pytest.mark.parametrize("name", TASKS)
def test_task_logic_matches_the_reference(oracle_factory, name):
    n = int(G[f"{name}/num_envs"])
    env = _registry()[name](num_envs=n, px_factory=oracle_factory)
    env.reset(seed=0)
    states, lf, rf = T(f"{name}/state"), T(f"{name}/lforce"), T(f"{name}/rforce")
    seen_success = seen_grasp = 0
    for k in range(len(states)):
        env.set_state(states[k])
        forces = {id(env._q_lgrasp): lf[k], id(env._q_rgrasp): rf[k]}
        zero = torch.zeros(n, 3)
        env.get_pairwise_contact_forces = lambda q: forces.get(id(q), zero)   # the recorded finger forces instead of the simulator's
        grasp = env.is_grasping()
        assert torch.equal(grasp, T(f"{name}/grasp")[k])                   # Panda.is_grasping (panda.py:237-265)
        info = env.get_info()
        obs = env.get_obs(info)
        rew = env.get_reward(obs, None, info)
        assert torch.equal(info["success"], T(f"{name}/success")[k]), k
        assert torch.allclose(obs, T(f"{name}/obs")[k], atol=1e-6), k
        assert torch.allclose(rew, T(f"{name}/reward")[k], atol=2e-6), k
        seen_success += int(info["success"].sum()); seen_grasp += int(grasp.sum())
    assert seen_success > 0 and (seen_grasp > 0 or name == "StackPyramid-v1")   # the fixture exercises the success and grasp branches


def test_pusht_matches_the_reference(oracle_factory):
    """PushT-v1 (push_t.py:256-540): the pseudo-render tables, the intersection ratio of every state, evaluate, observation and the
    pose-based reward, all computed by the reference's code on this package's states."""
    from maniskill_amd.envs.push_t import PushTEnv

    n = int(G["PushT-v1/num_envs"])
    env = PushTEnv(num_envs=n, px_factory=oracle_factory)
    env.reset(seed=0)
    assert torch.equal(env.tee_render.float(), T("PushT-tables/tee_render")) and torch.allclose(env.world_to_goal_trans, T("PushT-tables/world_to_goal_trans"), atol=1e-6)
    states = T("PushT-v1/state")
    for k in range(len(states)):
        env.set_state(states[k])
        assert torch.allclose(env.pseudo_render_intersection(), T("PushT-v1/intersection")[k], atol=1e-6), k
        info = env.get_info()
        assert torch.equal(info["success"], T("PushT-v1/success")[k])
        assert torch.allclose(env.get_obs(info), T("PushT-v1/obs")[k], atol=1e-6)
        assert torch.allclose(env.compute_normalized_dense_reward(info), T("PushT-v1/reward")[k], atol=2e-6)
    assert T("PushT-v1/success").any() and not T("PushT-v1/success").all()


def test_camera_matrices(oracle_factory):
    """RenderCamera.get_extrinsic_matrix (OpenCV, ros2opencv @ inv(pose)) and get_model_matrix (pose * POSE_GL_TO_ROS) of the
    reference for random camera poses (structs/render_camera.py:77-145)."""
    poses = T("camera/pose")
    env = PickCubeEnv(num_envs=len(poses), px_factory=oracle_factory, obs_mode="depth+segmentation")
    cam = env.camera
    cam.get_global_pose = lambda: poses
    cam._cached_extrinsic = cam._cached_model = None
    assert torch.allclose(cam.get_extrinsic_matrix(), T("camera/extrinsic"), atol=2e-6)
    assert torch.allclose(cam.get_model_matrix(), T("camera/model"), atol=2e-6)


@pytest.mark.parametrize("mode", ["pd_ee_delta_pos", "pd_ee_delta_pose", "pd_ee_target_delta_pos", "pd_ee_target_delta_pose", "pd_ee_pose"])
def test_ee_controllers_match_the_reference(oracle_factory, mode):
    """PDEEPos / PDEEPoseController.set_action + Kinematics.compute_ik (GPU branch) of the reference, fed with this package's
    Jacobian and link poses, against this package's controller: the arm joint targets of four consecutive control steps (the
    virtual-target modes carry their target pose from step to step)."""
    acts, states, want = T(f"ee/{mode}/action"), T(f"ee/{mode}/state"), T(f"ee/{mode}/target")
    env = PickCubeEnv(num_envs=acts.shape[1], px_factory=oracle_factory, control_mode=mode)
    env.reset(seed=4)
    for k in range(len(acts)):
        assert torch.equal(env.get_state(), states[k])                       # the same rollout as when the vectors were made
        env.step(acts[k])
        # the LM system (7 joints, 6 task dimensions, lambda 1e-4) is ill-conditioned, so fp32 rounding of the Euler-angle round
        # trips shows up at the 1e-4 level in the joint targets (measured: <= 9.5e-5 rad; 0 for the position-only delta mode)
        got, ref = env._target_qpos[:, :7], want[k]
        assert torch.allclose(got, ref, atol=3e-4), (k, (got - ref).abs().max())


@pytest.mark.parametrize("mode", ["pd_joint_delta_pos", "pd_joint_target_delta_pos", "pd_joint_pos", "pd_joint_vel", "pd_joint_pos_vel", "pd_joint_delta_pos_vel"])
def test_joint_controllers_match_the_reference(oracle_factory, mode):
    """PDJointPos / PDJointPosMimic / PDJointVel / PDJointPosVelController.set_action of the reference over three consecutive control
    steps: position targets of the arm and of the mimic gripper, velocity targets."""
    acts, qt, vt = T(f"joint/{mode}/action"), T(f"joint/{mode}/qpos_target"), T(f"joint/{mode}/qvel_target")
    env = PickCubeEnv(num_envs=acts.shape[1], px_factory=oracle_factory, control_mode=mode)
    env.reset(seed=6)
    for k in range(len(acts)):
        env.step(acts[k])
        if not torch.isnan(qt[k][:, 0]).any():          # the velocity controller sets no arm position target
            assert torch.allclose(env._target_qpos[:, :7], qt[k][:, :7], atol=1e-6), k
        assert torch.allclose(env._target_qpos[:, 7:9], qt[k][:, 7:9], atol=1e-7)
        if mode in ("pd_joint_vel", "pd_joint_pos_vel", "pd_joint_delta_pos_vel"):
            assert torch.allclose(env._target_qvel_buf[:, :7], vt[k], atol=1e-6)


def test_shader_texture_transforms(oracle_factory):
    """The minimal pack's texture transforms (render/shaders.py:66-84) applied by the reference to this package's textures ==
    what Camera.get_obs hands out here."""
    env = PickCubeEnv(num_envs=2, px_factory=oracle_factory, obs_mode="rgb+depth+segmentation")
    env.reset(seed=0)
    env.camera.take_picture()
    assert torch.equal(env.camera.get_picture_cuda().torch(), T("shader/position_segmentation"))      # the same picture as when recorded
    out = env.camera.get_obs(depth=True, segmentation=True, position=True, rgb=True)
    for k in ("depth", "segmentation", "position", "rgb"):
        assert torch.equal(out[k], T(f"shader/{k}")), k


def test_constants_and_scene_initialisation():
    """Registered episode lengths, task constants, the robots' rest keyframes and base / table placement as the reference's Python
    states them (registration.py, pick_cube_cfgs.py, the task classes, TableSceneBuilder.initialize with its noise switched off)."""
    from maniskill_amd.envs import scene_builders as sb
    from maniskill_amd.envs.lift_peg_upright import LiftPegUprightEnv
    from maniskill_amd.envs.peg_insertion_side import PegInsertionSideEnv
    from maniskill_amd.envs.poke_cube import PokeCubeEnv
    from maniskill_amd.envs.pull_cube import PullCubeEnv
    from maniskill_amd.envs.push_cube import PushCubeEnv
    from maniskill_amd.envs.push_t import PushTEnv
    reg = _registry()
    for name, steps in zip(G["const/names"].tolist(), G["const/max_episode_steps"].tolist()):
        assert reg[name].max_episode_steps == steps, name
    P = PickCubeEnv
    mine = [P.cube_half_size, P.goal_thresh, P.cube_spawn_half_size, P.max_goal_height, *P.cube_spawn_center, *P.camera_eye, *P.camera_target]
    assert np.allclose(mine, G["const/pickcube/values"])
    assert PushCubeEnv.goal_radius == float(G["const/tasks/push_goal_radius"]) and PullCubeEnv.goal_radius == float(G["const/tasks/pull_goal_radius"])
    K = PokeCubeEnv
    assert np.allclose([K.cube_half_size, K.peg_half_width, K.peg_half_length, K.goal_radius], G["const/tasks/poke"])
    assert np.allclose([LiftPegUprightEnv.peg_half_width, LiftPegUprightEnv.peg_half_length], G["const/tasks/liftpeg"])
    S = PushTEnv
    assert np.allclose([S.intersection_thresh, S.goal_z_rot, *S.goal_offset, S.tee_spawnbox_xlength, S.tee_spawnbox_ylength, S.tee_spawnbox_xoffset,
                        S.tee_spawnbox_yoffset], G["const/tasks/pusht"], atol=1e-7)
    assert np.allclose(sb.PANDA_REST_QPOS, G["const/panda/rest_qpos"]) and np.allclose(PegInsertionSideEnv.rest_qpos, G["const/panda_wristcam/rest_qpos"])
    assert np.allclose(G["const/panda/root_p"], [-0.615, 0, 0]) and np.allclose(G["const/panda/table_p"], [-0.12, 0, -sb.TABLE_HEIGHT])


INIT = {"PickCube-v1": dict(cube="_b_cube", goal_site="_b_goal"), "PushCube-v1": dict(cube="_b_cube", goal_region="_b_goal"),
        "PullCube-v1": dict(cube="_b_cube", goal_region="_b_goal"), "StackCube-v1": dict(cubeA="_b_cube", cubeB="_b_goal"),
        "LiftPegUpright-v1": dict(peg="_b_cube"), "PokeCube-v1": dict(peg="_b_cube", cube="_b_poked", goal_region="_b_goal"),
        "PullCubeTool-v1": dict(l_shape_tool="_b_cube", cube="_b_pulled"), "StackPyramid-v1": dict(cubeA="_b_cube", cubeB="_b_cubeB", cubeC="_b_cubeC"),
        "PushT-v1": dict(Tee="_b_tee", goal_Tee="_b_goal", goal_ee="_b_ee"), "PegInsertionSide-v1": dict(peg="_b_cube", box_with_hole="_b_goal")}


@pytest.mark.parametrize("name", sorted(INIT))
def test_episode_layout_statistics(oracle_factory, name):
    """Support and mean of every placed actor's position and yaw over 1500 fresh episodes against the reference's own
    _initialize_episode (4000 sub-scenes on a recording fake): this package draws from its own per-env RNG streams, so the samples
    differ but the distributions must not."""
    n = 1500
    env = _registry()[name](num_envs=n, px_factory=oracle_factory)
    env.reset(seed=123)
    for actor, attr in INIT[name].items():
        raw = env._pose(getattr(env, attr))
        R = lambda k: torch.from_numpy(np.atleast_1d(G[f"init/{name}/{actor}/{k}"]))   # noqa: E731
        span = (R("pmax") - R("pmin")).clamp(min=1e-3)
        tol = 0.03 * span + 1e-4
        assert ((raw[:, :3].min(0)[0] - R("pmin")).abs() <= tol).all(), (actor, raw[:, :3].min(0)[0], R("pmin"))
        assert ((raw[:, :3].max(0)[0] - R("pmax")).abs() <= tol).all(), (actor, raw[:, :3].max(0)[0], R("pmax"))
        assert ((raw[:, :3].mean(0) - R("pmean")).abs() <= 0.05 * span + 1e-4).all(), (actor, raw[:, :3].mean(0), R("pmean"))
        yaw = torch.remainder(2 * torch.atan2(raw[:, 6], raw[:, 3]) + np.pi, 2 * np.pi) - np.pi
        ys = float(R("yaw_max") - R("yaw_min"))
        if ys > 1e-3 and float(R("qxy_absmax")) < 1e-6:          # a yaw-only randomisation in the reference
            assert abs(float(yaw.min() - R("yaw_min"))) <= 0.03 * ys + 1e-3 and abs(float(yaw.max() - R("yaw_max"))) <= 0.03 * ys + 1e-3, actor
            assert raw[:, 4:6].abs().max() < 1e-6


@pytest.mark.parametrize("mode", ["sparse", "dense", "normalized_dense", "none"])
def test_reward_modes(oracle_factory, mode):
    """BaseEnv.get_reward / compute_sparse_reward (sapien_env.py:648-697): the four reward modes, with and without a fail flag (the
    reference's sparse reward of a task without a fail flag is the bool success tensor itself; here it is its float value)."""
    env = PickCubeEnv(num_envs=8, px_factory=oracle_factory, reward_mode=mode)
    env.compute_dense_reward = lambda obs, action, info: T("reward_mode/dense").clone()
    for tag, info in (("s", dict(success=T("reward_mode/success"))), ("sf", dict(success=T("reward_mode/success"), fail=T("reward_mode/fail")))):
        got = env.get_reward(None, None, info)
        assert got.dtype == torch.float32 and torch.allclose(got, T(f"reward_mode/{mode}_{tag}/out"), atol=1e-7), (mode, tag)
    with pytest.raises(NotImplementedError):
        PickCubeEnv(num_envs=1, px_factory=oracle_factory, reward_mode="shaped")


def test_base_camera_configs():
    """Every task's base_camera as its _default_sensor_configs states it in the reference: look_at eye / target, 128 x 128, fov pi/2,
    near 0.01, far 100."""
    reg = _registry()
    for name, row in zip(G["sensor/names"].tolist(), G["sensor/rows"]):
        cls = reg[name]
        assert np.allclose(cls.camera_eye, row[0:3], atol=1e-6) and np.allclose(cls.camera_target, row[3:6], atol=1e-6), name
        assert tuple(row[6:8]) == (128, 128) and abs(row[8] - np.pi / 2) < 1e-6 and row[9] == 0.01 and row[10] == 100, name


@pytest.mark.skipif(not os.path.isdir("/root/reference/mani_skill"), reason="needs the reference tree (present in the build container only)")
def test_committed_vectors_are_what_the_generator_writes(tmp_path):
    """tests/golden/reference_vectors.npz is exactly what tests/golden/make_reference_vectors.py produces today: neither the fixture nor
    the states this package feeds the reference's code have drifted."""
    import subprocess
    import sys

    out = str(tmp_path / "fresh.npz")
    subprocess.check_call([sys.executable, os.path.join(HERE, "golden", "make_reference_vectors.py"), out], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                          timeout=600)
    fresh = np.load(out)
    assert sorted(fresh.files) == sorted(G.files)
    for k in G.files:
        a, b = G[k], fresh[k]
        assert a.shape == b.shape and a.dtype == b.dtype, k
        assert np.array_equal(a, b, equal_nan=True) if a.dtype.kind in "fc" else np.array_equal(a, b), k
