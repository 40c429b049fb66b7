"""Pose / Actor / Link / Articulation views (maniskill_amd/structs.py; reference: mani_skill/utils/structs/*): reads equal the
envs' own gathers, writes land in the buffers and take effect after the apply, contact forces and states follow the reference's
shapes."""
import numpy as np
import pytest
import torch

from maniskill_amd.envs.pick_cube import PickCubeEnv
from maniskill_amd.envs.push_t import PushTEnv
from maniskill_amd.structs import Pose


def test_pose_algebra():
    gen = torch.Generator().manual_seed(0)
    q = torch.randn(5, 4, generator=gen); q = q / q.norm(dim=1, keepdim=True)
    a = Pose.create_from_pq(torch.randn(5, 3, generator=gen), q)
    q2 = torch.randn(5, 4, generator=gen); q2 = q2 / q2.norm(dim=1, keepdim=True)
    b = Pose.create_from_pq(torch.randn(5, 3, generator=gen), q2)
    ident = (a * a.inv()).raw_pose
    assert torch.allclose(ident[:, :3], torch.zeros(5, 3), atol=1e-6) and torch.allclose(ident[:, 3].abs(), torch.ones(5), atol=1e-6)
    # composition agrees with the homogeneous matrices
    assert torch.allclose((a * b).to_transformation_matrix(), a.to_transformation_matrix() @ b.to_transformation_matrix(), atol=1e-5)
    assert torch.allclose(a.inv().to_transformation_matrix(), torch.linalg.inv(a.to_transformation_matrix()), atol=1e-5)
    one = Pose.create_from_pq(p=[0.1, 0.2, 0.3])
    assert one.raw_pose.shape == (1, 7) and one.q.tolist() == [[1.0, 0.0, 0.0, 0.0]] and len(a[1:3]) == 2


def test_views_read_what_the_env_reads(oracle_factory):
    env = PickCubeEnv(num_envs=3, px_factory=oracle_factory)
    env.reset(seed=0)
    for _ in range(3):
        env.step(torch.full((3, 8), 0.3))
    sc = env.scene
    assert set(sc.actors) == {"table-workspace", "cube", "goal_site"} and list(sc.articulations) == ["panda"]
    cube, robot = sc.actors["cube"], env.robot
    assert torch.equal(cube.pose.raw_pose, env.cube_pose) and torch.equal(robot.links_map["panda_hand_tcp"].pose.raw_pose, env.tcp_pose)
    assert torch.equal(robot.qpos, env.qpos) and torch.equal(robot.qvel, env.qvel) and robot.dof_count == 9 and robot.max_dof == 9
    assert robot.get_active_joints()[:2] == ["panda_joint1", "panda_joint2"] and robot.get_active_joints()[-1] == "panda_finger_joint2"
    assert robot.qlimits.shape == (3, 9, 2) and torch.allclose(robot.qlimits[0, 7], torch.tensor([0.0, 0.04]))
    assert torch.equal(robot.drive_targets, env._target_qpos) and (robot.drive_targets[:, 0] - robot.qpos[:, 0]).abs().max() > 1e-3
    assert cube.px_body_type == "dynamic" and sc.actors["goal_site"].px_body_type == "kinematic"
    assert cube.per_scene_id.tolist() == [env._b_cube + 1] * 3 and abs(cube.mass[0].item() - 0.064) < 1e-6
    # state dict of the scene == the env's
    sd, ed = sc.get_sim_state(), env.get_state_dict()
    assert all(torch.equal(sd["actors"][k], ed["actors"][k]) for k in ed["actors"]) and torch.equal(sd["articulations"]["panda"], ed["articulations"]["panda"])
    assert [l.index for l in robot.links] == list(range(15)) and robot.find_link_by_name("panda_link3").index == 3


def test_writes_land_after_the_apply(oracle_factory):
    env = PickCubeEnv(num_envs=4, px_factory=oracle_factory)
    env.reset(seed=1)
    cube, robot, px = env.scene.actors["cube"], env.robot, env.px
    new = Pose.create_from_pq(torch.tensor([[0.3, 0.35, 0.3]]).repeat(2, 1))   # clear of the arm
    cube.set_pose(new, env_idx=[1, 3])
    cube.set_linear_velocity(torch.tensor([0.0, 0.0, 1.0]), env_idx=[1, 3])
    px.gpu_apply_rigid_dynamic_data(); px.gpu_fetch_all()
    p = cube.pose.p
    assert torch.allclose(p[[1, 3]], new.p, atol=1e-6) and (p[[0, 2], 2] < 0.03).all()          # the others stayed on the table
    px.step(); px.gpu_fetch_all()
    assert torch.allclose(cube.linear_velocity[[1, 3], 2], torch.full((2,), 1.0 - 9.81 * px.timestep), atol=1e-5)
    # joint state and drive targets
    q = robot.qpos
    q[:, 0] += 0.2
    robot.set_qpos(q); robot.set_qvel(torch.zeros_like(q))
    robot.set_joint_drive_targets(q[:, :2], joint_indices=[0, 1])
    px.gpu_apply_all(); px.gpu_update_articulation_kinematics(); px.gpu_fetch_all()
    assert torch.allclose(robot.qpos, q) and torch.allclose(robot.drive_targets[:, :2], q[:, :2])
    # a state round trip through Actor / Articulation.set_state
    s_c, s_r = cube.get_state(), robot.get_state()
    assert s_c.shape == (4, 13) and s_r.shape == (4, 13 + 18)
    for _ in range(3):
        px.step()
    cube.set_state(s_c); robot.set_state(s_r)
    px.gpu_apply_all(); px.gpu_update_articulation_kinematics(); px.gpu_fetch_all()
    assert torch.allclose(cube.get_state(), s_c, atol=1e-6) and torch.allclose(robot.get_state(), s_r, atol=1e-6)


def test_contact_forces_and_apply_force(oracle_factory):
    env = PickCubeEnv(num_envs=2, px_factory=oracle_factory)
    env.reset(seed=0)
    for _ in range(10):
        env.step(None)
    cube = env.scene.actors["cube"]
    f = cube.get_net_contact_forces()
    assert f.shape == (2, 3) and torch.allclose(f[:, 2], torch.full((2,), 0.064 * 9.81), rtol=2e-2) and cube.is_static().all()
    lf = env.robot.get_net_contact_forces(["panda_leftfinger", "panda_rightfinger"])
    assert lf.shape == (2, 2, 3) and lf.abs().max() < 1e-6                                    # the open gripper touches nothing
    assert env.robot.get_link_incoming_joint_forces().shape == (2, 15, 6)
    cube.apply_force(torch.tensor([0.0, 0.0, 0.064 * 9.81 * 3]))                              # 3 g upwards for one substep
    env.px.step(); env.px.gpu_fetch_all()
    assert torch.allclose(cube.linear_velocity[:, 2], torch.full((2,), 2 * 9.81 * env.px.timestep), rtol=1e-3)


def test_pusht_scene_view(oracle_factory):
    env = PushTEnv(num_envs=2, px_factory=oracle_factory)
    env.reset(seed=0)
    assert set(env.scene.actors) == {"table-workspace", "Tee", "goal_Tee", "goal_ee"} and env.robot.name == "panda_stick" and env.robot.dof_count == 7
    assert torch.equal(env.scene.actors["Tee"].pose.raw_pose, env._pose(env._b_tee))
