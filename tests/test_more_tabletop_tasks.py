"""PullCube-v1, LiftPegUpright-v1, PokeCube-v1 (mani_skill/envs/tasks/tabletop/{pull_cube,lift_peg_upright,poke_cube}.py): reset
layouts, observation slices and known answers of evaluate / reward on the CPU oracle; HIP parity of rollouts under -m gpu."""
import numpy as np
import pytest
import torch

from maniskill_amd.envs.lift_peg_upright import LiftPegUprightEnv
from maniskill_amd.envs.poke_cube import PokeCubeEnv
from maniskill_amd.envs.pull_cube import PullCubeEnv
import maniskill_amd


def _teleport(env, body, p=None, q=None):
    if p is not None:
        env._rbd[:, body, :3] = p + env._offsets
    if q is not None:
        env._rbd[:, body, 3:7] = q
    env._rbd[:, body, 7:13] = 0.0
    env.px.gpu_apply_all(); env.px.gpu_fetch_all()


def test_pull_cube(oracle_factory):
    env = PullCubeEnv(num_envs=4, px_factory=oracle_factory)
    obs, info = env.reset(seed=0)
    assert obs.shape == (4, 35) and not info["success"].any()
    cube, goal = env.cube_pose, env.goal_pos
    assert (cube[:, :2].abs() <= 0.1 + 1e-6).all()
    assert torch.allclose(cube[:, 0] - goal[:, 0], torch.full((4,), 0.2), atol=1e-6) and torch.allclose(goal[:, 1], cube[:, 1], atol=1e-6)   # behind the cube
    assert torch.allclose(obs[:, 18:25], env.tcp_pose) and torch.allclose(obs[:, 25:28], goal) and torch.allclose(obs[:, 28:35], cube)
    r0 = env.step(None)[1]
    assert (r0 > 0).all() and (r0 < 1 / 3).all()                       # reaching stage only
    # the reward's pull position is on the far side of the cube: a tcp there earns the full reaching stage + the place stage
    d_goal = torch.linalg.norm(cube[:, :2] - goal[:, :2], dim=1)
    _teleport(env, env._b_cube, p=torch.cat([goal[:, :2] + torch.tensor([0.05, 0.0]), cube[:, 2:3]], dim=1))
    obs, rew, term, trunc, info = env.step(None)
    assert info["success"].all() and term.all() and torch.allclose(rew, torch.ones(4)) and (d_goal > 0.19).all()


def test_lift_peg_upright(oracle_factory):
    env = LiftPegUprightEnv(num_envs=3, px_factory=oracle_factory)
    obs, info = env.reset(seed=0)
    assert obs.shape == (3, 32) and not info["success"].any()
    for _ in range(5):
        obs, rew, term, trunc, info = env.step(None)
    peg = env.peg_pose
    assert torch.allclose(peg[:, 2], torch.full((3,), 0.025), atol=1.5e-3) and (peg[:, :2].abs() <= 0.1 + 1e-3).all()   # lies flat, rolled by pi/2
    assert torch.allclose(peg[:, 3:7].abs(), torch.tensor([[np.cos(np.pi / 4), np.sin(np.pi / 4), 0, 0]], dtype=torch.float32).repeat(3, 1), atol=5e-3)
    assert torch.allclose(obs[:, 25:32], peg) and not info["success"].any()
    # lying: axis term ~ 0, height term 1 - tanh(5 * 0.095), reaching / 5
    assert (rew * 3 < 1 - np.tanh(5 * 0.095) + 0.2 + 0.02).all() and (rew > 0).all()
    # tilted up about the world y axis (keeping its roll): stands on its end, stays there, pays the maximum
    qy = torch.tensor([[np.cos(-np.pi / 4), 0.0, np.sin(-np.pi / 4), 0.0]], dtype=torch.float32).repeat(3, 1)
    qx = torch.tensor([[np.cos(np.pi / 4), np.sin(np.pi / 4), 0.0, 0.0]], dtype=torch.float32).repeat(3, 1)
    up = torch.cat([peg[:, :2] + torch.tensor([0.15, 0.0]), torch.full((3, 1), 0.12)], dim=1)      # away from the gripper
    _teleport(env, env._b_cube, p=up, q=env._qmul(qy, qx))
    for _ in range(10):
        obs, rew, term, trunc, info = env.step(None)
    assert info["success"].all() and term.all() and torch.allclose(rew, torch.ones(3))
    assert torch.allclose(env.peg_pose[:, 2], torch.full((3,), 0.12), atol=2e-3)
    assert env.get_state().shape == (3, 13 * 2 + 13 + 18) and list(env.get_state_dict()["actors"]) == ["table-workspace", "peg"]


def test_poke_cube(oracle_factory):
    env = PokeCubeEnv(num_envs=4, px_factory=oracle_factory)
    obs, info = env.reset(seed=0)
    assert obs.shape == (4, 54) and not info["success"].any()
    peg, cube, goal, tcp = env.peg_pose, env.poked_cube_pose, env.goal_pos, env.tcp_pose
    assert torch.allclose(cube[:, 0] - peg[:, 0], torch.full((4,), 0.22), atol=1e-6)               # 0.1 in front of the peg's head
    assert torch.allclose(goal[:, 0] - cube[:, 0], torch.full((4,), 0.1), atol=1e-6) and torch.allclose(goal[:, 1], cube[:, 1], atol=1e-6)
    yaw = 2 * torch.atan2(cube[:, 6], cube[:, 3])
    assert (yaw.abs() <= np.pi / 6 + 1e-5).all() and yaw.std() > 1e-3
    # observation layout (:143-157), including the reference's goal_pos = peg position
    assert torch.allclose(obs[:, 18:25], tcp) and torch.allclose(obs[:, 25:32], cube) and torch.allclose(obs[:, 32:39], peg)
    assert torch.allclose(obs[:, 39:42], peg[:, :3]) and torch.allclose(obs[:, 42:45], peg[:, :3] - tcp[:, :3])
    assert torch.allclose(obs[:, 45:48], cube[:, :3] - peg[:, :3]) and torch.allclose(obs[:, 48:51], goal - cube[:, :3])
    assert torch.allclose(obs[:, 51:54], env.peg_head_pos - cube[:, :3])
    assert torch.allclose(info["head_to_cube_dist"], torch.linalg.norm((env.peg_head_pos - cube[:, :3])[:, :2], dim=1))
    r0 = env.step(None)[1]
    assert (r0 > 0).all() and (r0 < 0.2).all()                         # 2 (1 - tanh(5 d)) / 10
    # the cube on the goal with the robot at rest: success, maximum reward
    _teleport(env, env._b_poked, p=torch.cat([goal[:, :2], cube[:, 2:3]], dim=1))
    for _ in range(3):
        obs, rew, term, trunc, info = env.step(None)
    assert info["is_cube_placed"].all() and info["success"].all() and term.all() and torch.allclose(rew, torch.ones(4))
    assert list(env.get_state_dict()["actors"]) == ["table-workspace", "cube", "peg", "goal_region"] and env.get_state().shape == (4, 83)


def test_pull_cube_tool(oracle_factory):
    """PullCubeTool-v1 (pull_cube_tool.py): the L-shaped tool is one actor of two boxes with the reference's densities (0.25 kg each,
    centre of mass between them); layout, observation slices, success once the cube is near the base."""
    from maniskill_amd.envs.pull_cube_tool import PullCubeToolEnv, _compound_box_mass

    m, com, I = _compound_box_mass([((0.1, 0, 0), (0.1, 0.025, 0.025), 500.0), ((0.175, 0.05, 0), (0.025, 0.05, 0.025), 1000.0)])
    assert abs(m - 0.5) < 1e-9 and np.allclose(com, (0.1375, 0.025, 0.0)) and I[2] > I[1] > I[0] > 0 and abs(I[3]) > 0   # xy product of inertia of an L
    env = PullCubeToolEnv(num_envs=4, px_factory=oracle_factory)
    obs, info = env.reset(seed=0)
    assert obs.shape == (4, 39) and not info["success"].any() and env.max_episode_steps == 100
    for _ in range(10):
        obs, rew, term, trunc, info = env.step(None)
    tool, cube = env.tool_pose, env.pulled_cube_pose
    assert ((tool[:, :2] <= -0.1 + 1e-3) & (tool[:, :2] >= -0.3 - 1e-3)).all() and torch.allclose(tool[:, 2], torch.full((4,), 0.025), atol=1.5e-3)
    assert ((cube[:, 0] >= 0.05 - 1e-3) & (cube[:, 0] <= 0.25 + 1e-3)).all() and torch.allclose(cube[:, 2], torch.full((4,), 0.02), atol=1.5e-3)
    assert torch.allclose(obs[:, 25:32], cube) and torch.allclose(obs[:, 32:39], tool) and env.px.get_overflow() == 0
    assert (tool[:, 3] > 0.9999).all()                                   # the L lies flat and does not tip about its offset centre of mass
    _teleport(env, env._b_pulled, p=torch.tensor([[-0.2, 0.05, 0.02]]).repeat(4, 1))
    obs, rew, term, trunc, info = env.step(None)
    assert info["success"].all() and term.all() and (rew > 1.0).all()    # 5 (success) + reaching, / 5
    assert list(env.get_state_dict()["actors"]) == ["table-workspace", "cube", "l_shape_tool"]


def test_registered(oracle_factory):
    for name, dim in (("PullCube-v1", 35), ("LiftPegUpright-v1", 32), ("PokeCube-v1", 54)):
        venv = maniskill_amd.make(name, num_envs=2, px_factory=oracle_factory)
        obs, _ = venv.reset(seed=1)
        obs, rew, term, trunc, info = venv.step(torch.zeros(2, 8))
        assert obs.shape == (2, dim) and rew.shape == (2,)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["PullCube-v1", "LiftPegUpright-v1", "PokeCube-v1", "StackPyramid-v1", "PullCubeTool-v1"])
def test_hip_matches_oracle_rollout(oracle_factory, name):
    from maniskill_amd.envs import registered as _registry
    cls = _registry()[name]
    n = 48
    gpu, cpu = cls(num_envs=n, device="cuda:0"), cls(num_envs=n, px_factory=oracle_factory)
    og, _ = gpu.reset(seed=2022); oc, _ = cpu.reset(seed=2022)
    assert torch.equal(og.cpu(), oc)
    gen = torch.Generator().manual_seed(4)
    for t in range(30):
        a = 2 * torch.rand(n, 8, generator=gen) - 1
        og, rg, tg, ug, _ = gpu.step(a.to("cuda:0"))
        oc, rc, tc, uc, _ = cpu.step(a)
        assert np.allclose(og.cpu().numpy(), oc.numpy(), rtol=1e-4, atol=1e-5), t
        assert np.allclose(rg.cpu().numpy(), rc.numpy(), atol=1e-5) and torch.equal(tg.cpu(), tc)


def test_stack_pyramid(oracle_factory):
    """StackPyramid-v1 (stack_pyramid.py): placements keep their distance, the built pyramid is a success once the gripper is
    away and it stands still, sparse reward."""
    from maniskill_amd.envs.stack_pyramid import StackPyramidEnv

    env = StackPyramidEnv(num_envs=6, px_factory=oracle_factory)
    obs, info = env.reset(seed=0)
    assert obs.shape == (6, 64) and not info["success"].any() and env.max_episode_steps == 250
    P = [env._pose(b) for b in (env._b_cube, env._b_cubeB, env._b_cubeC)]
    for i in range(3):
        assert (P[i][:, 0].abs() <= 0.1 + 1e-6).all() and (P[i][:, 1].abs() <= 0.2 + 1e-6).all() and torch.allclose(P[i][:, 2], torch.full((6,), 0.02), atol=1e-6)
        for j in range(i):
            assert (torch.linalg.norm(P[i][:, :2] - P[j][:, :2], dim=1) > 2 * np.linalg.norm([0.02, 0.02]) - 1e-6).all()
    tcp = env.tcp_pose
    assert torch.allclose(obs[:, 18:25], tcp) and torch.allclose(obs[:, 25:32], P[0]) and torch.allclose(obs[:, 39:46], P[2])
    assert torch.allclose(obs[:, 46:49], P[0][:, :3] - tcp[:, :3]) and torch.allclose(obs[:, 61:64], P[2][:, :3] - P[0][:, :3])
    with pytest.raises(NotImplementedError):
        StackPyramidEnv(num_envs=1, px_factory=oracle_factory, reward_mode="normalized_dense")
    # build the pyramid by hand, away from the arm: A and B side by side (5 mm gap), C centred on top of both
    base = torch.tensor([0.25, 0.3, 0.02])
    ident = torch.tensor([1.0, 0, 0, 0])
    for body, dp in ((env._b_cube, (-0.0225, 0.0, 0.0)), (env._b_cubeB, (0.0225, 0.0, 0.0)), (env._b_cubeC, (0.0, 0.0, 0.0405))):
        env._rbd[:, body, :3] = base + torch.tensor(dp) + env._offsets
        env._rbd[:, body, 3:7] = ident
        env._rbd[:, body, 7:13] = 0.0
    env.px.gpu_apply_all(); env.px.gpu_fetch_all()
    for _ in range(20):
        obs, rew, term, trunc, info = env.step(None)
    assert info["success"].all() and term.all() and torch.equal(rew, torch.ones(6))
    assert torch.allclose(env._pose(env._b_cubeC)[:, 2], torch.full((6,), 0.06), atol=2e-3)      # the top cube stays up there
    # the top cube beside the others instead: no success, reward 0
    env._rbd[:, env._b_cubeC, :3] = base + torch.tensor([0.0, 0.1, 0.0]) + env._offsets
    env.px.gpu_apply_all(); env.px.gpu_fetch_all()
    for _ in range(5):
        obs, rew, term, trunc, info = env.step(None)
    assert not info["success"].any() and torch.equal(rew, torch.zeros(6))
    assert list(env.get_state_dict()["actors"]) == ["table-workspace", "cubeA", "cubeB", "cubeC"] and env.get_state().shape == (6, 13 * 4 + 13 + 18)
