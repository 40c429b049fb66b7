"""API-conformance tests of the PickCube host mirror, modelled on the reference's own suite
(tests/test_gpu_envs.py, tests/test_sim_state.py, tests/test_envs.py), run on the CPU oracle."""
import numpy as np
import pytest
import torch

from maniskill_amd.envs.pick_cube import PickCubeEnv


def _env(oracle_factory, n=16, **kw):
    return PickCubeEnv(num_envs=n, px_factory=oracle_factory, **kw)


def test_obs_reward_shapes_and_dtypes(oracle_factory):
    """tests/test_gpu_envs.py:48-73: batched tensors; PickCube state obs is 42 wide (pick_cube.py:132-145)."""
    env = _env(oracle_factory)
    obs, info = env.reset(seed=2022)
    assert obs.shape == (16, 42) and obs.dtype == torch.float32
    obs, rew, term, trunc, info = env.step(torch.zeros(16, 8))
    assert obs.shape == (16, 42) and rew.shape == (16,) and term.dtype == torch.bool and trunc.dtype == torch.bool
    for k in ("success", "is_obj_placed", "is_robot_static", "is_grasped", "elapsed_steps"):
        assert info[k].shape == (16,)
    assert torch.isfinite(obs).all() and (rew >= 0).all() and (rew <= 1).all()


def test_state_shape(oracle_factory):
    """tests/test_sim_state.py:10-37: cube (N,13), panda (N,13+9*2), flat (N, 13*3+13+9*2)."""
    env = _env(oracle_factory)
    assert env.get_state().shape == (16, 13 * 3 + 13 + 9 * 2)


def test_seeded_reset_is_reproducible(oracle_factory):
    """tests/test_envs.py:151-184."""
    env = _env(oracle_factory, 8)
    a, _ = env.reset(seed=17)
    for _ in range(3):
        env.step(torch.rand(8, 8) * 2 - 1)
    b, _ = env.reset(seed=17)
    assert torch.equal(a, b)
    c, _ = env.reset(seed=18)
    assert not torch.equal(a, c)
    # initial conditions honour the task's randomisation ranges (pick_cube.py:106-130)
    cube = a[:, 29:36]
    assert (cube[:, :2].abs() <= 0.1 + 1e-6).all() and torch.allclose(cube[:, 2], torch.full((8,), 0.02))
    goal = a[:, 26:29]
    assert (goal[:, :2].abs() <= 0.1 + 1e-6).all() and (goal[:, 2] >= 0.02 - 1e-6).all() and (goal[:, 2] <= 0.32 + 1e-6).all()
    assert torch.allclose(a[:, 7:9], torch.full((8, 2), 0.04))


def test_partial_reset_isolates_envs(oracle_factory):
    """tests/test_gpu_envs.py:245-269."""
    env = _env(oracle_factory, 8)
    env.reset(seed=5)
    for _ in range(4):
        obs, *_ = env.step(torch.rand(8, 8) * 2 - 1)
    before = env.get_state().clone()
    idx = torch.tensor([1, 5])
    env.reset(options=dict(env_idx=idx))
    after = env.get_state()
    keep = torch.tensor([0, 2, 3, 4, 6, 7])
    assert torch.equal(before[keep], after[keep])
    assert not torch.equal(before[idx], after[idx])
    assert (env._elapsed_steps[idx] == 0).all() and (env._elapsed_steps[keep] == 4).all()


def test_truncation_after_max_episode_steps(oracle_factory):
    """tests/test_gpu_envs.py:272-285: 50 step(None) calls => truncated."""
    env = _env(oracle_factory, 4)
    env.reset(seed=0)
    for i in range(50):
        obs, rew, term, trunc, info = env.step(None)
        assert bool(trunc.all()) == (i == 49)


def test_state_roundtrip_reproduces_trajectory(oracle_factory):
    """tests/test_envs.py:196-212: get_state -> steps -> set_state -> same steps => identical obs.
    A teleport (apply of a changed pose / qpos) drops the env's contact warm-start cache, so two
    replays from the same saved state are bit-identical."""
    env = _env(oracle_factory, 4)
    env.reset(seed=3)
    acts = [torch.rand(4, 8) * 2 - 1 for _ in range(10)]
    for a in acts[:5]:
        env.step(a)
    st = env.get_state().clone()

    def replay():
        env.set_state(st)
        out = None
        for a in acts[5:]:
            out, *_ = env.step(a)
        return out.clone()

    replay()  # the first replay starts from a state equal to the saved one (no teleport); the next two teleport
    r1, r2 = replay(), replay()
    assert torch.equal(r1, r2)


def test_bad_action_shape_raises(oracle_factory):
    env = _env(oracle_factory, 4)
    with pytest.raises(AssertionError, match="expected shape"):
        env.step(torch.zeros(4, 7))


def test_actions_are_clipped(oracle_factory):
    """_clip_and_scale_action (utils/gym_utils.py:104-107): |a| > 1 behaves as |a| = 1."""
    e1, e2 = _env(oracle_factory, 2), _env(oracle_factory, 2)
    e1.reset(seed=1); e2.reset(seed=1)
    o1, *_ = e1.step(torch.full((2, 8), 5.0))
    o2, *_ = e2.step(torch.ones(2, 8))
    assert torch.equal(o1, o2)


def test_scene_offsets_do_not_leak_into_observations(oracle_factory):
    """Sub-scenes sit on a 5 m grid (sapien_env.py:1191-1202) but poses are reported scene-local."""
    env = _env(oracle_factory, 9)
    obs, _ = env.reset(seed=2)
    assert obs[:, 19:22].abs().max() < 2.0  # tcp position
    raw = env.px.cuda_rigid_body_data.torch().view(9, -1, 13)
    assert raw[:, env._b_cube, 0].max() - raw[:, env._b_cube, 0].min() > 4.0


def test_env_results_do_not_depend_on_batch_composition(oracle_factory):
    """Partition invariance: env g behaves the same alone or inside a larger batch."""
    big = _env(oracle_factory, 8)
    small = PickCubeEnv(num_envs=4, px_factory=oracle_factory, env_index_offset=4, total_envs=8)
    ob, _ = big.reset(seed=2022)
    os_, _ = small.reset(seed=2022)
    assert torch.equal(ob[4:], os_)
    gen = torch.Generator().manual_seed(1)
    for _ in range(5):
        a = torch.rand(8, 8, generator=gen) * 2 - 1
        ob, *_ = big.step(a)
        os_, *_ = small.step(a[4:])
    assert torch.equal(ob[4:], os_)


def test_action_and_observation_spaces(oracle_factory):
    """single_action_space / action_space per control mode (Panda._controller_configs, panda.py:77-211)."""
    import maniskill_amd

    for mode, dim in (("pd_joint_delta_pos", 8), ("pd_ee_delta_pose", 7), ("pd_joint_pos_vel", 15), ("pd_ee_pose", 7), ("pd_joint_pos", 8)):
        env = PickCubeEnv(num_envs=3, px_factory=oracle_factory, control_mode=mode)
        sp = env.single_action_space
        assert sp.shape == (dim,) == (env.action_dim,) and env.action_space.shape == (3, dim) and sp.dtype == np.float32
        assert sp.low[-1] == -1.0 and sp.high[-1] == 1.0                    # the mimic gripper is always normalised
        a = env.action_space.sample()
        assert a.shape == (3, dim) and env.action_space.contains(a)
        env.step(torch.from_numpy(a))
    env = PickCubeEnv(num_envs=2, px_factory=oracle_factory, control_mode="pd_joint_pos")
    assert np.allclose(env.single_action_space.low[:7], env.robot.qlimits[0, :7, 0].numpy()) and env.single_action_space.high[3] < 0   # joint 4's range is negative
    assert env.single_observation_space.shape == (42,) and env.observation_space.shape == (2, 42)
    venv = maniskill_amd.make("PushT-v1", num_envs=2, px_factory=oracle_factory)
    assert venv.single_action_space.shape == (7,) and venv.action_space.shape == (2, 7)


def test_edge_cases(oracle_factory):
    """Empty partial reset, single env with a flat action, numpy actions, unknown modes."""
    env = _env(oracle_factory, n=3)
    env.reset(seed=0)
    s0 = env.get_state().clone()
    env.reset(options=dict(env_idx=torch.tensor([], dtype=torch.long)))            # nobody is reset: nothing moves
    assert torch.equal(env.get_state(), s0)
    assert env.step(np.zeros((3, 8), dtype=np.float32))[1].shape == (3,)
    one = _env(oracle_factory, n=1)
    one.reset(seed=0)
    assert one.step(torch.zeros(8))[0].shape == (1, 42)
    for kw in (dict(obs_mode="pointcloud"), dict(control_mode="pd_base_vel"), dict(reward_mode="shaped")):
        with pytest.raises(NotImplementedError):
            _env(oracle_factory, n=1, **kw)
    with pytest.raises(RuntimeError):
        PickCubeEnv(num_envs=1, device="cpu")                                       # the product path has no CPU backend
