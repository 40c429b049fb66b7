"""One control step replayed as a captured HIP graph (maniskill_amd/graph.py) must give the eager step's results bit for bit:
same kernels, same order, same stream -- only the launch mechanism differs."""
import pytest
import torch

from maniskill_amd.graph import StepGraph


def test_step_graph_needs_a_gpu_env():
    with pytest.raises(RuntimeError, match="GPU"):
        StepGraph(lambda a: None, 4, 8, "cpu")


def _rollout_pair(make, steps, seed=3, scale=0.6):
    eager, graphed = make(), make()
    zero = torch.zeros(eager.num_envs, eager.action_dim, device=eager.device)
    for _ in range(2):           # enable_step_graph() runs two throw-away eager steps before the capture: mirror them
        eager.step(zero)
    g = graphed.enable_step_graph(warmup=2)
    o0, _ = eager.reset(seed=seed)
    o1, _ = graphed.reset(seed=seed)
    assert torch.equal(_state(o0), _state(o1))
    gen = torch.Generator(device="cpu").manual_seed(11)
    for k in range(steps):
        a = (scale * (2 * torch.rand(eager.num_envs, eager.action_dim, generator=gen) - 1)).to(eager.device)
        e = eager.step(a)
        r = graphed.step(a)
        assert torch.equal(_state(e[0]), _state(r[0])), f"obs differ at step {k}"
        assert torch.equal(e[1], r[1]) and torch.equal(e[2], r[2]) and torch.equal(e[3], r[3])
        for key in ("success", "elapsed_steps"):
            assert torch.equal(e[4][key], r[4][key])
    assert g.replays == steps
    return eager, graphed, e, r


def _state(obs):
    return obs["state"] if isinstance(obs, dict) else obs


@pytest.mark.gpu
def test_graph_replay_equals_eager_torch_task_path():
    from maniskill_amd.envs.peg_insertion_side import PegInsertionSideEnv
    eager, graphed, _, _ = _rollout_pair(lambda: PegInsertionSideEnv(num_envs=96, device="cuda:0", fused=False), steps=12)
    # a partial reset between replays is eager work on the same persistent state: no re-capture
    idx = torch.tensor([0, 5, 17, 95], device="cuda:0")
    eager.reset(seed=9, options=dict(env_idx=idx))
    graphed.reset(seed=9, options=dict(env_idx=idx))
    a = torch.full((96, eager.action_dim), 0.25, device="cuda:0")
    e, r = eager.step(a), graphed.step(a)
    assert torch.equal(e[0], r[0]) and torch.equal(e[1], r[1])
    assert (r[4]["elapsed_steps"][idx] == 1).all()


@pytest.mark.gpu
def test_graph_replay_equals_eager_fused_path_with_camera():
    from maniskill_amd.envs.pick_cube import PickCubeEnv
    make = lambda: PickCubeEnv(num_envs=64, device="cuda:0", obs_mode="depth+segmentation")
    eager, graphed, e, r = _rollout_pair(make, steps=6, scale=1.0)
    de, dr = e[0]["sensor_data"]["base_camera"], r[0]["sensor_data"]["base_camera"]
    assert torch.equal(de["depth"], dr["depth"]) and torch.equal(de["segmentation"], dr["segmentation"])
    assert (dr["segmentation"] > 0).any()


@pytest.mark.gpu
def test_graph_replay_with_a_mounted_camera_equals_eager():
    """PegInsertionSide's hand_camera rides on camera_link: its picture and its per-step camera matrices (pose product and rigid inverse,
    written without host-checked linear algebra) are part of the captured step."""
    from maniskill_amd.envs.peg_insertion_side import PegInsertionSideEnv
    make = lambda: PegInsertionSideEnv(num_envs=48, device="cuda:0", obs_mode="depth+segmentation")   # noqa: E731
    eager, graphed, e, r = _rollout_pair(make, steps=6, scale=1.0)
    for uid in ("base_camera", "hand_camera"):
        de, dr = e[0]["sensor_data"][uid], r[0]["sensor_data"][uid]
        assert torch.equal(de["depth"], dr["depth"]) and torch.equal(de["segmentation"], dr["segmentation"]), uid
        for k in ("extrinsic_cv", "cam2world_gl", "intrinsic_cv"):
            assert torch.equal(e[0]["sensor_param"][uid][k], r[0]["sensor_param"][uid][k]), (uid, k)
    moved = (r[0]["sensor_param"]["hand_camera"]["cam2world_gl"][:, :3, 3] - graphed.reset(seed=3)[0]["sensor_param"]["hand_camera"]["cam2world_gl"][:, :3, 3]).norm(dim=1)
    assert moved.min().item() > 1e-3


@pytest.mark.gpu
def test_outputs_of_a_replay_survive_the_next_replay():
    from maniskill_amd.envs.push_cube import PushCubeEnv
    env = PushCubeEnv(num_envs=32, device="cuda:0")
    env.enable_step_graph()
    env.reset(seed=1)
    a = torch.full((32, env.action_dim), 0.5, device="cuda:0")
    obs1, rew1, *_ = env.step(a)
    keep_o, keep_r = obs1.clone(), rew1.clone()
    obs2, rew2, *_ = env.step(-a)
    assert torch.equal(obs1, keep_o) and torch.equal(rew1, keep_r) and not torch.equal(obs1, obs2)


@pytest.mark.gpu
def test_camera_planes_and_info_of_a_replay_survive_the_next_replay():
    """the camera .clone() runs inside the capture, so sensor_data / info must be copied after the replay."""
    from maniskill_amd.envs.pick_cube import PickCubeEnv
    env = PickCubeEnv(num_envs=32, device="cuda:0", obs_mode="depth+segmentation")
    env.enable_step_graph()
    env.reset(seed=1)
    a = torch.full((32, env.action_dim), 0.8, device="cuda:0")
    obs1, _, _, _, info1 = env.step(a)
    d1 = obs1["sensor_data"]["base_camera"]["depth"]
    keep_d, keep_steps = d1.clone(), info1["elapsed_steps"].clone()
    for _ in range(4):
        obs2, _, _, _, info2 = env.step(-a)
    assert torch.equal(d1, keep_d) and torch.equal(info1["elapsed_steps"], keep_steps)
    assert not torch.equal(obs2["sensor_data"]["base_camera"]["depth"], keep_d)
    assert (info2["elapsed_steps"] == keep_steps + 4).all()


@pytest.mark.gpu
def test_apply_force_between_replays_acts_on_the_next_replay():
    """the external-wrench path is data-driven (k_dynamics consumes and clears the rows), so a force applied eagerly
    between two replays of a captured step acts during the next replay and only then."""
    from maniskill_amd.envs.push_cube import PushCubeEnv
    make = lambda: PushCubeEnv(num_envs=16, device="cuda:0")   # noqa: E731
    eager, graphed = make(), make()
    zero = torch.zeros(16, eager.action_dim, device="cuda:0")
    for _ in range(2):
        eager.step(zero)
    graphed.enable_step_graph(warmup=2)
    eager.reset(seed=4)
    graphed.reset(seed=4)
    cube = eager.template.body_id("cube")
    f = torch.tensor([3.0, 0.0, 0.0], device="cuda:0")
    for k in range(4):
        if k == 1:
            eager.px.apply_force(cube, f)
            graphed.px.apply_force(cube, f)
        e, r = eager.step(zero), graphed.step(zero)
        assert torch.equal(e[0], r[0]), f"step {k}"
    eager.px.gpu_fetch_all()
    vx = eager.px.cuda_rigid_body_data.torch().view(16, -1, 13)[:, cube, 0]
    ref = make()
    ref.reset(seed=4)
    for k in range(4):
        ref.step(zero)
    ref.px.gpu_fetch_all()
    assert (vx - ref.px.cuda_rigid_body_data.torch().view(16, -1, 13)[:, cube, 0]).abs().min() > 1e-5   # the push moved the cube
