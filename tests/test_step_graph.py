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
