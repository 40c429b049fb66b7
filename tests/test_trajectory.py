"""Trajectory recording / replay (maniskill_amd/trajectory.py; reference: utils/wrappers/record.py, trajectory/replay_trajectory.py):
layout of the recorded file, action replay from seeds, state replay, partial resets; HIP replays of an oracle-recorded trace
under -m gpu."""
import json
import os

import numpy as np
import pytest
import torch

from maniskill_amd.envs.pick_cube import PickCubeEnv
from maniskill_amd.envs.push_t import PushTEnv
from maniskill_amd.trajectory import RecordEpisode, load_trajectory, open_arrays, replay_trajectory, save_arrays

HERE = os.path.dirname(os.path.abspath(__file__))


def _record(env, out_dir, steps, seed=5, partial_at=None, name="trajectory"):
    rec = RecordEpisode(env, out_dir, trajectory_name=name, env_id="PickCube-v1", source_type="test", source_desc="random actions")
    rec.reset(seed=seed)
    gen = torch.Generator().manual_seed(3)
    for k in range(steps):
        if partial_at is not None and k == partial_at:
            rec.reset(options=dict(env_idx=torch.tensor([1, 2])))
        rec.step(0.5 * (2 * torch.rand(env.num_envs, env.action_dim, generator=gen) - 1))
    rec.close()
    return rec


def test_array_container_round_trip(tmp_path):
    tree = {"traj_0": {"actions": np.arange(6, dtype=np.float32).reshape(3, 2), "env_states": {"actors": {"cube": np.ones((4, 13), np.float32)}}}}
    p = str(tmp_path / "t.npz")
    save_arrays(p, tree)
    back = open_arrays(p)
    assert np.array_equal(back["traj_0"]["actions"], tree["traj_0"]["actions"])
    assert back["traj_0"]["env_states"]["actors"]["cube"].shape == (4, 13)


def test_state_dict_views(oracle_factory):
    env = PickCubeEnv(num_envs=3, px_factory=oracle_factory)
    env.reset(seed=0)
    sd = env.get_state_dict()
    assert list(sd["actors"]) == ["table-workspace", "cube", "goal_site"] and list(sd["articulations"]) == ["panda"]
    assert sd["actors"]["cube"].shape == (3, 13) and sd["articulations"]["panda"].shape == (3, 13 + 18)
    assert torch.equal(torch.hstack([*sd["actors"].values(), *sd["articulations"].values()]), env.get_state())
    moved = {"actors": {"cube": sd["actors"]["cube"].clone()}}
    moved["actors"]["cube"][:, 0] += 0.05
    env.set_state_dict(moved)                                      # missing entries keep their values
    now = env.get_state_dict()
    assert torch.allclose(now["actors"]["cube"][:, 0], sd["actors"]["cube"][:, 0] + 0.05, atol=1e-7)
    assert torch.equal(now["articulations"]["panda"], sd["articulations"]["panda"])
    t = PushTEnv(num_envs=2, px_factory=oracle_factory)
    t.reset(seed=0)
    assert list(t.get_state_dict()["actors"]) == ["table-workspace", "Tee", "goal_Tee", "goal_ee"]
    assert t.get_state_dict()["articulations"]["panda_stick"].shape == (2, 13 + 14)


def test_record_layout_and_replays(oracle_factory, tmp_path):
    n, T = 4, 12
    _record(PickCubeEnv(num_envs=n, px_factory=oracle_factory), str(tmp_path), T)
    meta, arrays = load_trajectory(str(tmp_path / "trajectory.npz"))
    assert meta["env_info"]["env_id"] == "PickCube-v1" and meta["env_info"]["max_episode_steps"] == 50 and meta["source_type"] == "test"
    assert [ep["episode_id"] for ep in meta["episodes"]] == [0, 1, 2, 3]
    ep = meta["episodes"][2]
    assert ep["elapsed_steps"] == T and ep["control_mode"] == "pd_joint_delta_pos" and ep["episode_seed"] == 5 + 2 and ep["reset_kwargs"] == {"seed": 7}
    tr = arrays["traj_2"]
    assert tr["actions"].shape == (T, 8) and tr["actions"].dtype == np.float32 and tr["rewards"].shape == (T,)
    assert tr["terminated"].dtype == bool and tr["truncated"].shape == (T,) and tr["success"].shape == (T,)
    assert tr["env_states"]["actors"]["cube"].shape == (T + 1, 13) and tr["env_states"]["articulations"]["panda"].shape == (T + 1, 31)
    # replay by actions from the recorded seeds: the simulation is deterministic, so every recorded state comes back exactly
    res = replay_trajectory(PickCubeEnv(num_envs=n, px_factory=oracle_factory), str(tmp_path / "trajectory.npz"))
    assert res.num_replays == 4 and res.max_state_error == 0.0
    # fewer envs than episodes: two batches; episodes recorded in grid cells 2 and 3 now run in cells 0 and 1, i.e. at
    # another scene offset, which moves the fp32 rounding of their world coordinates
    res2 = replay_trajectory(PickCubeEnv(num_envs=2, px_factory=oracle_factory), str(tmp_path / "trajectory.npz"))
    assert res2.num_replays == 4 and res2.max_state_error < 1e-5
    # with the recorded states re-imposed after every step: a state holds poses and velocities, not the solver's warm-start
    # cache (a set_state is a teleport and drops it) and it passes through the scene-offset subtraction, so one step from a
    # restored state differs from the recorded step in the 1e-4 decade -- the reference notes the same for its GPU replay
    # (replay_trajectory.py:208-211)
    # (in the recording's own grid cells the restored state is bit-identical to the current one, which is no teleport: exact)
    res3 = replay_trajectory(PickCubeEnv(num_envs=4, px_factory=oracle_factory), str(tmp_path / "trajectory.npz"), use_env_states=True)
    assert res3.num_replays == 4 and res3.max_state_error == 0.0
    res4 = replay_trajectory(PickCubeEnv(num_envs=2, px_factory=oracle_factory), str(tmp_path / "trajectory.npz"), use_env_states=True)
    assert res4.num_replays == 4 and res4.max_state_error < 1e-3


def test_partial_reset_splits_episodes(oracle_factory, tmp_path):
    n, T, cut = 4, 10, 4
    _record(PickCubeEnv(num_envs=n, px_factory=oracle_factory), str(tmp_path), T, partial_at=cut)
    meta, arrays = load_trajectory(str(tmp_path / "trajectory.npz"))
    lens = sorted((ep["env_index"], ep["elapsed_steps"], ep["episode_seed"]) for ep in meta["episodes"])
    # envs 1 and 2: an episode of `cut` steps, then one of T - cut steps whose seed is not a reset(seed=...) seed
    assert lens == [(0, T, 5), (1, cut, 6), (1, T - cut, -1), (2, cut, 7), (2, T - cut, -1), (3, T, 8)]
    first = [ep for ep in meta["episodes"] if ep["env_index"] == 1 and ep["elapsed_steps"] == cut][0]
    second = [ep for ep in meta["episodes"] if ep["env_index"] == 1 and ep["elapsed_steps"] == T - cut][0]
    a, b = arrays[f"traj_{first['episode_id']}"], arrays[f"traj_{second['episode_id']}"]
    assert a["env_states"]["actors"]["cube"].shape[0] == cut + 1 and b["env_states"]["actors"]["cube"].shape[0] == T - cut + 1
    assert not np.allclose(a["env_states"]["actors"]["cube"][-1], b["env_states"]["actors"]["cube"][0])   # a fresh episode
    # episodes without a reproducible seed replay from their first recorded state (rounding of the scene-offset round trip)
    res = replay_trajectory(PickCubeEnv(num_envs=n, px_factory=oracle_factory), str(tmp_path / "trajectory.npz"))
    assert res.num_replays == 6 and res.max_state_error < 1e-5


@pytest.mark.gpu
def test_hip_replays_a_trace_recorded_on_the_oracle(oracle_factory, tmp_path):
    """The golden-trace use of §8(f)3: a trajectory recorded on one backend, replayed by actions on the other."""
    n, T = 16, 25
    _record(PickCubeEnv(num_envs=n, px_factory=oracle_factory), str(tmp_path), T)
    res = replay_trajectory(PickCubeEnv(num_envs=n, device="cuda:0"), str(tmp_path / "trajectory.npz"))
    assert res.num_replays == n and res.max_state_error < 1e-4
    _record(PickCubeEnv(num_envs=n, device="cuda:0"), str(tmp_path), T, name="hip")
    again = replay_trajectory(PickCubeEnv(num_envs=n, device="cuda:0"), str(tmp_path / "hip.npz"))
    assert again.max_state_error == 0.0


def test_oracle_reproduces_its_committed_trace(oracle_factory):
    """tests/golden/pickcube_oracle_trace.{npz,json} (tests/golden/make_golden.py): drift detector in the recording layout."""
    res = replay_trajectory(PickCubeEnv(num_envs=8, px_factory=oracle_factory), os.path.join(HERE, "golden", "pickcube_oracle_trace.npz"))
    assert res.num_replays == 8 and res.max_state_error == 0.0


@pytest.mark.gpu
def test_hip_replays_the_committed_trace():
    res = replay_trajectory(PickCubeEnv(num_envs=8, device="cuda:0"), os.path.join(HERE, "golden", "pickcube_oracle_trace.npz"))
    assert res.num_replays == 8 and res.max_state_error < 1e-4


def test_recording_under_the_auto_resetting_vector_env(oracle_factory, tmp_path):
    """RecordEpisode(env) under same-step partial resets (what ManiSkillVectorEnv does, vector/wrappers/gymnasium.py:150-176): the
    resets cut the recorded episodes; every recorded episode is at most max_episode_steps long and replays exactly."""
    env = PickCubeEnv(num_envs=3, px_factory=oracle_factory)
    env.max_episode_steps = 6                                   # short episodes: several truncations within the rollout
    rec = RecordEpisode(env, str(tmp_path), env_id="PickCube-v1")
    rec.reset(seed=9)
    gen = torch.Generator().manual_seed(2)
    finals = 0
    for _ in range(15):
        _, _, term, trunc, _ = rec.step(0.4 * (2 * torch.rand(3, 8, generator=gen) - 1))
        done = term | trunc
        if done.any():
            finals += 1
            rec.reset(options=dict(env_idx=torch.nonzero(done).flatten()))
    rec.close()
    meta, arrays = load_trajectory(str(tmp_path / "trajectory.npz"))
    lens = [ep["elapsed_steps"] for ep in meta["episodes"]]
    assert finals == 2 and sorted(lens) == [3, 3, 3, 6, 6, 6, 6, 6, 6]          # 15 steps = 6 + 6 + 3 per env
    assert all(arrays[f"traj_{ep['episode_id']}"]["truncated"][-1] == (ep["elapsed_steps"] == 6) for ep in meta["episodes"])
    fresh = PickCubeEnv(num_envs=3, px_factory=oracle_factory)
    res = replay_trajectory(fresh, str(tmp_path / "trajectory.npz"))
    assert res.num_replays == 9 and res.max_state_error < 1e-5
