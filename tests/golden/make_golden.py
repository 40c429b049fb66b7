"""Generates the committed golden fixtures from the CPU oracle (oracle/liborc.so).

The reference's own arithmetic for this path (PhysX 5 inside the sapien wheel) is not runnable
anywhere in this project, and the reference tree holds no golden vectors for it
(no *.npz/*.h5/*.pkl), so these fixtures pin the ORACLE, not PhysX: parity unpinned.  They serve
(1) to detect drift of the oracle itself and (2) as inputs that travel to the GPU box.

Usage: python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle_backend import OraclePhysxSystem  # noqa: E402
from maniskill_amd.envs.pick_cube import PickCubeEnv  # noqa: E402


def rollout(n=8, steps=40, seed=2022):
    env = PickCubeEnv(num_envs=n, px_factory=lambda tpl, k, cfg: OraclePhysxSystem(tpl, k, cfg))
    obs, _ = env.reset(seed=seed)
    gen = torch.Generator().manual_seed(0)
    actions = (2 * torch.rand(steps, n, 8, generator=gen) - 1).numpy().astype(np.float32)
    out_obs = [obs.numpy().copy()]
    rew, contact_ids = [], []
    for t in range(steps):
        obs, r, term, trunc, info = env.step(torch.from_numpy(actions[t]))
        out_obs.append(obs.numpy().copy())
        rew.append(r.numpy().copy())
        ids = [env.px.get_contacts(e)[0][:, :2] for e in range(n)]
        contact_ids.append(np.array([len(i) for i in ids] + [int(x) for i in ids for x in i.reshape(-1)], dtype=np.int32))
    return dict(actions=actions, obs=np.stack(out_obs), rew=np.stack(rew),
                state=env.get_state().numpy().copy(), contact_ids=np.concatenate(contact_ids),
                contact_ids_offsets=np.cumsum([0] + [len(c) for c in contact_ids]).astype(np.int64))


def trace(n=8, steps=20, seed=31):
    """A trajectory in the reference's recording layout (maniskill_amd/trajectory.py): pickcube_oracle_trace.{npz,json}."""
    from maniskill_amd.trajectory import RecordEpisode

    env = PickCubeEnv(num_envs=n, px_factory=lambda tpl, k, cfg: OraclePhysxSystem(tpl, k, cfg))
    rec = RecordEpisode(env, HERE, trajectory_name="pickcube_oracle_trace", env_id="PickCube-v1", source_type="oracle",
                        source_desc="uniform random actions in [-0.7, 0.7], CPU oracle (oracle/liborc.so), tests/golden/make_golden.py")
    rec.reset(seed=seed)
    gen = torch.Generator().manual_seed(1)
    for _ in range(steps):
        rec.step(0.7 * (2 * torch.rand(n, 8, generator=gen) - 1))
    rec.close()


if __name__ == "__main__":
    trace()
    print("wrote pickcube_oracle_trace.npz / .json")
    g = rollout()
    np.savez_compressed(os.path.join(HERE, "pickcube_oracle_rollout.npz"), **g)
    print("wrote pickcube_oracle_rollout.npz", {k: v.shape for k, v in g.items()})
