"""On-GPU probe: per-kernel times of the physics substep under (a) zero actions (every env has the 4
cube-table contacts only) and (b) random actions (long contact tails), plus the contact histogram."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from maniskill_amd import _native as _N
if os.environ.get("MSK_LIB"):      # A/B runs of two builds on the same box: MSK_LIB=maniskill_amd/csrc/libmsk_b.so
    _N.DEFAULT_LIB = os.path.abspath(os.environ["MSK_LIB"])
from maniskill_amd.envs.pick_cube import PickCubeEnv

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
env = PickCubeEnv(num_envs=N, device="cuda:0")
env.reset(seed=2022)
torch.manual_seed(0)


def run(tag, steps, act):
    env.px.timing_enable(steps * 5)
    for _ in range(steps):
        env.step(act())
    t = env.px.timing_read()
    env.px.timing_enable(0)
    c = env.px.get_env_contact_counts()
    print(tag, {k: round(v[0] / max(v[1], 1) * 1e3, 1) for k, v in t.items()}, "us/launch; contacts mean %.2f max %d p99 %d"
          % (c.mean(), c.max(), np.percentile(c, 99)), "hist", np.bincount(c)[:12].tolist(), "classes", env.px.get_solver_class_counts().tolist(),
          "flags", env.px.get_overflow())


run("zero-actions  ", 20, lambda: torch.zeros(N, 8, device="cuda:0"))
for _ in range(int(os.environ.get("PROBE_ROUNDS", "4"))):
    run("random-actions", 60, lambda: 2 * torch.rand(N, 8, device="cuda:0") - 1)
