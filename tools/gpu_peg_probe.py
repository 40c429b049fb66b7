"""On-GPU probe: PegInsertionSide-v1 (config 4): per-kernel times, solver class counts and the contact histogram under random actions."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from maniskill_amd import _native as _N
if os.environ.get("MSK_LIB"):
    _N.DEFAULT_LIB = os.path.abspath(os.environ["MSK_LIB"])
from maniskill_amd.envs.peg_insertion_side import PegInsertionSideEnv
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
env = PegInsertionSideEnv(num_envs=N, device="cuda:0")
env.reset(seed=2022)
torch.manual_seed(0)
for r in range(4):
    env.px.timing_enable(60 * 5)
    for _ in range(60):
        env.step(2 * torch.rand(N, 8, device="cuda:0") - 1)
    t = env.px.timing_read()
    env.px.timing_enable(0)
    c = env.px.get_env_contact_counts()
    print({k: round(v[0] / max(v[1], 1) * 1e3, 1) for k, v in t.items()}, "contacts mean %.2f max %d p99 %d" % (c.mean(), c.max(), np.percentile(c, 99)),
          "hist", np.bincount(c)[:24].tolist(), "classes", env.px.get_solver_class_counts().tolist(), "flags", env.px.get_overflow())
