"""Per-kernel-slot HIP-event times of the physics substep (dynamics / collide / solve) early and late in a random rollout.
usage: python tools/gpu_kernel_probe.py [N=4096] [late_steps=600]   (tuning env vars: MSK_NP_NBOX, MSK_NP_NHULL, MSK_NP_BOXGROUP)"""
import os, sys, json
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from maniskill_amd.envs.pick_cube import PickCubeEnv
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
LATE = int(sys.argv[2]) if len(sys.argv) > 2 else 600
env = PickCubeEnv(num_envs=N, device="cuda:0")
env.reset(seed=2022)
torch.manual_seed(0)
out = {}


def measure(tag):
    env.px.timing_enable(20 * 5)
    for _ in range(20):
        env.step(2 * torch.rand(N, 8, device="cuda:0") - 1)
    torch.cuda.synchronize()
    k = env.px.timing_read()
    env.px.timing_enable(0)
    out[tag] = {n: round(v[0] / max(v[1], 1) * 1e3, 1) for n, v in k.items()}
    out[tag]["contacts_mean"] = float(env.px.get_env_contact_counts().mean())


for _ in range(20):
    env.step(2 * torch.rand(N, 8, device="cuda:0") - 1)
measure("early")
for _ in range(LATE):
    env.step(2 * torch.rand(N, 8, device="cuda:0") - 1)
measure("late")
tags = {k: os.environ[k] for k in ("MSK_NP_NBOX", "MSK_NP_NHULL", "MSK_NP_BOXGROUP", "MSK_NP_SKIP") if k in os.environ}
print(json.dumps({"N": N, "env": tags, **out}))
