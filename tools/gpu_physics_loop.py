"""Physics-only loop for profiling: N envs, K substeps after a few random control steps."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from maniskill_amd.envs.pick_cube import PickCubeEnv
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
K = int(sys.argv[2]) if len(sys.argv) > 2 else 50
env = PickCubeEnv(num_envs=N, device="cuda:0")
env.reset(seed=2022)
torch.manual_seed(0)
for _ in range(20):
    env.step(2 * torch.rand(N, 8, device="cuda:0") - 1)
torch.cuda.synchronize(); t = time.time()
for _ in range(K):
    env.px.step()
torch.cuda.synchronize(); dt = time.time() - t
print(f"N={N}: physics substep {dt/K*1e3:.3f} ms")
