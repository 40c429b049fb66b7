"""On the GPU box: cProfile of ManiSkillVectorEnv windows 0 and N of a long run (what grows?)."""
import cProfile, io, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from maniskill_amd.vector import ManiSkillVectorEnv
n = 4096
venv = ManiSkillVectorEnv("PickCube-v1", num_envs=n, device="cuda:0", record_metrics=True)
venv.reset(seed=7)
W = int(sys.argv[1]) if len(sys.argv) > 1 else 8
for w in range(W):
    prof = w in (0, 1, W - 1)
    pr = cProfile.Profile()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    if prof:
        pr.enable()
    for k in range(1000):
        venv.step(2 * torch.rand(n, 8, device="cuda:0") - 1)
    torch.cuda.synchronize()
    if prof:
        pr.disable()
    print(f"window {w}: {time.perf_counter() - t0:.3f} s", flush=True)
    if prof:
        s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(12); print(s.getvalue()[:3000], flush=True)
