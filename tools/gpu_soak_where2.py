"""On the GPU box: which call inside PickCubeEnv.reset gets slower from reset to reset?  cProfile of 20 auto resets early and 20 after 200 resets."""
import cProfile, io, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from maniskill_amd.envs.pick_cube import PickCubeEnv
n = 4096
env = PickCubeEnv(num_envs=n, device="cuda:0")
env.reset(seed=7)
idx = torch.arange(n, device="cuda:0")


def window(tag, k, profile):
    pr = cProfile.Profile()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    if profile:
        pr.enable()
    for _ in range(k):
        for _ in range(int(os.environ.get("STEPS_BETWEEN", "5"))):
            env.step(2 * torch.rand(n, 8, device="cuda:0") - 1)
        torch.cuda.synchronize()
        env.reset(options=dict(env_idx=idx))
        torch.cuda.synchronize()
    if profile:
        pr.disable()
    dt = time.perf_counter() - t0
    print(f"{tag}: {dt / k * 1e3:.2f} ms per (5 steps + reset)", flush=True)
    if profile:
        s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(14); print(s.getvalue()[:3500], flush=True)


window("resets 0..20", 20, True)
for w in range(9):
    window(f"resets {20 * (w + 1)}..{20 * (w + 2)}", 20, False)
window("resets 200..220", 20, True)
