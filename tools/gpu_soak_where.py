"""On the GPU box: WHERE does ManiSkillVectorEnv lose time over a long run?  Per 1000-step window: seconds inside env.step, inside the auto reset, and in the
book-keeping around them (each section closed by a device synchronisation), for the fused PickCube env with and without a step graph; plus the process's
memory (host RSS, torch's device allocator)."""
import os, sys, time, resource
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from maniskill_amd.vector import ManiSkillVectorEnv
n = 4096
W = int(sys.argv[1]) if len(sys.argv) > 1 else 10
for graph in (False, True):
    venv = ManiSkillVectorEnv("PickCube-v1", num_envs=n, device="cuda:0", record_metrics=True)
    env = venv.base_env
    if graph:
        env.enable_step_graph()
    venv.reset(seed=7)
    t_step = t_reset = 0.0
    orig_step, orig_reset = env.step, env.reset

    def step(a):
        global t_step
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = orig_step(a)
        torch.cuda.synchronize(); t_step += time.perf_counter() - t0
        return out

    def reset(seed=None, options=None):
        global t_reset
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = orig_reset(seed=seed, options=options)
        torch.cuda.synchronize(); t_reset += time.perf_counter() - t0
        return out
    env.step, env.reset = step, reset
    for w in range(W):
        t_step = t_reset = 0.0
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for k in range(1000):
            venv.step(2 * torch.rand(n, 8, device="cuda:0") - 1)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"graph {graph} window {w}: total {dt:.3f} s = env.step {t_step:.3f} + auto resets {t_reset:.3f} + book-keeping {dt - t_step - t_reset:.3f}; "
              f"host RSS {resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024:.0f} MB, device allocated {torch.cuda.memory_allocated() / 1e6:.0f} MB reserved {torch.cuda.memory_reserved() / 1e6:.0f} MB",
              flush=True)
    del venv, env
