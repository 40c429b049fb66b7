"""On the GPU box: HIP vs oracle over long rollouts from several seeds (a wider net than the -m gpu tests: every step of the early, middle
and late regime of the same envs).   python tools/gpu_fuzz_parity.py [envs=128] [steps=400] [seeds=1,2,3] [tasks=PickCube,Peg]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from oracle_backend import OraclePhysxSystem
from maniskill_amd.envs.pick_cube import PickCubeEnv
from maniskill_amd.envs.peg_insertion_side import PegInsertionSideEnv
from maniskill_amd.envs.push_t import PushTEnv
from maniskill_amd.envs.stack_cube import StackCubeEnv
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 400
seeds = [int(s) for s in (sys.argv[3] if len(sys.argv) > 3 else "1,2,3").split(",")]
tasks = dict(PickCube=PickCubeEnv, Peg=PegInsertionSideEnv, PushT=PushTEnv, StackCube=StackCubeEnv)
for cls in [tasks[k] for k in (sys.argv[4] if len(sys.argv) > 4 else "PickCube,Peg").split(",")]:
    for seed in seeds:
        gpu = cls(num_envs=n, device="cuda:0", fused=False)
        cpu = cls(num_envs=n, px_factory=lambda t, k, c: OraclePhysxSystem(t, k, c))
        gpu.reset(seed=seed); cpu.reset(seed=seed)
        gen = torch.Generator().manual_seed(100 + seed)
        bit, worst = 0, 0.0
        for t in range(steps):
            a = 2 * torch.rand(n, gpu.action_dim, generator=gen) - 1
            og, *_ = gpu.step(a.to("cuda:0")); oc, *_ = cpu.step(a)
            d = float((og.cpu() - oc).abs().max())
            worst = max(worst, d); bit += int(d == 0.0)
            if not np.isfinite(d) or d > 1e-3:
                print(f"{cls.__name__} seed {seed}: DIVERGED at step {t}: {d}"); break
        sg, sc = gpu.get_state().cpu(), cpu.get_state()
        print(f"{cls.__name__} seed {seed}: {bit}/{steps} steps bit-equal, worst |obs diff| {worst:.2e}, final state diff {float((sg - sc).abs().max()):.2e}, "
              f"flags hip {gpu.px.get_overflow()} oracle {cpu.px.get_overflow()}", flush=True)
