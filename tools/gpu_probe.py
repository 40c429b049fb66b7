"""Quick on-GPU probe: HIP vs oracle parity trace + a first timing. Run via gpurun."""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle_backend import OraclePhysxSystem
from maniskill_amd.envs.pick_cube import PickCubeEnv

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
print("torch", torch.__version__, "cuda", torch.cuda.is_available(), torch.cuda.get_device_name(0))
gpu = PickCubeEnv(num_envs=n, device="cuda:0")
cpu = PickCubeEnv(num_envs=n, px_factory=lambda tpl, k, cfg: OraclePhysxSystem(tpl, k, cfg))
og, _ = gpu.reset(seed=2022); oc, _ = cpu.reset(seed=2022)
print("reset diff", (og.cpu() - oc).abs().max().item())
gen = torch.Generator().manual_seed(0)
worst = 0.0
for i in range(steps):
    a = 2 * torch.rand(n, 8, generator=gen) - 1
    og, rg, tg, ug, ig = gpu.step(a.to("cuda:0"))
    oc, rc, tc, uc, ic = cpu.step(a)
    d = (og.cpu() - oc).abs()
    worst = max(worst, d.max().item())
    if i % 10 == 0 or i == steps - 1:
        bad = (d.max(dim=1)[0] > 1e-4).sum().item()
        # contact-pair index parity on a few envs
        same = 0
        for e in range(min(n, 16)):
            gi, gv = gpu.px.get_contacts(e); ci, cv = cpu.px.get_contacts(e)
            same += int(gi.shape == ci.shape and (gi == ci).all())
        print(f"step {i}: max|dobs|={d.max().item():.3e} envs>1e-4: {bad}/{n} contact-id-equal {same}/{min(n,16)} bitexact={bool((og.cpu()==oc).all())}")
print("worst", worst)
# timing
for N in (4096,):
    env = PickCubeEnv(num_envs=N, device="cuda:0")
    env.reset(seed=2022)
    a = 2 * torch.rand(N, 8, device="cuda:0") - 1
    for _ in range(3): env.step(a)
    torch.cuda.synchronize(); t = time.time()
    K = 20
    for _ in range(K):
        a = 2 * torch.rand(N, 8, device="cuda:0") - 1
        env.step(a)
    torch.cuda.synchronize(); dt = time.time() - t
    print(f"N={N}: {dt/K*1e3:.2f} ms/step, {N*K/dt:.0f} env-steps/s")
    # physics-only
    torch.cuda.synchronize(); t = time.time()
    for _ in range(K*5): env.px.step()
    torch.cuda.synchronize(); dt = time.time() - t
    print(f"N={N}: physics substep {dt/(K*5)*1e3:.3f} ms")
