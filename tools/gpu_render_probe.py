"""Development aid: times the rasteriser with parts switched off (variant builds in /tmp)."""
import os, subprocess, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
src = os.path.join(ROOT, "maniskill_amd", "csrc")
flag = sys.argv[1] if len(sys.argv) > 1 else ""
lib = "/tmp/libmsk_rt.so"
subprocess.check_call(f"cd {src} && hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -fvisibility=hidden -Wno-unused-value {flag} -o {lib} msk_physx.hip", shell=True)
from maniskill_amd import _native as N
N.DEFAULT_LIB = lib
from maniskill_amd.envs.pick_cube import PickCubeEnv
n = 4096
env = PickCubeEnv(num_envs=n, device="cuda:0", obs_mode="depth+segmentation")
env.reset(seed=2022); torch.manual_seed(0)
for _ in range(10): env.step(2 * torch.rand(n, 8, device="cuda:0") - 1)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
torch.cuda.synchronize(); ev[0].record()
for _ in range(20): env.camera.take_picture()
ev[1].record(); torch.cuda.synchronize()
print(flag or "full", "us/frame", ev[0].elapsed_time(ev[1]) / 20 * 1e3)
