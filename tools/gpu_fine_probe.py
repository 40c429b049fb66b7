"""Development aid: finer s_memtime stamps of k_csolve's prologue (temporary PHASE placement)."""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from maniskill_amd import _native as N
N.DEFAULT_LIB = os.path.join(ROOT, "maniskill_amd", "csrc", "libmsk_prof.so")
from maniskill_amd.envs.pick_cube import PickCubeEnv
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
env = PickCubeEnv(num_envs=n, device="cuda:0")
env.reset(seed=2022); torch.manual_seed(0)
dll = env.px.lib.dll
dll.msk_debug_phases.argtypes = [C.c_void_p, C.POINTER(C.c_longlong)]
for _ in range(100): env.step(2 * torch.rand(n, 8, device="cuda:0") - 1)
out = np.zeros(n * 16 + 64, dtype=np.int64)
dll.msk_debug_phases(env.px.ctx, out.ctypes.data_as(C.POINTER(C.c_longlong)))
t = out[:n * 8].reshape(n, 8)
d = np.diff(t, axis=1)
names = sys.argv[2].split(",") if len(sys.argv) > 2 else [f"p{i}" for i in range(7)]
print({k: int(v) for k, v in zip(names, d.mean(0))}, "total", int((t[:, 7] - t[:, 0]).mean()))
