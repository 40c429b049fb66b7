import sys, torch
sys.path.insert(0, '.')
from maniskill_amd.envs.pick_cube import PickCubeEnv
n = int(sys.argv[1])
env = PickCubeEnv(num_envs=n, device="cuda:0", fused=False)
torch.cuda.synchronize(); print("init ok", flush=True)
env.reset(seed=1); torch.cuda.synchronize(); print("reset ok", flush=True)
for i in range(3):
    env.px.step(); torch.cuda.synchronize(); print("substep", i, "ok", flush=True)
env.step(torch.zeros(n, 8, device="cuda:0")); torch.cuda.synchronize(); print("step ok", env.px.get_overflow(), flush=True)
