"""On the GPU box: a long rollout of the fused PickCube env behind ManiSkillVectorEnv (same-step partial auto resets, episode metrics):
    python tools/gpu_soak.py [steps=20000] [envs=4096]
Checks every 1000 steps that observations / rewards / simulator state are finite and that no solver scheduling flag was raised."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from maniskill_amd.vector import ManiSkillVectorEnv
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
venv = ManiSkillVectorEnv("PickCube-v1", num_envs=n, device="cuda:0", record_metrics=True)
env = venv.base_env
obs, _ = venv.reset(seed=7)
torch.manual_seed(1)
episodes = 0
t0 = time.perf_counter()
for k in range(steps):
    obs, rew, term, trunc, info = venv.step(2 * torch.rand(n, 8, device="cuda:0") - 1)
    if "_final_info" in info:
        episodes += int(info["_final_info"].sum())
    if k % 1000 == 999:
        ok = bool(torch.isfinite(obs).all() and torch.isfinite(rew).all() and torch.isfinite(env.get_state()).all())
        flags = env.px.get_overflow()
        print(f"step {k + 1}: finite {ok}, flags {flags}, episodes finished {episodes}, max |qvel| {float(env.qvel.abs().max()):.1f}, "
              f"{n * (k + 1) / (time.perf_counter() - t0) / 1e6:.2f} M env-steps/s incl. resets and checks", flush=True)
        assert ok and flags & 6 == 0
print("SOAK_OK")
