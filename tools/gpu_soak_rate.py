"""On the GPU box: does the step rate hold over a long run?  1000-step windows of (a) the fused PickCube env alone (graph replay, random actions, a full reset every 200 steps),
(b) the same behind ManiSkillVectorEnv (partial auto resets, metrics); per window: ms per step, and the GPU's clock / power as rocm-smi reports them.
    python tools/gpu_soak_rate.py [windows=12] [envs=4096]"""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from maniskill_amd.envs.pick_cube import PickCubeEnv
from maniskill_amd.vector import ManiSkillVectorEnv
W = int(sys.argv[1]) if len(sys.argv) > 1 else 12
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096


def smi():
    try:
        out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showtemp"], capture_output=True, text=True, timeout=20).stdout
        keep = [ln.strip() for ln in out.splitlines() if any(k in ln for k in ("sclk", "Power", "Temperature (Sensor junction)", "mclk"))]
        return " | ".join(k.split(":", 1)[-1].strip() if "sclk" not in k else k.split("sclk")[-1].strip() for k in keep)[:200]
    except Exception as e:   # noqa: BLE001
        return f"(rocm-smi: {e})"


env = PickCubeEnv(num_envs=n, device="cuda:0")
env.enable_step_graph()
env.reset(seed=1)
torch.manual_seed(0)
for w in range(W):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(1000):
        if k % 200 == 0:
            env.reset()
        env.step(2 * torch.rand(n, 8, device="cuda:0") - 1)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"bare env  window {w}: {dt:.3f} ms/step, {n * 1000 / dt / 1e6:.2f} M env-steps/s | {smi()}", flush=True)
# round 6: the env behind the wrapper replays its control step as a graph like the bare one, its auto resets come from the device-side mask (envs/_device_reset.py),
# and the episode phases are randomised at the start: the steady state (some env finishes at almost every step) from the first window on
venv = ManiSkillVectorEnv("PickCube-v1", num_envs=n, device="cuda:0", record_metrics=True)
if os.environ.get("SOAK_HOST_RESETS"):
    venv.base_env.device_reset = False
venv.base_env.enable_step_graph()
venv.reset(seed=7)
if not os.environ.get("SOAK_HOST_RESETS"):      # (a seeded reset voids the prepared episodes: a worker builds the ring again, resets are host-side meanwhile)
    venv.base_env._device_reset_wanted()
    if venv.base_env._dev_reset is not None:
        venv.base_env._dev_reset.wait_ready()
venv.base_env._elapsed_steps.copy_(torch.randint(0, 50, (n,), device="cuda:0", dtype=torch.int32))
for w in range(W):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(1000):
        venv.step(2 * torch.rand(n, 8, device="cuda:0") - 1)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    dr = venv.base_env._dev_reset
    print(f"vector env window {w}: {dt:.3f} ms/step, {n * 1000 / dt / 1e6:.2f} M env-steps/s | {smi()}" + (f" | ring refreshes {dr.refreshes}, episodes prepared {dr.images_made}" if dr else " | host-side resets"), flush=True)
