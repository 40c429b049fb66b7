"""On the GPU box: where does a ManiSkillVectorEnv step over the fused PickCube env spend its time?  Cumulative variants, 300 steps each, episode phases randomised
(some env finishes at almost every step); plus the host profile of the full wrapper.    python tools/gpu_vector_probe.py [envs=4096] [steps=300]"""
import cProfile, io, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from maniskill_amd.envs.pick_cube import PickCubeEnv
from maniskill_amd.vector import ManiSkillVectorEnv
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
K = int(sys.argv[2]) if len(sys.argv) > 2 else 300
D = "cuda:0"


def fresh(device_reset):
    env = PickCubeEnv(num_envs=n, device=D, device_reset=device_reset)
    env.enable_step_graph()
    env.reset(seed=3)
    if device_reset:      # (the ring of prepared episodes is built by a worker after a seeded reset: the probe is about the device path)
        env._device_reset_wanted()
        env._dev_reset.wait_ready()
    env._elapsed_steps.copy_(torch.randint(0, 50, (n,), device=D, dtype=torch.int32))
    return env


def timed(name, fn, warm=70):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(K):
        fn()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / K * 1e3
    print(f"{name:72s} {dt:.3f} ms/step  {n / dt / 1e3:.2f} M env-steps/s", flush=True)


act = lambda: 2 * torch.rand(n, 8, device=D) - 1
env = fresh(True)
timed("A  env.step (graph), nothing else; truncated envs just run on", lambda: env.step(act()))
def B():
    o, r, te, tr, i = env.step(act()); bool((te | tr).any())
timed("B  A + `done.any()` (the reference's wait)", B)
def C():
    o, r, te, tr, i = env.step(act()); d = te | tr; env.reset_mask(d); bool(d.any())
timed("C  B + reset_mask(done) issued before the wait (device-side resets)", C)
dr = env._dev_reset
print(f"   device reset: {dr.resets} issued, {dr.refreshes} ring refreshes, {dr.images_made} episodes prepared, {dr.nent} words per image, ring of {dr.slots}")
import maniskill_amd.envs._device_reset as DRM
acc = [0.0, 0]
orig = DRM.DeviceReset.refresh
def refresh(self):
    t = time.perf_counter(); orig(self); acc[0] += time.perf_counter() - t; acc[1] += 1
DRM.DeviceReset.refresh = refresh
timed("C  again, the ring refreshes timed on the host", C, warm=0)
print(f"   {acc[1]} refreshes took {acc[0] * 1e3:.1f} ms in {K} steps = {acc[0] / K * 1e3:.3f} ms per step")
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
d = torch.zeros(n, dtype=torch.bool, device=D); d[::50] = True
torch.cuda.synchronize(); ev[0].record()
for _ in range(20):
    env.reset_mask(d)
ev[1].record(); torch.cuda.synchronize()
print(f"   reset_mask alone (82 envs named: kernel + observe + copy), device time: {ev[0].elapsed_time(ev[1]) / 20 * 1e3:.1f} us")
venv = ManiSkillVectorEnv(fresh(True), record_metrics=True)
timed("D  ManiSkillVectorEnv(record_metrics) over the same env, device-side resets", lambda: venv.step(act()))
venv0 = ManiSkillVectorEnv(fresh(True), record_metrics=False)
timed("D' the same without record_metrics", lambda: venv0.step(act()))
venvh = ManiSkillVectorEnv(fresh(False), record_metrics=True)
timed("E  ManiSkillVectorEnv(record_metrics), host-side resets", lambda: venvh.step(act()))
pr = cProfile.Profile(); pr.enable()
for _ in range(K):
    venv.step(act())
torch.cuda.synchronize(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(18); print(s.getvalue()[:4000])
