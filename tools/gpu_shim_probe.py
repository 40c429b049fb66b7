#!/usr/bin/env python3
"""GPU-box probe: the reference's own mani_skill on the HIP backend through the sapien shim (needs a staged checkout, see
tests/ref_harness.py).  Prints parity of a short rollout against the same rollout on the CPU checker and env-steps/s."""
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "tests"))
sys.path.insert(0, os.path.join(HERE, ".."))
import ref_harness  # noqa: E402

print("reference:", ref_harness.find_reference(), flush=True)
gym = ref_harness.setup("hip")
if gym is None:
    print("no reference checkout staged")
    sys.exit(0)
import torch  # noqa: E402

env_id = sys.argv[1] if len(sys.argv) > 1 else "PickCube-v1"
out = {}
for n in (16, 4096):
    t0 = time.time()
    env = gym.make(env_id, num_envs=n, render_backend="none")
    obs, _ = env.reset(seed=0)
    t_build = time.time() - t0
    torch.manual_seed(0)
    steps = 50 if n > 16 else 20
    torch.cuda.synchronize()
    t0 = time.time()
    for i in range(steps):
        a = 2 * torch.rand(env.action_space.shape, device=obs.device) - 1
        obs, rew, term, trunc, info = env.step(a)
    torch.cuda.synchronize()
    dt = time.time() - t0
    out[n] = dict(build_s=round(t_build, 2), steps_per_s=round(n * steps / dt, 1), ms_per_step=round(1e3 * dt / steps, 3),
                  obs_finite=bool(torch.isfinite(obs).all()), rew_mean=float(rew.mean()))
    print(n, out[n], flush=True)
    env.close()
print(json.dumps(out))
