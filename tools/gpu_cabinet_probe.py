#!/usr/bin/env python3
"""BASELINE config 5 on the GPU box: the reference's own OpenCabinetDrawer-v1 (Fetch + a different synthetic cabinet per sub-scene,
tools/make_synthetic_partnet.py) on the HIP backend through the sapien shim.  Prints (1) HIP vs the CPU checker over 50 control
steps on 32 envs, (2) env-steps/s at the env counts given."""
import json
import os
import subprocess
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import ref_harness  # noqa: E402

ref = ref_harness.find_reference()
if ref is None:
    print("no reference build staged")
    sys.exit(0)
assets = "/tmp/ms_assets_synth"
meta = os.path.join(ref, "mani_skill", "assets", "partnet_mobility", "meta")
max_drawers = os.environ.get("MSK_CABINET_DRAWERS", "1")
subprocess.check_call([sys.executable, os.path.join(HERE, "make_synthetic_partnet.py"), "--out", assets, "--max-drawers", max_drawers, "--ids-from",
                       os.path.join(meta, "info_cabinet_drawer_train.json"), "--placeholder-ids-from",
                       os.path.join(meta, "info_cabinet_door_train.json")], stdout=subprocess.DEVNULL)
os.environ["MS_ASSET_DIR"] = assets
mode = sys.argv[1] if len(sys.argv) > 1 else "bench"
import torch  # noqa: E402

if mode == "parity":
    # two processes' worth of global state (the shim's backend switch is process-wide): run the oracle side in a child
    if len(sys.argv) > 2 and sys.argv[2] == "child":
        gym = ref_harness.setup("oracle")
        env = gym.make("OpenCabinetDrawer-v1", num_envs=32, render_backend="gpu")
        env.reset(seed=0)
        g = torch.Generator().manual_seed(3)
        out = [env.unwrapped.get_state().clone()]      # the episode's initial state (torch's CPU and GPU generators differ: handed over)
        for k in range(50):
            a = 2 * torch.rand(32, 13, generator=g) - 1
            env.step(a)
            out.append(env.unwrapped.get_state().clone())
        torch.save(torch.stack(out), "/tmp/cab_oracle.pt")
        sys.exit(0)
    subprocess.check_call([sys.executable, __file__, "parity", "child"])
    gym = ref_harness.setup("hip")
    env = gym.make("OpenCabinetDrawer-v1", num_envs=32)
    env.reset(seed=0)
    g = torch.Generator().manual_seed(3)
    ref_states = torch.load("/tmp/cab_oracle.pt")
    env.unwrapped.set_state(ref_states[0].cuda())
    worst, first_bad = 0.0, None
    for k in range(50):
        a = 2 * torch.rand(32, 13, generator=g) - 1
        env.step(a.cuda())
        s = env.unwrapped.get_state().cpu()
        assert torch.isfinite(s).all() and torch.isfinite(ref_states[k + 1]).all(), k
        err = float(((s - ref_states[k + 1]).abs() / (1 + ref_states[k + 1].abs())).max())
        if err > 1e-3 and first_bad is None:
            first_bad = (k, err)
        worst = max(worst, err)
    print("first step above 1e-3:", first_bad)
    print(json.dumps({"parity": "OpenCabinetDrawer-v1 32 envs x 50 steps HIP vs oracle", "max_rel_err": worst, "groups": len(env.unwrapped.scene.px._groups)}))
    sys.exit(0 if worst < 1e-3 else 1)

gym = ref_harness.setup("hip")
if mode == "profile":   # where a control step of the multi-group scene spends its host time
    import cProfile, pstats
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
    env = gym.make("OpenCabinetDrawer-v1", num_envs=n)
    obs, _ = env.reset(seed=0)
    for _ in range(3):
        env.step(2 * torch.rand(env.action_space.shape, device=obs.device) - 1)
    torch.cuda.synchronize()
    prof = cProfile.Profile(); prof.enable()
    t0 = time.time()
    for _ in range(20):
        env.step(2 * torch.rand(env.action_space.shape, device=obs.device) - 1)
    t_launch = time.time() - t0
    torch.cuda.synchronize()
    dt = time.time() - t0
    prof.disable()
    print(f"{n} envs: {1e3 * dt / 20:.2f} ms per step, host returns after {1e3 * t_launch / 20:.2f} ms per step")
    pstats.Stats(prof).sort_stats("cumulative").print_stats(45)
    pstats.Stats(prof).sort_stats("tottime").print_stats(25)
    sys.exit(0)
res = {}
for n in [int(x) for x in sys.argv[2:]] or [256, 1024]:
    t0 = time.time()
    env = gym.make("OpenCabinetDrawer-v1", num_envs=n)
    obs, _ = env.reset(seed=0)
    tb = time.time() - t0
    torch.manual_seed(0)
    for _ in range(3):
        env.step(2 * torch.rand(env.action_space.shape, device=obs.device) - 1)
    torch.cuda.synchronize()
    t0 = time.time()
    steps = 30
    for _ in range(steps):
        obs, rew, *_ = env.step(2 * torch.rand(env.action_space.shape, device=obs.device) - 1)
    torch.cuda.synchronize()
    dt = time.time() - t0
    res[n] = dict(build_s=round(tb, 1), env_steps_per_s=round(n * steps / dt, 1), ms_per_step=round(1e3 * dt / steps, 2),
                  groups=len(env.unwrapped.scene.px._groups), finite=bool(torch.isfinite(obs).all()))
    print(n, res[n], flush=True)
    env.close()
print(json.dumps({"workload": "OpenCabinetDrawer-v1 (reference task code over the shim), Fetch + synthetic cabinets, max drawers " + max_drawers, "results": res}))
