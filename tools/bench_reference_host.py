#!/usr/bin/env python3
"""env-steps/s of the REFERENCE'S OWN host Python (mani_skill's BaseEnv, controllers, structs, task code, unmodified) on this
backend through the sapien shim -- the drop-in path, next to bench.py's number for the hand-written fused host.  Needs a reference
build (a checkout, or oracle/_ref/maniskill from oracle/build_ref.py); prints one JSON line.

    python tools/bench_reference_host.py [--env PickCube-v1] [--envs 4096] [--steps 200] [--accelerate control|task|graph]

--accelerate: maniskill_amd.fused_step.accelerate(env) first -- "control": the fused controller under the reference's own task code; "task": the task plugin
where one exists (OpenCabinetDrawer-v1); "graph": the control step as one HIP graph replay.  OpenCabinetDrawer-v1 needs PartNet-Mobility cabinets (a download):
--synthetic-partnet writes tools/make_synthetic_partnet.py's substitutes first.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
import ref_harness  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--env", default="PickCube-v1")
    ap.add_argument("--envs", type=int, default=4096)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--obs-mode", default="state")
    ap.add_argument("--accelerate", default="none", choices=["none", "control", "task", "graph"])
    ap.add_argument("--synthetic-partnet", type=int, default=0, metavar="MAX_DRAWERS", help="write synthetic cabinets with up to this many drawers and point MS_ASSET_DIR at them")
    ap.add_argument("--cprofile", type=int, default=0, help="print the N most expensive host functions (cumulative) of the timed loop to stderr")
    a = ap.parse_args()
    if ref_harness.find_reference() is None:
        print(json.dumps({"error": "no reference build present"}))
        return
    if a.synthetic_partnet:
        import subprocess
        assets = "/tmp/ms_assets_synth_bench"
        meta = os.path.join(ref_harness.find_reference(), "mani_skill", "assets", "partnet_mobility", "meta")
        subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "make_synthetic_partnet.py"), "--out", assets, "--max-drawers", str(a.synthetic_partnet),
                               "--ids-from", os.path.join(meta, "info_cabinet_drawer_train.json"), "--placeholder-ids-from",
                               os.path.join(meta, "info_cabinet_door_train.json")], stdout=subprocess.DEVNULL)
        os.environ["MS_ASSET_DIR"] = assets
    gym = ref_harness.setup(os.environ.get("MSK_REF_BACKEND", "hip"))      # ("oracle": a smoke run of this script without a GPU)
    t0 = time.perf_counter()
    kw = dict(render_backend="none") if (a.obs_mode == "state" and not a.env.startswith(("OpenCabinet", "PushT"))) else {}      # (the cabinet and PushT tasks read render shapes)
    env = gym.make(a.env, num_envs=a.envs, obs_mode=a.obs_mode, **kw)
    level = "none"
    if a.accelerate != "none":
        from maniskill_amd.fused_step import accelerate
        level = accelerate(env, graph=a.accelerate == "graph", task=a.accelerate != "control").level + ("+graph" if a.accelerate == "graph" else "")
    obs, _ = env.reset(seed=2022)
    build_s = time.perf_counter() - t0
    dev = env.unwrapped.device
    sync = torch.cuda.synchronize if dev.type == "cuda" else (lambda: None)
    torch.manual_seed(0)
    with torch.inference_mode(a.accelerate != "graph"):
        for _ in range(5):
            env.step(2 * torch.rand(env.action_space.shape, device=dev) - 1)
        sync()
        prof = None
        if a.cprofile:
            import cProfile
            prof = cProfile.Profile()
            prof.enable()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            env.step(2 * torch.rand(env.action_space.shape, device=dev) - 1)
        sync()
        dt = time.perf_counter() - t0
        if prof is not None:
            import pstats
            prof.disable()
            pstats.Stats(prof, stream=sys.stderr).sort_stats("cumulative").print_stats(a.cprofile)
            pstats.Stats(prof, stream=sys.stderr).sort_stats("tottime").print_stats(a.cprofile // 2)
    print(json.dumps({"metric": f"env steps/sec, {a.envs} parallel {a.env} envs, reference host Python over the sapien shim", "value": a.envs * a.steps / dt,
                      "unit": "env-steps/s", "ms_per_step": dt / a.steps * 1e3, "steps": a.steps, "obs_mode": a.obs_mode, "build_s": round(build_s, 1), "accelerate": level,
                      "host": "mani_skill (unmodified): BaseEnv.step, controllers, structs, task evaluate / obs / reward as eager torch ops" if level == "none" else
                              "mani_skill builds, resets and owns the env; maniskill_amd.fused_step runs its control step (" + level + ")",
                      "backend": "libmsk_physx.so (HIP, gfx950) through maniskill_amd/shim/sapien"}))


if __name__ == "__main__":
    main()
