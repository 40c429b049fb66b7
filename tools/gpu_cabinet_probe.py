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
import numpy as np  # noqa: E402
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

if mode == "parity_physics":
    # physics only: the oracle side is fed the drive targets the HIP side's (reference, torch-on-GPU) controllers produced, so that the
    # comparison does not see torch's CPU / GPU transcendentals differ by an ulp inside the reference's controller code:
    #   python tools/gpu_cabinet_probe.py parity_physics N STEPS
    N, STEPS = int(sys.argv[2]), int(sys.argv[3])
    if len(sys.argv) > 4 and sys.argv[4] == "child":
        gym = ref_harness.setup("oracle")
        env = gym.make("OpenCabinetDrawer-v1", num_envs=N, render_backend="gpu")
        env.reset(seed=0)
        blob = torch.load("/tmp/cab_phys_in.pt")
        base = env.unwrapped
        px = base.scene.px
        base.set_state(blob["state"])
        out, cc = [], []
        for k in range(STEPS):
            px.cuda_articulation_target_qpos.torch()[:] = blob["tq"][k]
            px.cuda_articulation_target_qvel.torch()[:] = blob["tv"][k]
            px.gpu_apply_articulation_target_position(); px.gpu_apply_articulation_target_velocity()
            for _ in range(base._sim_steps_per_control):
                base.scene.step()
            base._after_control_step()      # the task moves its goal marker (a kinematic actor without collision) to the handle
            base.scene._gpu_fetch_all()
            out.append(base.get_state().clone())
            cc.append(np.concatenate([grp.engine.get_env_contact_counts() for grp in px._groups]))
        torch.save(dict(states=torch.stack(out), groups=len(px._groups), cc=torch.from_numpy(np.stack(cc)), genvs=[[int(e) for e in grp.envs] for grp in px._groups]), "/tmp/cab_phys_out.pt")
        sys.exit(0)
    gym = ref_harness.setup("hip")
    env = gym.make("OpenCabinetDrawer-v1", num_envs=N)
    env.reset(seed=0)
    base = env.unwrapped
    px = base.scene.px
    g = torch.Generator().manual_seed(3)
    s0 = base.get_state().clone()
    base.set_state(s0)
    hip_states, tq, tv, hcc = [], [], [], []
    for k in range(STEPS):
        env.step((2 * torch.rand(N, 13, generator=g) - 1).cuda())
        hcc.append(np.concatenate([grp.engine.get_env_contact_counts() for grp in px._groups]))
        px.gpu_fetch_articulation_target_qpos(); px.gpu_fetch_articulation_target_qvel()
        tq.append(px.cuda_articulation_target_qpos.torch().cpu().clone()); tv.append(px.cuda_articulation_target_qvel.torch().cpu().clone())
        hip_states.append(base.get_state().cpu().clone())
    torch.save(dict(state=s0.cpu(), tq=torch.stack(tq), tv=torch.stack(tv)), "/tmp/cab_phys_in.pt")
    flags = sorted({int(grp.engine.get_overflow()) for grp in px._groups})
    subprocess.check_call([sys.executable, __file__, "parity_physics", str(N), str(STEPS), "child"])
    ref_out = torch.load("/tmp/cab_phys_out.pt")
    worst, nbit, told = 0.0, 0, False
    order = [e for ge in ref_out["genvs"] for e in ge]      # env of every entry of the per-group contact-count vectors
    # the goal marker's pose is computed by the reference's own torch code (on the GPU on one side, on the CPU on the other: an ulp apart) and
    # touches nothing: its 13 columns are left out of the comparison
    names = list(base.get_state_dict()["actors"].keys())
    keep = torch.ones(hip_states[0].shape[1], dtype=torch.bool)
    for i, nm in enumerate(names):
        if "goal" in nm:
            keep[13 * i:13 * i + 13] = False
    for k in range(STEPS):
        s, r = hip_states[k][:, keep], ref_out["states"][k][:, keep]
        assert torch.isfinite(s).all() and torch.isfinite(r).all(), k
        worst = max(worst, float(((s - r).abs() / (1 + r.abs())).max()))
        nbit += int(torch.equal(s, r))
        if not torch.equal(s, r) and not told:
            told = True
            bad = (s != r).any(dim=1).nonzero().flatten().tolist()
            dcc = (hcc[k] != ref_out["cc"][k].numpy()).nonzero()[0].tolist()
            print(f"first step not bit-equal: {k}; envs {bad[:8]}; columns of the first: {(s[bad[0]] != r[bad[0]]).nonzero().flatten().tolist()[:16]}; max abs diff "
                  f"{float((s - r).abs().max()):.3e}; contact counts differ at entries {dcc[:8]} = envs {[order[i] for i in dcc[:8]]}: hip {hcc[k][dcc[:8]].tolist()} oracle "
                  f"{ref_out['cc'][k].numpy()[dcc[:8]].tolist()}; hip contacts of env {bad[0]}: {int(hcc[k][order.index(bad[0])])}, step before: "
                  f"{int(hcc[k - 1][order.index(bad[0])]) if k else -1} / oracle {int(ref_out['cc'][k][order.index(bad[0])])}, {int(ref_out['cc'][k - 1][order.index(bad[0])]) if k else -1}")
    print(json.dumps({"parity": f"OpenCabinetDrawer-v1 physics only (drive targets handed over), {N} envs x {STEPS} steps HIP vs oracle", "max_rel_err": worst,
                      "bit_equal_steps": nbit, "groups": len(px._groups), "flags": flags}))
    sys.exit(0 if worst < 1e-4 else 1)

if mode == "parity_scale":
    # HIP at the config's scale (N sub-scenes, every structural group), the oracle on the first M of them, STEPS control steps from the
    # HIP run's post-reset state:  python tools/gpu_cabinet_probe.py parity_scale N M STEPS
    N, M, STEPS = (int(x) for x in sys.argv[2:5])
    if len(sys.argv) > 5 and sys.argv[5] == "child":
        gym = ref_harness.setup("oracle")
        env = gym.make("OpenCabinetDrawer-v1", num_envs=M, render_backend="gpu")
        env.reset(seed=0)
        blob = torch.load("/tmp/cab_scale_in.pt")
        assert env.unwrapped.get_state().shape == blob["state"].shape, (env.unwrapped.get_state().shape, blob["state"].shape)
        env.unwrapped.set_state(blob["state"])
        out = []
        for k in range(STEPS):
            env.step(blob["actions"][k])
            out.append(env.unwrapped.get_state().clone())
        torch.save(dict(states=torch.stack(out), groups=len(env.unwrapped.scene.px._groups)), "/tmp/cab_scale_out.pt")
        sys.exit(0)
    gym = ref_harness.setup("hip")
    env = gym.make("OpenCabinetDrawer-v1", num_envs=N)
    env.reset(seed=0)
    g = torch.Generator().manual_seed(3)
    actions = 2 * torch.rand(STEPS, N, 13, generator=g) - 1
    s0 = env.unwrapped.get_state().clone()
    torch.save(dict(state=s0[:M].cpu(), actions=actions[:, :M].clone()), "/tmp/cab_scale_in.pt")
    env.unwrapped.set_state(s0)
    hip_states = []
    for k in range(STEPS):
        env.step(actions[k].cuda())
        hip_states.append(env.unwrapped.get_state()[:M].cpu().clone())
    ngroups = len(env.unwrapped.scene.px._groups)
    flags = sorted({int(grp.engine.get_overflow()) for grp in env.unwrapped.scene.px._groups})
    subprocess.check_call([sys.executable, __file__, "parity_scale", str(N), str(M), str(STEPS), "child"])
    ref_out = torch.load("/tmp/cab_scale_out.pt")
    worst, first_bad = 0.0, None
    for k in range(STEPS):
        s, r = hip_states[k], ref_out["states"][k]
        assert torch.isfinite(s).all() and torch.isfinite(r).all(), k
        err = float(((s - r).abs() / (1 + r.abs())).max())
        if err > 1e-4 and first_bad is None:
            rel = (s - r).abs() / (1 + r.abs())
            e_bad = int(rel.max(dim=1)[0].argmax())
            cols = (rel[e_bad] > 1e-4).nonzero().flatten().tolist()
            first_bad = (k, err, "env", e_bad, "state columns", cols[:12], "hip", [round(float(s[e_bad, c]), 5) for c in cols[:6]],
                         "oracle", [round(float(r[e_bad, c]), 5) for c in cols[:6]], "envs off", int((rel.max(dim=1)[0] > 1e-4).sum()))
        worst = max(worst, err)
        if k < 6 or k % 10 == 0:
            ne = (s != r)
            print(f"step {k}: entries not bit-equal {int(ne.sum())} of {ne.numel()} in {int(ne.any(dim=1).sum())} envs, max abs diff {float((s - r).abs().max()):.3e}")
    print("first step above 1e-4:", first_bad)
    print(json.dumps({"parity": f"OpenCabinetDrawer-v1 HIP {N} envs vs oracle on the first {M}, {STEPS} steps", "max_rel_err": worst, "groups": ngroups,
                      "oracle_groups": ref_out["groups"], "flags": flags}))
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
