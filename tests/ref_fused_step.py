"""Runs inside a fresh interpreter (tests/test_fused_step.py): envs built by the REFERENCE's own code over the sapien shim, one stepped by the reference's
unmodified BaseEnv.step, its twin by maniskill_amd.fused_step.accelerate -- same seeds, same actions.  Prints one line ``FUSED {json}``.

    python tests/ref_fused_step.py <oracle|emu|hip> <case> [num_envs] [steps]
cases: cabinet (task plugin, several structural groups), cabinet_graph (hip only: the control step as one HIP graph), panda:<control_mode> (control level),
       graph:<env id> (hip: the reference's own task code behind the fused controller, captured), unsupported,
       graph_safe:<env id> (the op stream of two consecutive steps holds nothing a graph capture / replay gets wrong), speed (hip: env-steps/s of the forms)
"""
import json
import os
import subprocess
import sys
import time

import ref_harness

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _cabinet_assets(max_drawers="2"):
    ref = ref_harness.find_reference()
    assets = os.environ.get("MSK_SYNTH_ASSETS", "/tmp/ms_assets_synth_fused")
    meta = os.path.join(ref, "mani_skill", "assets", "partnet_mobility", "meta")
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "make_synthetic_partnet.py"), "--out", assets, "--max-drawers", max_drawers,
                           "--ids-from", os.path.join(meta, "info_cabinet_drawer_train.json"), "--placeholder-ids-from",
                           os.path.join(meta, "info_cabinet_door_train.json")], stdout=subprocess.DEVNULL)
    os.environ["MS_ASSET_DIR"] = assets


def _flat(x):
    """observations as one float tensor (camera modes hand out nested dicts)"""
    import torch
    if isinstance(x, dict):      # (by key: the reference's texture order follows a set's iteration order, sensors/camera.py:206)
        return torch.cat([_flat(x[k]) for k in sorted(x)], dim=-1) if x else torch.zeros(0)
    return x.reshape(x.shape[0], -1).float()


def _state(env):
    """env.get_state(); where the reference's own flatten fails for several sub-scenes (RotateValve / RotateSingleObjectInHand register one articulation per
    sub-scene) the simulator's buffers themselves"""
    import torch
    base = env.unwrapped
    try:
        return base.get_state()
    except Exception:      # noqa: BLE001
        px = base.scene.px
        raw = [px.cuda_rigid_body_data.torch().reshape(1, -1)]
        if len(base.scene.articulations):
            raw += [px.cuda_articulation_qpos.torch().reshape(1, -1), px.cuda_articulation_qvel.torch().reshape(1, -1)]
        return torch.cat(raw, dim=1).clone()


def _compare(gym, eid, n, steps, kw, acc_kw, atol=0.0, history=0):
    import torch
    from maniskill_amd.fused_step import accelerate
    a, b = gym.make(eid, num_envs=n, **kw), gym.make(eid, num_envs=n, **kw)
    dev = a.unwrapped.device
    acc = accelerate(b, **acc_kw)
    for _ in range(history):      # what a capture does to an env before its first reset: throw-away steps with a zero action
        b.unwrapped.step(torch.zeros(b.action_space.shape, device=dev))
    # ... and the reference's reset does not undo all of it (drive targets stay in the simulation until the next action; OpenCabinetDrawer's _initialize_episode
    # steps the physics once under them): the env that is compared against gets the same history
    for _ in range(getattr(acc, "throwaway_steps", 0) + history):
        a.unwrapped.step(torch.zeros(a.action_space.shape, device=dev))
    torch.manual_seed(7)
    oa, _ = a.reset(seed=3)
    torch.manual_seed(7)
    ob, _ = b.reset(seed=3)
    res = dict(level=acc.level, graph=acc.graph is not None, reset_equal=bool(torch.equal(_flat(oa), _flat(ob))), groups=len(getattr(a.unwrapped.scene.px, "_groups", [0])))
    g = torch.Generator().manual_seed(1)
    worst_obs = worst_rew = worst_state = 0.0
    flags = True
    trace = []
    for k in range(steps):
        act = 2 * torch.rand(a.action_space.shape, generator=g) - 1
        if k % 7 == 3:
            act = act * 3                       # outside the box: clipped
        act = act.to(dev)
        torch.manual_seed(1000 + k)      # (tasks that draw from torch's global generator inside a step -- SO100GraspCube's camera mount -- draw the same in both envs)
        ra = a.step(act)
        torch.manual_seed(1000 + k)
        rb = b.step(act)
        sa, sb = _state(a), _state(b)
        worst_state = max(worst_state, float((sa - sb).abs().max()))
        if len(trace) < 6 and float((sa - sb).abs().max()) > 0:      # where a mismatch starts: step, env, state column
            d = (sa - sb).abs()
            trace.append((k, int(d.max(dim=1).values.argmax()), int(d.max(dim=0).values.argmax()), float(d.max()), int((d.max(dim=1).values > 0).sum())))
        if _flat(ra[0]).numel():      # (a camera task built with render_backend="none" has no observation to compare)
            worst_obs = max(worst_obs, float((_flat(ra[0]) - _flat(rb[0])).abs().max()))
        assert ra[1].dtype == rb[1].dtype and type(ra[0]) is type(rb[0])
        worst_rew = max(worst_rew, float((ra[1].float() - rb[1].float()).abs().max()))
        flags = flags and bool(torch.equal(ra[2], rb[2])) and bool(torch.equal(ra[3], rb[3])) and bool(torch.equal(ra[4]["elapsed_steps"], rb[4]["elapsed_steps"]))
        if "success" in ra[4]:
            flags = flags and bool(torch.equal(ra[4]["success"], rb[4]["success"]))
    # a partial reset in the middle of the rollout, then on
    idx = torch.arange(0, n, 2, device=dev)
    torch.manual_seed(8); a.reset(seed=5, options=dict(env_idx=idx))
    torch.manual_seed(8); b.reset(seed=5, options=dict(env_idx=idx))
    for k in range(3):
        act = (2 * torch.rand(a.action_space.shape, generator=g) - 1).to(dev)
        torch.manual_seed(2000 + k)
        ra = a.step(act)
        torch.manual_seed(2000 + k)
        rb = b.step(act)
        if _flat(ra[0]).numel():
            worst_obs = max(worst_obs, float((_flat(ra[0]) - _flat(rb[0])).abs().max()))
        worst_state = max(worst_state, float((_state(a) - _state(b)).abs().max()))
    res.update(worst_obs=worst_obs, worst_rew=worst_rew, worst_state=worst_state, flags=flags, finite=bool(torch.isfinite(_flat(ra[0])).all() and torch.isfinite(ra[1]).all()), mismatch_trace=trace)
    acc.restore()
    res["restored"] = "step" not in b.unwrapped.__dict__ and "_step_action" not in b.unwrapped.__dict__
    return res


def main():
    backend, case = sys.argv[1], sys.argv[2]
    n = int(sys.argv[3]) if len(sys.argv) > 3 else 6
    steps = int(sys.argv[4]) if len(sys.argv) > 4 else 12
    if case.startswith("cabinet") or "OpenCabinet" in case or case == "speed":
        _cabinet_assets()
    gym = ref_harness.setup(backend)
    import torch
    from maniskill_amd.fused_step import Unsupported, accelerate
    if case == "cabinet":
        res = _compare(gym, "OpenCabinetDrawer-v1", n, steps, {}, {})
    elif case == "cabinet_history":        # the eager plugin on envs that have stepped before their first reset, like a captured one (both: see _compare)
        res = _compare(gym, "OpenCabinetDrawer-v1", n, steps, {}, {}, history=3)
    elif case == "cabinet_graph":
        res = _compare(gym, "OpenCabinetDrawer-v1", n, steps, {}, dict(graph=True))
    elif case.startswith(("kernel:", "kernel_graph:")):      # kernel[_graph]:<env id>[:obs mode]: the plugins on the fused task kernels (emu | hip) against the reference's own step
        parts = case.split(":")
        kw = dict(obs_mode=parts[2]) if len(parts) > 2 else {}
        if kw.get("obs_mode", "state") in ("state", "state_dict") and parts[1] != "PushT-v1":      # (PushT reads the render shapes it has just attached)
            kw["render_backend"] = "none"
        res = _compare(gym, parts[1], n, steps, kw, dict(graph=True) if parts[0] == "kernel_graph" else {})
    elif case == "kernel_hidden":             # an actor the reference 'hid' (moved 99999 m away, its pose remembered: structs/actor.py:176-201): the kernels step aside for the reference's own step
        a, b = gym.make("PickCube-v1", num_envs=n, render_backend="none"), gym.make("PickCube-v1", num_envs=n, render_backend="none")
        acc = accelerate(b)
        a.reset(seed=3); b.reset(seed=3)
        g = torch.Generator().manual_seed(4)
        worst, phases = 0.0, []
        for phase in ("shown", "hidden", "shown"):
            for e in (a, b):
                (e.unwrapped.goal_site.hide_visual if phase == "hidden" else e.unwrapped.goal_site.show_visual)()
            phases.append(bool(acc.plugin._usable()))
            for _ in range(steps):
                act = 2 * torch.rand(a.action_space.shape, generator=g) - 1
                ra, rb = a.step(act), b.step(act)
                worst = max(worst, float((ra[0] - rb[0]).abs().max()), float((ra[1] - rb[1]).abs().max()), float((a.unwrapped.get_state() - b.unwrapped.get_state()).abs().max()))
        res = dict(level=acc.level, worst=worst, usable=phases)
    elif case.startswith("graph:"):           # the reference's OWN task code behind the fused controller, captured (tasks whose step is graph-safe)
        res = _compare(gym, case.split(":", 1)[1], n, steps, dict(render_backend="none"), dict(graph=True, task=False))
    elif case in ("dry_rgbd", "graph_rgbd"):  # camera observations: the plugin steps aside, the reference's own step (take_picture, texture transforms) is what is captured
        res = _compare(gym, "PickCube-v1", n, steps, dict(obs_mode="rgbd"), dict(graph="dry" if case == "dry_rgbd" else True))
    elif case in ("dry_pusht", "dry_pusht_cam", "graph_pusht"):    # PushT-v1 (BASELINE config 3's task): its intersection 'renderer' patched mask-free, the rest of its own step captured
        kw = dict(obs_mode="state") if case == "dry_pusht" else dict(obs_mode="rgb+depth+segmentation")
        res = _compare(gym, "PushT-v1", n, steps, kw, dict(graph=True if case == "graph_pusht" else "dry", task=False))
    elif case.startswith("dry:"):             # ... the same path without the capture (CPU checker): the results have to be the reference's
        res = _compare(gym, case.split(":", 1)[1], n, steps, dict(render_backend="none"), dict(graph="dry", task=False))
    elif case.startswith("panda:"):
        res = _compare(gym, "PickCube-v1", n, steps, dict(render_backend="none", control_mode=case.split(":")[1]), dict(task=False))
    elif case == "pickcube_grasp":                # the grasp branch of is_grasping: the cube put between the open fingers, the gripper closed on it, the arm lifted
        from mani_skill.utils.structs.pose import Pose
        from maniskill_amd.fused_step import accelerate
        a, b = gym.make("PickCube-v1", num_envs=n, render_backend="none"), gym.make("PickCube-v1", num_envs=n, render_backend="none")
        acc = accelerate(b)
        a.reset(seed=3); b.reset(seed=3)
        for e in (a, b):
            base = e.unwrapped
            base.cube.set_pose(Pose.create_from_pq(p=base.agent.tcp.pose.p.clone()))
            base.scene._gpu_apply_all(); base.scene.px.gpu_update_articulation_kinematics(); base.scene._gpu_fetch_all()
        worst, grasped, rewards = 0.0, 0, []
        for k in range(steps):
            act = torch.zeros(a.action_space.shape)
            act[:, -1] = -1.0
            if k >= 4:
                act[:, 1] = -0.3          # shoulder back: the hand goes up with the cube
            ra, rb = a.step(act), b.step(act)
            worst = max(worst, float((ra[0] - rb[0]).abs().max()), float((ra[1] - rb[1]).abs().max()), float((a.unwrapped.get_state() - b.unwrapped.get_state()).abs().max()))
            assert torch.equal(ra[4]["is_grasped"], rb[4]["is_grasped"]) and torch.equal(ra[4]["is_robot_static"], rb[4]["is_robot_static"])
            grasped = max(grasped, int(ra[4]["is_grasped"].sum()))
            rewards.append(round(float(rb[1].max()), 4))
        res = dict(level=acc.level, worst=worst, grasped=grasped, max_reward=max(rewards))
    elif case.startswith("pickcube"):             # pickcube[:reward_mode]: the task plugin
        kw = dict(render_backend="none")
        if ":" in case:
            kw["reward_mode"] = case.split(":")[1]
        # (the torch restatement: on a library with the fused task kernels it is the second choice, asked for by task="torch")
        res = _compare(gym, "PickCube-v1", n, steps, kw, dict(graph=True, task="torch") if case.startswith("pickcube_graph") else dict(task="torch"))
    elif case == "changing_constant":
        from maniskill_amd.fused_step import DeviceConstants
        mode = DeviceConstants("cpu")

        def task_code(k):
            return torch.tensor([float(k), 0.0], device="cpu")
        def step(k):                # one `with` = one step (the k-th evaluation of a call path INSIDE a step is a constant of its own: loops)
            with mode:
                return task_code(k)
        outs, raised = [], False
        for k in (1.0, 1.0, 2.0):   # (one call line: the call path is part of the key -- in the product it ends at the step's entry in fused_step.py)
            try:
                outs.append(step(k))
            except Unsupported:
                raised = True
        a, b = outs[0], outs[1]
        raised = raised and len(outs) == 2
        with mode:                  # two callers of one helper line inside one step: two constants, not one that changes (Pose.create -> common.to_tensor)
            c1, c2 = task_code(5.0), task_code(6.0)
        raised = raised and float(c1[0]) == 5.0 and float(c2[0]) == 6.0
        res = dict(raised=raised, served=mode.served, clones=a.data_ptr() != b.data_ptr(), equal=bool(torch.equal(a, b)))
    elif case == "auto":
        import warnings
        import maniskill_amd.shim as shim
        orig = shim.auto_accelerate("task")
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            a = gym.make("PickCube-v1", num_envs=2, render_backend="none")
            b = gym.make("PickCube-v1", num_envs=2, render_backend="none", control_mode="pd_ee_delta_pose")
        gym.make = orig
        c = gym.make("PickCube-v1", num_envs=2, render_backend="none")
        res = dict(accelerated=getattr(a.unwrapped, "_msk_accelerated", None) is not None and a.unwrapped._msk_accelerated.level == "task",
                   left_alone="_step_action" not in b.unwrapped.__dict__, warned=any("not accelerated" in str(x.message) for x in w),
                   undone="_step_action" not in c.unwrapped.__dict__)
        for e in (a, c):          # (b: the reference's EE controller takes its CPU route on the checker's cpu tensors; nothing of this test)
            e.reset(seed=0)
            e.step(torch.as_tensor(e.action_space.sample()))
    elif case == "reconfigure":
        # a reconfigured env has a new scene and a new px: the fused step rebuilds its index tensors at the next step and goes on with the reference's bits
        a, b = gym.make("PickCube-v1", num_envs=n, render_backend="none"), gym.make("PickCube-v1", num_envs=n, render_backend="none")
        acc = accelerate(b)
        g = torch.Generator().manual_seed(2)
        worst = 0.0
        for rnd in range(2):
            opts = dict(reconfigure=True) if rnd else {}
            a.reset(seed=4 + rnd, options=opts); b.reset(seed=4 + rnd, options=opts)
            for _ in range(4):
                act = 2 * torch.rand(a.action_space.shape, generator=g) - 1
                ra, rb = a.step(act), b.step(act)
                worst = max(worst, float((ra[0] - rb[0]).abs().max()), float((a.unwrapped.get_state() - b.unwrapped.get_state()).abs().max()))
        # ... and another control mode (a new controller object) is picked up the same way
        for e in (a, b):
            e.unwrapped.agent.set_control_mode("pd_joint_pos")
            e.unwrapped.agent.controller.reset()
        for _ in range(3):
            act = 0.5 * (2 * torch.rand(n, 8, generator=g) - 1)
            ra, rb = a.step(act), b.step(act)
            worst = max(worst, float((ra[0] - rb[0]).abs().max()), float((a.unwrapped.get_state() - b.unwrapped.get_state()).abs().max()))
        res = dict(worst=worst, rebuilds=acc.rebuilds, level=acc.level, same_scene=acc.scene is b.unwrapped.scene)
    elif case == "plugin_refused":            # PickCube with camera observations: the plugin (state observations only) steps aside, the fused controller stays
        a, b = gym.make("PickCube-v1", num_envs=2, obs_mode="rgbd"), gym.make("PickCube-v1", num_envs=2, obs_mode="rgbd")
        acc = accelerate(b)
        a.reset(seed=1); b.reset(seed=1)
        act = torch.as_tensor(a.action_space.sample())
        ra, rb = a.step(act), b.step(act)
        same = bool(torch.equal(ra[0]["sensor_data"]["base_camera"]["rgb"], rb[0]["sensor_data"]["base_camera"]["rgb"])) and bool(torch.equal(ra[1], rb[1]))
        res = dict(level=acc.level, refused=acc.plugin_refused, same=same)
    elif case == "not_verified":              # a task whose own step keeps state in Python (the number of dots drawn so far picks this step's actor): no graph
        env = gym.make("TableTopFreeDraw-v1", num_envs=2, render_backend="none")
        try:
            accelerate(env, graph=True)
            res = dict(raised=False)
        except Unsupported as e:
            res = dict(raised=True, message=str(e)[:400], untouched="_step_action" not in env.unwrapped.__dict__ and "step" not in env.unwrapped.__dict__)
    elif case == "unsupported":
        env = gym.make("PickCube-v1", num_envs=2, render_backend="none", control_mode="pd_ee_delta_pose")
        try:
            accelerate(env)
            res = dict(raised=False)
        except Unsupported as e:
            res = dict(raised=True, message=str(e), untouched="_step_action" not in env.unwrapped.__dict__)
    elif case.startswith("graph_safe:"):
        # What a HIP-graph replay of the step needs, checked on the op stream (no GPU needed): (1) nothing that synchronises, (2) no state that travels
        # from one step to the next through a tensor the earlier step ALLOCATED (a replay re-reads the memory that was current at capture time; state has to
        # live in tensors that persist and are updated in place)
        eid = case.split(":", 1)[1]
        kw = {} if eid.startswith("OpenCabinet") or eid == "PushT-v1" else dict(render_backend="none")      # (PushT reads the render shapes it has just attached)
        env = gym.make(eid, num_envs=n, **kw)
        env.reset(seed=0)
        acc = accelerate(env, graph="watch")          # (a task plugin where there is one -- then no verdict is needed --, else the reference's own step under DeviceConstants, watched)
        v = acc.safety or dict(sync=[], flow=[], host_data=[])
        c = getattr(acc, "constants", None)
        res = dict(level=acc.level, sync=v["sync"], flow=v["flow"], host_data=v["host_data"], constants_served=getattr(c, "served", 0), rewritten=getattr(c, "rewritten", 0),
                   masked=getattr(c, "masked", 0))
    elif case == "speed":
        out = {}
        for form, kw in (("reference", None), ("control", dict(task=False)), ("task", {}), ("graph", dict(graph=True))):
            env = gym.make("OpenCabinetDrawer-v1", num_envs=n)
            if kw is not None:
                accelerate(env, **kw)
            env.reset(seed=0)
            dev = env.unwrapped.device
            with torch.inference_mode(form != "graph"):
                for _ in range(5):
                    env.step(2 * torch.rand(env.action_space.shape, device=dev) - 1)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(steps):
                    env.step(2 * torch.rand(env.action_space.shape, device=dev) - 1)
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
            out[form] = dict(env_steps_per_s=round(n * steps / dt, 1), ms_per_step=round(dt / steps * 1e3, 3))
            env.close()
        res = dict(num_envs=n, steps=steps, forms=out)
    else:
        raise SystemExit(f"unknown case {case}")
    print("FUSED " + json.dumps(res))


if __name__ == "__main__":
    main()
