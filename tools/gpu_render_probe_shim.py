"""Development aid: where does a picture of the DROP-IN path go?  The reference's own PushT-v1 (mani_skill, unmodified) built over the sapien shim -- the template the
shim compiles from the reference's visual meshes (MSK_RENDER_TRI_BUDGET) -- and `take_picture` of its camera group timed with the render workgroup cut off after phase k
(MSK_RENDER_CUT on libmsk_prof.so, the cuts of tools/gpu_render_probe.py), planes only and without Color as the task plugin asks for them.
    python tools/gpu_render_probe_shim.py [env id] [envs]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 3 and sys.argv[3] == "child":
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import ref_harness
    gym = ref_harness.setup("hip")
    n = int(sys.argv[2])
    env = gym.make(sys.argv[1], num_envs=n, obs_mode="depth+segmentation")
    env.reset(seed=2022); torch.manual_seed(0)
    base = env.unwrapped
    for _ in range(5):
        env.step(2 * torch.rand(env.action_space.shape, device=base.device) - 1)
    g = next(iter(base.scene.camera_groups.values()))
    mode = os.environ.get("PROBE_OUTPUTS", "planes")
    if mode == "planes":
        g.set_outputs(False, color=False)
    elif mode == "planes+color":
        g.set_outputs(False, color=True)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    for _ in range(3): g.take_picture()
    torch.cuda.synchronize(); ev[0].record()
    for _ in range(10): g.take_picture()
    ev[1].record(); torch.cuda.synchronize()
    size = getattr(getattr(base.scene, "render_system_group", None), "simplification", None)
    print(f"{sys.argv[1]} x {n} [{mode}] budget {os.environ.get('MSK_RENDER_TRI_BUDGET', 'default')} cut {os.environ.get('MSK_RENDER_CUT', '0')} "
          f"lib {os.path.basename(os.environ.get('MSK_LIB', 'libmsk_physx.so'))}: {ev[0].elapsed_time(ev[1]) / 10 * 1e3:.1f} us per picture  template {size}", flush=True)
else:
    task = sys.argv[1] if len(sys.argv) > 1 else "PushT-v1"
    n = sys.argv[2] if len(sys.argv) > 2 else "4096"
    prof = os.path.join(ROOT, "maniskill_amd", "csrc", "libmsk_prof.so")
    run = lambda **e: subprocess.call([sys.executable, __file__, task, n, "child"], env=dict(os.environ, **e), stderr=subprocess.DEVNULL)
    for cut in (1, 2, 3, 4, 5, 6, 7, 8, 9, 0):
        run(MSK_RENDER_CUT=str(cut), MSK_LIB=prof)
    run(MSK_RENDER_CUT="0")
    run(MSK_RENDER_CUT="0", PROBE_OUTPUTS="planes+color")
    run(MSK_RENDER_CUT="0", PROBE_OUTPUTS="all")
