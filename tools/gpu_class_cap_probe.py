"""On the GPU box: does the packed solver class (four envs per wavefront, up to 16 blocks) hold envs it should hand to the one-env-per-wavefront classes?  k_csolve is as long as its
slowest wavefront -- a packed one whose four envs are swept as far as the largest of them (profiles/r06_launch_position_probe.log: mean wave 29 us, slowest 55-59).  Same rollout
(seed, actions) per capacity of class 0 (msk_set_solver_classes): rate over 1000 steps and over the last 200, the classes' populations at the end.
    python tools/gpu_class_cap_probe.py [env=PickCube|Peg] [envs=4096] [caps=16,12,10,8,6]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from maniskill_amd.envs.pick_cube import PickCubeEnv
from maniskill_amd.envs.peg_insertion_side import PegInsertionSideEnv
name = sys.argv[1] if len(sys.argv) > 1 else "PickCube"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
caps = [int(c) for c in (sys.argv[3] if len(sys.argv) > 3 else "16,12,10,8,6").split(",")]
env = (PegInsertionSideEnv if name == "Peg" else PickCubeEnv)(num_envs=n, device="cuda:0")
adim = env.action_dim
env.enable_step_graph()
for rep in range(2):
    for c0 in caps:
        env.px.set_solver_classes([c0, 20, 32])
        env.reset(seed=2022)
        torch.manual_seed(0)
        for _ in range(20):
            env.step(2 * torch.rand(n, adim, device="cuda:0") - 1)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for k in range(1000):
            if k == 800:
                torch.cuda.synchronize(); t8 = time.perf_counter()
            env.step(2 * torch.rand(n, adim, device="cuda:0") - 1)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        print(f"{name} {n} envs, class 0 up to {c0:2d} blocks: {n * 1000 / (t1 - t0) / 1e6:.3f} M over 1000 steps ({(t1 - t0):.3f} ms/step), last 200: {n * 200 / (t1 - t8) / 1e6:.3f} M; "
              f"classes at the end {list(env.px.get_solver_class_counts())}", flush=True)
