"""Fresh-interpreter helper: robots of the reference's registry that no locally testable task uses, loaded by the reference's own agent
classes into Empty-v1 over the shim (URDF / SRDF parsing, mimic joints, package:// paths, capacities), reset and stepped with small
actions.    python tests/ref_robot_sweep.py <oracle|hip> [--random STEPS] [uid ...]  -> prints ROB {json}
--random STEPS: full-scale samples of the action space for STEPS control steps (the floating hands: saturated drives on light links, the
round-2 blow-up) and the largest joint speed seen is reported with the verdict."""
import json
import sys

import ref_harness

ROBOTS = ["fetch", "xarm7_ability", "koch-v1.1", "floating_panda_gripper", "fixed_inspire_hand_right", "allegro_hand_left", "humanoid"]


def main():
    gym = ref_harness.setup(sys.argv[1])
    import torch
    res = {}
    args = sys.argv[2:]
    scale, steps = 0.1, 5
    if args and args[0] == "--random":
        scale, steps = 1.0, int(args[1])
        args = args[2:]
    for uid in args or ROBOTS:
        try:
            env = gym.make("Empty-v1", num_envs=2, robot_uids=uid, render_backend="none")
            env.reset(seed=0)
            dev = env.unwrapped.device
            px = env.unwrapped.scene.px
            ok, fastest = True, 0.0
            for _ in range(steps):
                a = env.action_space.sample()
                a = {k: scale * torch.as_tensor(v, device=dev) for k, v in a.items()} if isinstance(a, dict) else scale * torch.as_tensor(a, device=dev)
                env.step(a)
                ok = ok and all(bool(torch.isfinite(t).all()) for t in (px.cuda_rigid_body_data.torch(), px.cuda_articulation_qpos.torch(), px.cuda_articulation_qvel.torch()))
                if not ok:
                    break
                fastest = max(fastest, float(px.cuda_articulation_qvel.torch().abs().max()))
            res[uid] = ("ok" if scale < 1.0 else f"ok (max |qvel| {fastest:.1f})") if ok else "non-finite state"
            env.close()
        except BaseException as ex:  # noqa: BLE001 -- the report is the point
            res[uid] = f"{type(ex).__name__}: {str(ex)[:160]}"
    print("ROB " + json.dumps(res))


if __name__ == "__main__":
    main()
