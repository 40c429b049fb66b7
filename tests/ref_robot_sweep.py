"""Fresh-interpreter helper: robots of the reference's registry that no locally testable task uses, loaded by the reference's own agent
classes into Empty-v1 over the shim (URDF / SRDF parsing, mimic joints, package:// paths, capacities), reset and stepped with small
actions.    python tests/ref_robot_sweep.py <oracle|hip> [uid ...]  -> prints ROB {json}"""
import json
import sys

import ref_harness

ROBOTS = ["fetch", "xarm7_ability", "koch-v1.1", "floating_panda_gripper", "fixed_inspire_hand_right", "allegro_hand_left", "humanoid"]


def main():
    gym = ref_harness.setup(sys.argv[1])
    import torch
    res = {}
    for uid in sys.argv[2:] or ROBOTS:
        try:
            env = gym.make("Empty-v1", num_envs=2, robot_uids=uid, render_backend="none")
            env.reset(seed=0)
            dev = env.unwrapped.device
            for _ in range(5):
                a = env.action_space.sample()
                a = {k: 0.1 * torch.as_tensor(v, device=dev) for k, v in a.items()} if isinstance(a, dict) else 0.1 * torch.as_tensor(a, device=dev)
                env.step(a)
            px = env.unwrapped.scene.px
            ok = all(bool(torch.isfinite(t).all()) for t in (px.cuda_rigid_body_data.torch(), px.cuda_articulation_qpos.torch(), px.cuda_articulation_qvel.torch()))
            res[uid] = "ok" if ok else "non-finite state"
            env.close()
        except BaseException as ex:  # noqa: BLE001 -- the report is the point
            res[uid] = f"{type(ex).__name__}: {str(ex)[:160]}"
    print("ROB " + json.dumps(res))


if __name__ == "__main__":
    main()
