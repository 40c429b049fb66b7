"""Fresh-interpreter helper: agent.set_control_mode after gpu_init (what mani_skill/examples/benchmarking/gpu_sim.py does for its fixed
trajectories) re-programs the joint drives of a finalised context (msk_set_drive after msk_finalize).
    python tests/ref_control_mode_switch.py <oracle|hip>  -> prints CMS {json}"""
import json
import sys

import ref_harness


def main():
    gym = ref_harness.setup(sys.argv[1])
    import torch
    env = gym.make("PickCube-v1", num_envs=3, render_backend="none", control_mode="pd_joint_delta_pos")
    env.reset(seed=0)
    u = env.unwrapped
    dev = u.device
    env.step(torch.zeros(env.action_space.shape, device=dev))
    u.agent.set_control_mode("pd_joint_pos")
    u.agent.controller.reset()
    target = torch.tensor([0.0, 0.68, 0.0, -1.9292649, 0.0, 2.627549, 0.7840855, 0.04], device=dev)
    for _ in range(40):
        env.step(target.repeat(3, 1))
    q = u.agent.robot.get_qpos()[:, :7].cpu()
    print("CMS " + json.dumps(dict(arm_error=float((q - target[:7].cpu()).abs().max()), control_mode=u.agent.control_mode)))


if __name__ == "__main__":
    main()
