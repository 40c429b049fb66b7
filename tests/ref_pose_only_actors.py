"""Fresh-interpreter helper: kinematic actors without a collision shape beyond the engine's body capacity (the Draw tasks keep 300-1000
"dot" actors per env, mani_skill/envs/tasks/drawing/draw_triangle.py:193-223) live as pose-only rows behind the engine's rows of the
unified rigid-body buffer: poses written through Actor.set_pose stay where they were put, across apply / step / fetch and across a
partial reset of another env.
    python tests/ref_pose_only_actors.py <oracle|hip>  -> prints POA {json}"""
import json
import sys

import ref_harness


def main():
    gym = ref_harness.setup(sys.argv[1])
    import torch
    from mani_skill.utils.structs.pose import Pose
    env = gym.make("DrawTriangle-v1", num_envs=3, render_backend="none")
    env.reset(seed=0)
    u = env.unwrapped
    dev = u.device
    g = u.scene.px._groups[0]
    want = torch.tensor([[0.1, 0.2, 0.3], [0.4, 0.5, 0.6], [0.7, 0.8, 0.9]], device=dev)
    picks = (5, len(u.dots) - 1)                 # an engine-side kinematic body and a pose-only one
    for k in picks:
        u.dots[k].set_pose(Pose.create_from_pq(want))
    u.scene._gpu_apply_all()
    for _ in range(3):
        env.step(torch.as_tensor(env.action_space.sample(), device=dev))
    err = max(float((u.dots[k].pose.p - want).abs().max()) for k in picks)
    drawn = [u.dots[k].pose.p[0].cpu().tolist() for k in range(3)]      # the task itself moved dots 0..2 (brush above the canvas: parked at z = -DOT_THICKNESS)
    env.reset(seed=1, options=dict(env_idx=torch.tensor([1], device=dev)))
    after = u.dots[picks[1]].pose.p.cpu()
    print("POA " + json.dumps(dict(engine_bodies=int(g.nb), pose_only=int(g.npassive), rows=int(u.scene.px.cuda_rigid_body_data.torch().shape[0]),
                                    pose_error=err, drawn_z=[d[2] for d in drawn], kept_env0=float((after[0] - want[0].cpu()).abs().max()),
                                    reset_env1_z=float(after[1, 2]))))


if __name__ == "__main__":
    main()
