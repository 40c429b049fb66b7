"""Fresh-interpreter helper: TriFingerRotateCubeLevel0-v1 (mani_skill/envs/tasks/rotate_cube.py:86-120) has a non-convex static arena wall
(`high_table_boundary.stl`, a conical ring around the fingers) and a robot whose three fingers keep a dozen speculative contacts among
themselves.  The cube has to stay on the table: the ring must not be one solid hull (the cube and the fingers live inside it), and an env
over its contact capacity must lose link-against-link contacts of the robot, not the cube's.
    python tests/ref_trifinger_scene.py <oracle|hip>  -> prints TRI {json}"""
import json
import sys

import ref_harness


def main():
    gym = ref_harness.setup(sys.argv[1])
    import torch
    env = gym.make("TriFingerRotateCubeLevel0-v1", num_envs=4, render_backend="none")
    env.reset(seed=0)
    u = env.unwrapped
    dev = u.device
    zs, vmax = [], 0.0
    for t in range(60):
        env.step(torch.as_tensor(env.action_space.sample(), device=dev))
        zs.append(u.obj.pose.p[:, 2].min().item())
        vmax = max(vmax, float(u.obj.linear_velocity.norm(dim=1).max()))
    px = u.scene.px
    wall = [s for c in px._components for s in getattr(c, "collision_shapes", []) if type(s).__name__ == "PhysxCollisionShapeTriangleMesh"]
    print("TRI " + json.dumps(dict(cube_z_min=min(zs), cube_z_last=zs[-1], cube_speed_max=vmax, half_size=float(u.size / 2),
                                    wall_pieces=len(wall[0]._hulls), wall_error=float(wall[0].decomposition_error))))


if __name__ == "__main__":
    main()
