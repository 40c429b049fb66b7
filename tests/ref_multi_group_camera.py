"""Fresh-interpreter helper of tests/test_reference_conformance.py: RotateValveLevel1-v1 builds a different valve per sub-scene, so the scene runs
as several structural groups -- each with its own rasteriser template and camera; the textures must come back in sub-scene order.
    python tests/ref_multi_group_camera.py <oracle|hip>   -> prints MGC {json}"""
import json
import sys

import ref_harness


def main():
    gym = ref_harness.setup(sys.argv[1])
    import torch
    n = 6
    env = gym.make("RotateValveLevel1-v1", num_envs=n, obs_mode="rgb+depth+segmentation")
    obs, _ = env.reset(seed=0)
    dev = env.unwrapped.device
    obs, *_ = env.step(torch.zeros(env.action_space.shape, device=dev))
    u = env.unwrapped
    px = u.scene.px
    cam = list(obs["sensor_data"].keys())[0]

    def depth_now():
        o = u.get_obs()
        return o["sensor_data"][cam]["depth"].float().cpu().clone(), o["sensor_data"][cam]

    base, d = depth_now()
    # move the valve of ONE sub-scene out of view: exactly that sub-scene's picture must change
    changed_only_k = []
    for k in range(n):
        pose = u.valve.root_pose
        p = pose.p.clone(); q = pose.q.clone()
        p[k, 2] += 5.0
        from mani_skill.utils.structs.pose import Pose
        u.valve.set_root_pose(Pose.create_from_pq(p, q))
        u.scene._gpu_apply_all(); px.gpu_update_articulation_kinematics(); u.scene._gpu_fetch_all()
        now, _ = depth_now()
        delta = (now - base).abs().flatten(1).max(1).values
        changed_only_k.append(bool(delta[k] > 0 and (delta[torch.arange(n) != k] == 0).all()))
        p[k, 2] -= 5.0
        u.valve.set_root_pose(Pose.create_from_pq(p, q))
        u.scene._gpu_apply_all(); px.gpu_update_articulation_kinematics(); u.scene._gpu_fetch_all()
    back, _ = depth_now()
    print("MGC " + json.dumps(dict(shapes={k: list(v.shape) for k, v in d.items()}, groups=len(px._groups), covered=[float((base[i] > 0).float().mean()) for i in range(n)],
                                  changed_only_k=changed_only_k, restored=bool((back - base).abs().max() <= 2))))


if __name__ == "__main__":
    main()
