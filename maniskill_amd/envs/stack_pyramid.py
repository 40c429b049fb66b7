"""StackPyramid-v1 on the MI355X-native backend: the reference task (mani_skill/envs/tasks/tabletop/stack_pyramid.py:24-196).

PandaWristCam, table and three 4 cm cubes: put the red cube next to the green one and the blue one on top of both.  Reward modes
"none" / "sparse" only, as in the reference.  Host code is torch over the backend's buffers (graph-capturable).
"""
from __future__ import annotations

import numpy as np
import torch

from ..physx import SceneTemplate
from . import scene_builders as sb
from .pick_cube import PickCubeEnv, _quat_to_y_axis, compute_angle_between


class StackPyramidEnv(PickCubeEnv):
    state_actor_names = ("table-workspace", "cubeA", "cubeB", "cubeC")
    state_articulation_name = "panda_wristcam"
    max_episode_steps = 250
    max_reward = 1.0
    obs_dim = 64
    camera_eye, camera_target = (0.3, 0.0, 0.4), (-0.05, 0.0, 0.1)     # base_camera (:56-59)

    def __init__(self, *args, **kw):
        kw["fused"] = False
        kw.setdefault("reward_mode", "sparse")
        if kw["reward_mode"] not in ("sparse", "none"):
            raise NotImplementedError("StackPyramid-v1 supports the reward modes 'none' and 'sparse' (stack_pyramid.py:44)")
        super().__init__(*args, **kw)

    def _build_template(self, arm_stiffness=None):
        tpl = SceneTemplate()
        art = sb.add_panda(tpl, arm_stiffness=arm_stiffness, asset="panda_v3.json")
        table = sb.add_table_scene(tpl)
        a = sb.add_cube(tpl, "cubeA", 0.02, (0, 0, 0.2))
        b = sb.add_cube(tpl, "cubeB", 0.02, (1, 0, 0.2))
        c = sb.add_cube(tpl, "cubeC", 0.02, (-1, 0, 0.2))
        for body, rgb in ((a, (1.0, 0.0, 0.0)), (b, (0.0, 1.0, 0.0)), (c, (0.0, 0.0, 1.0))):
            tpl.set_body_color(body, rgb + (1.0,))
        for k, name in enumerate(tpl.body_names):
            if name.startswith("panda_") or name.startswith("camera_"):
                tpl.set_body_color(k, (0.9, 0.9, 0.9, 1.0))
        self._b_cubeB, self._b_cubeC = b, c
        return tpl, dict(art=art, table=table, cube=a, goal_site=b)

    def _hidden_bodies(self):
        return ()

    def _state_actor_bodies(self):
        return [self._b_table, self._b_cube, self._b_cubeB, self._b_cubeC]

    def _initialize_episode(self, env_idx, idx_np, off, f32):
        """:91-135: three placements from one UniformPlacementSampler over [-0.1, 0.1] x [-0.2, 0.2] (each new cube at least two
        half-diagonals from the earlier ones, 100 tries: samplers.py:52-73), yaw uniform in [0, 2 pi)."""
        b = len(idx_np)
        radius = float(np.linalg.norm([0.02, 0.02]))
        lo, rng = np.array([-0.1, -0.2]), np.array([0.2, 0.4])
        placed = [self._rng.uniform(idx_np, 2) * rng + lo]
        for _ in range(2):
            pos = np.zeros((b, 2))
            done = np.zeros(b, dtype=bool)
            for _try in range(100):
                cand = self._rng.uniform(idx_np, 2) * rng + lo
                ok = ~done
                for prev in placed:
                    ok &= np.linalg.norm(cand - prev, axis=1) > 2 * radius
                pos[ok] = cand[ok]
                done |= ok
                if done.all():
                    break
            placed.append(pos)
        yaw = self._rng.uniform(idx_np, 3) * (2 * np.pi)
        for k, bid in enumerate((self._b_cube, self._b_cubeB, self._b_cubeC)):
            xyz = np.concatenate([placed[k], np.full((b, 1), 0.02)], axis=1)
            q = np.zeros((b, 4)); q[:, 0] = np.cos(yaw[:, k] / 2); q[:, 3] = np.sin(yaw[:, k] / 2)
            self._rbd[env_idx, bid, :3] = f32(xyz) + off
            self._rbd[env_idx, bid, 3:7] = f32(q)
            self._rbd[env_idx, bid, 7:13] = 0.0

    # ---- task ------------------------------------------------------------------------------------------------------------
    def _is_grasping_body(self, body, min_force=0.5, max_angle=85):
        """Panda.is_grasping(object) (panda.py:237-269) for any of the three cubes."""
        if not hasattr(self, "_grasp_q"):   # one (finger, cube) impulse query pair per cube, created on first use
            px, f1, f2 = self.px, self._b_f1, self._b_f2
            self._grasp_q = {b: (px.gpu_create_contact_pair_impulse_query([(f1, b)]), px.gpu_create_contact_pair_impulse_query([(f2, b)]))
                             for b in (self._b_cube, self._b_cubeB, self._b_cubeC)}
        ql, qr = self._grasp_q[body]
        lf, rf = self.get_pairwise_contact_forces(ql), self.get_pairwise_contact_forces(qr)
        ldir = _quat_to_y_axis(self._rbd[:, self._b_f1, 3:7])
        rdir = -_quat_to_y_axis(self._rbd[:, self._b_f2, 3:7])
        lflag = (torch.linalg.norm(lf, dim=1) >= min_force) & (torch.rad2deg(compute_angle_between(ldir, lf)) <= max_angle)
        rflag = (torch.linalg.norm(rf, dim=1) >= min_force) & (torch.rad2deg(compute_angle_between(rdir, rf)) <= max_angle)
        return lflag & rflag

    def _static(self, body):
        r = self._rbd[:, body]
        return (r[:, 7:10].norm(dim=1) <= 1e-2) & (r[:, 10:13].norm(dim=1) <= 0.5)

    def evaluate(self):
        """:137-176."""
        self._fresh()
        A, B, C = (self._pose(b)[:, :3] for b in (self._b_cube, self._b_cubeB, self._b_cubeC))
        reach = float(np.linalg.norm([0.04, 0.04])) + 0.005

        def placed(offset, body, top):
            flag = torch.linalg.norm(offset[:, :2], dim=1) <= reach
            if top:
                flag = flag & (torch.abs(offset[:, 2]) > 0.02)
            return flag & self._static(body) & ~self._is_grasping_body(body)
        ab = placed(A - B, self._b_cube, False)
        cb = placed(B - C, self._b_cubeC, True)
        ca = placed(A - C, self._b_cubeC, True)
        return {"success": ab & cb & ca}

    def get_obs(self, info):
        """:178-194."""
        tcp = self.tcp_pose
        A, B, C = (self._pose(b) for b in (self._b_cube, self._b_cubeB, self._b_cubeC))
        t = tcp[:, :3]
        return torch.hstack([self.qpos, self.qvel, tcp, A, B, C, A[:, :3] - t, B[:, :3] - t, C[:, :3] - t,
                             B[:, :3] - A[:, :3], C[:, :3] - B[:, :3], C[:, :3] - A[:, :3]])

    def get_reward(self, obs, action, info):
        if self.reward_mode == "none":
            return torch.zeros(self.num_envs, device=self.device)
        return info["success"].float()          # sparse (sapien_env.py:1005-1012)
