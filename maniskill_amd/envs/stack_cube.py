"""StackCube-v1 on the MI355X-native backend: the reference task (mani_skill/envs/tasks/tabletop/stack_cube.py:23-200).

Scene: Panda, table, two 4 cm cubes (cubeA red: the one to pick; cubeB green: the base).  Success = cubeA on cubeB (xy offset
within the half diagonal + 5 mm, z offset = one cube height +- 5 mm), static (<= 1 cm/s, <= 0.5 rad/s), and released.
State obs (48): qpos 9, qvel 9, tcp pose 7, cubeA pose 7, cubeB pose 7, tcp->A 3, tcp->B 3, A->B 3.  Host code is torch over the
backend's buffers; physics and camera are the same HIP kernels as PickCube.
"""
from __future__ import annotations

import numpy as np
import torch

from ..physx import SceneTemplate
from . import scene_builders as sb
from .pick_cube import PickCubeEnv


class StackCubeEnv(PickCubeEnv):
    state_actor_names = ("table-workspace", "cubeA", "cubeB")
    max_episode_steps = 50
    max_reward = 8.0
    obs_dim = 48

    def __init__(self, *args, **kw):
        kw["fused"] = False
        super().__init__(*args, **kw)
        self._b_cubeB = self._b_goal

    def _build_template(self, arm_stiffness=None):
        tpl = SceneTemplate()
        art = sb.add_panda(tpl, arm_stiffness=arm_stiffness)
        table = sb.add_table_scene(tpl)
        a = sb.add_cube(tpl, "cubeA", self.cube_half_size, (0, 0, 0.1))
        b = sb.add_cube(tpl, "cubeB", self.cube_half_size, (1, 0, 0.1))
        tpl.set_body_color(a, (1.0, 0.0, 0.0, 1.0))
        tpl.set_body_color(b, (0.0, 1.0, 0.0, 1.0))
        for k in range(len(tpl.body_names)):
            if tpl.body_names[k].startswith("panda_"):
                tpl.set_body_color(k, (0.9, 0.9, 0.9, 1.0))
        return tpl, dict(art=art, table=table, cube=a, goal_site=b)   # the base cube takes the slot of PickCube's goal marker

    def _hidden_bodies(self):
        return ()

    def _initialize_episode(self, env_idx, idx_np, off, f32):
        """stack_cube.py:78-113: a common random centre in [-0.1, 0.1]^2 plus two offsets from a UniformPlacementSampler over
        [-0.1, 0.1] x [-0.2, 0.2] that keeps the cubes at least two radii (half diagonal + 1 mm each) apart; yaw random."""
        b, dev = len(idx_np), self.device
        self._rbd[env_idx, self._b_goal, 7:13] = 0.0                          # _clear_sim_state for the second cube
        radius = float(np.linalg.norm([0.02, 0.02]) + 0.001)
        lo, rng = np.array([-0.1, -0.2]), np.array([0.2, 0.4])
        xy = self._rng.uniform(idx_np, 2) * 0.2 - 0.1
        pa = self._rng.uniform(idx_np, 2) * rng + lo
        pb = np.zeros((b, 2))
        done = np.zeros(b, dtype=bool)
        for _ in range(100):                                                  # rejection sampling, per env (samplers.py:52-73)
            cand = self._rng.uniform(idx_np, 2) * rng + lo
            ok = (np.linalg.norm(cand - pa, axis=1) > 2 * radius) & ~done
            pb[ok] = cand[ok]
            done |= ok
            if done.all():
                break
        yaw = self._rng.uniform(idx_np, 2) * (2 * np.pi)
        for bid, pxy, k in ((self._b_cube, xy + pa, 0), (self._b_goal, xy + pb, 1)):
            xyz = np.concatenate([pxy, np.full((b, 1), 0.02)], axis=1)
            q = np.zeros((b, 4)); q[:, 0] = np.cos(yaw[:, k] / 2); q[:, 3] = np.sin(yaw[:, k] / 2)
            self._rbd[env_idx, bid, :3] = f32(xyz) + off
            self._rbd[env_idx, bid, 3:7] = f32(q)

    @property
    def cubeB_pose(self):
        return self._pose(self._b_goal)

    def evaluate(self):
        self._fresh()
        pa, pb = self.cube_pose[:, :3], self.cubeB_pose[:, :3]
        offset = pa - pb
        hs = self.cube_half_size
        xy_flag = torch.linalg.norm(offset[:, :2], dim=1) <= float(np.linalg.norm([hs, hs])) + 0.005
        z_flag = torch.abs(offset[:, 2] - hs * 2) <= 0.005
        on = xy_flag & z_flag
        lin = self._rbd[:, self._b_cube, 7:10].norm(dim=1)
        ang = self._rbd[:, self._b_cube, 10:13].norm(dim=1)
        static = (lin <= 1e-2) & (ang <= 0.5)
        grasped = self.is_grasping()
        return {"is_cubeA_grasped": grasped, "is_cubeA_on_cubeB": on, "is_cubeA_static": static, "success": on & static & ~grasped}

    def get_obs(self, info):
        a, b, tcp = self.cube_pose, self.cubeB_pose, self.tcp_pose
        return torch.hstack([self.qpos, self.qvel, tcp, a, b, a[:, :3] - tcp[:, :3], b[:, :3] - tcp[:, :3], b[:, :3] - a[:, :3]])

    def compute_dense_reward(self, obs, action, info):
        tcp, pa, pb = self.tcp_pose[:, :3], self.cube_pose[:, :3], self.cubeB_pose[:, :3]
        reward = 2 * (1 - torch.tanh(5 * torch.linalg.norm(tcp - pa, dim=1)))
        goal = torch.hstack([pb[:, :2], (pb[:, 2] + self.cube_half_size * 2)[:, None]])
        place = 1 - torch.tanh(5.0 * torch.linalg.norm(goal - pa, dim=1))
        g = info["is_cubeA_grasped"]
        reward = torch.where(g, 4 + place, reward)
        width = 0.04 * 2                                                        # get_qlimits()[0, -1, 1] * 2 (panda finger joint upper limit)
        ungrasp = torch.where(g, self.qpos[:, -2:].sum(dim=1) / width, torch.ones_like(reward))
        v = self._rbd[:, self._b_cube, 7:10].norm(dim=1)
        av = self._rbd[:, self._b_cube, 10:13].norm(dim=1)
        static_reward = 1 - torch.tanh(v * 10 + av)
        reward = torch.where(info["is_cubeA_on_cubeB"], 6 + (ungrasp + static_reward) / 2.0, reward)
        return torch.where(info["success"], torch.full_like(reward, 8.0), reward)
