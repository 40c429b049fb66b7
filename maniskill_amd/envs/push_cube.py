"""PushCube-v1 on the MI355X-native backend: the reference task (mani_skill/envs/tasks/tabletop/push_cube.py:36-260) on the
PickCube scene (Panda, table, 4 cm cube, a kinematic goal marker without collision).

  * _initialize_episode   push_cube.py:150-171   cube uniform in [-0.1, 0.1]^2 flat on the table, goal region 0.1 + r in front of it
  * evaluate               :173-184               cube centre within goal_radius of the region's centre and still on the table
  * _get_obs_extra         :186-207               tcp pose, goal position, cube pose  (state obs: 9 + 9 + 7 + 3 + 7 = 35)
  * compute_dense_reward   :209-247               reach the push pose behind the cube, then place, then keep it flat; success = 4

Host code is torch over the backend's buffers (the fused task kernels of include/msk_task.h are PickCube's and PushT's);
physics and camera are the same HIP kernels.
"""
from __future__ import annotations

import numpy as np
import torch

from ..graph import const

from . import scene_builders as sb
from .pick_cube import PickCubeEnv


class PushCubeEnv(PickCubeEnv):
    state_actor_names = ("table-workspace", "cube", "goal_region")
    max_episode_steps = 50
    max_reward = 4.0
    goal_radius = 0.1
    obs_dim = 35

    def __init__(self, *args, **kw):
        kw["fused"] = False
        super().__init__(*args, **kw)

    def _build_template(self, arm_stiffness=None):
        tpl, ids = sb.build_pick_cube_template(self.cube_half_size, arm_stiffness=arm_stiffness)
        goal = ids["goal_site"]
        ang = np.arange(24) * (2 * np.pi / 24)     # goal_region: red / white disc of radius goal_radius, 1e-5 thick, visual only
        disc = np.concatenate([np.stack([np.full(24, s * 1e-5), self.goal_radius * np.cos(ang), self.goal_radius * np.sin(ang)], axis=1)
                               for s in (-1.0, 1.0)])
        tpl.add_visual(goal, sb.N.SHAPE_CONVEX, verts=disc)
        tpl.set_body_color(goal, (194 / 255, 19 / 255, 22 / 255, 1.0))
        tpl.set_body_color(ids["cube"], (12 / 255, 42 / 255, 160 / 255, 1.0))
        return tpl, ids

    def _hidden_bodies(self):
        return ()     # the goal region is part of the picture

    def _initialize_episode(self, env_idx, idx_np, off, f32):
        b, dev = len(idx_np), self.device
        u = self._rng.uniform(idx_np, 2)
        xyz = np.zeros((b, 3))
        xyz[:, :2] = u * 0.2 - 0.1
        xyz[:, 2] = self.cube_half_size
        self._rbd[env_idx, self._b_cube, :3] = f32(xyz) + off
        self._rbd[env_idx, self._b_cube, 3:7] = torch.tensor([1.0, 0, 0, 0], device=dev)
        goal = xyz + np.array([0.1 + self.goal_radius, 0.0, 0.0])
        goal[:, 2] = 1e-3
        self._rbd[env_idx, self._b_goal, :3] = f32(goal) + off
        self._rbd[env_idx, self._b_goal, 3:7] = f32([np.cos(np.pi / 4), 0, np.sin(np.pi / 4), 0])   # euler2quat(0, pi/2, 0)

    def evaluate(self):
        cube, goal = self.cube_pose[:, :3], self.goal_pos
        placed = (torch.linalg.norm(cube[:, :2] - goal[:, :2], dim=1) < self.goal_radius) & (cube[:, 2] < self.cube_half_size + 5e-3)
        return {"success": placed}

    def get_obs(self, info):
        return torch.hstack([self.qpos, self.qvel, self.tcp_pose, self.goal_pos, self.cube_pose])

    def compute_dense_reward(self, obs, action, info):
        cube, tcp, goal = self.cube_pose[:, :3], self.tcp_pose[:, :3], self.goal_pos
        push = cube + const((-self.cube_half_size - 0.005, 0, 0), self.device)
        d = torch.linalg.norm(push - tcp, dim=1)
        reward = 1 - torch.tanh(5 * d)
        reached = d < 0.01
        place = 1 - torch.tanh(5 * torch.linalg.norm(cube[:, :2] - goal[:, :2], dim=1))
        reward = reward + place * reached
        z_reward = 1 - torch.tanh(5 * torch.abs(cube[:, 2] - self.cube_half_size))
        reward = reward + place * z_reward * reached
        return torch.where(info["success"], torch.full_like(reward, 4.0), reward)
