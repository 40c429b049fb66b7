"""PullCube-v1 on the MI355X-native backend: the reference task (mani_skill/envs/tasks/tabletop/pull_cube.py:23-151).

Same scene as PushCube-v1 (Panda, table, 4 cm cube, red / white goal disc without collision); the goal lies behind the cube
(towards the robot) and the reward asks the tcp to the far side of the cube.  Host code is torch over the backend's buffers
(graph-capturable: maniskill_amd/graph.py).
"""
from __future__ import annotations

import numpy as np
import torch

from ..graph import const
from .push_cube import PushCubeEnv


class PullCubeEnv(PushCubeEnv):
    max_reward = 3.0
    camera_eye, camera_target = (-0.5, 0.0, 0.25), (0.2, 0.0, -0.5)    # base_camera (:52-55)

    def _initialize_episode(self, env_idx, idx_np, off, f32):
        """:98-121: cube xy uniform in [-0.1, 0.1]^2, goal = cube - (0.1 + goal_radius, 0), 1 mm above the table."""
        b = len(idx_np)
        u = self._rng.uniform(idx_np, 2)
        xyz = np.zeros((b, 3))
        xyz[:, :2] = u * 0.2 - 0.1
        xyz[:, 2] = self.cube_half_size
        self._rbd[env_idx, self._b_cube, :3] = f32(xyz) + off
        self._rbd[env_idx, self._b_cube, 3:7] = const((1.0, 0.0, 0.0, 0.0), self.device)
        goal = xyz - np.array([0.1 + self.goal_radius, 0.0, 0.0])
        goal[:, 2] = 1e-3
        self._rbd[env_idx, self._b_goal, :3] = f32(goal) + off
        self._rbd[env_idx, self._b_goal, 3:7] = f32([np.cos(np.pi / 4), 0, np.sin(np.pi / 4), 0])   # euler2quat(0, pi/2, 0)

    def evaluate(self):
        """:123-134: the cube's xy within goal_radius of the goal's."""
        cube, goal = self.cube_pose[:, :3], self.goal_pos
        return {"success": torch.linalg.norm(cube[:, :2] - goal[:, :2], dim=1) < self.goal_radius}

    def compute_dense_reward(self, obs, action, info):
        """:147-165: reach the pull position (far side of the cube), then the cube-to-goal distance; 3 on success."""
        cube, tcp, goal = self.cube_pose[:, :3], self.tcp_pose[:, :3], self.goal_pos
        pull = cube + const((self.cube_half_size + 2 * 0.005, 0, 0), self.device)
        d = torch.linalg.norm(pull - tcp, dim=1)
        reward = 1 - torch.tanh(5 * d)
        reached = d < 0.01
        place = 1 - torch.tanh(5 * torch.linalg.norm(cube[:, :2] - goal[:, :2], dim=1))
        reward = reward + place * reached
        return torch.where(info["success"], torch.full_like(reward, 3.0), reward)
