"""PokeCube-v1 on the MI355X-native backend: the reference task (mani_skill/envs/tasks/tabletop/poke_cube.py:20-230).

Panda, table, a 24 x 5 x 5 cm peg, a 4 cm cube in front of the peg's head and a red / white goal disc (no collision) behind the
cube: grasp the peg and poke the cube onto the goal.  Host code is torch over the backend's buffers (graph-capturable).
"""
from __future__ import annotations

import numpy as np
import torch

from ..physx import SceneTemplate
from .. import _native as N
from . import scene_builders as sb
from .pick_cube import PickCubeEnv


class PokeCubeEnv(PickCubeEnv):
    state_actor_names = ("table-workspace", "cube", "peg", "goal_region")
    max_episode_steps = 50
    max_reward = 10.0
    obs_dim = 54
    cube_half_size = 0.02
    peg_half_width = 0.025
    peg_half_length = 0.12
    goal_radius = 0.05
    camera_eye, camera_target = (0.3, 0.0, 0.6), (-0.1, 0.0, 0.1)      # base_camera (:50-53)

    def __init__(self, *args, **kw):
        kw["fused"] = False
        super().__init__(*args, **kw)

    def _build_template(self, arm_stiffness=None):
        tpl = SceneTemplate()
        art = sb.add_panda(tpl, arm_stiffness=arm_stiffness)
        table = sb.add_table_scene(tpl)
        cube = sb.add_cube(tpl, "cube", self.cube_half_size, (1.0, 0, self.cube_half_size))
        half = (self.peg_half_length, self.peg_half_width, self.peg_half_width)
        m, I = sb.box_mass_properties(half, 1000.0)
        peg = tpl.add_actor("peg", N.BODY_DYNAMIC, p=(0, 0, self.peg_half_width), mass=m, inertia6=I)
        tpl.add_shape(peg, N.SHAPE_BOX, params=half)
        goal = sb.add_site(tpl, "goal_region")
        ang = np.arange(24) * (2 * np.pi / 24)     # red / white disc of radius goal_radius, 1e-5 thick, visual only
        disc = np.concatenate([np.stack([np.full(24, s * 1e-5), self.goal_radius * np.cos(ang), self.goal_radius * np.sin(ang)], axis=1)
                               for s in (-1.0, 1.0)])
        tpl.add_visual(goal, N.SHAPE_CONVEX, verts=disc)
        tpl.set_body_color(goal, (194 / 255, 19 / 255, 22 / 255, 1.0))
        tpl.set_body_color(cube, (1.0, 0.0, 0.0, 1.0))
        tpl.set_body_color(peg, (12 / 255, 42 / 255, 160 / 255, 1.0))
        for k, name in enumerate(tpl.body_names):
            if name.startswith("panda_"):
                tpl.set_body_color(k, (0.9, 0.9, 0.9, 1.0))
        self._b_poked = cube
        return tpl, dict(art=art, table=table, cube=peg, goal_site=goal)   # "cube" = what the gripper's contact queries refer to: the peg

    def _hidden_bodies(self):
        return ()

    def _state_actor_bodies(self):
        return [self._b_table, self._b_poked, self._b_cube, self._b_goal]

    def _initialize_episode(self, env_idx, idx_np, off, f32):
        """:111-141: peg xy uniform in [-0.1, 0.1]^2 along x; cube 0.1 in front of the peg's head, y uniform, yaw in +-pi/6;
        goal = cube + (0.05 + goal_radius, 0)."""
        b = len(idx_np)
        u = self._rng.uniform(idx_np, 4)
        peg = np.zeros((b, 3))
        peg[:, :2] = u[:, :2] * 0.2 - 0.1
        peg[:, 2] = self.peg_half_width
        self._rbd[env_idx, self._b_cube, :3] = f32(peg) + off
        self._rbd[env_idx, self._b_cube, 3:7] = f32([1.0, 0, 0, 0])
        cube = np.zeros((b, 3))
        cube[:, 0] = peg[:, 0] + self.peg_half_length + 0.1
        cube[:, 1] = u[:, 2] * 0.2 - 0.1
        cube[:, 2] = self.cube_half_size
        yaw = u[:, 3] * (np.pi / 3) - np.pi / 6
        q = np.zeros((b, 4)); q[:, 0] = np.cos(yaw / 2); q[:, 3] = np.sin(yaw / 2)
        self._rbd[env_idx, self._b_poked, :3] = f32(cube) + off
        self._rbd[env_idx, self._b_poked, 3:7] = f32(q)
        self._rbd[env_idx, self._b_poked, 7:13] = 0.0
        goal = cube + np.array([0.05 + self.goal_radius, 0.0, 0.0])
        goal[:, 2] = 1e-3
        self._rbd[env_idx, self._b_goal, :3] = f32(goal) + off
        self._rbd[env_idx, self._b_goal, 3:7] = f32([np.cos(np.pi / 4), 0, np.sin(np.pi / 4), 0])   # euler2quat(0, pi/2, 0)

    @property
    def peg_pose(self): return self.cube_pose
    @property
    def poked_cube_pose(self): return self._pose(self._b_poked)
    @property
    def peg_head_pos(self):
        """:104-105: the peg's position plus the head offset along the WORLD x axis (as the reference adds it)."""
        p = self.peg_pose[:, :3].clone()
        p[:, 0] += self.peg_half_length
        return p

    def evaluate(self):
        """:159-190."""
        cube, peg, goal = self.poked_cube_pose, self.peg_pose, self.goal_pos
        placed = torch.linalg.norm(cube[:, :2] - goal[:, :2], dim=1) < self.goal_radius
        angle_diff = torch.abs(self._quat_to_euler_xyz(peg[:, 3:7])[:, 2] - self._quat_to_euler_xyz(cube[:, 3:7])[:, 2])
        head_to_cube = torch.linalg.norm(self.peg_head_pos[:, :2] - cube[:, :2], dim=1)
        fit = (angle_diff < 0.05) & (head_to_cube <= self.cube_half_size + 0.005)
        static = self.is_static(0.2)
        return {"success": placed & static, "is_cube_placed": placed, "is_peg_cube_fit": fit, "is_peg_grasped": self.is_grasping(),
                "angle_diff": angle_diff, "head_to_cube_dist": head_to_cube}

    def get_obs(self, info):
        """:143-157 (goal_pos is the peg's position there, and here)."""
        tcp, cube, peg, goal = self.tcp_pose, self.poked_cube_pose, self.peg_pose, self.goal_pos
        return torch.hstack([self.qpos, self.qvel, tcp, cube, peg, peg[:, :3], peg[:, :3] - tcp[:, :3], cube[:, :3] - peg[:, :3],
                             goal - cube[:, :3], self.peg_head_pos - cube[:, :3]])

    def compute_dense_reward(self, obs, action, info):
        """:192-222."""
        tcp, peg, cube, goal = self.tcp_pose[:, :3], self.peg_pose[:, :3], self.poked_cube_pose[:, :3], self.goal_pos
        d = torch.linalg.norm(tcp - peg, dim=1)
        reached = d < 0.01
        reward = 2 * (1 - torch.tanh(5.0 * d))
        align = 1 - torch.tanh(5.0 * info["angle_diff"])
        close = 1 - torch.tanh(5.0 * info["head_to_cube_dist"])
        grasped = info["is_peg_grasped"] & reached
        reward = torch.where(grasped, 4 + close + align, reward)
        place = 1 - torch.tanh(5 * torch.linalg.norm(goal - cube, dim=1))
        fit = info["is_peg_cube_fit"] & grasped
        reward = torch.where(fit, 7 + place, reward)
        static = 1 - torch.tanh(5 * torch.linalg.norm(self.qvel[:, :-2], dim=1))
        reward = torch.where(info["is_cube_placed"], reward + static, reward)
        return torch.where(info["success"], torch.full_like(reward, 10.0), reward)
