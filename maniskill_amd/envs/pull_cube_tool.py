"""PullCubeTool-v1 on the MI355X-native backend: the reference task (mani_skill/envs/tasks/tabletop/pull_cube_tool.py:20-282).

Panda, table, a cube out of the arm's reach and an L-shaped tool (handle 20 x 5 x 5 cm at density 500, hook 5 x 10 x 5 cm at
density 1000: one dynamic actor with two boxes) within reach: grasp the tool, hook the cube, pull it towards the base.  Host code
is torch over the backend's buffers (graph-capturable: maniskill_amd/graph.py).
"""
from __future__ import annotations

import numpy as np
import torch

from ..graph import const
from ..physx import SceneTemplate
from .. import _native as N
from . import scene_builders as sb
from .pick_cube import PickCubeEnv


def _compound_box_mass(boxes):
    """Mass, centre of mass and inertia about it (xx, yy, zz, xy, xz, yz) of boxes [(centre, half sizes, density)] in one frame."""
    m = np.array([d * 8 * h[0] * h[1] * h[2] for _, h, d in boxes])
    c = np.array([ctr for ctr, _, _ in boxes], dtype=np.float64)
    com = (m[:, None] * c).sum(0) / m.sum()
    I = np.zeros((3, 3))
    for (ctr, h, _), mi in zip(boxes, m):
        hx, hy, hz = h
        I += np.diag([mi / 3 * (hy * hy + hz * hz), mi / 3 * (hx * hx + hz * hz), mi / 3 * (hx * hx + hy * hy)])
        r = np.asarray(ctr, dtype=np.float64) - com
        I += mi * (np.dot(r, r) * np.eye(3) - np.outer(r, r))
    return float(m.sum()), tuple(com), (I[0, 0], I[1, 1], I[2, 2], I[0, 1], I[0, 2], I[1, 2])


class PullCubeToolEnv(PickCubeEnv):
    state_actor_names = ("table-workspace", "cube", "l_shape_tool")
    max_episode_steps = 100
    max_reward = 5.0
    obs_dim = 39
    goal_radius = 0.3
    cube_half_size = 0.02
    handle_length, hook_length, width, height = 0.2, 0.05, 0.05, 0.05
    cube_size = 0.02
    arm_reach = 0.35
    grasp_max_angle = 20.0
    camera_eye, camera_target = (0.3, 0.0, 0.5), (-0.1, 0.0, 0.1)      # base_camera (:66-80)

    def __init__(self, *args, **kw):
        kw["fused"] = False
        super().__init__(*args, **kw)

    def _build_template(self, arm_stiffness=None):
        tpl = SceneTemplate()
        art = sb.add_panda(tpl, arm_stiffness=arm_stiffness)
        table = sb.add_table_scene(tpl)
        cube = sb.add_cube(tpl, "cube", self.cube_half_size, (0, 0, self.cube_half_size))
        hl, kl, w, h = self.handle_length, self.hook_length, self.width, self.height
        boxes = [((hl / 2, 0.0, 0.0), (hl / 2, w / 2, h / 2), 500.0),              # handle (:101-105)
                 ((hl - kl / 2, w, 0.0), (kl / 2, w, h / 2), 1000.0)]              # hook   (:112-115)
        m, com, I = _compound_box_mass(boxes)
        tool = tpl.add_actor("l_shape_tool", N.BODY_DYNAMIC, p=(0, 0, h / 2), mass=m, com=com, inertia6=I)
        for ctr, half, _ in boxes:
            tpl.add_shape(tool, N.SHAPE_BOX, p=ctr, params=half)
        tpl.set_body_color(cube, (12 / 255, 42 / 255, 160 / 255, 1.0))
        tpl.set_body_color(tool, (1.0, 0.0, 0.0, 1.0))
        for k, name in enumerate(tpl.body_names):
            if name.startswith("panda_"):
                tpl.set_body_color(k, (0.9, 0.9, 0.9, 1.0))
        self._b_pulled = cube
        return tpl, dict(art=art, table=table, cube=tool, goal_site=cube)   # "cube" = what the gripper's contact queries refer to: the tool

    def _hidden_bodies(self):
        return ()

    def _state_actor_bodies(self):
        return [self._b_table, self._b_pulled, self._b_cube]

    def _initialize_episode(self, env_idx, idx_np, off, f32):
        """:146-179: tool in [-0.3, -0.1]^2, unrotated; cube x in arm_reach - 0.3 + [0, handle_length), y in [-0.25, 0.05), yaw +-pi/6."""
        b = len(idx_np)
        u = self._rng.uniform(idx_np, 5)
        tool = np.stack([-u[:, 0] * 0.2 - 0.1, -u[:, 1] * 0.2 - 0.1, np.full(b, self.height / 2)], axis=1)
        self._rbd[env_idx, self._b_cube, :3] = f32(tool) + off
        self._rbd[env_idx, self._b_cube, 3:7] = const((1.0, 0.0, 0.0, 0.0), self.device)
        cube = np.stack([self.arm_reach + u[:, 2] * self.handle_length - 0.3, u[:, 3] * 0.3 - 0.25, np.full(b, self.cube_size / 2 + 0.015)], axis=1)
        yaw = u[:, 4] * (np.pi / 3) - np.pi / 6
        q = np.zeros((b, 4)); q[:, 0] = np.cos(yaw / 2); q[:, 3] = np.sin(yaw / 2)
        self._rbd[env_idx, self._b_pulled, :3] = f32(cube) + off
        self._rbd[env_idx, self._b_pulled, 3:7] = f32(q)
        self._rbd[env_idx, self._b_pulled, 7:13] = 0.0

    @property
    def tool_pose(self): return self.cube_pose
    @property
    def pulled_cube_pose(self): return self._pose(self._b_pulled)

    def evaluate(self):
        """:191-214: the cube within 0.6 m of the robot's base (xy)."""
        cube, base = self.pulled_cube_pose[:, :3], self._pose(self._b_root)[:, :3]
        return {"success": torch.linalg.norm(cube[:, :2] - base[:, :2], dim=1) < 0.6}

    def get_obs(self, info):
        return torch.hstack([self.qpos, self.qvel, self.tcp_pose, self.pulled_cube_pose, self.tool_pose])

    def compute_dense_reward(self, obs, action, info):
        """:216-270."""
        tcp, cube, tool = self.tcp_pose[:, :3], self.pulled_cube_pose[:, :3], self.tool_pose[:, :3]
        base = self._pose(self._b_root)[:, :3]
        dev = self.device
        reaching = 2.0 * (1 - torch.tanh(5.0 * torch.linalg.norm(tcp - (tool + const((0.02, 0, 0), dev)), dim=1)))
        grasp = self.is_grasping(max_angle=20)
        reward = reaching + 2.0 * grasp
        ideal = cube + const((-(self.hook_length + self.cube_half_size), -0.067, 0), dev)
        d_pos = torch.linalg.norm(tool - ideal, dim=1)
        reward = reward + 1.5 * (1 - torch.tanh(3.0 * d_pos)) * grasp
        target = base + const((0.05, 0, 0), dev)
        d_ws = torch.linalg.norm(cube - target, dim=1)
        d0 = torch.linalg.norm(const((self.arm_reach + 0.1, 0, self.cube_size / 2), dev) - target, dim=1)
        reward = reward + 3.0 * ((d0 - d_ws) / d0) * (d_pos < 0.05) * grasp
        reward = torch.where(cube[:, 0] > self.arm_reach + 0.15, reward - 2.0, reward)
        return torch.where(info["success"], reward + 5.0, reward)
