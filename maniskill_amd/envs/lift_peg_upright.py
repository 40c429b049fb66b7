"""LiftPegUpright-v1 on the MI355X-native backend: the reference task (mani_skill/envs/tasks/tabletop/lift_peg_upright.py:20-144).

Panda, table and a 24 x 5 x 5 cm peg (one collision box, actors/common.py:230-261) lying on the table; success = the peg stands on
its end.  Host code is torch over the backend's buffers (graph-capturable: maniskill_amd/graph.py).
"""
from __future__ import annotations

import numpy as np
import torch

from ..graph import const
from ..physx import SceneTemplate
from .. import _native as N
from . import scene_builders as sb
from .pick_cube import PickCubeEnv


class LiftPegUprightEnv(PickCubeEnv):
    state_actor_names = ("table-workspace", "peg")
    max_episode_steps = 50
    max_reward = 3.0
    obs_dim = 32
    peg_half_width = 0.025
    peg_half_length = 0.12
    camera_eye, camera_target = (0.3, 0.0, 0.6), (-0.1, 0.0, 0.1)      # base_camera (:46-49)

    def __init__(self, *args, **kw):
        kw["fused"] = False
        super().__init__(*args, **kw)

    def _build_template(self, arm_stiffness=None):
        tpl = SceneTemplate()
        art = sb.add_panda(tpl, arm_stiffness=arm_stiffness)
        table = sb.add_table_scene(tpl)
        half = (self.peg_half_length, self.peg_half_width, self.peg_half_width)
        m, I = sb.box_mass_properties(half, 1000.0)
        peg = tpl.add_actor("peg", N.BODY_DYNAMIC, p=(0, 0, 0.1), mass=m, inertia6=I)
        tpl.add_shape(peg, N.SHAPE_BOX, params=half)
        tpl.set_body_color(peg, (176 / 255, 14 / 255, 14 / 255, 1.0))    # color_1 of the two-colour peg (one colour per body here)
        for k, name in enumerate(tpl.body_names):
            if name.startswith("panda_"):
                tpl.set_body_color(k, (0.9, 0.9, 0.9, 1.0))
        return tpl, dict(art=art, table=table, cube=peg, goal_site=peg)   # no goal actor in this task: both roles are the peg's

    def _hidden_bodies(self):
        return ()

    def _state_actor_bodies(self):
        return [self._b_table, self._b_cube]

    def _initialize_episode(self, env_idx, idx_np, off, f32):
        """:78-90: xy uniform in [-0.1, 0.1]^2, lying flat: rolled 90 deg about its own axis, z = half width."""
        b = len(idx_np)
        u = self._rng.uniform(idx_np, 2)
        xyz = np.zeros((b, 3))
        xyz[:, :2] = u * 0.2 - 0.1
        xyz[:, 2] = self.peg_half_width
        self._rbd[env_idx, self._b_cube, :3] = f32(xyz) + off
        self._rbd[env_idx, self._b_cube, 3:7] = const((float(np.cos(np.pi / 4)), float(np.sin(np.pi / 4)), 0.0, 0.0), self.device)   # euler2quat(pi/2, 0, 0)

    @property
    def peg_pose(self): return self.cube_pose

    def evaluate(self):
        """:92-103: |euler_XYZ[2]| within 0.08 of pi/2 and the centre at half the length above the table."""
        peg = self.peg_pose
        euler = self._quat_to_euler_xyz(peg[:, 3:7])
        upright = torch.abs(torch.abs(euler[:, 2]) - np.pi / 2) < 0.08
        close = torch.abs(peg[:, 2] - self.peg_half_length) < 0.005
        return {"success": upright & close}

    def get_obs(self, info):
        return torch.hstack([self.qpos, self.qvel, self.tcp_pose, self.peg_pose])

    def compute_dense_reward(self, obs, action, info):
        """:115-138: |peg axis . z| + height term + reaching / 5 (1 / 5 when grasped); 3 on success."""
        peg, tcp = self.peg_pose, self.tcp_pose[:, :3]
        w, x, y, z = peg[:, 3:7].unbind(-1)
        axis_z = 2 * (x * z - w * y)                    # z component of R e_x
        reward = axis_z.abs()
        reward = reward + 1 - torch.tanh(5 * torch.abs(peg[:, 2] - self.peg_half_length))
        reach = 1 - torch.tanh(5 * torch.linalg.norm(peg[:, :3] - tcp, dim=1))
        reach = torch.where(self.is_grasping(), torch.ones_like(reach), reach) / 5
        reward = reward + reach
        return torch.where(info["success"], torch.full_like(reward, 3.0), reward)
