"""PegInsertionSide-v1 (BASELINE.json config 4) on the MI355X-native backend: the reference task
(mani_skill/envs/tasks/tabletop/peg_insertion_side.py:48-360).

Every sub-scene has its own peg (half length 0.085-0.125, radius 0.015-0.025) and its own box with a hole (four slabs
around a hole of that radius + 3 mm, offset from the box centre); the reference builds one actor per sub-scene and merges
them (`:133-187`), here the template's peg box and the four slabs are per-env instances (include/msk_physx.h:
msk_declare_env_box / msk_declare_env_mass).  Robot: PandaWristCam (panda_v3.urdf: the Panda with a camera link on the hand).

  * _load_scene / sizes           :109-187     * _initialize_episode   :189-241
  * has_peg_inserted / evaluate   :259-277     * _get_obs_extra        :279-288   state obs: 9 + 9 + 7 + 7 + 3 + 7 + 1 = 43
  * compute_dense_reward          :290-354     reach, grasp (max_angle 20), align, insert; success = 10
"""
from __future__ import annotations

import numpy as np
import torch

from ..graph import const

from ..physx import SceneTemplate
from .. import _native as N
from . import scene_builders as sb
from .pick_cube import PickCubeEnv


def _pose_mul(a, b):
    """(N,7) x (N,7) poses [p, q wxyz]"""
    return torch.cat([a[:, :3] + PickCubeEnv._qrot(a[:, 3:7], b[:, :3]), PickCubeEnv._qmul(a[:, 3:7], b[:, 3:7])], dim=-1)


def _pose_inv(a):
    qi = a[:, 3:7] * const((1.0, -1.0, -1.0, -1.0), a.device)
    return torch.cat([-PickCubeEnv._qrot(qi, a[:, :3]), qi], dim=-1)


def _from_p(p):
    q = torch.zeros(p.shape[0], 4, device=p.device); q[:, 0] = 1.0
    return torch.cat([p, q], dim=-1)


class PegInsertionSideEnv(PickCubeEnv):
    state_actor_names = ("table-workspace", "peg", "box_with_hole")
    state_articulation_name = "panda_wristcam"
    max_episode_steps = 100
    max_reward = 10.0
    obs_dim = 43
    clearance = 0.003
    camera_eye, camera_target = (0.0, -0.3, 0.2), (0.0, 0.0, 0.1)     # base_camera (:93-96)
    rest_qpos = np.array([0.0, np.pi / 8, 0, -np.pi * 5 / 8, 0, np.pi * 3 / 4, -np.pi / 4, 0.04, 0.04])   # :225-236

    grasp_max_angle = 20.0      # is_grasping(max_angle=20) in the reward (:289)

    def _init_fused_task(self):
        # the controller kernels and control_step are PickCube's (cube = peg); evaluate / obs / reward are the peg's
        import ctypes as C
        fp = C.POINTER(C.c_float)
        arrs = [np.ascontiguousarray(t.cpu().numpy(), dtype=np.float32) for t in (self.peg_half_sizes, self._hole_offset, self.box_hole_radii)]
        L = self.px.lib
        L.check(self.px.ctx, L.task_peg_init(self.px.ctx, *[a.ctypes.data_as(fp) for a in arrs]), "task_peg_init")

    def _fused_observe(self, advance: bool):
        import ctypes as C
        L, px = self.px.lib, self.px
        N, dev = self.num_envs, self.device
        from ..graph import alloc_step_outputs
        obs, rew, fl, elapsed, head = alloc_step_outputs(N, self.obs_dim, dev, extra_floats=3)
        L.check(px.ctx, L.task_peg_observe(px.ctx, C.c_void_p(obs.data_ptr()), C.c_void_p(rew.data_ptr()), C.c_void_p(fl.data_ptr()),
                                           C.c_void_p(self._elapsed_steps.data_ptr()), C.c_void_p(head.data_ptr()),
                                           1 if advance else 0, px._stream()), "task_peg_observe")
        elapsed.copy_(self._elapsed_steps)
        info = dict(elapsed_steps=elapsed, success=fl[:, 0], peg_head_pos_at_hole=head)
        return self._with_sensor_data(obs), self._fused_reward(rew, info), fl[:, 4], fl[:, 5], info

    # ---- scene -----------------------------------------------------------------------------------------------------------
    def _build_template(self, arm_stiffness=None):
        tpl = SceneTemplate()
        art = sb.add_panda(tpl, arm_stiffness=arm_stiffness, asset="panda_v3.json")
        table = sb.add_table_scene(tpl)
        m, I = sb.box_mass_properties([0.1, 0.02, 0.02], 1000.0)
        peg = tpl.add_actor("peg", N.BODY_DYNAMIC, p=(0, 0, 0.1), mass=m, inertia6=I)
        self._s_peg = tpl.add_shape(peg, N.SHAPE_BOX, params=(0.1, 0.02, 0.02))
        box = tpl.add_actor("box_with_hole", N.BODY_KINEMATIC, p=(0, 1, 0.1))
        self._s_slabs = [tpl.add_shape(box, N.SHAPE_BOX, params=(0.1, 0.04, 0.1)) for _ in range(4)]
        tpl.declare_env_box(self._s_peg)
        tpl.declare_env_mass(peg)
        for s in self._s_slabs:
            tpl.declare_env_box(s)
        tpl.set_body_color(peg, (0xEC / 255, 0x73 / 255, 0x57 / 255, 1.0))
        tpl.set_body_color(box, (0xFF / 255, 0xD2 / 255, 0x89 / 255, 1.0))
        for k in range(len(tpl.body_names)):
            if tpl.body_names[k].startswith("panda_") or tpl.body_names[k].startswith("camera_"):
                tpl.set_body_color(k, (0.9, 0.9, 0.9, 1.0))
        return tpl, dict(art=art, table=table, cube=peg, goal_site=box)   # the peg is what the gripper queries refer to

    def _hidden_bodies(self):
        return ()

    def _camera_configs(self, tpl):
        """base_camera (:93-96) and the agent's hand_camera on camera_link (agents/robots/panda/panda_wristcam.py:19-32)."""
        from ..render import CameraConfig
        hand = CameraConfig("hand_camera", (0.0, 0.0, 0.0), (1.0, 0.0, 0.0, 0.0), 128, 128, np.pi / 2, 0.01, 100.0, mount=tpl.body_id("camera_link"))
        return super()._camera_configs(tpl) + [hand]

    def _after_gpu_init(self):
        """Sizes drawn once per env from its own seed (the reference draws them at reconfiguration, :114-131)."""
        n = self.num_envs
        from .pick_cube import BatchedRNG
        rng = BatchedRNG((2022 + self.env_index_offset + np.arange(n)).astype(np.uint64) * np.uint64(7919) + np.uint64(17))
        idx = np.arange(n)
        u = rng.uniform(idx, 4)
        lengths = 0.085 + u[:, 0] * (0.125 - 0.085)
        radii = 0.015 + u[:, 1] * (0.025 - 0.015)
        centers = 0.5 * (lengths - radii)[:, None] * (u[:, 2:4] * 2 - 1)
        self.peg_half_sizes = torch.tensor(np.stack([lengths, radii, radii], axis=1), dtype=torch.float32, device=self.px.device)
        self.box_hole_radii = torch.tensor(radii + self.clearance, dtype=torch.float32, device=self.px.device)
        self._hole_offset = torch.tensor(np.concatenate([np.zeros((n, 1)), centers], axis=1), dtype=torch.float32, device=self.px.device)
        self._np_lengths, self._np_radii = lengths, radii
        # peg: box (length, r, r), density 1000
        half = np.stack([lengths, radii, radii], axis=1)
        mass = 8 * half.prod(1) * 1000.0
        inertia = np.stack([mass / 3 * (half[:, 1] ** 2 + half[:, 2] ** 2), mass / 3 * (half[:, 0] ** 2 + half[:, 2] ** 2),
                            mass / 3 * (half[:, 0] ** 2 + half[:, 1] ** 2)], axis=1)
        self.px.set_env_boxes(self._s_peg, half)
        self.px.set_env_masses(self._b_peg_body(), mass, inertia)
        # box with hole (_build_box_with_hole, :19-45): inner radius r + clearance, outer radius = depth = length
        inner, outer, depth = radii + self.clearance, lengths, lengths
        th = (outer - inner) * 0.5
        hc = centers * 0.5
        offs = th + inner
        slabs = [(np.stack([depth, th - hc[:, 0], outer], 1), np.stack([np.zeros(n), offs + hc[:, 0], np.zeros(n)], 1)),
                 (np.stack([depth, th + hc[:, 0], outer], 1), np.stack([np.zeros(n), -offs + hc[:, 0], np.zeros(n)], 1)),
                 (np.stack([depth, outer, th - hc[:, 1]], 1), np.stack([np.zeros(n), np.zeros(n), offs + hc[:, 1]], 1)),
                 (np.stack([depth, outer, th + hc[:, 1]], 1), np.stack([np.zeros(n), np.zeros(n), -offs + hc[:, 1]], 1))]
        for s, (hs, lp) in zip(self._s_slabs, slabs):
            self.px.set_env_boxes(s, hs, lp)

    def _b_peg_body(self):
        return self.ids["cube"]

    # ---- episode ---------------------------------------------------------------------------------------------------------
    def _initialize_episode(self, env_idx, idx_np, off, f32):
        b = len(idx_np)
        u = self._rng.uniform(idx_np, 6)
        r, L = self._np_radii[idx_np], self._np_lengths[idx_np]

        def zrot(a):
            q = np.zeros((b, 4)); q[:, 0] = np.cos(a / 2); q[:, 3] = np.sin(a / 2)
            return q
        peg = np.stack([u[:, 0] * 0.2 - 0.1, u[:, 1] * 0.3 - 0.3, r], axis=1)
        self._rbd[env_idx, self._b_cube, :3] = f32(peg) + off
        self._rbd[env_idx, self._b_cube, 3:7] = f32(zrot(np.pi / 2 - np.pi / 3 + u[:, 2] * (2 * np.pi / 3)))
        box = np.stack([u[:, 3] * 0.1 - 0.05, 0.2 + u[:, 4] * 0.2, L], axis=1)
        self._rbd[env_idx, self._b_goal, :3] = f32(box) + off
        self._rbd[env_idx, self._b_goal, 3:7] = f32(zrot(np.pi / 2 - np.pi / 8 + u[:, 5] * (np.pi / 4)))

    # ---- task ------------------------------------------------------------------------------------------------------------
    @property
    def peg_pose(self): return self.cube_pose
    @property
    def box_pose(self): return self._pose(self._b_goal)
    @property
    def peg_head_pose(self):
        head = torch.zeros(self.num_envs, 3, device=self.device); head[:, 0] = self.peg_half_sizes[:, 0]
        return _pose_mul(self.peg_pose, _from_p(head))
    @property
    def box_hole_pose(self): return _pose_mul(self.box_pose, _from_p(self._hole_offset))
    @property
    def goal_pose(self):
        head = torch.zeros(self.num_envs, 3, device=self.device); head[:, 0] = self.peg_half_sizes[:, 0]
        return _pose_mul(self.box_hole_pose, _pose_inv(_from_p(head)))

    def has_peg_inserted(self):
        p = _pose_mul(_pose_inv(self.box_hole_pose), self.peg_head_pose)[:, :3]
        rr = self.box_hole_radii
        ok = (-0.015 <= p[:, 0]) & (-rr <= p[:, 1]) & (p[:, 1] <= rr) & (-rr <= p[:, 2]) & (p[:, 2] <= rr)
        return ok, p

    def evaluate(self):
        self._fresh()
        ok, p = self.has_peg_inserted()
        return dict(success=ok, peg_head_pos_at_hole=p)

    def get_obs(self, info):
        return torch.hstack([self.qpos, self.qvel, self.tcp_pose, self.peg_pose, self.peg_half_sizes, self.box_hole_pose,
                             self.box_hole_radii[:, None]])

    def compute_dense_reward(self, obs, action, info):
        grip = self.tcp_pose[:, :3]
        off = torch.zeros(self.num_envs, 3, device=self.device); off[:, 0] = -0.06
        tgt = _pose_mul(self.peg_pose, _from_p(off))
        reaching = 1 - torch.tanh(4.0 * torch.linalg.norm(grip - tgt[:, :3], dim=1))
        grasped = self.is_grasping(max_angle=20)
        reward = reaching + grasped
        ginv = _pose_inv(self.goal_pose)
        head_yz = torch.linalg.norm(_pose_mul(ginv, self.peg_head_pose)[:, 1:3], dim=1)
        peg_yz = torch.linalg.norm(_pose_mul(ginv, self.peg_pose)[:, 1:3], dim=1)
        pre = 3 * (1 - torch.tanh(0.5 * (head_yz + peg_yz) + 4.5 * torch.maximum(head_yz, peg_yz)))
        reward = reward + pre * grasped
        pre_inserted = (head_yz < 0.01) & (peg_yz < 0.01)
        inside = _pose_mul(_pose_inv(self.box_hole_pose), self.peg_head_pose)[:, :3]
        insertion = 5 * (1 - torch.tanh(5.0 * torch.linalg.norm(inside, dim=1)))
        reward = reward + insertion * (grasped & pre_inserted)
        return torch.where(info["success"], torch.full_like(reward, 10.0), reward)
