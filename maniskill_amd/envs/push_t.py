"""PushT-v1 on the MI355X-native backend: host-side mirror of the reference task.

Mirrors, with the same names and semantics:
  * PushTEnv                                   mani_skill/envs/tasks/tabletop/push_t.py:69-540
      _load_scene (Tee, goal_Tee, goal_ee, pseudo-render tables)   :159-320
      pseudo_render_intersection                                    :343-431
      _initialize_episode                                           :433-482
      evaluate / _get_obs_extra / compute_dense_reward              :484-540
  * WhiteTableSceneBuilder.initialize (panda_stick keyframe)        push_t.py:28-50
  * PandaStick, pd_joint_delta_pos (7 arm joints, +-0.1 rad)        mani_skill/agents/robots/panda/panda_stick.py:16-165
  * base_camera 128x128, fov pi/2                                   push_t.py:133-145
  * BaseEnv.reset / step / obs                                      mani_skill/envs/sapien_env.py:857-978,1042-1132,501-634

Everything here is torch indexing over the backend's zero-copy buffers; the physics is ``px.step()`` and the
camera is ``RenderCameraGroup.take_picture()`` (HIP kernels behind include/msk_physx.h and include/msk_render.h).
"""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch

from ..physx import PhysxGpuSystem, SimConfig
from . import scene_builders as sb
from .pick_cube import BatchedRNG


from ._device_reset import DeviceResetMixin      # noqa: E402


class PushTEnv(DeviceResetMixin):
    """PushT-v1, ``pd_joint_delta_pos`` control, PandaStick; obs_mode 'state' or 'depth+segmentation'."""

    max_episode_steps = 100
    camera_eye, camera_target = (0.3, 0.0, 0.6), (-0.1, 0.0, 0.1)      # base_camera (push_t.py:133-145)
    tee_spawnbox_xlength, tee_spawnbox_ylength = 0.2, 0.3
    tee_spawnbox_xoffset, tee_spawnbox_yoffset = -0.1, -0.1
    goal_offset = (-0.156, -0.1)
    goal_z_rot = (5 / 3) * np.pi
    ee_starting_pos2D = (-0.321, 0.284, 1e-3)
    intersection_thresh = 0.90
    arm_delta = 0.1
    action_dim = 7

    def __init__(self, num_envs: int = 1, device: Optional[str] = None, sim_config: Optional[SimConfig] = None,
                 robot_init_qpos_noise: float = 0.02, obs_mode: str = "state", env_index_offset: int = 0,
                 total_envs: Optional[int] = None, px_factory=None, fused: Optional[bool] = None, reward_mode: str = "normalized_dense",
                 device_reset: Optional[bool] = None):
        self.device_reset = device_reset      # partial resets from a device-side mask (envs/_device_reset.py); None: on for the fused env on a GPU
        if reward_mode not in ("normalized_dense", "dense", "sparse", "none"):
            raise NotImplementedError(f"reward_mode {reward_mode!r}: one of 'normalized_dense', 'dense', 'sparse', 'none' (sapien_env.py:648-670)")
        self.reward_mode = reward_mode
        self.num_envs = int(num_envs)
        self.sim_config = sim_config or SimConfig()
        self.robot_init_qpos_noise = robot_init_qpos_noise
        self.env_index_offset = int(env_index_offset)
        self._sim_steps_per_control = self.sim_config.sim_freq // self.sim_config.control_freq
        tpl, ids = sb.build_push_t_template()
        self.template, self.ids = tpl, ids
        if px_factory is None:
            dev = torch.device(device or "cuda")
            if dev.type != "cuda":
                raise RuntimeError("maniskill_amd runs its physics on an AMD GPU (device 'cuda[:k]'); there is no CPU backend")
            self.px = PhysxGpuSystem(dev, tpl, self.num_envs, self.sim_config)
        else:
            self.px = px_factory(tpl, self.num_envs, self.sim_config)
        self.device = self.px.device
        self.px.gpu_init()
        side = int(np.ceil(np.sqrt(total_envs if total_envs is not None else self.num_envs)))
        g = np.arange(self.num_envs) + self.env_index_offset
        offsets = np.stack([(g % side - side // 2) * self.sim_config.spacing, (g // side - side // 2) * self.sim_config.spacing,
                            np.zeros(self.num_envs)], axis=1)
        self.px.set_scene_offsets(offsets)
        N, NB, dev = self.num_envs, self.px.bodies_per_env, self.device
        self._rbd = self.px.cuda_rigid_body_data.torch().view(N, NB, 13)
        self._qpos = self.px.cuda_articulation_qpos.torch().view(N, -1)
        self._qvel = self.px.cuda_articulation_qvel.torch().view(N, -1)
        self._target_qpos_buf = self.px.cuda_articulation_target_qpos.torch().view(N, -1)
        self._offsets = self.px.scene_offsets
        from ..structs import SceneView
        self.scene = SceneView(self.px, fresh=self._fresh)   # Actor / Link / Articulation views (structs.py)
        self.robot = self.scene.articulations[tpl.art_names[0]]
        from .. import spaces
        self.single_action_space = spaces.Box(-np.ones(7, np.float32), np.ones(7, np.float32), dtype=np.float32)   # pd_joint_delta_pos of PandaStick
        self.action_space = spaces.batch_space(self.single_action_space, self.num_envs)
        self._b_tee, self._b_goal, self._b_ee, self._b_table = ids["tee"], ids["goal_tee"], ids["goal_ee"], ids["table"]
        self._b_root, self._b_tcp = tpl.body_id("panda_link0"), tpl.body_id("panda_hand_tcp")
        self._table_pose = torch.tensor([-0.12, 0.0, -sb.TABLE_HEIGHT, np.cos(np.pi / 4), 0, 0, np.sin(np.pi / 4)], dtype=torch.float32, device=dev)
        self._root_pose = torch.tensor([-0.615, 0.0, 0.0, 1, 0, 0, 0], dtype=torch.float32, device=dev)
        self._elapsed_steps = torch.zeros(N, dtype=torch.int32, device=dev)
        self._target_qpos = torch.zeros(N, 7, dtype=torch.float32, device=dev)
        self._main_seeds = 2022 + self.env_index_offset + np.arange(N)
        self._rng = BatchedRNG(self._main_seeds)
        self._episode_count = np.zeros(N, dtype=np.uint64)
        self._setup_pseudo_render()
        if obs_mode not in ("state", "depth+segmentation", "rgb", "rgbd", "rgb+depth+segmentation"):
            raise NotImplementedError(f"obs_mode {obs_mode!r}: this backend provides 'state', 'depth+segmentation', 'rgb', 'rgbd' "
                                      "and 'rgb+depth+segmentation'")
        self.obs_mode = obs_mode
        # obs mode -> textures (sapien_env.py:120-160 parse_obs_mode_to_struct)
        self._textures = dict(rgb="rgb" in obs_mode, depth=("depth" in obs_mode or obs_mode == "rgbd"), segmentation="segmentation" in obs_mode)
        self._want_color = self._textures["rgb"]
        self.camera = None
        if obs_mode != "state":
            from ..render import CameraConfig, RenderCameraGroup, attach_template_visuals, look_at
            attach_template_visuals(self.px, tpl)
            p, q = look_at(eye=list(self.camera_eye), target=list(self.camera_target))
            self.camera = RenderCameraGroup(self.px, CameraConfig("base_camera", p, q, 128, 128, np.pi / 2, 0.01, 100.0))
            if self._want_color:
                self.camera.enable_color()
            self.camera.set_outputs(position_texture=False)      # no obs mode of this env hands out `position`: the planes (and Color) are all a step needs
        self.obs_dim = 7 + 7 + 7 + 3 + 7
        # fused task kernels (include/msk_task.h): controller, evaluate / obs / reward as two launches instead of ~100 torch ops
        can_fuse = getattr(self.px.lib, "has_task_kernels", False)      # (the CPU oracle has none; the emulated HIP library of tests/hipemu does)
        self.fused = (can_fuse and not self.px.host_memory) if fused is None else bool(fused)
        self._buffers_stale = False
        if self.fused:
            if not can_fuse:
                raise RuntimeError("fused task kernels need the HIP backend")
            from .. import _native as NN
            import ctypes as C
            w2g = self.world_to_goal_trans.cpu().numpy().astype(np.float32)
            d = NN.PushTDesc(tee=self._b_tee, goal=self._b_goal, tcp=self._b_tcp, arm_dofs=7, arm_delta=self.arm_delta,
                             goal_xy=(C.c_float * 2)(*self.goal_offset), goal_z_rot=float(self.goal_z_rot),
                             world_to_goal=(C.c_float * 6)(*[float(x) for x in w2g[:2].reshape(-1)]),
                             uv_scale=float(np.float32((self.res / 2) / self.uv_half_width)),
                             intersection_thresh=self.intersection_thresh, max_episode_steps=self.max_episode_steps)
            mask = np.ascontiguousarray(self.tee_render.cpu().numpy().astype(np.uint8))
            self.px.lib.check(self.px.ctx, self.px.lib.task_pusht_init(self.px.ctx, C.byref(d), mask.ctypes.data_as(C.POINTER(C.c_uint8))),
                              "task_pusht_init")
        self.reset(seed=None)
        self._constructed = True

    # ---------------------------------------------------------------- the 64 x 64 "pseudo render" tables (push_t.py:264-320)
    def _setup_pseudo_render(self):
        dev = self.device
        res, hw = 64, 0.15
        self.res, self.uv_half_width = res, hw
        grid = torch.arange(res, dtype=torch.float32).view(1, res).repeat(res, 1) - (res / 2)
        uv = (torch.cat([grid.unsqueeze(0), (-1 * grid.T).unsqueeze(0)], dim=0) + 0.5) / ((res / 2) / hw)
        self.uv_grid = uv.to(dev)
        self.homo_uv = torch.cat([self.uv_grid, torch.ones_like(self.uv_grid[0]).unsqueeze(0)], dim=0)
        box1 = torch.tensor([[-0.1, 0.025], [0.1, 0.025], [-0.1, -0.025], [0.1, -0.025]])
        box2 = torch.tensor([[-0.025, 0.175], [0.025, 0.175], [-0.025, 0.025], [0.025, 0.025]])
        box1[:, 1] -= sb.TEE_COM_Y
        box2[:, 1] -= sb.TEE_COM_Y
        box1 = (box1 * ((res / 2) / hw) + res / 2).long()
        box2 = (box2 * ((res / 2) / hw) + res / 2).long()
        tee = torch.zeros(res, res)
        tee.T[box1[0, 0]:box1[1, 0], box1[2, 1]:box1[0, 1]] = 1
        tee.T[box2[0, 0]:box2[1, 0], box2[2, 1]:box2[0, 1]] = 1
        self.tee_render = tee.flip(0).to(dev)
        c, s = np.cos(self.goal_z_rot), np.sin(self.goal_z_rot)  # quat_to_zrot of [cos(a/2), 0, 0, 0] reduces to a z rotation by a
        goal_trans = torch.tensor([[c, -s, self.goal_offset[0]], [s, c, self.goal_offset[1]], [0, 0, 1]], dtype=torch.float32)
        self.world_to_goal_trans = torch.linalg.inv(goal_trans).to(dev)

    @staticmethod
    def quat_to_z_euler(quats):
        signs = torch.ones_like(quats[:, -1])
        signs[quats[:, -1] < 0] = -1.0
        return 2 * (quats[:, 0] * signs).clamp(-1, 1).acos()

    def quat_to_zrot(self, quats):
        a = self.quat_to_z_euler(quats)
        R = torch.zeros(quats.shape[0], 3, 3, device=quats.device)
        R[:, 2, 2] = 1
        R[:, 0, 0] = a.cos(); R[:, 1, 1] = a.cos(); R[:, 0, 1] = -a.sin(); R[:, 1, 0] = a.sin()
        return R

    def pseudo_render_intersection(self):
        tee = self._pose(self._b_tee)
        T = self.quat_to_zrot(tee[:, 3:7])
        T[:, 0:2, 2] = tee[:, :2]
        T = self.world_to_goal_trans @ T
        b, res = T.shape[0], self.res
        pts = (T @ self.homo_uv.view(3, -1)).view(b, 3, res, res)
        pts = pts[:, 0:2] / pts[:, -1].unsqueeze(1)
        coords = pts[:, :, self.tee_render == 1].view(b, 2, -1)
        idx = (coords * ((res / 2) / self.uv_half_width) + (res / 2)).long().view(b, 2, -1)
        bad = (idx[:, 0] < 0) | (idx[:, 0] >= res) | (idx[:, 1] < 0) | (idx[:, 1] >= res)
        idx[:, 0][bad] = 0
        idx[:, 1][bad] = 0
        final = torch.zeros(b, res, res, device=self.device)
        bi = torch.arange(b, device=self.device).view(-1, 1).repeat(1, idx.shape[-1])
        final[bi, idx[:, 0], idx[:, 1]] = 1
        final = final.permute(0, 2, 1).flip(1)
        inter = (final.bool() & self.tee_render.bool()).sum(dim=[-1, -2]).float()
        return inter / self.tee_render.bool().sum().float()

    # ---------------------------------------------------------------- struct-style views
    def _fresh(self):
        """Fused mode publishes the sapien-style buffers lazily: refresh them before any host-side read."""
        if self.fused and self._buffers_stale:
            self.px.gpu_fetch_all()
            self._buffers_stale = False

    def _pose(self, body):
        self._fresh()
        raw = self._rbd[:, body, :7].clone()
        raw[:, :3] -= self._offsets
        return raw

    @property
    def qpos(self):
        self._fresh()
        return self._qpos[:, :7]
    @property
    def qvel(self):
        self._fresh()
        return self._qvel[:, :7]
    @property
    def tcp_pose(self): return self._pose(self._b_tcp)

    # ---------------------------------------------------------------- reset
    def reset(self, seed=None, options: Optional[dict] = None):
        options = options or {}
        on_device = self._reset_on_device(seed, options)
        if on_device is not None:
            return on_device
        dev = self.device
        env_idx = torch.as_tensor(options["env_idx"], device=dev, dtype=torch.long) if "env_idx" in options else torch.arange(self.num_envs, device=dev)
        idx_np = env_idx.cpu().numpy()
        self._host_reset_begins()
        self._fresh()
        if seed is not None:
            seeds = (np.asarray(seed).reshape(-1) if not np.isscalar(seed) else np.array([seed])).astype(np.int64)
            if len(seeds) == 1:
                seeds = seeds[0] + self.env_index_offset + idx_np
            self._main_seeds[idx_np] = seeds
            self._episode_count[idx_np] = 0
        self._rng.reseed(idx_np, self._main_seeds[idx_np].astype(np.uint64) * np.uint64(1000003) + self._episode_count[idx_np])
        self._episode_count[idx_np] += np.uint64(1)
        self._elapsed_steps[env_idx] = 0
        b = len(idx_np)
        f32 = getattr(self, "_stage", None)      # every host-made row of this reset goes through a pinned buffer of its own (envs/_stage.py)
        if f32 is None:
            from ._stage import HostStage
            f32 = self._stage = HostStage(dev, self.num_envs)
        f32.begin()
        off = self._offsets[env_idx]
        self._rbd[env_idx, self._b_tee, 7:13] = 0.0
        self._qvel[env_idx] = 0.0
        # WhiteTableSceneBuilder.initialize
        table = self._table_pose.repeat(b, 1); table[:, :3] += off
        self._rbd[env_idx, self._b_table, :7] = table
        qpos = self._rng.normal(idx_np, 7) * self.robot_init_qpos_noise + sb.PANDA_STICK_REST_QPOS
        self._qpos[env_idx, :7] = f32(qpos)
        root = self._root_pose.repeat(b, 1); root[:, :3] += off
        self._rbd[env_idx, self._b_root, :7] = root
        # PushTEnv._initialize_episode: goal tee fixed, tee uniform in the spawn box with a random z rotation
        gq = np.array([np.cos(self.goal_z_rot / 2), 0, 0, np.sin(self.goal_z_rot / 2)])
        goal = np.tile([self.goal_offset[0], self.goal_offset[1], 1e-3], (b, 1))
        self._rbd[env_idx, self._b_goal, :3] = f32(goal) + off
        self._rbd[env_idx, self._b_goal, 3:7] = f32(gq)
        u = self._rng.uniform(idx_np, 3)
        xyz = goal.copy()
        xyz[:, 0] += u[:, 0] * self.tee_spawnbox_xlength + self.tee_spawnbox_xoffset
        xyz[:, 1] += u[:, 1] * self.tee_spawnbox_ylength + self.tee_spawnbox_yoffset
        xyz[:, 2] = 0.04 / 2 + 1e-3
        ang = u[:, 2] * (2 * np.pi)
        q = np.zeros((b, 4)); q[:, 0] = np.cos(ang / 2); q[:, 3] = np.sin(ang / 2)
        self._rbd[env_idx, self._b_tee, :3] = f32(xyz) + off
        self._rbd[env_idx, self._b_tee, 3:7] = f32(q)
        self._rbd[env_idx, self._b_ee, :3] = f32(np.tile(self.ee_starting_pos2D, (b, 1))) + off
        self._rbd[env_idx, self._b_ee, 3:7] = f32([np.cos(np.pi / 4), 0, np.sin(np.pi / 4), 0])   # euler2quat(0, pi/2, 0)
        self._target_qpos[env_idx] = self._qpos[env_idx, :7]
        self._target_qpos_buf[env_idx, :7] = self._qpos[env_idx, :7]
        self.px.gpu_apply_all()
        self.px.gpu_update_articulation_kinematics()
        self.px.gpu_fetch_all()
        self._buffers_stale = False
        self._host_reset_ends(idx_np, reseeded=seed is not None)
        if self.fused:
            obs, _, _, _, info = self._fused_observe(False)
            return obs, info
        info = self.get_info()
        return self.get_obs(info), info

    # ---------------------------------------------------------------- step
    def _fused_observe(self, advance: bool):
        import ctypes as C
        L, px, N, dev = self.px.lib, self.px, self.num_envs, self.device
        od = self.obs_dim if self.camera is None else 21
        from ..graph import alloc_step_outputs
        obs, rew, fl, elapsed, _ = alloc_step_outputs(N, od, dev)
        L.check(px.ctx, L.task_pusht_observe(px.ctx, C.c_void_p(obs.data_ptr()), od, C.c_void_p(rew.data_ptr()), C.c_void_p(fl.data_ptr()),
                                             C.c_void_p(self._elapsed_steps.data_ptr()), 1 if advance else 0, px._stream()), "task_pusht_observe")
        elapsed.copy_(self._elapsed_steps)
        info = dict(elapsed_steps=elapsed, success=fl[:, 0])
        if self.camera is not None:
            self.camera.take_picture()
            obs = dict(state=obs, sensor_data=dict(base_camera=self.camera.get_obs(**self._textures)),
                       sensor_param=dict(base_camera=self.camera.get_params()))
        return obs, self._mode_reward(rew, info), fl[:, 4], fl[:, 5], info

    def _mode_reward(self, normalized, info):
        """BaseEnv.get_reward (sapien_env.py:648-670) from the normalized dense reward (max 3) and the success flag."""
        mode = self.reward_mode
        if mode == "normalized_dense":
            return normalized
        if mode == "dense":
            return normalized * 3.0
        if mode == "sparse":
            return info["success"].float()
        return torch.zeros_like(normalized)

    def _fused_step(self, action):
        import ctypes as C
        L, px = self.px.lib, self.px
        if action is not None:
            action = torch.as_tensor(action, dtype=torch.float32, device=self.device)
            if action.ndim == 1:
                action = action[None]
            if action.shape != (self.num_envs, self.action_dim):
                raise AssertionError(f"Received action of shape {tuple(action.shape)} but expected shape ({self.num_envs}, {self.action_dim})")
            action = action.contiguous()
            L.check(px.ctx, L.task_pusht_set_action(px.ctx, C.c_void_p(action.data_ptr()), px._stream()), "task_pusht_set_action")
        L.check(px.ctx, L.control_step(px.ctx, self._sim_steps_per_control, px._stream()), "control_step")
        self._buffers_stale = True
        return self._fused_observe(True)

    def enable_step_graph(self, warmup: int = 2):
        """One control step captured as a HIP graph (maniskill_amd/graph.py); call ``reset`` afterwards."""
        from ..graph import StepGraph
        self._step_graph = None
        self._step_graph = StepGraph(self._step_eager, self.num_envs, self.action_dim, self.device, warmup)
        return self._step_graph

    def disable_step_graph(self):
        self._step_graph = None

    def step(self, action):
        g = getattr(self, "_step_graph", None)
        if g is not None and action is not None:
            if self.fused:
                self._buffers_stale = True
            return g(action)
        return self._step_eager(action)

    def _step_eager(self, action):
        if self.fused:
            return self._fused_step(action)
        if action is not None:
            action = torch.as_tensor(action, dtype=torch.float32, device=self.device)
            if action.ndim == 1:
                action = action[None]
            if action.shape != (self.num_envs, self.action_dim):
                raise AssertionError(f"Received action of shape {tuple(action.shape)} but expected shape ({self.num_envs}, {self.action_dim})")
            a = torch.clip(action, -1.0, 1.0)
            self._target_qpos[:] = self.qpos + self.arm_delta * a
            self._target_qpos_buf[:, :7] = self._target_qpos
            self.px.gpu_apply_articulation_target_position()
        self.px.step_n(self._sim_steps_per_control)     # the substeps of one control step: nothing acts between them (msk_step_n)
        self.px.gpu_fetch_all()
        self._elapsed_steps += 1
        info = self.get_info()
        obs = self.get_obs(info)
        reward = self._mode_reward(self.compute_normalized_dense_reward(info), info)
        terminated = info["success"].clone()
        truncated = self._elapsed_steps >= self.max_episode_steps
        return obs, reward, terminated, truncated, info

    # ---------------------------------------------------------------- task
    def evaluate(self):
        return {"success": self.pseudo_render_intersection() >= self.intersection_thresh}

    def get_info(self):
        info = dict(elapsed_steps=self._elapsed_steps.clone())
        info.update(self.evaluate())
        return info

    def get_obs(self, info):
        tcp = self.tcp_pose
        if self.camera is None:   # state: agent (qpos, qvel) + extra (tcp_pose, goal_pos, obj_pose)
            return torch.hstack([self.qpos, self.qvel, tcp, self._pose(self._b_goal)[:, :3], self._pose(self._b_tee)])
        self.camera.take_picture()
        return dict(state=torch.hstack([self.qpos, self.qvel, tcp]), sensor_data=dict(base_camera=self.camera.get_obs(**self._textures)),
                    sensor_param=dict(base_camera=self.camera.get_params()))

    def compute_dense_reward(self, info):
        tee = self._pose(self._b_tee)
        rot_rew = (self.quat_to_z_euler(tee[:, 3:7]) - self.goal_z_rot).cos()
        reward = (((rot_rew + 1) / 2) ** 2) / 2
        d_goal = torch.linalg.norm(tee[:, 0:2] - self._pose(self._b_goal)[:, 0:2], dim=1)
        reward = reward + ((1 - torch.tanh(5 * d_goal)) ** 2) / 2
        d_tcp = torch.linalg.norm(tee[:, :3] - self.tcp_pose[:, :3], dim=1)
        reward = reward + ((1 - torch.tanh(5 * d_tcp)).sqrt()) / 20
        return torch.where(info["success"], torch.full_like(reward, 3.0), reward)

    def compute_normalized_dense_reward(self, info):
        return self.compute_dense_reward(info) / 3.0

    def get_state(self):
        self._fresh()

        def actor(bid):
            s = self._rbd[:, bid, :].clone()
            s[:, :3] -= self._offsets
            return s
        return torch.hstack([actor(self._b_table), actor(self._b_tee), actor(self._b_goal), actor(self._b_ee), actor(self._b_root), self.qpos, self.qvel])

    def set_state(self, state, env_idx=None):
        """BaseEnv.set_state (sapien_env.py:1299-1325): the layout get_state returns."""
        if env_idx is None:
            env_idx = torch.arange(self.num_envs, device=self.device)
        state = torch.as_tensor(state, dtype=torch.float32, device=self.device)
        self._fresh()
        off = self._offsets[env_idx]
        for k, bid in enumerate([self._b_table, self._b_tee, self._b_goal, self._b_ee, self._b_root]):
            s = state[:, 13 * k: 13 * (k + 1)].clone()
            s[:, :3] += off
            self._rbd[env_idx, bid, :] = s
        self._qpos[env_idx, :7] = state[:, 65:72]
        self._qvel[env_idx, :7] = state[:, 72:79]
        self.px.gpu_apply_all()
        self.px.gpu_update_articulation_kinematics()
        self.px.gpu_fetch_all()
        self._buffers_stale = False

    state_actor_names = ("table-workspace", "Tee", "goal_Tee", "goal_ee")
    state_articulation_name = "panda_stick"
    # the state-dict view of the flat state is the same code for every env of this package
    from .pick_cube import PickCubeEnv as _P
    state_layout, get_state_dict, set_state_dict = _P.state_layout, _P.get_state_dict, _P.set_state_dict
    del _P

    def close(self):
        self.px.close()
