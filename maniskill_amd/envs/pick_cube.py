"""PickCube-v1 on the MI355X-native backend: host-side mirror of the reference hot path.

Mirrors, with the same names and semantics (rows A1-A8 of SURVEY.md §8a):
  * BaseEnv.reset / step / _step_action / get_obs / get_state   mani_skill/envs/sapien_env.py:857-978,1042-1132,501-560,1272-1325
  * TimeLimitWrapper (max_episode_steps=50)                     mani_skill/utils/registration.py:127-168
  * PickCubeEnv                                                 mani_skill/envs/tasks/tabletop/pick_cube.py:35-195
  * Panda.is_grasping / is_static / tcp_pose                    mani_skill/agents/robots/panda/panda.py:237-277
  * pd_joint_delta_pos = PDJointPos(use_delta) arm + PDJointPosMimic gripper
                                                                mani_skill/agents/controllers/pd_joint_pos.py:76-93,207-228
  * TableSceneBuilder.initialize                                mani_skill/utils/scene_builder/table/scene_builder.py:67-103

Everything here is torch indexing over the backend's zero-copy buffers; the physics is
``self.px.step()`` (HIP kernels behind include/msk_physx.h).  Because rows are env-major,
``rigid_body_data.view(N, 18, 13)[:, body]`` is a strided view, not a gather.
"""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch

from ..graph import const

from ..physx import PhysxGpuSystem, SimConfig
from . import scene_builders as sb


# --------------------------------------------------------------------------------------
# counter-based RNG: partition-invariant episode randomness (seed 2022 + global env index,
# reference: sapien_env.py:321,327 and envs/utils/randomization/batched_rng.py:13-70)
# --------------------------------------------------------------------------------------
def _splitmix64(x: np.ndarray) -> np.ndarray:
    x = (x + np.uint64(0x9E3779B97F4A7C15)).astype(np.uint64)
    z = x
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


class BatchedRNG:
    """One independent stream per env, keyed by (seed, episode counter, draw counter)."""

    def __init__(self, seeds: np.ndarray):
        self.seeds = np.asarray(seeds, dtype=np.uint64)
        self.counter = np.zeros_like(self.seeds)

    def reseed(self, idx: np.ndarray, seeds: np.ndarray):
        self.seeds[idx] = np.asarray(seeds, dtype=np.uint64)
        self.counter[idx] = 0

    def _bits(self, idx: np.ndarray, n: int) -> np.ndarray:
        with np.errstate(over="ignore"):
            base = _splitmix64(self.seeds[idx] * np.uint64(0x2545F4914F6CDD1D) + self.counter[idx])
            k = np.arange(1, n + 1, dtype=np.uint64)[None, :]
            out = _splitmix64(base[:, None] + k * np.uint64(0xD1342543DE82EF95))
        self.counter[idx] += np.uint64(n)
        return out

    def uniform(self, idx: np.ndarray, n: int) -> np.ndarray:
        """(len(idx), n) float64 in [0, 1)."""
        return (self._bits(idx, n) >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)

    def normal(self, idx: np.ndarray, n: int) -> np.ndarray:
        u = self.uniform(idx, 2 * n)
        u1, u2 = 1.0 - u[:, :n], u[:, n:]
        return np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)


def _quat_to_y_axis(q: torch.Tensor) -> torch.Tensor:
    """Second column of the rotation matrix of wxyz quaternions (Pose.to_transformation_matrix()[..., :3, 1])."""
    w, x, y, z = q.unbind(-1)
    return torch.stack([2 * (x * y - w * z), 1 - 2 * (x * x + z * z), 2 * (y * z + w * x)], dim=-1)


def compute_angle_between(x1: torch.Tensor, x2: torch.Tensor) -> torch.Tensor:
    """utils/common.py:300-304 (normalize_vector uses eps 1e-8)."""
    n1 = x1 / torch.linalg.norm(x1, dim=1, keepdim=True).clamp_min(1e-8)
    n2 = x2 / torch.linalg.norm(x2, dim=1, keepdim=True).clamp_min(1e-8)
    return torch.arccos(torch.clip((n1 * n2).sum(1), -1, 1))


from ._device_reset import DeviceResetMixin      # noqa: E402


class PickCubeEnv(DeviceResetMixin):
    """PickCube-v1, state observations, ``pd_joint_delta_pos`` control, Panda.

    ``num_envs`` sub-scenes live on one device in one ``PhysxGpuSystem``.  The constructor
    argument ``px_factory`` exists for the test-suite (it injects the CPU oracle); the
    product path always builds ``PhysxGpuSystem`` on a GPU and raises if the HIP library
    is missing.
    """

    max_episode_steps = 50
    max_reward = 5.0
    camera_eye, camera_target = (0.3, 0.0, 0.6), (-0.1, 0.0, 0.1)   # base_camera (pick_cube.py:64-71)
    rest_qpos = sb.PANDA_REST_QPOS       # TableSceneBuilder.initialize keyframe (scene_builder/table/scene_builder.py:67-103)
    goal_thresh = 0.025
    grasp_max_angle = 85.0    # is_grasping's max_angle in the task's reward (panda.py:237)
    cube_half_size = 0.02
    cube_spawn_half_size = 0.1
    cube_spawn_center = (0.0, 0.0)
    max_goal_height = 0.3
    arm_delta = 0.1                      # arm_pd_joint_delta_pos lower/upper (panda.py:90-98)
    gripper_low, gripper_high = -0.01, 0.04  # gripper_pd_joint_pos (panda.py:177-185)
    action_dim = 8
    obs_dim = 42

    def __init__(self, num_envs: int = 1, device: Optional[str] = None, sim_config: Optional[SimConfig] = None,
                 robot_init_qpos_noise: float = 0.02, reward_mode: str = "normalized_dense",
                 env_index_offset: int = 0, total_envs: Optional[int] = None, px_factory=None,
                 fused: Optional[bool] = None, obs_mode: str = "state", control_mode: str = "pd_joint_delta_pos", device_reset: Optional[bool] = None):
        self.num_envs = int(num_envs)
        self.device_reset = device_reset      # partial resets from a device-side mask (envs/_device_reset.py); None: on for the fused envs on a GPU
        self.sim_config = sim_config or SimConfig()
        self.robot_init_qpos_noise = robot_init_qpos_noise
        if reward_mode not in ("normalized_dense", "dense", "sparse", "none"):
            raise NotImplementedError(f"reward_mode {reward_mode!r}: one of 'normalized_dense', 'dense', 'sparse', 'none' (sapien_env.py:648-670)")
        self.reward_mode = reward_mode
        # control modes of Panda._controller_configs (panda.py:187-200): joint deltas (default) or end-effector deltas through IK
        dims = {"pd_joint_delta_pos": 8, "pd_joint_pos": 8, "pd_joint_target_delta_pos": 8, "pd_joint_vel": 8,
                "pd_joint_pos_vel": 15, "pd_joint_delta_pos_vel": 15,
                "pd_ee_delta_pos": 4, "pd_ee_delta_pose": 7, "pd_ee_pose": 7, "pd_ee_target_delta_pos": 4, "pd_ee_target_delta_pose": 7}
        if control_mode not in dims:
            raise NotImplementedError(f"control_mode {control_mode!r}: this backend provides {sorted(dims)}")
        self.control_mode = control_mode
        self.action_dim = dims[control_mode]
        self.env_index_offset = int(env_index_offset)
        assert self.sim_config.sim_freq % self.sim_config.control_freq == 0
        self._sim_steps_per_control = self.sim_config.sim_freq // self.sim_config.control_freq
        tpl, ids = self._build_template(arm_stiffness=0.0 if control_mode == "pd_joint_vel" else None)
        self.template, self.ids = tpl, ids
        if px_factory is None:
            if device is None:
                device = "cuda"
            dev = torch.device(device)
            if dev.type != "cuda":
                raise RuntimeError("maniskill_amd runs its physics on an AMD GPU (device 'cuda[:k]'); there is no CPU backend")
            self.px = PhysxGpuSystem(dev, tpl, self.num_envs, self.sim_config)
        else:
            self.px = px_factory(tpl, self.num_envs, self.sim_config)
        self.device = self.px.device
        self.px.gpu_init()
        self._after_gpu_init()
        # sub-scene grid offsets (sapien_env.py:1191-1202)
        # a shard of a larger job (dist.py) keeps the GLOBAL grid cell of each env, so that the
        # published fp32 rows (env-frame pose + offset) do not depend on the partitioning
        side = int(np.ceil(np.sqrt(total_envs if total_envs is not None else self.num_envs)))
        g = np.arange(self.num_envs) + self.env_index_offset
        offsets = np.stack([(g % side - side // 2) * self.sim_config.spacing,
                            (g // side - side // 2) * self.sim_config.spacing, np.zeros(self.num_envs)], axis=1)
        self.px.set_scene_offsets(offsets)
        N, NB = self.num_envs, self.px.bodies_per_env
        self._rbd = self.px.cuda_rigid_body_data.torch().view(N, NB, 13)
        self._qpos = self.px.cuda_articulation_qpos.torch().view(N, -1)
        self._qvel = self.px.cuda_articulation_qvel.torch().view(N, -1)
        self._target_qpos_buf = self.px.cuda_articulation_target_qpos.torch().view(N, -1)
        b = tpl.body_id
        self._b_cube, self._b_goal, self._b_table = ids["cube"], ids["goal_site"], ids["table"]
        self._b_root = b("panda_link0")
        self._b_tcp = b("panda_hand_tcp")
        self._b_f1, self._b_f2 = b("panda_leftfinger"), b("panda_rightfinger")
        self._q_lgrasp = self.px.gpu_create_contact_pair_impulse_query([(self._b_f1, self._b_cube)])
        self._q_rgrasp = self.px.gpu_create_contact_pair_impulse_query([(self._b_f2, self._b_cube)])
        self._offsets = self.px.scene_offsets  # (N, 3)
        from ..structs import SceneView
        self.scene = SceneView(self.px, fresh=self._fresh)   # Actor / Link / Articulation views (structs.py)
        self.robot = self.scene.articulations[tpl.art_names[0]]
        from .. import spaces
        lim = np.array([tpl.joint_limits[b] for b in tpl.art_active[0]], dtype=np.float32)
        self.single_action_space = spaces.panda_action_space(control_mode, lim)
        self.action_space = spaces.batch_space(self.single_action_space, self.num_envs)
        self.single_observation_space = spaces.Box(-np.inf, np.inf, (self.obs_dim,), np.float32)    # state part (obs_mode "state")
        self.observation_space = spaces.batch_space(self.single_observation_space, self.num_envs)
        dev = self.device
        self._rest_qpos = torch.tensor(sb.PANDA_REST_QPOS, dtype=torch.float32, device=dev)
        self._table_pose = torch.tensor([-0.12, 0.0, -sb.TABLE_HEIGHT, np.cos(np.pi / 4), 0, 0, np.sin(np.pi / 4)],
                                        dtype=torch.float32, device=dev)
        self._root_pose = torch.tensor([-0.615, 0.0, 0.0, 1, 0, 0, 0], dtype=torch.float32, device=dev)
        self._elapsed_steps = torch.zeros(N, dtype=torch.int32, device=dev)
        self._target_qpos = torch.zeros(N, 9, dtype=torch.float32, device=dev)
        self._target_qvel_buf = self.px.cuda_articulation_target_qvel.torch().view(N, -1)
        self._target_pose = None
        self._main_seeds = 2022 + self.env_index_offset + np.arange(N)
        self._rng = BatchedRNG(self._main_seeds)
        self._episode_count = np.zeros(N, dtype=np.uint64)
        self.action_low = -torch.ones(self.action_dim, device=dev)
        self.action_high = torch.ones(self.action_dim, device=dev)
        # fused task kernels (include/msk_task.h): same arithmetic as the torch code below, two launches per step
        # instead of ~120.  Default on the HIP backend; the torch path stays the readable reference (fused=False).
        can_fuse = getattr(self.px.lib, "has_task_kernels", False)      # (the CPU oracle has none; the emulated HIP library of tests/hipemu does)
        self.fused = (can_fuse and not self.px.host_memory) if fused is None else bool(fused)
        self._setup_ee_controller()
        if self.fused:
            if not can_fuse:
                raise RuntimeError("fused task kernels need the HIP backend")
            from .. import _native as NN
            import ctypes as C
            d = NN.PickCubeDesc(cube=self._b_cube, goal=self._b_goal, tcp=self._b_tcp, left_finger=self._b_f1, right_finger=self._b_f2,
                                arm_dofs=7, arm_delta=self.arm_delta, gripper_mid=0.5 * (self.gripper_high + self.gripper_low),
                                gripper_half=0.5 * (self.gripper_high - self.gripper_low),
                                goal_thresh=self.goal_thresh, min_force=0.5, max_angle_deg=self.grasp_max_angle, static_thresh=0.2,
                                max_episode_steps=self.max_episode_steps)
            self.px.lib.check(self.px.ctx, self.px.lib.task_pickcube_init(self.px.ctx, C.byref(d)), "task_pickcube_init")
            self._init_fused_task()
        # sensors: PickCube-v1's base_camera (pick_cube.py:64-71), 128 x 128, fov pi/2, depth + segmentation textures
        if obs_mode not in ("state", "depth+segmentation", "rgb", "rgbd", "rgb+depth+segmentation"):
            raise NotImplementedError(f"obs_mode {obs_mode!r}: this backend provides 'state', 'depth+segmentation', 'rgb', 'rgbd' "
                                      "and 'rgb+depth+segmentation'")
        self.obs_mode = obs_mode
        # obs mode -> textures (sapien_env.py:120-160 parse_obs_mode_to_struct)
        self._textures = dict(rgb="rgb" in obs_mode, depth=("depth" in obs_mode or obs_mode == "rgbd"), segmentation="segmentation" in obs_mode)
        self._want_color = self._textures["rgb"]
        self.camera, self.cameras = None, {}
        if obs_mode != "state":
            from ..render import RenderCameraGroup, attach_template_visuals
            attach_template_visuals(self.px, tpl, hidden_bodies=self._hidden_bodies())
            for cfg in self._camera_configs(tpl):      # the env's sensors, then the agent's (sapien_env.py _setup_sensors)
                self.cameras[cfg.uid] = RenderCameraGroup(self.px, cfg)
                if self._want_color:
                    self.cameras[cfg.uid].enable_color()
                self.cameras[cfg.uid].set_outputs(position_texture=False)      # no obs mode of these envs hands out `position`
            self.camera = self.cameras["base_camera"]
        self.reset(seed=None)
        self._constructed = True

    def _camera_configs(self, tpl):
        from ..render import CameraConfig, look_at
        p, q = look_at(eye=self.camera_eye, target=self.camera_target)
        return [CameraConfig("base_camera", p, q, 128, 128, np.pi / 2, 0.01, 100.0)]

    def _with_sensor_data(self, state_obs):
        """_get_obs_with_sensor_data (sapien_env.py:627-634): update_render, take_picture, texture transforms."""
        if self.camera is None:
            return state_obs
        from .. import graph as _graph
        for cam in self.cameras.values():
            cam.take_picture()
        # inside a step-graph capture the replay snapshots every output once (graph._clone_tree): no first copy of the planes here
        return dict(state=state_obs,
                    sensor_data={uid: cam.get_obs(copy=not _graph.CAPTURING, **self._textures) for uid, cam in self.cameras.items()},
                    sensor_param={uid: cam.get_params() for uid, cam in self.cameras.items()})

    # ---------------------------------------------------------------- struct-style views
    def _fresh(self):
        """Fused mode publishes the sapien-style buffers lazily: refresh them before any host-side read."""
        if getattr(self, "fused", False) and getattr(self, "_buffers_stale", False):
            self.sync_buffers()

    def _pose(self, body):  # Actor.pose / Link.pose .raw_pose with the scene offset removed (actor.py:341-365)
        self._fresh()
        raw = self._rbd[:, body, :7].clone()
        raw[:, :3] -= self._offsets
        return raw

    @property
    def cube_pose(self): return self._pose(self._b_cube)
    @property
    def goal_pos(self):
        self._fresh()
        return self._rbd[:, self._b_goal, :3] - self._offsets
    @property
    def tcp_pose(self): return self._pose(self._b_tcp)
    @property
    def qpos(self):
        self._fresh()
        return self._qpos[:, :9]
    @property
    def qvel(self):
        self._fresh()
        return self._qvel[:, :9]

    # ---------------------------------------------------------------- reset
    def reset(self, seed=None, options: Optional[dict] = None):
        options = options or {}
        on_device = self._reset_on_device(seed, options)
        if on_device is not None:
            return on_device
        dev = self.device
        if "env_idx" in options:
            env_idx = torch.as_tensor(options["env_idx"], device=dev, dtype=torch.long)
        else:
            env_idx = torch.arange(self.num_envs, device=dev)
        idx_np = env_idx.cpu().numpy()
        self._host_reset_begins()
        if getattr(self, "fused", False) and getattr(self, "_buffers_stale", False):
            self.sync_buffers()   # the masked setters below write into the torch-visible buffers
        if seed is not None:
            seeds = (np.asarray(seed).reshape(-1) if not np.isscalar(seed) else np.array([seed])).astype(np.int64)
            if len(seeds) == 1:
                seeds = seeds[0] + self.env_index_offset + idx_np
            self._main_seeds[idx_np] = seeds
            self._episode_count[idx_np] = 0
        # episode seed = f(main seed, episode counter)
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
        # _clear_sim_state (sapien_env.py:1023-1036)
        self._rbd[env_idx, self._b_cube, 7:13] = 0.0
        self._qvel[env_idx] = 0.0
        if "reset_to_env_states" in options:
            self.set_state(options["reset_to_env_states"]["env_states"], env_idx)
        else:
            # TableSceneBuilder.initialize
            table = self._table_pose.repeat(b, 1)
            table[:, :3] += off
            self._rbd[env_idx, self._b_table, :7] = table
            qpos = self._rng.normal(idx_np, 9) * self.robot_init_qpos_noise + self.rest_qpos
            qpos[:, -2:] = 0.04
            self._qpos[env_idx, :9] = f32(qpos)
            root = self._root_pose.repeat(b, 1)
            root[:, :3] += off
            self._rbd[env_idx, self._b_root, :7] = root
            self._initialize_episode(env_idx, idx_np, off, f32)
        # controller.reset(): targets = current qpos (pd_joint_pos.py:54-69)
        self._target_qpos[env_idx] = self._qpos[env_idx, :9]
        self._target_qpos_buf[env_idx, :9] = self._qpos[env_idx, :9]
        if self.control_mode in ("pd_joint_vel", "pd_joint_pos_vel", "pd_joint_delta_pos_vel"):
            self._target_qvel_buf[env_idx] = 0.0
        self.px.gpu_apply_all()
        self.px.gpu_update_articulation_kinematics()
        self.px.gpu_fetch_all()
        self._host_reset_ends(idx_np, reseeded=seed is not None)
        if "target_delta" in self.control_mode and self.control_mode.startswith("pd_ee"):   # controller.reset(): target = current ee pose
            cur = self.ee_pose_at_base()
            if getattr(self, "_target_pose", None) is None:
                self._target_pose = cur.clone()
            else:
                self._target_pose[env_idx] = cur[env_idx]
        if getattr(self, "fused", False):
            obs, _, _, _, info = self._fused_observe(False)
            return obs, info
        info = self.get_info()
        obs = self._with_sensor_data(self.get_obs(info))
        return obs, info

    def _initialize_episode(self, env_idx, idx_np, off, f32):
        """PickCubeEnv._initialize_episode (pick_cube.py:106-130)."""
        b, dev = len(idx_np), self.device
        # all episode randomness is evaluated on the host in float64 and rounded once, so that the
        # initial state is bit-identical whatever device the env lives on
        u = self._rng.uniform(idx_np, 6)
        hs, cc = self.cube_spawn_half_size, self.cube_spawn_center
        xyz = np.zeros((b, 3))
        xyz[:, 0] = u[:, 0] * hs * 2 - hs + cc[0]
        xyz[:, 1] = u[:, 1] * hs * 2 - hs + cc[1]
        xyz[:, 2] = self.cube_half_size
        yaw = u[:, 2] * (2 * np.pi)  # random_quaternions(lock_x, lock_y): rotation about z
        qs = np.zeros((b, 4))
        qs[:, 0] = np.cos(yaw / 2)
        qs[:, 3] = np.sin(yaw / 2)
        self._rbd[env_idx, self._b_cube, :3] = f32(xyz) + off
        self._rbd[env_idx, self._b_cube, 3:7] = f32(qs)
        goal = np.zeros((b, 3))
        goal[:, 0] = u[:, 3] * hs * 2 - hs + cc[0]
        goal[:, 1] = u[:, 4] * hs * 2 - hs + cc[1]
        goal[:, 2] = u[:, 5] * self.max_goal_height + xyz[:, 2]
        self._rbd[env_idx, self._b_goal, :3] = f32(goal) + off
        self._rbd[env_idx, self._b_goal, 3:7] = torch.tensor([1.0, 0, 0, 0], device=dev)

    # ---------------------------------------------------------------- step
    def _set_action(self, action: torch.Tensor):
        """CombinedController.set_action -> arm PDJointPos(use_delta) + gripper PDJointPosMimic."""
        a = torch.clip(action, -1.0, 1.0)
        qpos = self.qpos
        # _clip_and_scale_action: 0.5*(high+low) + 0.5*(high-low)*a  (utils/gym_utils.py:104-107)
        self._target_qpos[:, :7] = qpos[:, :7] + self.arm_delta * a[:, :7]
        g = 0.5 * (self.gripper_high + self.gripper_low) + 0.5 * (self.gripper_high - self.gripper_low) * a[:, 7:8]
        self._target_qpos[:, 7:9] = g
        self._target_qpos_buf[:, :9] = self._target_qpos

    def _init_fused_task(self):
        """Hook: a task with its own fused evaluate / obs / reward kernel binds it here (the pickcube binding is in place)."""

    def _build_template(self, arm_stiffness=None):
        return sb.build_pick_cube_template(self.cube_half_size, arm_stiffness=arm_stiffness)

    def _hidden_bodies(self):
        return (self._b_goal,)   # goal_site is in _hidden_objects (pick_cube.py:104)

    def _after_gpu_init(self):
        """Hook: per-env instance parameters (tasks that build a different actor per sub-scene)."""

    # ---- end-effector control (agents/controllers/pd_ee_pose.py:24-262, utils/kinematics.py:185-259) ---------------------
    ee_pos_bound = 0.1      # pos_lower / pos_upper of arm_pd_ee_delta_pos(e) (panda.py:103-124)
    ee_rot_lower = -0.1     # rot_lower: the reference scales the (norm-clipped) rotation action by rot_LOWER (pd_ee_pose.py:236)
    ik_damping = 1e-4       # levenberg_marquardt lambda (kinematics.py:237)

    def _setup_ee_controller(self):
        """Joint frames of the 7 arm joints in their parent links (pose_in_parent of the template's add_link records)."""
        par, xp = [], []
        for op, a in self.template.ops:
            if op == "add_link" and a[2] != 0 and len(par) < 7:      # a = (art, parent, joint_type, pose_in_parent, ...)
                par.append(int(a[1]))
                xp.append([float(x) for x in a[3]])
        self._ee_parent = torch.tensor(par, dtype=torch.long, device=self.device)
        self._ee_xp = torch.tensor(xp, dtype=torch.float32, device=self.device)           # (7, 7) p, q(wxyz)

    @staticmethod
    def _qrot(q, v):
        """rotate v (..., 3) by quaternion q (..., 4 wxyz)"""
        w, u = q[..., :1], q[..., 1:]
        t = 2.0 * torch.cross(u, v, dim=-1)
        return v + w * t + torch.cross(u, t, dim=-1)

    @staticmethod
    def _qmul(a, b):
        w1, x1, y1, z1 = a.unbind(-1)
        w2, x2, y2, z2 = b.unbind(-1)
        q = torch.stack([w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2, w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2,
                         w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2, w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2], dim=-1)
        return torch.where(q[..., :1] < 0, -q, q)   # quaternion_multiply standardises to a non-negative real part (rotation_conversions.py)

    def ee_jacobian(self) -> torch.Tensor:
        """(N, 6, 7) geometric Jacobian [linear; angular] of panda_hand_tcp in the root link's frame (what
        pk_chain.jacobian(q) returns, kinematics.py:232): column k = [z_k x (p_ee - o_k); z_k] of arm joint k."""
        self._fresh()
        poses = self._rbd[:, :, :7]
        ppar = poses[:, self._ee_parent]                                     # (N, 7, 7) parent link poses (world)
        xp = self._ee_xp[None]
        o = ppar[..., :3] + self._qrot(ppar[..., 3:7], xp[..., :3].expand(self.num_envs, -1, -1))
        qj = self._qmul(ppar[..., 3:7], xp[..., 3:7].expand(self.num_envs, -1, -1))
        ex = torch.zeros_like(o); ex[..., 0] = 1.0
        z = self._qrot(qj, ex)                                               # joint axis = local x of the joint frame
        pee = poses[:, self._b_tcp, :3][:, None]
        Jv = torch.cross(z, pee - o, dim=-1)
        # into the root frame
        qr = poses[:, self._b_root, 3:7][:, None]
        qri = qr * const((1.0, -1.0, -1.0, -1.0), qr.device)
        Jv, Jw = self._qrot(qri, Jv), self._qrot(qri, z)
        return torch.cat([Jv, Jw], dim=-1).transpose(1, 2)                  # (N, 6, 7)

    def _ee_delta(self, action: torch.Tensor) -> torch.Tensor:
        """_clip_and_scale_action of PDEEPos / PDEEPoseController (pd_ee_pose.py:224-237): (N, 6) delta pose in the root frame."""
        pos = self.ee_pos_bound * torch.clip(action[:, :3], -1.0, 1.0)
        if self.control_mode in ("pd_ee_delta_pos", "pd_ee_target_delta_pos"):   # PDEEPosController: translation only
            return torch.hstack([pos, torch.zeros_like(pos)])
        rot = action[:, 3:6].clone()
        nrm = torch.linalg.norm(rot, dim=1)
        rot[nrm > 1] = (rot / nrm[:, None])[nrm > 1]
        return torch.hstack([pos, rot * self.ee_rot_lower])

    def _set_action_ee(self, action: torch.Tensor):
        """compute_ik with is_delta_pose (kinematics.py:229-245): one Levenberg-Marquardt step, target = q0 + dq."""
        delta = self._ee_delta(action)
        J = self.ee_jacobian()
        # the reference solves (J^T J + lambda I) dq = J^T delta (7 x 7, rank 6 up to lambda = 1e-4: rounding in the null-space direction
        # is amplified 1e4-fold); the identical step in its dual form dq = J^T (J J^T + lambda I)^-1 delta is a well-conditioned 6 x 6
        # system whose solution lies in J's row space by construction -- what the fused kernel computes too, so the two agree to 1e-5
        JT = J.transpose(1, 2)
        M = torch.bmm(J, JT) + self.ik_damping * torch.eye(6, device=J.device)
        dq = torch.bmm(JT, torch.linalg.solve(M, delta.unsqueeze(-1))).squeeze(-1)
        na = action.shape[1]
        self._target_qpos[:, :7] = self.qpos[:, :7] + dq
        g = 0.5 * (self.gripper_high + self.gripper_low) + 0.5 * (self.gripper_high - self.gripper_low) * torch.clip(action[:, na - 1:na], -1.0, 1.0)
        self._target_qpos[:, 7:9] = g
        self._target_qpos_buf[:, :9] = self._target_qpos

    # ---- the other controllers of Panda._controller_configs (panda.py:81-211) --------------------------------------------
    def _gripper_target(self, a_last):
        return 0.5 * (self.gripper_high + self.gripper_low) + 0.5 * (self.gripper_high - self.gripper_low) * torch.clip(a_last, -1.0, 1.0)

    @staticmethod
    def _euler_xyz_to_quat(r):
        """matrix_to_quaternion(euler_angles_to_matrix(r, "XYZ")): R = Rx(a) Ry(b) Rz(c)  ->  q = qx * qy * qz (wxyz)."""
        h = 0.5 * r
        cx, cy, cz, sx, sy, sz = h[:, 0].cos(), h[:, 1].cos(), h[:, 2].cos(), h[:, 0].sin(), h[:, 1].sin(), h[:, 2].sin()
        return torch.stack([cx * cy * cz - sx * sy * sz, sx * cy * cz + cx * sy * sz, cx * sy * cz - sx * cy * sz, cx * cy * sz + sx * sy * cz], dim=-1)

    @classmethod
    def _quat_to_euler_xyz(cls, q):
        """matrix_to_euler_angles(quaternion_to_matrix(q), "XYZ")."""
        w, x, y, z = q.unbind(-1)
        r02 = 2 * (x * z + w * y)
        r12, r22 = 2 * (y * z - w * x), 1 - 2 * (x * x + y * y)
        r01, r00 = 2 * (x * y - w * z), 1 - 2 * (y * y + z * z)
        return torch.stack([torch.atan2(-r12, r22), torch.asin(torch.clamp(r02, -1.0, 1.0)), torch.atan2(-r01, r00)], dim=-1)

    def ee_pose_at_base(self):
        """to_base * ee_pose (pd_ee_pose.py:70-73): (N, 7) tcp pose in the root link's frame."""
        self._fresh()
        root, tcp = self._rbd[:, self._b_root, :7], self._rbd[:, self._b_tcp, :7]
        qri = root[:, 3:7] * const((1.0, -1.0, -1.0, -1.0), root.device)
        return torch.cat([self._qrot(qri, tcp[:, :3] - root[:, :3]), self._qmul(qri, tcp[:, 3:7])], dim=-1)

    def _set_action_any(self, action: torch.Tensor):
        """CombinedController.set_action for the control mode of this env; returns which target buffer to commit."""
        mode, na = self.control_mode, action.shape[1]
        if mode == "pd_joint_delta_pos":
            self._set_action(action)
        elif mode in ("pd_ee_delta_pos", "pd_ee_delta_pose"):
            self._set_action_ee(action)
        elif mode == "pd_joint_pos":          # absolute arm targets, not normalised (normalize_action=False, panda.py:81-89)
            self._target_qpos[:, :7] = action[:, :7]
            self._target_qpos[:, 7:9] = self._gripper_target(action[:, 7:8])
        elif mode == "pd_joint_target_delta_pos":   # use_target: the delta accumulates on the previous target (pd_joint_pos.py:84-87)
            self._target_qpos[:, :7] = self._target_qpos[:, :7] + self.arm_delta * torch.clip(action[:, :7], -1.0, 1.0)
            self._target_qpos[:, 7:9] = self._gripper_target(action[:, 7:8])
        elif mode == "pd_joint_vel":          # PDJointVelController: velocity targets in [-1, 1] rad/s on damping-only drives
            self._target_qvel_buf[:, :7] = torch.clip(action[:, :7], -1.0, 1.0)
            self._target_qpos[:, 7:9] = self._gripper_target(action[:, 7:8])
            self._target_qpos[:, :7] = self.qpos[:, :7]       # stiffness 0: the position target of the arm joints is inert
        elif mode in ("pd_joint_pos_vel", "pd_joint_delta_pos_vel"):   # PDJointPosVelController (pd_joint_pos_vel.py:40-66): [pos 7 | vel 7 | gripper]
            if mode == "pd_joint_pos_vel":        # absolute, not normalised
                self._target_qpos[:, :7] = action[:, :7]
                self._target_qvel_buf[:, :7] = action[:, 7:14]
            else:                                 # normalised: position deltas in +-0.1 rad, velocities in +-1 rad/s
                a = torch.clip(action[:, :14], -1.0, 1.0)
                self._target_qpos[:, :7] = self.qpos[:, :7] + self.arm_delta * a[:, :7]
                self._target_qvel_buf[:, :7] = a[:, 7:14]
            self._target_qpos[:, 7:9] = self._gripper_target(action[:, 14:15])
        else:                                 # virtual / absolute target pose in the root frame (pd_ee_pose.py:104-129,239-262)
            if mode == "pd_ee_pose":              # use_delta=False, normalize_action=False: [xyz | XYZ euler] of the target itself
                target = torch.cat([action[:, :3], self._euler_xyz_to_quat(action[:, 3:6])], dim=-1)
            else:                                 # pd_ee_target_delta_pos / pose: the delta accumulates on the previous target
                delta = self._ee_delta(action)
                prev = self._target_pose
                q = self._qmul(self._euler_xyz_to_quat(delta[:, 3:6]), prev[:, 3:7])    # root_aligned_body_rotation
                target = torch.cat([prev[:, :3] + delta[:, :3], q], dim=-1)            # root_translation
            self._target_pose = target
            cur = self.ee_pose_at_base()
            qci = cur[:, 3:7] * const((1.0, -1.0, -1.0, -1.0), cur.device)
            d6 = torch.cat([target[:, :3] - cur[:, :3], self._quat_to_euler_xyz(self._qmul(target[:, 3:7], qci))], dim=-1)   # kinematics.py:218-228
            J = self.ee_jacobian()
            JT = J.transpose(1, 2)
            A = torch.bmm(JT, J) + self.ik_damping * torch.eye(7, device=J.device)
            dq = torch.linalg.solve(A, torch.bmm(JT, d6.unsqueeze(-1))).squeeze(-1)
            self._target_qpos[:, :7] = self.qpos[:, :7] + dq
            self._target_qpos[:, 7:9] = self._gripper_target(action[:, na - 1:na])
        self._target_qpos_buf[:, :9] = self._target_qpos
        self.px.gpu_apply_articulation_target_position()
        if mode in ("pd_joint_vel", "pd_joint_pos_vel", "pd_joint_delta_pos_vel"):
            self.px.gpu_apply_articulation_target_velocity()

    def _step_action(self, action):
        if action is not None:
            action = torch.as_tensor(action, dtype=torch.float32, device=self.device)
            if action.ndim == 1:
                action = action[None]
            if action.shape != (self.num_envs, self.action_dim):
                raise AssertionError(f"Received action of shape {tuple(action.shape)} but expected shape ({self.num_envs}, {self.action_dim})")
            self._set_action_any(action)
        self.px.step_n(self._sim_steps_per_control)     # the substeps of one control step: nothing acts between them (msk_step_n)
        self.px.gpu_fetch_all()
        return action

    def _fused_observe(self, advance: bool):
        import ctypes as C
        L, px = self.px.lib, self.px
        # the kernel writes straight into this step's fresh output tensors (no staging buffers, no copies); the flag bytes
        # are 0 / 1, i.e. valid torch.bool storage
        N, dev = self.num_envs, self.device
        from ..graph import alloc_step_outputs
        obs, rew, fl, elapsed, _ = alloc_step_outputs(N, self.obs_dim, dev)     # one allocation: a graph replay snapshots it with one copy
        L.check(px.ctx, L.task_pickcube_observe(px.ctx, C.c_void_p(obs.data_ptr()), C.c_void_p(rew.data_ptr()),
                                                C.c_void_p(fl.data_ptr()), C.c_void_p(self._elapsed_steps.data_ptr()),
                                                1 if advance else 0, px._stream()), "task_pickcube_observe")
        elapsed.copy_(self._elapsed_steps)
        info = dict(elapsed_steps=elapsed, success=fl[:, 0], is_obj_placed=fl[:, 1], is_robot_static=fl[:, 2],
                    is_grasped=fl[:, 3])
        return self._with_sensor_data(obs), self._fused_reward(rew, info), fl[:, 4], fl[:, 5], info

    def _fused_step(self, action):
        """BaseEnv.step on the fused kernels: controller, substeps + link frames, evaluate/obs/reward."""
        import ctypes as C
        L, px = self.px.lib, self.px
        if action is not None:
            action = torch.as_tensor(action, dtype=torch.float32, device=self.device)
            if action.ndim == 1:
                action = action[None]
            if action.shape != (self.num_envs, self.action_dim):
                raise AssertionError(f"Received action of shape {tuple(action.shape)} but expected shape ({self.num_envs}, {self.action_dim})")
            action = action.contiguous()
            if self.control_mode == "pd_joint_delta_pos":
                L.check(px.ctx, L.task_pickcube_set_action(px.ctx, C.c_void_p(action.data_ptr()), px._stream()), "task_pickcube_set_action")
            elif self.control_mode not in ("pd_ee_delta_pos", "pd_ee_delta_pose"):   # the less common controllers stay in torch
                self._fresh()
                self._set_action_any(action)
            else:   # end-effector control: Jacobian + Levenberg-Marquardt step per env in one kernel
                L.check(px.ctx, L.task_pickcube_set_action_ee(px.ctx, C.c_void_p(action.data_ptr()), self.action_dim, self._b_root,
                                                              self.ee_pos_bound, self.ee_rot_lower, self.ik_damping, px._stream()),
                        "task_pickcube_set_action_ee")
        L.check(px.ctx, L.control_step(px.ctx, self._sim_steps_per_control, px._stream()), "control_step")
        self._buffers_stale = True   # the sapien-style external buffers are refreshed on demand (sync_buffers)
        return self._fused_observe(True)

    def sync_buffers(self):
        """Publishes the simulator state into the torch-visible sapien buffers (scene._gpu_fetch_all)."""
        self.px.gpu_fetch_all()
        self._buffers_stale = False

    def enable_step_graph(self, warmup: int = 2):
        """Captures one control step (controller, substeps, fetch, task code, camera) as a HIP graph; ``step`` then
        replays it with a single launch (maniskill_amd/graph.py).  Runs ``warmup`` + 1 throw-away steps: call
        ``reset`` afterwards.  Needs a GPU env."""
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
        action = self._step_action(action)
        self._elapsed_steps += 1
        info = self.get_info()
        sobs = self.get_obs(info)
        reward = self.get_reward(sobs, action, info)
        obs = self._with_sensor_data(sobs)
        terminated = info["success"].clone()
        truncated = self._elapsed_steps >= self.max_episode_steps  # TimeLimitWrapper
        return obs, reward, terminated, truncated, info

    # ---------------------------------------------------------------- task
    def get_pairwise_contact_forces(self, query):
        self.px.gpu_query_contact_pair_impulses(query)
        return query.cuda_impulses.torch().clone() / self.px.timestep

    def is_grasping(self, min_force=0.5, max_angle=85):
        self._fresh()
        lf = self.get_pairwise_contact_forces(self._q_lgrasp)
        rf = self.get_pairwise_contact_forces(self._q_rgrasp)
        lforce, rforce = torch.linalg.norm(lf, dim=1), torch.linalg.norm(rf, dim=1)
        ldir = _quat_to_y_axis(self._rbd[:, self._b_f1, 3:7])
        rdir = -_quat_to_y_axis(self._rbd[:, self._b_f2, 3:7])
        langle = compute_angle_between(ldir, lf)
        rangle = compute_angle_between(rdir, rf)
        lflag = torch.logical_and(lforce >= min_force, torch.rad2deg(langle) <= max_angle)
        rflag = torch.logical_and(rforce >= min_force, torch.rad2deg(rangle) <= max_angle)
        return torch.logical_and(lflag, rflag)

    def is_static(self, threshold=0.2):
        return torch.max(torch.abs(self.qvel[:, :-2]), 1)[0] <= threshold

    def evaluate(self):
        is_obj_placed = torch.linalg.norm(self.goal_pos - self.cube_pose[:, :3], dim=1) <= self.goal_thresh
        is_grasped = self.is_grasping()
        is_robot_static = self.is_static(0.2)
        return {"success": is_obj_placed & is_robot_static, "is_obj_placed": is_obj_placed,
                "is_robot_static": is_robot_static, "is_grasped": is_grasped}

    def get_info(self):
        info = dict(elapsed_steps=self._elapsed_steps.clone())
        info.update(self.evaluate())
        return info

    def get_obs(self, info):
        """flatten_state_dict(dict(agent=dict(qpos, qvel), extra=dict(...))) -> (N, 42) (pick_cube.py:132-145)."""
        cube, tcp, goal = self.cube_pose, self.tcp_pose, self.goal_pos
        return torch.hstack([
            self.qpos, self.qvel, info["is_grasped"][:, None].float(), tcp, goal, cube,
            cube[:, :3] - tcp[:, :3], goal - cube[:, :3],
        ])

    def compute_dense_reward(self, obs, action, info):
        cube_p, tcp_p, goal = self.cube_pose[:, :3], self.tcp_pose[:, :3], self.goal_pos
        reaching_reward = 1 - torch.tanh(5 * torch.linalg.norm(cube_p - tcp_p, dim=1))
        reward = reaching_reward
        is_grasped = info["is_grasped"]
        reward = reward + is_grasped
        place_reward = 1 - torch.tanh(5 * torch.linalg.norm(goal - cube_p, dim=1))
        reward = reward + place_reward * is_grasped
        static_reward = 1 - torch.tanh(5 * torch.linalg.norm(self.qvel[:, :-2], dim=1))
        reward = reward + static_reward * info["is_obj_placed"]
        reward = torch.where(info["success"], torch.full_like(reward, 5.0), reward)
        return reward

    def get_reward(self, obs, action, info):
        """BaseEnv.get_reward (sapien_env.py:648-670): 'normalized_dense' (default), 'dense', 'sparse' (success - fail as float;
        compute_sparse_reward :672-697) or 'none'."""
        mode = self.reward_mode
        if mode == "none":
            return torch.zeros(self.num_envs, dtype=torch.float32, device=self.device)
        if mode == "sparse":
            r = info["success"].float() if "success" in info else torch.zeros(self.num_envs, device=self.device)
            return r - info["fail"].float() if "fail" in info else r
        if mode not in ("dense", "normalized_dense"):
            raise NotImplementedError(mode)
        r = self.compute_dense_reward(obs, action, info)
        return r / self.max_reward if mode == "normalized_dense" else r

    def _fused_reward(self, rew, info):
        """The fused task kernels pay the normalized dense reward; the other modes follow from it and the success flag."""
        mode = self.reward_mode
        if mode == "normalized_dense":
            return rew
        if mode == "dense":
            return rew * self.max_reward
        if mode == "sparse":
            return info["success"].float()
        if mode == "none":
            return torch.zeros_like(rew)
        raise NotImplementedError(mode)

    # ---------------------------------------------------------------- state (sapien_env.py:1272-1325)
    def get_state(self):
        """(N, 13*3 + 13 + 9*2): actors [table, cube, goal] then articulation [root pose/vel, qpos, qvel]
        (tests/test_sim_state.py:10-37)."""
        if getattr(self, "fused", False) and getattr(self, "_buffers_stale", False):
            self.sync_buffers()

        def actor(bid):
            s = self._rbd[:, bid, :].clone()
            s[:, :3] -= self._offsets
            return s
        root = actor(self._b_root)
        return torch.hstack([actor(b) for b in self._state_actor_bodies()] + [root, self.qpos, self.qvel])

    def _state_actor_bodies(self):
        """Body ids behind ``state_actor_names`` (every non-static actor of the scene, in build order)."""
        return [self._b_table, self._b_cube, self._b_goal]

    def set_state(self, state, env_idx=None):
        if env_idx is None:
            env_idx = torch.arange(self.num_envs, device=self.device)
        state = torch.as_tensor(state, dtype=torch.float32, device=self.device)
        if getattr(self, "fused", False) and getattr(self, "_buffers_stale", False):
            self.sync_buffers()
        off = self._offsets[env_idx]
        bodies = self._state_actor_bodies() + [self._b_root]
        for k, bid in enumerate(bodies):
            s = state[:, 13 * k: 13 * (k + 1)].clone()
            s[:, :3] += off
            self._rbd[env_idx, bid, :] = s
        o = 13 * len(bodies)
        self._qpos[env_idx, :9] = state[:, o:o + 9]
        self._qvel[env_idx, :9] = state[:, o + 9:o + 18]
        self.px.gpu_apply_all()
        self.px.gpu_update_articulation_kinematics()
        self.px.gpu_fetch_all()

    # names of the state-dict entries, in get_state order (scene.get_sim_state: actors in build order, then articulations)
    state_actor_names = ("table-workspace", "cube", "goal_site")
    state_articulation_name = "panda"

    def state_layout(self):
        """[(group, name, start, size)] of the flat state vector: 13 per actor, 13 + 2 * dof for the articulation."""
        out, start = [], 0
        for name in self.state_actor_names:
            out.append(("actors", name, start, 13)); start += 13
        ndof = self.get_state().shape[1] - start - 13
        out.append(("articulations", self.state_articulation_name, start, 13 + ndof))
        return out

    def get_state_dict(self):
        """BaseEnv.get_state_dict (sapien_env.py:1272-1283): {"actors": {name: (N, 13)}, "articulations": {name: (N, 13 + 2 dof)}}."""
        flat = self.get_state()
        out = {"actors": {}, "articulations": {}}
        for group, name, start, size in self.state_layout():
            out[group][name] = flat[:, start:start + size].clone()
        return out

    def set_state_dict(self, state: dict, env_idx=None):
        """BaseEnv.set_state_dict (sapien_env.py:1293-1303); entries that are missing keep their current value."""
        flat = self.get_state()
        if env_idx is not None:
            flat = flat[torch.as_tensor(env_idx, device=self.device, dtype=torch.long)]
        for group, name, start, size in self.state_layout():
            if group in state and name in state[group]:
                v = torch.as_tensor(state[group][name], dtype=torch.float32, device=self.device)
                flat[:, start:start + size] = v if v.ndim == 2 else v[None]
        self.set_state(flat, env_idx)

    def close(self):
        self.px.close()
