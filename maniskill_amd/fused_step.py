"""A fused control step for envs built by the reference's own code (``gym.make`` of mani_skill over the sapien shim).

The drop-in path is bound by the host: an unmodified ``BaseEnv.step`` of OpenCabinetDrawer-v1 issues ~320 eager torch ops and 28
boundary calls per control step (tools: the op count of a step under a TorchDispatchMode), and ``Articulation.set_joint_drive_targets``
indexes with a boolean mask (utils/structs/articulation.py:888-893: ``gx[self.scene._reset_mask[...]]`` -- a device->host
synchronisation), so the step cannot be captured into a HIP graph either.  ``accelerate(env)`` takes the step of such an env over:

  * ``FusedControl`` -- ``BaseEnv._step_action`` (envs/sapien_env.py:1073-1132) for the controller classes the robots of the benchmarked tasks use
    (PDJointPos [delta | target | absolute], PDJointPosMimic, PDJointVel, PDBaseVel, PDBaseForwardVel, Passive:
    agents/controllers/pd_joint_pos.py:76-93,207-228, pd_joint_vel.py:40-44, pd_base_vel.py:18-72): one clip-and-scale over the whole action,
    one write per target buffer with index tensors prepared once (no mask), the physics steps.  Works on any task; the task's own
    evaluate / observation / reward code stays the reference's.
  * task plugins (``OpenCabinetDrawerStep``, ``PickCubeStep``) -- the rest of ``BaseEnv.step`` (sapien_env.py:1042-1071) for tasks whose evaluate / obs / reward
    were restated here without boolean-mask indexing: the whole control step then holds no synchronisation and can be replayed as one HIP graph
    (``accelerate(env, graph=True)``; maniskill_amd/graph.py).

Everything is written against the SAPIEN API the reference uses (``px.cuda_*`` buffers, ``px.gpu_apply_* / gpu_fetch_* / step``) and the reference's
struct attributes; where the backend offers its masked apply / fetch extension (shim/_system.py: gpu_apply_masked) the eight apply and seven fetch calls
of ``scene._gpu_apply_all`` / ``_gpu_fetch_all`` are one boundary call each.  Results are the reference's: tests/test_fused_step.py steps an accelerated env
and an unmodified twin side by side (simulation state bit-equal, observation / reward / flags equal).
"""
from __future__ import annotations

import torch


class Unsupported(NotImplementedError):
    """accelerate(): this env uses something the fused step does not restate; the env is left untouched."""


def _overridden(obj, name, base_cls):
    return getattr(type(obj), name) is not getattr(base_cls, name)


# --------------------------------------------------------------------------------------------------------------------- boundary helpers
class _Boundary:
    """scene._gpu_apply_all / _gpu_fetch_all (envs/scene.py:950-986) with the backend's masked calls where it has them."""

    def __init__(self, scene):
        self.scene, self.px = scene, scene.px
        self.masked = hasattr(self.px, "gpu_apply_masked") and hasattr(self.px, "gpu_fetch_masked")
        self.fetch_mask = 0
        if self.masked:
            if len(scene.non_static_actors) > 0:
                self.fetch_mask |= 1
            if len(scene.articulations) > 0:
                self.fetch_mask |= self.px.FETCH_ALL_MASK

    def apply_all(self):
        if self.masked:
            assert not self.scene._needs_fetch
            self.px.gpu_apply_masked(self.px.APPLY_ALL_MASK)
            self.scene._needs_fetch = True
        else:
            self.scene._gpu_apply_all()

    def fetch_all(self):
        if self.masked:
            self.px.gpu_fetch_masked(self.fetch_mask)
            self.scene._needs_fetch = False
        else:
            self.scene._gpu_fetch_all()


# --------------------------------------------------------------------------------------------------------------------- controllers
class FusedControl:
    """``BaseEnv._step_action`` for one agent whose controller is a CombinedController / single controller of the supported classes."""

    def __init__(self, base):
        from mani_skill.agents.controllers.base_controller import CombinedController, DictController
        from mani_skill.agents.controllers.passive_controller import PassiveController
        from mani_skill.agents.controllers.pd_base_vel import PDBaseForwardVelController, PDBaseVelController
        from mani_skill.agents.controllers.pd_joint_pos import PDJointPosController, PDJointPosMimicController
        from mani_skill.agents.controllers.pd_joint_vel import PDJointVelController
        from mani_skill.agents.multi_agent import MultiAgent
        from mani_skill.envs.sapien_env import BaseEnv

        self.base = base
        agent = base.agent
        if agent is None or isinstance(agent, MultiAgent):
            raise Unsupported("one agent per env")
        if not base.gpu_sim_enabled:
            raise Unsupported("the batched (GPU) simulation path only")
        # the task's hooks run where the reference runs them (sapien_env.py:1122-1128); whether a step with them may be replayed as a graph is the watch's call
        self.hook_before_control = _overridden(base, "_before_control_step", BaseEnv)
        self.hooks_per_substep = _overridden(base, "_before_simulation_step", BaseEnv) or _overridden(base, "_after_simulation_step", BaseEnv)
        ctrl = agent.controller
        if isinstance(ctrl, CombinedController):
            subs = [(c, ctrl.action_mapping[uid]) for uid, c in ctrl.controllers.items()]
        elif isinstance(ctrl, DictController):
            raise Unsupported("dict action spaces")
        else:
            subs = [(ctrl, (0, ctrl.single_action_space.shape[0]))]
        self.ctrl, self.robot, self.scene, self.px = ctrl, agent.robot, base.scene, base.scene.px
        dev = base.device
        self.adim = sum(e - s for _, (s, e) in subs)
        self.sim_steps = base._sim_steps_per_control
        # one clip-and-scale for the whole action: dims of controllers that do not normalise pass through (bounds +-inf, 0 + 1 * a)
        lo = torch.full((self.adim,), -float("inf"), device=dev)
        hi = torch.full((self.adim,), float("inf"), device=dev)
        c0 = torch.zeros(self.adim, device=dev)
        c1 = torch.ones(self.adim, device=dev)
        self.any_norm = False
        self.plan = []          # (kind, controller, slice, ...)
        pos_cols, vel_cols = [], []
        for c, (s, e) in subs:
            if getattr(c, "_normalize_action", False):
                self.any_norm = True
                low, high = c.action_space_low.to(dev), c.action_space_high.to(dev)
                lo[s:e], hi[s:e] = -1.0, 1.0
                c0[s:e] = 0.5 * (high + low)            # gym_utils.clip_and_scale_action: 0.5 * (high + low) + 0.5 * (high - low) * action
                c1[s:e] = 0.5 * (high - low)
            cols = torch.as_tensor(c.active_joint_indices, device=dev).long()
            t = type(c)
            if t is PassiveController or (e - s == 0 and not getattr(c, "sets_target_qpos", False) and not getattr(c, "sets_target_qvel", False)):
                continue
            if t is PDJointPosMimicController:
                if c.config.interpolate:
                    raise Unsupported("interpolated targets")
                self.plan.append(("mimic", c, s, e, cols, len(pos_cols)))
                pos_cols.append(cols)
            elif t is PDJointPosController:
                if c.config.interpolate:
                    raise Unsupported("interpolated targets")
                self.plan.append(("pos", c, s, e, cols, len(pos_cols)))
                pos_cols.append(cols)
            elif t is PDJointVelController:
                self.plan.append(("vel", c, s, e, cols, len(vel_cols)))
                vel_cols.append(cols)
            elif t in (PDBaseVelController, PDBaseForwardVelController):
                self.plan.append(("base" if t is PDBaseVelController else "base_fwd", c, s, e, cols, int(cols[2])))     # the 3rd joint is the orientation
                vel_cols.append(cols)
            else:
                raise Unsupported(f"controller {t.__name__}")
        self.lo, self.hi, self.c0, self.c1 = lo, hi, c0, c1
        rows = self.robot._data_index.long()
        self.rows = rows
        self.max_dof = self.robot.max_dof
        self.pos_gx = self.pos_gy = self.vel_gx = self.vel_gy = None
        if pos_cols:
            self.pos_gx, self.pos_gy = torch.meshgrid(rows, torch.cat(pos_cols), indexing="ij")
        if vel_cols:
            self.vel_gx, self.vel_gy = torch.meshgrid(rows, torch.cat(vel_cols), indexing="ij")
        self.sets_qpos, self.sets_qvel = bool(ctrl.sets_target_qpos), bool(ctrl.sets_target_qvel)

    @staticmethod
    def _keep(c, tgt):
        """the controller's target state, updated IN PLACE where it exists: a captured step reads and writes the same memory at every replay, and the
        reference's controller.reset() writes the rows of re-initialised envs into that very tensor (pd_joint_pos.py:57-69)"""
        cur = c._target_qpos
        if isinstance(cur, torch.Tensor) and cur.shape == tgt.shape and cur.dtype == tgt.dtype:
            cur.copy_(tgt)
        else:
            c._target_qpos = tgt

    def set_action(self, action):
        """agent.set_action(action): drive targets into px.cuda_articulation_target_qpos / _qvel (not yet applied)."""
        if self.any_norm:
            a = self.c0 + self.c1 * torch.clip(action, self.lo, self.hi)
        else:
            a = action
        q = self.px.cuda_articulation_qpos.torch()[self.rows, :self.max_dof]       # robot.get_qpos(): the state of the last fetch
        pos, vel = [], []
        for kind, c, s, e, cols, ori_col in self.plan:
            act = a[:, s:e]
            if kind == "pos":
                start = q[:, cols]
                if c.config.use_delta:
                    tgt = (c._target_qpos + act) if c.config.use_target else (start + act)
                else:
                    tgt = torch.broadcast_to(act, start.shape).clone()
                c._step, c._start_qpos = self.sim_steps, start
                self._keep(c, tgt)
                pos.append(tgt)
            elif kind == "mimic":
                start = q[:, cols]
                tgt = c._target_qpos.clone()        # persists across steps (pd_joint_pos.py:207-223)
                ci = c.control_joint_indices
                if c.config.use_delta:
                    tgt[:, ci] = (tgt[:, ci] + act) if c.config.use_target else (start[:, ci] + act)
                else:
                    tgt[:, ci] = act
                tgt[:, c.mimic_joint_indices] = tgt[:, c.mimic_control_joint_indices] * c._multiplier[None, :] + c._offset[None, :]
                c._step, c._start_qpos = self.sim_steps, start
                self._keep(c, tgt)
                pos.append(tgt)
            elif kind == "vel":
                vel.append(act)
            else:                                    # ego-centric base velocity (pd_base_vel.py:18-72)
                ori = q[:, ori_col]
                cs, sn = torch.cos(ori), torch.sin(ori)
                ax = act[:, 0].float()
                ay = act[:, 1].float() if kind == "base" else torch.zeros_like(ax)
                vx, vy = cs * ax + (-sn) * ay, sn * ax + cs * ay        # rot_mat @ [ax, ay]
                rest = act[:, 2:] if kind == "base" else act[:, 1:]
                vel.append(torch.hstack([vx[:, None], vy[:, None], rest]))
        if pos:
            self.px.cuda_articulation_target_qpos.torch()[self.pos_gx, self.pos_gy] = pos[0] if len(pos) == 1 else torch.cat(pos, dim=1)
        if vel:
            self.px.cuda_articulation_target_qvel.torch()[self.vel_gx, self.vel_gy] = vel[0] if len(vel) == 1 else torch.cat(vel, dim=1)

    def apply_targets(self):
        """sapien_env.py:1106-1118: the target buffers the controllers wrote go to the simulation"""
        if self.boundary.masked and self.sets_qpos and self.sets_qvel:
            self.px.gpu_apply_masked(16 | 32)
            return
        if self.sets_qpos:
            self.px.gpu_apply_articulation_target_position()
        if self.sets_qvel:
            self.px.gpu_apply_articulation_target_velocity()

    def __call__(self, action):
        """BaseEnv._step_action for a batched action tensor (None: step without a new action)."""
        base = self.base
        if action is not None:
            if not isinstance(action, torch.Tensor):
                action = torch.as_tensor(action)
            action = action.to(base.device)
            if action.ndim == 1 and base.num_envs == 1:
                action = action[None]
            if tuple(action.shape) != (base.num_envs, self.adim):
                raise AssertionError(f"Received action of shape {tuple(action.shape)} but expected shape ({base.num_envs}, {self.adim})")
            self.set_action(action)
            self.apply_targets()
        if self.hook_before_control:
            base._before_control_step()
        if self.hooks_per_substep:
            for _ in range(self.sim_steps):
                base._before_simulation_step()
                self.scene.step()
                base._after_simulation_step()
        else:
            for _ in range(self.sim_steps):
                self.scene.step()
        base._after_control_step()
        self.boundary.fetch_all()
        return action

    boundary: _Boundary = None


# --------------------------------------------------------------------------------------------------------------------- task plugins
class _PointOfLink:
    """p_world = R(q) p_local + t for the wxyz poses of one link per env: the arithmetic of Pose.to_transformation_matrix (utils/geometry/rotation_conversions.py:44-73,
    quaternion_to_matrix) followed by geometry.transform_points (geometry.py:134-140) -- the same products, sums and the same bmm, so the result has the
    reference's bits -- with the nine matrix entries formed side by side: entry = c0 + c1 * two_s * (q_a q_b + sgn q_c q_d)."""

    # (a, b, c, d, sgn, diagonal) per entry of the row-major matrix; r, i, j, k = 0, 1, 2, 3
    _E = [(2, 2, 3, 3, 1, 1), (1, 2, 3, 0, -1, 0), (1, 3, 2, 0, 1, 0),
          (1, 2, 3, 0, 1, 0), (1, 1, 3, 3, 1, 1), (2, 3, 1, 0, -1, 0),
          (1, 3, 2, 0, -1, 0), (2, 3, 1, 0, 1, 0), (1, 1, 2, 2, 1, 1)]

    def __init__(self, local, device):
        E = self._E
        self.ia, self.ib, self.ic, self.id = (torch.tensor([e[k] for e in E], device=device) for k in range(4))
        self.sgn = torch.tensor([float(e[4]) for e in E], device=device)
        self.c0 = torch.tensor([float(e[5]) for e in E], device=device)
        self.c1 = torch.tensor([-1.0 if e[5] else 1.0 for e in E], device=device)
        self.local = local[:, None, :].contiguous()

    def __call__(self, pose7):
        q = pose7[:, 3:7]
        two_s = 2.0 / (q * q).sum(-1)
        m = q[:, self.ia] * q[:, self.ib] + self.sgn * (q[:, self.ic] * q[:, self.id])
        R = (self.c0 + self.c1 * (two_s[:, None] * m)).view(-1, 3, 3)
        return torch.bmm(self.local, R.transpose(2, 1))[:, 0, :] + pose7[:, :3]


class OpenCabinetDrawerStep:
    """``BaseEnv.step`` of OpenCabinetDrawer-v1 / OpenCabinetDoor-v1 (envs/tasks/mobile_manipulation/open_cabinet_drawer.py:221-360), state observations."""

    env_ids = ("OpenCabinetDrawer-v1", "OpenCabinetDoor-v1")

    def __init__(self, base, control: FusedControl):
        if base.obs_mode not in ("state", "state_dict"):
            raise Unsupported("state observations only")
        if base.reward_mode not in ("normalized_dense", "dense", "sparse", "none"):
            raise Unsupported(f"reward mode {base.reward_mode}")
        if len(base.agent.controller.get_state()) > 0:
            raise Unsupported("controllers with state in the observation")
        if base.scene.parallel_in_single_scene:      # Link.pose / Actor.pose subtract scene_offsets there (structs/link.py:240, actor.py:357): raw rows would differ
            raise Unsupported("sub-scenes laid out in one scene")
        self.base, self.control, self.px, self.scene = base, control, base.scene.px, base.scene
        self.boundary = control.boundary
        dev = base.device
        hl = base.handle_link
        self.hl_rows = hl._body_data_index.long()
        self.tcp_rows = base.agent.tcp._body_data_index.long()
        self.goal_rows = base.handle_link_goal._body_data_index.long()
        self.handle_point = _PointOfLink(torch.as_tensor(base.handle_link_pos, device=dev).float().clone(), dev)
        j = hl.joint
        self.jrow, self.jcol = j._data_index.long(), j.active_index.long()
        self.target_qpos = base.target_qpos
        robot = base.agent.robot
        self.rrows, self.rdof = robot._data_index.long(), robot.max_dof
        self.unit_q = torch.tensor([1.0, 0.0, 0.0, 0.0], device=dev).expand(base.num_envs, 4)
        self.flat = base.obs_mode == "state"
        self._consts = tuple(torch.tensor(v, device=dev) for v in (2.0, 3.0, 5.0))

    def step(self, action):
        base, px = self.base, self.px
        # _step_action up to the physics steps, then _after_control_step (open_cabinet_drawer.py:294-305)
        ctl = self.control
        if action is not None:
            if not isinstance(action, torch.Tensor):
                action = torch.as_tensor(action)
            action = action.to(base.device)
            if tuple(action.shape) != (base.num_envs, ctl.adim):
                raise AssertionError(f"Received action of shape {tuple(action.shape)} but expected shape ({base.num_envs}, {ctl.adim})")
            ctl.set_action(action)
            ctl.apply_targets()
        for _ in range(ctl.sim_steps):
            self.scene.step()
        px.gpu_update_articulation_kinematics()
        self.boundary.fetch_all()
        rb = px.cuda_rigid_body_data.torch()
        hl = rb[self.hl_rows]
        handle_pos = self.handle_point(hl)
        rb[self.goal_rows, :7] = torch.cat([handle_pos, self.unit_q], dim=1)       # handle_link_goal.set_pose(Pose.create_from_pq(p=...))
        self.boundary.apply_all()
        self.boundary.fetch_all()
        base._elapsed_steps += 1
        # evaluate (open_cabinet_drawer.py:307-321).  The reference reads the buffers again behind the second fetch; apply + fetch leave the rows of
        # links as they were (tests/test_fused_step.py compares with the reference's bits), so `hl` and `handle_pos` are still current
        jq = px.cuda_articulation_qpos.torch()[self.jrow, self.jcol]
        open_enough = jq >= self.target_qpos
        static = (torch.linalg.norm(hl[:, 10:13], dim=1) <= 1) & (torch.linalg.norm(hl[:, 7:10], dim=1) <= 0.1)
        success = open_enough & static
        info = dict(elapsed_steps=base._elapsed_steps.clone(), success=success, handle_link_pos=handle_pos, open_enough=open_enough)
        tcp = rb[self.tcp_rows, :7]
        qpos = px.cuda_articulation_qpos.torch()[self.rrows, :self.rdof]
        qvel = px.cuda_articulation_qvel.torch()[self.rrows, :self.rdof]
        if self.flat:        # common.flatten_state_dict: agent (qpos, qvel), extra (tcp_pose, tcp_to_handle_pos, target_link_qpos, target_handle_pos)
            obs = torch.cat([qpos, qvel, tcp, handle_pos - tcp[:, :3], jq[:, None] if jq.ndim == 1 else jq, handle_pos], dim=1)
        else:
            obs = dict(agent=dict(qpos=qpos.clone(), qvel=qvel.clone()),
                       extra=dict(tcp_pose=tcp, tcp_to_handle_pos=handle_pos - tcp[:, :3], target_link_qpos=jq, target_handle_pos=handle_pos))
        mode = base.reward_mode
        if mode == "none":
            reward = torch.zeros(base.num_envs, device=base.device)
        elif mode == "sparse":
            reward = success              # compute_sparse_reward (sapien_env.py:683-690): info["success"] itself
        else:        # compute_dense_reward (open_cabinet_drawer.py:336-352) with selects for the masked assignments
            dist = torch.linalg.norm(tcp[:, :3] - handle_pos, dim=1)
            reaching = 1 - torch.tanh(5 * dist)
            left = torch.div(self.target_qpos - jq, self.target_qpos)
            open_reward = 2 * (1 - left)
            two, three, five = self._consts            # device constants made once: a host scalar inside a stream capture would be a copy
            reaching = torch.where(left < 0.999, two, reaching)
            open_reward = torch.where(open_enough, three, open_reward)
            reward = torch.where(success, five, reaching + open_reward)
            if mode == "normalized_dense":
                reward = reward / 5.0
        terminated = success.clone()
        truncated = torch.zeros(base.num_envs, dtype=torch.bool, device=base.device)
        base._last_obs = obs
        return obs, reward, terminated, truncated, info


class PickCubeStep:
    """``BaseEnv.step`` of PickCube-v1 with the Panda (envs/tasks/tabletop/pick_cube.py:132-190; Panda.is_grasping / is_static: agents/robots/panda/panda.py:237-269),
    state observations.  The same arithmetic as the reference's -- its own ``common.compute_angle_between`` on both fingers at once, the second column of
    ``quaternion_to_matrix`` entry by entry, one contact query for both finger pairs -- in ~60 launches where the reference's step spends ~300."""

    env_ids = ("PickCube-v1",)

    def __init__(self, base, control: FusedControl):
        from mani_skill.utils import common
        if base.obs_mode not in ("state", "state_dict"):
            raise Unsupported("state observations only")
        if base.reward_mode not in ("normalized_dense", "dense", "sparse", "none"):
            raise Unsupported(f"reward mode {base.reward_mode}")
        if base.robot_uids != "panda" or type(base).__name__ != "PickCubeEnv":
            raise Unsupported("the Panda PickCube task only")
        if base.scene.parallel_in_single_scene:
            raise Unsupported("sub-scenes laid out in one scene")
        if len(base.agent.controller.get_state()) > 0:
            raise Unsupported("controllers with state in the observation")
        from mani_skill.envs.sapien_env import BaseEnv
        if _overridden(base, "_after_control_step", BaseEnv):
            raise Unsupported("the task overrides _after_control_step")
        self.base, self.control, self.px, self.scene = base, control, base.scene.px, base.scene
        self.boundary = control.boundary
        self._angle = common.compute_angle_between
        agent = base.agent
        idx = lambda o: o._body_data_index.long()      # noqa: E731
        self.cube_rows, self.goal_rows, self.tcp_rows = idx(base.cube), idx(base.goal_site), idx(agent.tcp)
        self.finger_rows = torch.cat([idx(agent.finger1_link), idx(agent.finger2_link)])
        pairs = list(zip(agent.finger1_link._bodies, base.cube._bodies)) + list(zip(agent.finger2_link._bodies, base.cube._bodies))
        self.query = self.px.gpu_create_contact_pair_impulse_query(pairs)      # rows: finger1-cube of every env, then finger2-cube
        self.n = base.num_envs
        self.sign = torch.cat([torch.ones(self.n, 1, device=base.device), -torch.ones(self.n, 1, device=base.device)])
        robot = agent.robot
        self.rrows, self.rdof = robot._data_index.long(), robot.max_dof
        self.flat = base.obs_mode == "state"
        self.goal_thresh = float(base.goal_thresh)
        self.five = torch.tensor(5.0, device=base.device)

    def step(self, action):
        base, px, ctl, n = self.base, self.px, self.control, self.n
        if action is not None:
            if not isinstance(action, torch.Tensor):
                action = torch.as_tensor(action)
            action = action.to(base.device)
            if tuple(action.shape) != (n, ctl.adim):
                raise AssertionError(f"Received action of shape {tuple(action.shape)} but expected shape ({n}, {ctl.adim})")
            ctl.set_action(action)
            ctl.apply_targets()
        for _ in range(ctl.sim_steps):
            self.scene.step()
        self.boundary.fetch_all()
        base._elapsed_steps += 1
        rb = px.cuda_rigid_body_data.torch()
        cube, goal, tcp = rb[self.cube_rows, :7], rb[self.goal_rows, :3], rb[self.tcp_rows, :7]
        qpos = px.cuda_articulation_qpos.torch()[self.rrows, :self.rdof]
        qvel = px.cuda_articulation_qvel.torch()[self.rrows, :self.rdof]
        # evaluate
        obj_to_goal = goal - cube[:, :3]
        obj_to_goal_dist = torch.linalg.norm(obj_to_goal, axis=1)
        is_obj_placed = obj_to_goal_dist <= self.goal_thresh
        px.gpu_query_contact_pair_impulses(self.query)
        forces = self.query.cuda_impulses.torch().clone() / px.timestep                     # [2n, 3]
        fnorm = torch.linalg.norm(forces, axis=1)
        q = rb[self.finger_rows, 3:7]
        r, i, j, k = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
        two_s = 2.0 / (q * q).sum(-1)
        ydir = torch.stack([two_s * (i * j - k * r), 1 - two_s * (i * i + k * k), two_s * (j * k + i * r)], dim=1) * self.sign      # +y of finger 1, -y of finger 2
        angle = self._angle(ydir, forces)
        flag = torch.logical_and(fnorm >= 0.5, torch.rad2deg(angle) <= 85)
        is_grasped = torch.logical_and(flag[:n], flag[n:])
        arm_qvel = qvel[..., :-2]
        is_robot_static = torch.max(torch.abs(arm_qvel), 1)[0] <= 0.2
        success = is_obj_placed & is_robot_static
        info = dict(elapsed_steps=base._elapsed_steps.clone(), success=success, is_obj_placed=is_obj_placed, is_robot_static=is_robot_static, is_grasped=is_grasped)
        tcp_to_obj = cube[:, :3] - tcp[:, :3]
        if self.flat:
            obs = torch.hstack([qpos, qvel, is_grasped[:, None], tcp, goal, cube, tcp_to_obj, obj_to_goal])
        else:
            obs = dict(agent=dict(qpos=qpos, qvel=qvel), extra=dict(is_grasped=is_grasped, tcp_pose=tcp, goal_pos=goal, obj_pose=cube, tcp_to_obj_pos=tcp_to_obj,
                                                                     obj_to_goal_pos=obj_to_goal))
        mode = base.reward_mode
        if mode == "none":
            reward = torch.zeros(n, device=base.device)
        elif mode == "sparse":
            reward = success              # compute_sparse_reward (sapien_env.py:683-690): info["success"] itself
        else:
            reward = 1 - torch.tanh(5 * torch.linalg.norm(tcp_to_obj, axis=1))
            reward = reward + is_grasped
            reward = reward + (1 - torch.tanh(5 * obj_to_goal_dist)) * is_grasped
            reward = reward + (1 - torch.tanh(5 * torch.linalg.norm(arm_qvel, axis=1))) * is_obj_placed
            reward = torch.where(success, self.five, reward)
            if mode == "normalized_dense":
                reward = reward / 5.0
        base._last_obs = obs
        return obs, reward, success.clone(), torch.zeros(n, dtype=torch.bool, device=base.device), info


# --------------------------------------------------------------------------------------------------------------------- task plugins on the fused task kernels
def _single_engine(base):
    """the one library context behind the env's ``px`` (sub-scenes of one structural group), if that library has the fused task kernels (include/msk_task.h)"""
    px = base.scene.px
    eng = getattr(px, "_engine", None)
    if eng is None or getattr(px, "_multi", 0) is not None:
        raise Unsupported("the fused task kernels need every sub-scene in one structural group (one library context)")
    if not getattr(eng.lib, "has_task_kernels", False):
        raise Unsupported("this backend has no fused task kernels")
    return eng


def _template_body(struct, what):
    """the template body id of an Actor / Link (the same body of every sub-scene of the group)"""
    ids = {int(b._body_id) for b in struct._bodies}
    if len(ids) != 1 or min(ids) < 0:
        raise Unsupported(f"{what} is not one body of the scene template")
    return ids.pop()


def _delta_arm_controller(control, n_arm, gripper):
    """-> (arm_delta, gripper_mid, gripper_half) when the env's controller is what k_pickcube_set_action / k_pusht_set_action compute: a normalised
    PDJointPos(use_delta) on coordinates 0..n_arm-1 (pd_joint_pos.py:76-93) [+ a normalised absolute PDJointPosMimic on the two finger joints,
    :207-228]; the constants are read off FusedControl's own clip-and-scale tensors (0.5 (high + low), 0.5 (high - low) in the reference's fp32 order of
    operations), so the targets have the reference's bits."""
    plan = control.plan
    if len(plan) != (2 if gripper else 1):
        raise Unsupported("the fused task kernels restate pd_joint_delta_pos only")
    kind, c, s, e, cols, _ = plan[0]
    if not (kind == "pos" and c.config.use_delta and not c.config.use_target and getattr(c, "_normalize_action", False) and (s, e) == (0, n_arm)
            and cols.tolist() == list(range(n_arm))):
        raise Unsupported("the fused task kernels restate pd_joint_delta_pos only")
    c0, c1 = control.c0[:n_arm], control.c1[:n_arm]
    if bool((c0 != 0).any()) or bool((c1 != c1[0]).any()):
        raise Unsupported("an arm delta range that is not symmetric and uniform")
    mid = half = 0.0
    if gripper:
        kind, g, s, e, cols, _ = plan[1]
        if not (kind == "mimic" and not g.config.use_delta and getattr(g, "_normalize_action", False) and (s, e) == (n_arm, n_arm + 1)
                and cols.tolist() == [n_arm, n_arm + 1] and g.control_joint_indices.tolist() == [0] and g.mimic_joint_indices.tolist() == [1]
                and g.mimic_control_joint_indices.tolist() == [0] and g._multiplier.tolist() == [1.0] and g._offset.tolist() == [0.0]):
            raise Unsupported("the fused task kernels restate the Panda's mimic gripper controller only")
        mid, half = float(control.c0[n_arm]), float(control.c1[n_arm])
    if control.adim != n_arm + (1 if gripper else 0) or control.sets_qvel:
        raise Unsupported("the fused task kernels restate pd_joint_delta_pos only")
    return float(c1[0]), mid, half


class _KernelStep:
    """Shared by the plugins that run ``BaseEnv.step`` on the library's fused task kernels (include/msk_task.h) -- the launches the fused hosts of
    maniskill_amd.envs issue, here for an env the REFERENCE built and owns: set-action kernel (controller + commit of the drive targets), the substeps, ONE fetch
    (the sapien buffers stay what the reference's structs read), the task's observe kernel (evaluate + observation + normalised dense reward + flags + step
    counter).  The simulation state has the reference's bits (same targets, same substeps); observation and reward are the kernels' fp32 restatement of the
    task's arithmetic -- equal to the reference's own torch code within 2e-6 (libm tanhf / sqrtf against torch's; tests/test_fused_step.py states the bound),
    flags equal except on exact threshold ties."""

    level = "task-kernel"
    hidden_actors = ()

    def _setup(self, base, control):
        import ctypes
        from mani_skill.envs.sapien_env import BaseEnv
        if base.scene.parallel_in_single_scene:
            raise Unsupported("sub-scenes laid out in one scene")
        if len(base.agent.controller.get_state()) > 0:
            raise Unsupported("controllers with state in the observation")
        for hook in ("_after_control_step", "_before_control_step", "_before_simulation_step", "_after_simulation_step"):
            if _overridden(base, hook, BaseEnv):
                raise Unsupported(f"the task overrides {hook}")
        if base.reward_mode not in ("normalized_dense", "dense", "sparse", "none"):
            raise Unsupported(f"reward mode {base.reward_mode}")
        if base._elapsed_steps.dtype != torch.int32 or not base._elapsed_steps.is_contiguous():
            raise Unsupported("the step counter is not a contiguous int32 tensor")
        self.base, self.control, self.scene, self.boundary = base, control, base.scene, control.boundary
        self.eng = _single_engine(base)
        self.L, self.ctx, self.C = self.eng.lib, self.eng.ctx, ctypes
        self.n = base.num_envs

    def _ptr(self, t):
        return self.C.c_void_p(t.data_ptr())

    def _action(self, action):
        base = self.base
        if not isinstance(action, torch.Tensor):
            action = torch.as_tensor(action)
        action = action.to(base.device)
        if action.ndim == 1 and self.n == 1:
            action = action[None]
        if tuple(action.shape) != (self.n, self.control.adim):
            raise AssertionError(f"Received action of shape {tuple(action.shape)} but expected shape ({self.n}, {self.control.adim})")
        return action.float().contiguous()

    def _physics(self, action, set_action):
        """_step_action: targets (kernel), substeps.  The observe kernel comes next (PickCube's computes the link frames of the new state in the same
        launch), then ``_publish``: ONE fetch of everything scene._gpu_fetch_all() fetches."""
        L, eng = self.L, self.eng
        if action is not None:
            action = self._action(action)
            L.check(self.ctx, set_action(self.ctx, self._ptr(action), eng._stream()), "task set_action")
        L.check(self.ctx, L.control_step(self.ctx, self.control.sim_steps, eng._stream()), "control_step")

    def _publish(self):
        self.boundary.fetch_all()

    def step(self, action):
        """the env's ``step``: the kernels, or -- while the kernels cannot see what the reference's structs report (``_usable``) -- the reference's own step"""
        if not self._usable():
            return type(self.base).step(self.base, action)
        return self.kernel_step(action)

    def _usable(self):
        """what the kernels cannot see from inside the simulation: an actor the reference 'hid' by moving it 99999 m away (Actor.hide_visual in GPU
        simulation, structs/actor.py:176-201) reports its remembered pose -- then the reference's own step runs"""
        return not any(a.hidden for a in self.hidden_actors)

    def _reward(self, rew, success, scale):
        mode = self.base.reward_mode
        if mode == "normalized_dense":
            return rew
        if mode == "dense":
            return rew * scale
        if mode == "sparse":
            return success                 # compute_sparse_reward (sapien_env.py:683-690): info["success"] itself
        return torch.zeros(self.n, device=self.base.device)


class PickCubeKernelStep(_KernelStep):
    """``BaseEnv.step`` of PickCube-v1 (Panda, pd_joint_delta_pos, state observations) on msk_task_pickcube_set_action / msk_control_step /
    msk_task_pickcube_observe (envs/tasks/tabletop/pick_cube.py:132-190; panda.py:237-269)."""

    env_ids = ("PickCube-v1",)

    def __init__(self, base, control: FusedControl):
        if base.obs_mode not in ("state", "state_dict"):
            raise Unsupported("state observations only")
        if base.robot_uids != "panda" or type(base).__name__ != "PickCubeEnv":
            raise Unsupported("the Panda PickCube task only")
        self._setup(base, control)
        from . import _native as NN
        delta, mid, half = _delta_arm_controller(control, 7, gripper=True)
        agent = base.agent
        d = NN.PickCubeDesc(cube=_template_body(base.cube, "the cube"), goal=_template_body(base.goal_site, "the goal site"), tcp=_template_body(agent.tcp, "the tcp link"),
                            left_finger=_template_body(agent.finger1_link, "finger 1"), right_finger=_template_body(agent.finger2_link, "finger 2"),
                            arm_dofs=7, arm_delta=delta, gripper_mid=mid, gripper_half=half, goal_thresh=float(base.goal_thresh), min_force=0.5,
                            max_angle_deg=85.0, static_thresh=0.2, max_episode_steps=2 ** 31 - 1)      # (truncation is the TimeLimitWrapper's: BaseEnv.step returns False)
        self.L.check(self.ctx, self.L.task_pickcube_init(self.ctx, self.C.byref(d)), "task_pickcube_init")
        self.hidden_actors = (base.goal_site,)
        self.flat = base.obs_mode == "state"

    def kernel_step(self, action):
        from .graph import alloc_step_outputs
        base, L = self.base, self.L
        self._physics(action, L.task_pickcube_set_action)
        obs, rew, fl, elapsed, _ = alloc_step_outputs(self.n, 42, base.device)
        L.check(self.ctx, L.task_pickcube_observe(self.ctx, self._ptr(obs), self._ptr(rew), self._ptr(fl), self._ptr(base._elapsed_steps), 1, self.eng._stream()),
                "task_pickcube_observe")
        self._publish()
        elapsed.copy_(base._elapsed_steps)
        info = dict(elapsed_steps=elapsed, success=fl[:, 0], is_obj_placed=fl[:, 1], is_robot_static=fl[:, 2], is_grasped=fl[:, 3])
        if not self.flat:
            obs = dict(agent=dict(qpos=obs[:, 0:9], qvel=obs[:, 9:18]),
                       extra=dict(is_grasped=fl[:, 3], tcp_pose=obs[:, 19:26], goal_pos=obs[:, 26:29], obj_pose=obs[:, 29:36], tcp_to_obj_pos=obs[:, 36:39],
                                  obj_to_goal_pos=obs[:, 39:42]))
        base._last_obs = obs
        return obs, self._reward(rew, fl[:, 0], 5.0), fl[:, 4], fl[:, 5], info


class PegInsertionSideKernelStep(_KernelStep):
    """``BaseEnv.step`` of PegInsertionSide-v1 (BASELINE config 4's task; panda_wristcam, pd_joint_delta_pos, state observations) on the pickcube controller
    kernel and msk_task_peg_observe (envs/tasks/tabletop/peg_insertion_side.py:248-337): the peg is the 'cube' of the binding, box_with_hole its 'goal'; per-env
    peg sizes, hole offsets and hole radii are the env's own tensors (:114-131)."""

    env_ids = ("PegInsertionSide-v1",)

    def __init__(self, base, control: FusedControl):
        import numpy as np
        if base.obs_mode not in ("state", "state_dict"):
            raise Unsupported("state observations only")
        if type(base).__name__ != "PegInsertionSideEnv" or base.robot_uids not in ("panda_wristcam", "panda"):
            raise Unsupported("the Panda PegInsertionSide task only")
        self._setup(base, control)
        from . import _native as NN
        delta, mid, half = _delta_arm_controller(control, 7, gripper=True)
        agent = base.agent
        d = NN.PickCubeDesc(cube=_template_body(base.peg, "the peg"), goal=_template_body(base.box, "the box"), tcp=_template_body(agent.tcp, "the tcp link"),
                            left_finger=_template_body(agent.finger1_link, "finger 1"), right_finger=_template_body(agent.finger2_link, "finger 2"),
                            arm_dofs=7, arm_delta=delta, gripper_mid=mid, gripper_half=half, goal_thresh=0.0, min_force=0.5, max_angle_deg=20.0,
                            static_thresh=0.2, max_episode_steps=2 ** 31 - 1)
        self.L.check(self.ctx, self.L.task_pickcube_init(self.ctx, self.C.byref(d)), "task_pickcube_init")
        f32 = lambda t: np.ascontiguousarray(t.detach().cpu().numpy(), dtype=np.float32)      # noqa: E731
        halfs, holes, radii = f32(base.peg_half_sizes), f32(base.box_hole_offsets.p), f32(base.box_hole_radii)
        if halfs.shape != (self.n, 3) or holes.shape != (self.n, 3) or radii.shape != (self.n,):
            raise Unsupported("unexpected per-env peg tables")
        FP = self.C.POINTER(self.C.c_float)
        self.L.check(self.ctx, self.L.task_peg_init(self.ctx, halfs.ctypes.data_as(FP), holes.ctypes.data_as(FP), radii.ctypes.data_as(FP)), "task_peg_init")
        self.flat = base.obs_mode == "state"

    def kernel_step(self, action):
        from .graph import alloc_step_outputs
        base, L = self.base, self.L
        self._physics(action, L.task_pickcube_set_action)
        obs, rew, fl, elapsed, head = alloc_step_outputs(self.n, 43, base.device, extra_floats=3)
        L.check(self.ctx, L.task_peg_observe(self.ctx, self._ptr(obs), self._ptr(rew), self._ptr(fl), self._ptr(base._elapsed_steps), self._ptr(head), 1,
                                             self.eng._stream()), "task_peg_observe")
        self._publish()
        elapsed.copy_(base._elapsed_steps)
        info = dict(elapsed_steps=elapsed, success=fl[:, 0], peg_head_pos_at_hole=head)
        if not self.flat:
            obs = dict(agent=dict(qpos=obs[:, 0:9], qvel=obs[:, 9:18]),
                       extra=dict(tcp_pose=obs[:, 18:25], peg_pose=obs[:, 25:32], peg_half_size=obs[:, 32:35], box_hole_pose=obs[:, 35:42], box_hole_radius=obs[:, 42]))
        base._last_obs = obs
        return obs, self._reward(rew, fl[:, 0], 10.0), fl[:, 4], fl[:, 5], info


class PushTKernelStep(_KernelStep):
    """``BaseEnv.step`` of PushT-v1 (BASELINE config 3's task; panda_stick, pd_joint_delta_pos) on msk_task_pusht_set_action / msk_control_step /
    msk_task_pusht_observe (envs/tasks/tabletop/push_t.py:343-431 the 64 x 64 intersection 'renderer', :484-540 evaluate / observation / reward), state
    observations or the minimal pack's camera observations (rgb / depth / segmentation): the pictures are the shim's own take_picture, depth and segmentation
    handed out from the rasteriser's planes (what render/shaders.py:75-83 computes from PositionSegmentation, without the int16 x 4 texture in between)."""

    env_ids = ("PushT-v1",)

    def __init__(self, base, control: FusedControl):
        import numpy as np
        if type(base).__name__ != "PushTEnv" or base.robot_uids != "panda_stick":
            raise Unsupported("the panda_stick PushT task only")
        self._setup(base, control)
        from . import _native as NN
        C = self.C
        st = base.obs_mode_struct
        self.cameras = []
        if base.obs_mode in ("state", "state_dict"):
            self.obs_dim = 31
        else:
            vis = st.visual
            if st.state or st.state_dict or vis.position or vis.normal or vis.albedo or base.obs_mode in ("pointcloud", "sensor_data"):
                raise Unsupported(f"observation mode {base.obs_mode}")
            from mani_skill.sensors.camera import Camera
            self.scene.update_render(update_sensors=True, update_human_render_cameras=False)      # (the camera groups exist from here on)
            for name, sensor in self.scene.sensors.items():
                group = self.scene.camera_groups.get(name) if isinstance(sensor, Camera) else None
                if group is None or not hasattr(group, "planes") or sensor.config.shader_config.shader_pack != "minimal":
                    raise Unsupported("a sensor that is not a minimal-pack camera of this backend")
                depth, seg = group.planes()
                color = group.get_picture_cuda("Color").torch() if vis.rgb else None
                self.cameras.append((name, group, depth, seg, color))
            self.want = (bool(vis.rgb), bool(vis.depth), bool(vis.segmentation))
            self.obs_dim = 21
        delta, _, _ = _delta_arm_controller(control, 7, gripper=False)
        w2g = base.world_to_goal_trans.detach().cpu().numpy().astype(np.float32)
        mask = np.ascontiguousarray(base.tee_render.detach().cpu().numpy() == 1, dtype=np.uint8)
        if mask.shape != (64, 64) or int(base.res) != 64:
            raise Unsupported("the intersection kernel is written for the task's 64 x 64 grid")
        goal_xy = [float(v) for v in base.goal_offset]
        d = NN.PushTDesc(tee=_template_body(base.tee, "the T"), goal=_template_body(base.goal_tee, "the goal T"), tcp=_template_body(base.agent.tcp, "the tcp link"),
                         arm_dofs=7, arm_delta=delta, goal_xy=(C.c_float * 2)(*goal_xy), goal_z_rot=float(base.goal_z_rot),
                         world_to_goal=(C.c_float * 6)(*[float(x) for x in w2g[:2].reshape(-1)]),
                         uv_scale=float(np.float32((base.res / 2) / base.uv_half_width)), intersection_thresh=float(base.intersection_thresh),
                         max_episode_steps=2 ** 31 - 1)
        self.L.check(self.ctx, self.L.task_pusht_init(self.ctx, C.byref(d), mask.ctypes.data_as(C.POINTER(C.c_uint8))), "task_pusht_init")
        for _, group, *_ in self.cameras:
            group.set_outputs(False, color=self.want[0])      # what the observation mode reads, nothing else (SAPIEN fills every texture of the pack)
        self.flat = base.obs_mode == "state"
        self.consts = DeviceConstants(base.device)

    def restore(self):
        for _, group, *_ in self.cameras:
            group.set_outputs(True, True)

    def kernel_step(self, action):
        from . import graph as _graph
        base, L = self.base, self.L
        self._physics(action, L.task_pusht_set_action)
        obs, rew, fl, elapsed, _ = _graph.alloc_step_outputs(self.n, self.obs_dim, base.device)
        L.check(self.ctx, L.task_pusht_observe(self.ctx, self._ptr(obs), self.obs_dim, self._ptr(rew), self._ptr(fl), self._ptr(base._elapsed_steps), 1,
                                               self.eng._stream()), "task_pusht_observe")
        self._publish()
        elapsed.copy_(base._elapsed_steps)
        info = dict(elapsed_steps=elapsed, success=fl[:, 0])
        if self.cameras:          # _get_obs_with_sensor_data (sapien_env.py:627-634): agent, extra, sensor_param, sensor_data
            rgb, depth, seg = self.want
            keep = (lambda t: t) if _graph.CAPTURING else (lambda t: t.clone())      # (a replay snapshots every output once)
            data = {}
            for name, group, dplane, splane, color in self.cameras:
                group.take_picture()
                pics = {}
                if rgb:
                    pics["rgb"] = keep(color[..., :3])
                if depth:
                    pics["depth"] = keep(dplane)
                if seg:
                    pics["segmentation"] = keep(splane)
                data[name] = pics
            with self.consts:      # (a mounted camera's matrices are made from host constants at every call: render_camera.py:77-155)
                params = base.get_sensor_params()
            obs = dict(agent=dict(qpos=obs[:, 0:7], qvel=obs[:, 7:14]), extra=dict(tcp_pose=obs[:, 14:21]), sensor_param=params, sensor_data=data)
        elif not self.flat:
            obs = dict(agent=dict(qpos=obs[:, 0:7], qvel=obs[:, 7:14]), extra=dict(tcp_pose=obs[:, 14:21], goal_pos=obs[:, 21:24], obj_pose=obs[:, 24:31]))
        base._last_obs = obs
        return obs, self._reward(rew, fl[:, 0], 3.0), fl[:, 4], fl[:, 5], info


_PLUGINS = [OpenCabinetDrawerStep, PickCubeKernelStep, PickCubeStep, PegInsertionSideKernelStep, PushTKernelStep]


# --------------------------------------------------------------------------------------------------------------------- method patches
def _pusht_pseudo_render_intersection(base):
    """PushT-v1's ``pseudo_render_intersection`` (envs/tasks/tabletop/push_t.py:343-432) without what a capture forbids: the T's pixels are selected with the
    indices of the (constant) mask ``tee_render == 1`` instead of the mask, out-of-range indices are zeroed with a select instead of masked assignments, the
    work tensors are made on the device.  Same products (the same full matmul over the uv map), same truncation, same element order: the reference's bits."""
    dev = base.device
    mask = (base.tee_render == 1).reshape(-1)
    tidx = torch.nonzero(mask).reshape(-1)                       # once, outside any step: row-major order = the order boolean indexing yields
    res = base.uv_grid.shape[1]
    scale = (res / 2) / base.uv_half_width
    tee_bool = base.tee_render.bool()
    state = {}

    def pseudo_render_intersection():
        q, p = base.tee.pose.q, base.tee.pose.p
        b = q.shape[0]
        alphas = base.quat_to_z_euler(q)
        T = torch.zeros(b, 3, 3, device=dev)
        T[:, 2, 2] = 1
        T[:, 0, 0] = alphas.cos()
        T[:, 1, 1] = alphas.cos()
        T[:, 0, 1] = -alphas.sin()
        T[:, 1, 0] = alphas.sin()
        T[:, 0:2, 2] = p[:, :2]
        tee_to_goal = base.world_to_goal_trans @ T
        tees = (tee_to_goal @ base.homo_uv.view(3, -1)).view(b, 3, res, res)
        tees = tees[:, 0:2, :, :] / tees[:, -1, :, :].unsqueeze(1)
        coords = tees.reshape(b, 2, -1)[:, :, tidx]
        ind = (coords * scale + (res / 2)).long().view(b, 2, -1)
        bad = (ind[:, 0, :] < 0) | (ind[:, 0, :] >= base.res) | (ind[:, 1, :] < 0) | (ind[:, 1, :] >= base.res)
        zero = ind.new_zeros(())
        ix, iy = torch.where(bad, zero, ind[:, 0, :]), torch.where(bad, zero, ind[:, 1, :])
        if state.get("b") != b:
            state["b"] = b
            state["batch"] = torch.arange(b, device=dev).view(-1, 1).repeat(1, tidx.numel())
            state["one"] = torch.ones((), device=dev)
        final = torch.zeros(b, res, res, device=dev)
        final[state["batch"], ix, iy] = state["one"]
        final = final.permute(0, 2, 1).flip(1)
        inter = (final.bool() & tee_bool).sum(dim=[-1, -2]).float()
        return inter / tee_bool.sum().float()
    return pseudo_render_intersection


# env id -> {method name: factory(base) -> replacement}: single methods of a task whose results are restated bit for bit so that the rest of the task's OWN
# step can be captured (used by the generic graph level; installed as instance attributes, removed by restore())
_METHOD_PATCHES = {"PushT-v1": {"pseudo_render_intersection": _pusht_pseudo_render_intersection}}


# --------------------------------------------------------------------------------------------------------------------- masked assignments
class _MaskedSelection(torch.Tensor):
    """``y[mask]`` for a boolean ``mask`` over the leading dimensions, not evaluated yet (DeviceConstants hands it out while a step is warmed up, watched and
    captured).  Its one cheap use is the reference's idiom ``x[mask] = y[mask]`` (stack_cube.py:161, place_sphere.py:230, poke_cube.py:208, ...), which
    DeviceConstants turns into ``x <- where(mask, y, x)``: the same values in the same places, no ``nonzero()`` -- no host synchronisation, capturable.  ANY
    other use makes it the real selection first (the data-dependent shape a capture forbids: such a task fails the capture as before)."""

    @staticmethod
    def make(src, mask):
        sel = src.as_subclass(_MaskedSelection)
        sel._msk_src, sel._msk_mask = src, mask
        sel._msk_versions = (src._version, mask._version)      # the selection ALIASES its source: it is only y[mask] while neither was written since
        return sel

    def _unchanged(self):
        """eager ``y[mask]`` is a copy; this one is not -- a source or mask written between the selection and its use (``old = x[m]; x[m] = 0; y[m] = old``)
        would hand out the new values.  Such a step is refused, loudly, instead of replayed wrong."""
        if (self._msk_src._version, self._msk_mask._version) != self._msk_versions:
            raise Unsupported("a masked selection `y[mask]` was used after its source or mask had been written in place (a save / restore idiom): the lazy "
                              "selection of fused_step aliases its source, eager torch copies it -- this step cannot run under DeviceConstants")
        return self

    def _real(self):
        self._unchanged()
        with torch._C.DisableTorchFunctionSubclass():
            return torch.Tensor.__getitem__(self._msk_src, self._msk_mask)

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        def real(a):
            if isinstance(a, _MaskedSelection):
                return a._real()
            if isinstance(a, (list, tuple)):
                return type(a)(real(v) for v in a)
            if isinstance(a, dict):
                return {k: real(v) for k, v in a.items()}
            return a
        with torch._C.DisableTorchFunctionSubclass():
            return func(*real(args), **real(kwargs or {}))


# elementwise arithmetic on unevaluated selections of ONE mask stays unevaluated: a[m] op b[m] == (a op b)[m], element for element (`reward[m] += bonus[m]`,
# poke_cube.py:221, is __getitem__, __iadd__, __setitem__); (function, operation on the sources, operands swapped)
_T = torch.Tensor
_SELECTION_ARITHMETIC = {f: (op, swap) for op, fs, swap in (
    (torch.add, (_T.__add__, _T.__iadd__, _T.add, _T.add_, torch.add), False), (torch.add, (_T.__radd__,), True),
    (torch.sub, (_T.__sub__, _T.__isub__, _T.sub, _T.sub_, torch.sub), False), (torch.sub, (_T.__rsub__,), True),
    (torch.mul, (_T.__mul__, _T.__imul__, _T.mul, _T.mul_, torch.mul), False), (torch.mul, (_T.__rmul__,), True),
    (torch.true_divide, (_T.__truediv__, _T.__itruediv__, _T.div, _T.div_, torch.div, torch.true_divide), False), (torch.true_divide, (_T.__rtruediv__,), True)) for f in fs}


_SELECTION_UNARY = {f: getattr(torch, n) for n in ("sin", "cos", "tan", "tanh", "exp", "log", "sqrt", "abs", "neg", "square", "sigmoid", "asin", "acos", "atan")
                    for f in (getattr(torch, n), getattr(_T, n))}
_SELECTION_UNARY.update({_T.__neg__: torch.neg, _T.__abs__: torch.abs})
_MASK_NOT = (_T.__invert__, torch.logical_not, _T.logical_not, torch.bitwise_not, _T.bitwise_not)


def _selection_arithmetic(func, args, kwargs):
    """the unevaluated result, or None when the operands are not selections of one mask (and plain numbers)"""
    if len(args) != 2 or kwargs:
        return None
    op, swap = _SELECTION_ARITHMETIC[func]
    sels = [a for a in args if isinstance(a, _MaskedSelection)]
    mask = sels[0]._msk_mask
    srcs = []
    for a in args:
        if isinstance(a, _MaskedSelection):
            if a._msk_mask is not mask:
                return None
            if a._msk_src.shape != sels[0]._msk_src.shape:      # x[m] / n[m].view(-1, 1) (sapien_utils.py:354): sources that broadcast behind the mask's dimensions
                sa, sb = a._msk_src.shape, sels[0]._msk_src.shape
                if len(sa) != len(sb) or sa[:mask.ndim] != sb[:mask.ndim] or any(p != q and 1 not in (p, q) for p, q in zip(sa[mask.ndim:], sb[mask.ndim:])):
                    return None
            srcs.append(a._unchanged()._msk_src)
        elif isinstance(a, (int, float)) and not isinstance(a, bool):
            srcs.append(a)
        else:
            return None
    if swap:
        srcs.reverse()
    with torch._C.DisableTorchFunctionSubclass():
        return _MaskedSelection.make(op(*srcs), mask)


def _leading_mask(x, idx):
    return isinstance(idx, torch.Tensor) and idx.dtype == torch.bool and 1 <= idx.ndim <= x.ndim and tuple(idx.shape) == tuple(x.shape[:idx.ndim]) and idx.device == x.device


def _broadcast_mask(mask, x):
    return mask if mask.ndim == x.ndim else mask[(...,) + (None,) * (x.ndim - mask.ndim)]


# --------------------------------------------------------------------------------------------------------------------- host constants inside a step
class DeviceConstants(torch.overrides.TorchFunctionMode):
    """Active while a step is warmed up and captured.  Task code of the reference turns host data into device tensors inside the step --
    ``torch.tensor([-self.cube_half_size - 0.005, 0, 0], device=self.device)`` (envs/tasks/tabletop/push_cube.py:211), ``torch.tensor([1, -1, -1, -1], device=...)``
    (utils/geometry/rotation_conversions.py:440), ``common.to_tensor(array)`` (utils/common.py:158-167), a Python list as an index (``data[..., [2]]``,
    render/shaders.py:80) -- a host->device copy that a stream capture forbids.  Here
    the first evaluation of such an expression (during the eager warm-up) is kept on the device, keyed by its call site, and later evaluations get a device-side
    clone of it.  A call site whose host data CHANGES from one step to the next cannot be baked into a graph: that raises ``Unsupported``."""

    def __init__(self, device):
        super().__init__()
        self.device = torch.device(device)
        self.cache = {}
        self.served = 0
        self.masked = self.rewritten = self.syncs_skipped = 0      # y[mask] handed out unevaluated / masked assignments turned into selects / device waits skipped
        self._seen, self._nots = {}, {}
        self._depth, self._sync = 0, None
        self._one_hot = {}      # id(tensor) -> (tensor, its version, idx, idx's version): results of F.one_hot(idx, C) and of `that > 0.5`
        # masks that are all True for the whole step, by construction: `scene._reset_mask` outside a reset (the reference indexes through it in every pose setter,
        # utils/structs/actor.py:389-391 `idx[reset_mask[scene_idxs]]`: a boolean index, i.e. nonzero() and a wait, for a selection that selects everything).
        # all_true(): the current mask object; a selection through it -- or through a gather of it -- is the whole source.  Checked once per mask object (a wait,
        # outside any capture: the warm-up steps see every such object first).
        self.all_true = None
        self._true_ok, self._true_derived = {}, {}

    def __enter__(self):
        self._depth += 1
        if self._depth > 1:             # re-entered (a step that calls a step): one set of per-step tables, one patch of synchronize, undone by the outermost exit
            return super().__enter__()
        self._seen = {}                 # one step = one `with`: the k-th evaluation of a call path inside a step is its own constant (loops)
        self._nots = {}
        # `torch.cuda.synchronize()` inside the step (sapien_env.py:624, behind take_picture: "prevents the GPU from making poor scheduling decisions") is
        # a scheduling hint there, not a data dependency -- everything of a step is ordered on one stream -- and a stream capture forbids it (first seen on
        # hardware in round 5: the CPU checker has no device to wait for): a no-op while the step is warmed up, watched, captured
        self._sync = torch.cuda.synchronize
        torch.cuda.synchronize = self._no_sync
        return super().__enter__()

    def __exit__(self, *exc):
        self._depth -= 1
        if self._depth == 0:
            torch.cuda.synchronize = self._sync
        return super().__exit__(*exc)

    def _no_sync(self, *a, **k):
        self.syncs_skipped += 1

    def _is_all_true(self, mask) -> bool:
        if self.all_true is None or not isinstance(mask, torch.Tensor) or mask.dtype != torch.bool:
            return False
        hit = self._true_derived.get(id(mask))
        if hit is not None and hit[0] is mask and hit[1] == mask._version:
            return True
        cur = self.all_true()
        if mask is not cur:
            return False
        ok = self._true_ok.get(id(cur))
        if ok is None or ok[0] is not cur or ok[1] != cur._version:
            from . import graph as _graph
            if _graph.CAPTURING and torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
                return False          # (an object the warm-up never showed: no wait inside a capture -- the ordinary path decides)
            with torch._C.DisableTorchFunctionSubclass():
                ok = self._true_ok[id(cur)] = (cur, cur._version, bool(cur.all()))
        return ok[2]

    def _one_hot_index(self, x, index):
        """idx when ``index`` is ``(F.one_hot(idx, C) > t, slice(None)...)`` over the two leading dimensions of x, else None"""
        if not (isinstance(index, tuple) and len(index) >= 1 and isinstance(index[0], torch.Tensor) and index[0].dtype == torch.bool and index[0].ndim == 2
                and all(isinstance(r, slice) and r == slice(None) for r in index[1:]) and x.ndim >= 2 and tuple(index[0].shape) == tuple(x.shape[:2])):
            return None
        rec = self._one_hot.get(id(index[0]))
        if rec is None or rec[0] is not index[0] or rec[1] != index[0]._version or rec[2]._version != rec[3] or rec[2].shape[0] != x.shape[0]:
            return None
        return rec[2]

    @staticmethod
    def _site():
        """The call path of the expression: (file, line) of the frames outside torch and this module, innermost first.  The innermost frame alone is not
        enough: ``Pose.create`` -> ``common.to_tensor`` (utils/common.py:167) is one line that every caller's host data goes through -- keyed by it, two
        callers looked like one site whose data changes (PegInsertionSide-v1's reward, first seen on hardware in round 5)."""
        import sys
        f = sys._getframe(2)
        here = __file__
        path = []
        while f is not None and len(path) < 24:
            fn = f.f_code.co_filename
            if fn == here:          # the step's entry (captured_step / a plugin's step): what lies outside differs between warm-up, capture and replay
                if f.f_code.co_name not in ("__torch_function__", "_serve", "_site", "conv"):
                    if not path:
                        path.append((fn, f.f_lineno))
                    break
            elif "/torch/" not in fn:
                path.append((fn, f.f_lineno))
            f = f.f_back
        return tuple(path) if path else (("?", 0),)

    def _serve(self, site_key, content, make):
        k = self._seen.get(site_key, 0)
        self._seen[site_key] = k + 1
        site_key = site_key + (k,)
        hit = self.cache.get(site_key)
        if hit is None:
            hit = self.cache[site_key] = (content, make())
        elif hit[0] != content:
            where = site_key[0]
            raise Unsupported(f"host data that changes from step to step is uploaded inside the step at {where[0]}:{where[1]}: a replayed graph would keep the first value")
        self.served += 1
        return hit[1].clone()

    def __torch_function__(self, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        if (func is torch.tensor or func is torch.as_tensor) and args and not isinstance(args[0], torch.Tensor):
            dev = kwargs.get("device")
            if dev is not None and torch.device(dev).type == self.device.type:
                import numpy as np
                a = np.asarray(args[0])
                if a.dtype != object and a.size <= 65536:
                    return self._serve(self._site() + ("tensor", str(kwargs.get("dtype"))), (a.dtype.str, a.shape, a.tobytes()), lambda: func(*args, **kwargs))
        elif func is torch.Tensor.to and args and isinstance(args[0], torch.Tensor) and args[0].device.type == "cpu" and self.device.type != "cpu":
            t = args[0]
            target = kwargs.get("device", args[1] if len(args) > 1 and isinstance(args[1], (str, torch.device)) else None)
            if target is not None and torch.device(target).type == self.device.type and t.numel() <= 65536 and not t.requires_grad:
                tc = t.detach().contiguous()
                content = (str(tc.dtype), tuple(tc.shape), tc.numpy().tobytes() if tc.dtype != torch.bfloat16 else tc.float().numpy().tobytes())
                return self._serve(self._site() + ("to", str(kwargs.get("dtype"))), content, lambda: func(*args, **kwargs))
        elif func in _SELECTION_UNARY and len(args) == 1 and not kwargs and isinstance(args[0], _MaskedSelection):
            with torch._C.DisableTorchFunctionSubclass():      # f(a[m]) == f(a)[m], element for element (what f makes of the unselected elements is never looked at)
                return _MaskedSelection.make(_SELECTION_UNARY[func](args[0]._unchanged()._msk_src), args[0]._msk_mask)
        elif func in (torch.Tensor.view, torch.Tensor.reshape, torch.Tensor.unsqueeze) and args and isinstance(args[0], _MaskedSelection) and not kwargs \
                and args[0]._msk_src.ndim == args[0]._msk_mask.ndim and (tuple(args[1:]) in ((-1, 1), ((-1, 1),)) if func is not torch.Tensor.unsqueeze else tuple(args[1:]) == (-1,)):
            with torch._C.DisableTorchFunctionSubclass():      # n[m].view(-1, 1) == n.unsqueeze(-1)[m]
                return _MaskedSelection.make(args[0]._unchanged()._msk_src.unsqueeze(-1), args[0]._msk_mask)
        elif func is torch.nn.functional.one_hot and args and isinstance(args[0], torch.Tensor) and args[0].ndim == 1:
            ncls = kwargs.get("num_classes", args[1] if len(args) > 1 else -1)
            if not isinstance(ncls, int) or ncls < 1:
                return func(*args, **kwargs)                      # (the class count taken from the data: a read-back)
            # F.one_hot(idx, C), without the range check that reads the data back on a host tensor (ATen skips it on a GPU): remembered, so that
            # `x[one_hot > 0.5, :]` can be the gather it is
            out = (args[0].unsqueeze(-1) == torch.arange(ncls, device=args[0].device)).to(torch.int64)
            self._one_hot[id(out)] = (out, out._version, args[0], args[0]._version)
            return out
        elif func in (torch.Tensor.__gt__, torch.gt, torch.Tensor.gt) and len(args) == 2 and id(args[0]) in self._one_hot and isinstance(args[1], float) and 0.0 < args[1] < 1.0:
            rec = self._one_hot[id(args[0])]
            out = func(*args)
            if rec[0] is args[0] and rec[1] == args[0]._version:
                self._one_hot[id(out)] = (out, out._version, rec[2], rec[3])
                if len(self._one_hot) > 64:
                    self._one_hot = {id(out): self._one_hot[id(out)]}
            return out
        elif func is torch.Tensor.__getitem__ and len(args) == 2 and type(args[0]) is torch.Tensor and self._one_hot_index(args[0], args[1]) is not None:
            # x[F.one_hot(idx, C) > 0.5, :] (rotation_conversions.py:161-163: the best-conditioned of four candidates per row): exactly one row of x[n] per n, in order
            idx = self._one_hot_index(args[0], args[1])
            self.masked += 1
            x = args[0]
            return torch.take_along_dim(x, idx.reshape((-1, 1) + (1,) * (x.ndim - 2)), dim=1).squeeze(1)
        elif func in _MASK_NOT and len(args) == 1 and not kwargs and type(args[0]) is torch.Tensor and args[0].dtype == torch.bool:
            # `x[~m] = f(y[~m])` evaluates ~m once per use (rotation_conversions.py:549-552): the uses have to be ONE mask to be recognised as one selection
            hit = self._nots.get(id(args[0]))
            if hit is None or hit[0] is not args[0] or hit[2] != args[0]._version or hit[3] != hit[1]._version:      # (the cached result itself written in place: `nm = ~m; nm[1] = False`)
                out = func(args[0])
                hit = self._nots[id(args[0])] = (args[0], out, args[0]._version, out._version)
            return hit[1]
        elif func in _SELECTION_ARITHMETIC and any(isinstance(a, _MaskedSelection) for a in args):
            out = _selection_arithmetic(func, args, kwargs)
            if out is not None:
                return out
        elif func in (torch.Tensor.__getitem__, torch.Tensor.__setitem__) and len(args) >= 2 and type(args[0]) is torch.Tensor and isinstance(args[1], tuple) and len(args[1]) >= 2 \
                and isinstance(args[1][0], torch.Tensor) and _leading_mask(args[0], args[1][0]) and all(isinstance(r, (slice, int)) and not isinstance(r, bool) for r in args[1][1:]) \
                and args[1][0].ndim + len(args[1]) - 1 <= args[0].ndim:
            # x[mask, :2] / x[mask, 2] (drawing/draw.py:178-183): the mask over the leading dimensions of the VIEW x[:, :2] -- the forms below, on that view
            mask, rest = args[1][0], args[1][1:]
            view = torch.Tensor.__getitem__(args[0], (slice(None),) * mask.ndim + rest)
            if func is torch.Tensor.__getitem__:
                return self.__torch_function__(func, types, (view, mask))
            v = args[2]
            if isinstance(v, (int, float)) and not isinstance(v, bool):
                view.masked_fill_(_broadcast_mask(mask, view), v)      # x[mask, 2] = number
                self.rewritten += 1
                return None
            return self.__torch_function__(func, types, (view, mask, v))
        elif func is torch.Tensor.__getitem__ and len(args) == 2 and type(args[0]) is torch.Tensor and args[0].dtype == torch.bool and self._is_all_true(args[0]) \
                and isinstance(args[1], torch.Tensor) and args[1].dtype in (torch.int64, torch.int32):
            out = func(*args)                                       # reset_mask[scene_idxs]: a gather of an all-true mask is all true
            self._true_derived[id(out)] = (out, out._version)
            if len(self._true_derived) > 256:
                self._true_derived = {id(out): (out, out._version)}
            return out
        elif func is torch.Tensor.__getitem__ and len(args) == 2 and type(args[0]) is torch.Tensor and _leading_mask(args[0], args[1]) and self._is_all_true(args[1]):
            self.masked += 1
            return args[0].clone()                                  # y[all-true mask] is y, row for row (eager indexing copies: so does this)
        elif func is torch.Tensor.__setitem__ and len(args) == 3 and type(args[0]) is torch.Tensor and _leading_mask(args[0], args[1]) and self._is_all_true(args[1]) \
                and isinstance(args[2], torch.Tensor) and not isinstance(args[2], _MaskedSelection):
            x, mask, v = args
            x.copy_(torch.broadcast_to(v.to(x.dtype), x.shape))      # x[all-true mask] = v
            self.rewritten += 1
            return None
        elif func is torch.Tensor.__getitem__ and len(args) == 2 and type(args[0]) is torch.Tensor and _leading_mask(args[0], args[1]) and args[0].device.type == self.device.type:
            self.masked += 1
            return _MaskedSelection.make(args[0], args[1])          # y[mask]: evaluated only if something other than `x[mask] = ...` wants it
        elif func is torch.Tensor.__setitem__ and len(args) == 3 and type(args[0]) is torch.Tensor and _leading_mask(args[0], args[1]):
            x, mask, v = args
            if isinstance(v, _MaskedSelection):
                if v._msk_mask is mask and v._msk_src.shape == x.shape and v._msk_src.dtype == x.dtype:       # x[mask] = y[mask]
                    v._unchanged()
                    with torch._C.DisableTorchFunctionSubclass():
                        x.copy_(torch.where(_broadcast_mask(mask, x), v._msk_src, x))
                    self.rewritten += 1
                    return None
                return func(x, mask, v._real())
            if isinstance(v, torch.Tensor) and v.numel() == 1 and v.device == x.device and v.dtype == x.dtype:      # x[mask] = a one-element device tensor
                x.copy_(torch.where(_broadcast_mask(mask, x), v.reshape(()), x))
                self.rewritten += 1
                return None
            if isinstance(v, torch.Tensor) and mask.ndim < x.ndim and tuple(v.shape) == tuple(x.shape[mask.ndim:]) and v.device == x.device and v.dtype == x.dtype:
                x.copy_(torch.where(_broadcast_mask(mask, x), v, x))      # x[mask] = one row for every selected row (sapien_utils.py:353)
                self.rewritten += 1
                return None
        elif func is torch.Tensor.__getitem__ and len(args) == 2 and isinstance(args[1], (list, tuple)):
            # data[..., [2]] (render/shaders.py:80): a Python list as an index becomes an index tensor, made on the host and uploaded, at every evaluation
            idx, dev, site = args[1], args[0].device, None

            def conv(ix, k):
                nonlocal site
                if isinstance(ix, list) and ix and all(type(v) is int for v in ix):
                    site = site or self._site()
                    return self._serve(site + ("index", k), tuple(ix), lambda: torch.tensor(ix, dtype=torch.long, device=dev))
                return ix
            new_idx = conv(idx, 0) if isinstance(idx, list) else tuple(conv(ix, k) for k, ix in enumerate(idx))
            if site is not None:
                return func(args[0], new_idx)
        return func(*args, **kwargs)


# --------------------------------------------------------------------------------------------------------------------- is a step safe to replay?
def graph_safety(step_fn, action, settle: int = 2) -> dict:
    """What a HIP-graph replay of ``step_fn(action)`` would get wrong, found on the operator stream of two consecutive eager steps (after ``settle`` unwatched
    ones; works on any device, the CPU checker included):

    * ``sync``: operators that wait for the device or size their result by its data (``nonzero``, boolean-mask indexing, ``.item()``): a capture refuses them;
    * ``flow``: an operator of the second step reads a tensor the FIRST step allocated -- state handed from one step to the next through fresh memory.  This one
      is silent: the capture succeeds and every replay re-reads the memory that was current at capture time (RotateSingleObjectInHand keeps its previous unit
      vector that way).  State has to live in tensors that persist and are updated in place.

    An empty ``sync`` and ``flow`` is what ``accelerate(env, graph=True)`` requires before it captures the reference's own step (round 4 consulted a hand-kept list
    of task ids instead)."""
    import os
    import traceback
    from torch.utils._python_dispatch import TorchDispatchMode

    def tensors(x, out):
        if isinstance(x, torch.Tensor):
            out.append(x)
        elif isinstance(x, (list, tuple)):
            for v in x:
                tensors(v, out)
        elif isinstance(x, dict):
            for v in x.values():
                tensors(v, out)
        return out

    def site():
        for f in reversed(traceback.extract_stack(limit=60)[:-2]):
            if "/torch/" not in f.filename and f.filename != __file__ and f.name != "__torch_function__":
                return f"{os.path.basename(f.filename)}:{f.lineno}"
        return "?"

    class Watch(TorchDispatchMode):
        def __init__(self, earlier):
            super().__init__()
            self.earlier, self.made, self.keep, self.sync, self.flow, self.host = earlier, set(), [], [], [], []
            self.hostmade = set()      # storages made from host data WITHOUT a device (torch.tensor([body.mass ...]), structs/base.py:274): they live on the host on any backend

        def __torch_dispatch__(self, func, types, args=(), kwargs=None):
            name = str(func)
            ins = tensors([args, kwargs or {}], [])
            if any(t in name for t in ("_local_scalar_dense", "nonzero", "masked_select", "aten.equal", "is_nonzero", "unique")):
                # (.item() of host data the step has just made -- `link.mass[0].item()`, control/hopper.py:196 -- reads host memory: no wait for the device)
                if not ("_local_scalar_dense" in name and ins and ins[0].untyped_storage().data_ptr() in self.hostmade):
                    self.sync.append(f"{name} @ {site()}")
            if any(t in name for t in ("aten.index.Tensor", "aten.index_put")) and len(args) > 1 and isinstance(args[1], (list, tuple)):      # (index_select / index_add / ...: args[1] is a dim)
                for ix in args[1]:
                    if isinstance(ix, torch.Tensor) and ix.dtype in (torch.bool, torch.uint8):
                        v = args[2] if len(args) > 2 else None
                        if "index_put" in name and v is not None and v.numel() == 1 and v.device.type == "cpu" and len(args[1]) == 1:
                            continue        # x[mask] = scalar: dispatched to masked_fill, no nonzero()
                        self.sync.append(f"{name} with a mask @ {site()}")
            if "lift_fresh" in name and ins and ins[0].ndim > 0:
                # host data turned into a tensor inside the step: with `device=` in the same call DeviceConstants serves it (then this op does not appear);
                # `torch.tensor(array).to(device)` makes the host tensor here and uploads in `.to`, which DeviceConstants serves on a GPU: listed, not counted
                self.host.append(f"host data of shape {tuple(ins[0].shape)} @ {site()}")
                if ins[0].device.type == "cpu":
                    self.hostmade.add(ins[0].untyped_storage().data_ptr())
            for t in ins:
                if t.untyped_storage().nbytes() > 0 and t.untyped_storage().data_ptr() in self.earlier:      # (empty tensors share the null address)
                    self.flow.append(f"{name} reads a tensor the previous step allocated @ {site()}")
            out = func(*args, **(kwargs or {}))
            inp = {t.untyped_storage().data_ptr() for t in ins}
            for t in tensors(out, []):
                if t.untyped_storage().nbytes() > 0 and t.untyped_storage().data_ptr() not in inp:
                    self.made.add(t.untyped_storage().data_ptr())
                    self.keep.append(t)            # alive until the next step was watched: its address is not handed out again
            return out
    for _ in range(settle):
        step_fn(action)
    w1 = Watch(set())
    with w1:
        step_fn(action)
    w2 = Watch(w1.made)
    with w2:
        step_fn(action)
    return dict(sync=sorted(set(w1.sync + w2.sync)), flow=sorted(set(w2.flow)), host_data=sorted(set(w1.host + w2.host)))


def _verdict(step_fn, action) -> dict:
    """graph_safety for accelerate(): whatever the watch itself trips over (an operator signature it does not know, an error of the task's step under
    DeviceConstants) becomes ``Unsupported`` -- the caller gets a verdict or a refusal, never a TypeError out of the watch."""
    try:
        return graph_safety(step_fn, action)
    except Unsupported:
        raise
    except Exception as e:      # noqa: BLE001
        raise Unsupported(f"the step could not be watched for graph safety: {type(e).__name__}: {str(e).splitlines()[0][:300] if str(e) else ''}") from e


# --------------------------------------------------------------------------------------------------------------------- entry point
class Accelerated:
    """What accelerate() installed on an env: ``.level`` ("control" | "task" | "graph": the reference's own task code, captured | "graph-dry"), ``.graph`` (the
    StepGraph or None), ``.restore()``.  The index tensors, the plugin and the graph belong to ONE scene: when the env was reconfigured since
    (``reset(options=dict(reconfigure=True))``, ``reconfiguration_freq``: sapien_env.py:900-915 builds a new scene and a new ``px``) the next step rebuilds them for the
    new scene -- without the graph (its capture would run throw-away steps in the middle of an episode; call ``accelerate(env, graph=True)`` again after a
    reconfiguration to have one)."""

    def __init__(self, env, graph, task):
        self.env, self.base = env, env.unwrapped
        self._want_graph, self._want_task = graph, task
        self._saved = [(n, n in self.base.__dict__, self.base.__dict__.get(n)) for n in ("_step_action", "step")]
        self.graph = self.plugin = self.constants = self.plugin_refused = self.safety = None
        self.rebuilds = 0
        self.throwaway_steps = 0         # control steps (zero action) the build ran on the env: the watch's 4, a capture's warm-up steps
        try:
            self._build(graph)
        except BaseException:
            self.restore()
            raise

    def restore(self):
        for name, had, val in self._saved:
            if had:
                setattr(self.base, name, val)
            elif name in self.base.__dict__:
                delattr(self.base, name)
        if self.plugin is not None and hasattr(self.plugin, "restore"):
            self.plugin.restore()
        self.graph = None

    # the two entry points the env sees
    def _stale(self):
        """a reconfigured env (new scene, new px) or another control mode (agent.set_control_mode: another controller object)"""
        return self.base.scene is not self.scene or self.base.agent.controller is not self.control.ctrl

    def _reference(self, name, action):
        """the reference's own method, with the instance-level replacements out of the way for the call ({'control_mode': ..., 'action': ...} actions:
        sapien_env.py:1086-1093 switches the controller, then ``_stale()`` rebuilds for the new one)"""
        base, mine = self.base, {}
        for n in ("_step_action", "step"):
            if n in base.__dict__:
                mine[n] = base.__dict__.pop(n)
        try:
            return getattr(base, name)(action)
        finally:
            base.__dict__.update(mine)

    def _step_action(self, action):
        if isinstance(action, dict):
            return self._reference("_step_action", action)
        if self._stale():
            self._rebuild()
        return self._control_fn(action)

    def _step(self, action):
        if isinstance(action, dict):
            return self._reference("step", action)
        if self._stale():
            self._rebuild()
            return self.base.step(action)
        return self._step_fn(action)

    def _rebuild(self):
        import warnings
        self.rebuilds += 1
        had_graph = self.graph is not None
        try:
            self._build("dry" if self._want_graph == "dry" else False)
            if had_graph:
                warnings.warn("maniskill_amd.fused_step: the env was reconfigured; its control step runs fused but no longer as a HIP graph (accelerate(env, graph=True) again)")
        except Unsupported as e:
            self.restore()
            warnings.warn(f"maniskill_amd.fused_step: the reconfigured env is not accelerated any more: {e}")

    def _eid(self):
        return getattr(getattr(self.base, "spec", None), "id", None) or getattr(getattr(self.env, "spec", None), "id", None)

    def _build(self, graph):
        base, env = self.base, self.env
        for n, _, _ in self._saved:                  # while building, the env is the reference's again
            base.__dict__.pop(n, None)
        self.graph = self.plugin = self.constants = None
        control = self.control = FusedControl(base)
        control.boundary = _Boundary(base.scene)
        self.scene = base.scene
        self._control_fn = control
        plugin = None
        if self._want_task:
            eid = self._eid()
            for P in _PLUGINS:       # the first that takes the env: a plugin on the fused task kernels before its torch restatement (task="torch": the latter only)
                if eid in P.env_ids and not (self._want_task == "torch" and issubclass(P, _KernelStep)):
                    try:
                        plugin = P(base, control)
                        self.plugin_refused = None
                        break
                    except Unsupported as e:       # (camera observations, another robot, ...): the task's own code stays, behind the fused controller
                        self.plugin_refused = str(e)
        base._step_action = self._step_action
        self.level = "control"
        cls_step = type(base).step
        if plugin is not None:
            self.level, self.plugin = getattr(plugin, "level", "task"), plugin
            self._step_fn = plugin.step
            if graph and graph not in ("dry", "watch"):      # (a plugin's step is written for replay: no verdict needed)
                from .graph import StepGraph
                try:
                    g = self.graph = StepGraph(getattr(plugin, "kernel_step", plugin.step), base.num_envs, control.adim, base.device)
                    self.throwaway_steps += g.executed_steps      # the warm-up steps run; the captured one is recorded, not executed
                except Exception as e:      # noqa: BLE001  (not a GPU env, host-memory backend, a capture error): the caller falls back to the task level
                    if base.device.type == "cuda":
                        torch.cuda.synchronize()
                    if isinstance(e, Unsupported):
                        raise
                    raise Unsupported(f"the task plugin's step cannot be captured as a HIP graph: {str(e).splitlines()[0][:300]}") from e
                usable = getattr(plugin, "_usable", lambda: True)      # (decided per step on the host: a replay cannot)
                self._step_fn = lambda action: g(action) if (action is not None and usable()) else plugin.step(action)
            base.step = self._step
        elif graph:
            for name, factory in _METHOD_PATCHES.get(self._eid(), {}).items():
                if name not in [n for n, _, _ in self._saved]:
                    self._saved.append((name, name in base.__dict__, base.__dict__.get(name)))
                setattr(base, name, factory(base))
            # no plugin: the reference's own BaseEnv.step (its get_info / get_obs / get_reward) behind the fused controller.  Capturable when the task's code does
            # not synchronise inside the step (PickCube-v1, RollBall-v1, PushCube-v1, PegInsertionSide-v1, ...: tests/ref_fused_step.py graph_safe lists what a task
            # does); host constants made inside the step are served from the device (DeviceConstants); a task that synchronises (StackCube-v1: `reward[mask] =
            # tensor`) fails the capture and is left as the reference built it
            consts = self.constants = DeviceConstants(base.device)
            consts.all_true = lambda: base.scene._reset_mask      # all True outside a reset (sapien_env.py:880-882, 975: a fresh all-true tensor ends every reset)

            # State a task hands from one step to the next by REBINDING an attribute to a tensor the step has just made (`self.prev_unit_vector = new_unit_vector`,
            # dexterity/rotate_single_object_in_hand.py:261; the Draw tasks' dot bookkeeping): a replayed graph would re-read the memory that was current when it was
            # captured.  For the attributes found to do that (below), the value moves into ONE persistent tensor at the end of every step -- a copy the capture records
            # -- and the attribute is bound to it: same values at every step, the state now lives where a replay finds it.  A reset may rebind the attribute too
            # (ibid. :209): the buffer adopts the value before the next step runs.
            persist = self.persist = {}

            def adopt():
                for name, buf in persist.items():
                    cur = base.__dict__.get(name)
                    if cur is not buf and isinstance(cur, torch.Tensor) and cur.shape == buf.shape and cur.dtype == buf.dtype and cur.device == buf.device:
                        buf.copy_(cur)
                        base.__dict__[name] = buf

            def captured_step(a):
                with consts:
                    out = cls_step(base, a)
                    if persist:
                        adopt()
                    return out

            def host_state():
                return {k: x for k, x in base.__dict__.items() if type(x) in (int, float, bool, str)}

            def verdict_with_persistent_state():
                zero = torch.zeros(base.num_envs, control.adim, device=base.device)
                h0 = host_state()
                v = _verdict(captured_step, zero)
                self.throwaway_steps += 4
                # the third hazard, as silent as the second: state the step keeps in PYTHON (the Draw tasks count their dots in `self.draw_step` and pick this step's
                # actor with it, drawing/draw.py:185-188) -- a replay runs no Python, it would move the dot the capture saw for ever
                moved = sorted(k for k, x in host_state().items() if k in h0 and h0[k] != x)
                if moved:
                    v = dict(v, flow=v["flow"] + [f"the step changes the Python attribute `{k}` ({h0[k]!r} -> {base.__dict__[k]!r}): a replay would not" for k in moved])
                    return v
                if v["flow"] and not v["sync"]:
                    before = {k: t for k, t in base.__dict__.items() if isinstance(t, torch.Tensor)}
                    captured_step(zero)
                    self.throwaway_steps += 1
                    for k, t in list(base.__dict__.items()):
                        o = before.get(k)
                        if isinstance(t, torch.Tensor) and o is not None and t is not o and t.shape == o.shape and t.dtype == o.dtype and t.device == o.device \
                                and not t.requires_grad:
                            persist[k] = base.__dict__[k] = t.clone()
                    if persist:
                        v2 = _verdict(captured_step, zero)
                        self.throwaway_steps += 4
                        if not v2["flow"] and not v2["sync"]:
                            return v2
                        for k in list(persist):      # (it did not help: the env stays as the task wrote it)
                            base.__dict__[k] = persist.pop(k).clone()
                return v
            if graph == "dry":          # everything the capture would run, eagerly at every step (no GPU needed: the CPU suite checks the results and the op stream)
                self.level = "graph-dry"
                self._step_fn = captured_step
                verdict_with_persistent_state()      # (state attributes move into persistent tensors here too: the eager run checks what a capture would run)
            elif graph == "watch":      # the verdict alone (any device): what graph=True decides on
                self.level = "graph-dry"
                self._step_fn = captured_step
                self.safety = verdict_with_persistent_state()
            else:
                if graph is True:       # (graph="force" captures without asking)
                    verdict = self.safety = verdict_with_persistent_state()
                    if verdict["sync"] or verdict["flow"]:
                        what = "; ".join((verdict["sync"] + verdict["flow"])[:3])
                        raise Unsupported(f"the task's own step is not safe to replay as a graph: {what}" + (" (a capture refuses the first kind; the second kind -- state "
                                          "handed to the next step through a freshly allocated tensor -- would capture and replay stale)" if verdict["flow"] else ""))
                from .graph import StepGraph
                try:
                    g = self.graph = StepGraph(captured_step, base.num_envs, control.adim, base.device)
                    self.throwaway_steps += g.executed_steps      # the warm-up steps run; the captured one is recorded, not executed
                except Exception as e:      # noqa: BLE001  (a capture error: HIP reports the forbidden call)
                    if base.device.type == "cuda":
                        torch.cuda.synchronize()
                    if isinstance(e, Unsupported):
                        raise
                    culprit = ""
                    if base.device.type == "cuda":
                        from .graph import probe_capture
                        try:
                            culprit = probe_capture(captured_step, torch.zeros(base.num_envs, control.adim, device=base.device), base.device)
                        except Exception as pe:   # noqa: BLE001 -- a diagnostic must not mask the error it explains
                            culprit = f"(probe failed: {str(pe).splitlines()[0][:120]})"
                    raise Unsupported(f"the task's step cannot be captured as a HIP graph: {str(e).splitlines()[0][:300]}" + (f" -- first offender: {culprit}" if culprit else "")) from e
                self.level = "graph"
                # The tensors the env's attributes named when the step was captured are the ones a replay reads and writes.  A reset may bind an attribute to a new
                # tensor (`self.cum_rotation_angle = torch.zeros((b,))` when every sub-scene is reset, rotate_single_object_in_hand.py:210, while a step updates it in
                # place, :276): before a replay such an attribute's value is copied into the captured tensor and the attribute bound to it again.
                held = self._held = {k: t for k, t in base.__dict__.items() if isinstance(t, torch.Tensor) and t.device == base.device}

                def replayed(action):
                    d = base.__dict__
                    for name, t0 in held.items():
                        cur = d.get(name)
                        if cur is not t0 and isinstance(cur, torch.Tensor) and cur.shape == t0.shape and cur.dtype == t0.dtype and cur.device == t0.device:
                            t0.copy_(cur)
                            d[name] = t0
                    return g(action) if action is not None else cls_step(base, None)
                self._step_fn = replayed
            base.step = self._step
        base._msk_accelerated = self


def accelerate(env, graph=False, task: bool = True) -> Accelerated:
    """Install the fused control step on ``env`` (anything ``gym.make`` returned: the wrappers stay, ``env.unwrapped`` gets instance-level
    replacements of ``_step_action`` and -- where a task plugin exists and ``task`` is true -- of ``step``).  Raises ``Unsupported`` and leaves the env as
    it was when the env uses a controller / hook / observation mode that is not restated here.  ``graph=True`` (task level, GPU) additionally captures the
    control step as one HIP graph (with a task plugin: the plugin's step; without: the reference's own ``BaseEnv.step`` behind the fused controller, for tasks
    whose code is capturable); call ``env.reset`` afterwards (the capture and the watch before it run ``.throwaway_steps`` throw-away steps with a zero action --
    a reset of the REFERENCE does not undo everything a step did: drive targets stay in the simulation until the next action, and OpenCabinetDrawer-v1's
    ``_initialize_episode`` steps the physics once (open_cabinet_drawer.py:276-283), so an env that has stepped resets to a slightly different state than a
    fresh one -- on PhysX as here).  ``graph="dry"``: what the capture would run, run eagerly at
    every step -- for the CPU suite and for debugging; ``graph="watch"``: the same plus ``.safety``, the verdict of ``graph_safety`` that ``graph=True`` asks
    for before it captures a task's own step; ``graph="force"``: capture without asking."""
    return Accelerated(env, graph, task)
