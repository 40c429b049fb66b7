"""Batched views of the simulator's buffers under the reference's struct names.

``Pose`` / ``Actor`` / ``Link`` / ``Articulation`` of mani_skill/utils/structs/{pose,actor,link,articulation,base}.py: every
property is a gather from (or a masked write into) the torch-visible ``sapien``-style buffers of ``PhysxGpuSystem`` — row
``env * bodies_per_env + body`` of ``cuda_rigid_body_data`` instead of the reference's ``_body_data_index`` tensor, because every
sub-scene is an instance of one template.  As in the reference's GPU mode, setters only write the buffers: the
caller commits them with ``px.gpu_apply_*`` (``scene._gpu_apply_all()``, envs/scene.py:950-966) and reads fresh values after
``px.gpu_fetch_*``.  Poses are reported without the sub-scene offset (structs/actor.py:341-365).

A ``SceneView`` builds these objects for every body / articulation of a template; the envs of this package expose it as
``env.scene`` next to their own fused or torch task code, so a task written against the reference's structs reads the same here.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Union

import numpy as np
import torch

from . import _native as N


# ------------------------------------------------------------------------------------------------ Pose
def _qmul(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    aw, ax, ay, az = a.unbind(-1)
    bw, bx, by, bz = b.unbind(-1)
    q = torch.stack([aw * bw - ax * bx - ay * by - az * bz, aw * bx + ax * bw + ay * bz - az * by,
                     aw * by - ax * bz + ay * bw + az * bx, aw * bz + ax * by - ay * bx + az * bw], dim=-1)
    return torch.where(q[..., :1] < 0, -q, q)   # as quaternion_multiply of the reference: non-negative real part


def _qrot(q: torch.Tensor, v: torch.Tensor) -> torch.Tensor:
    u = q[..., 1:]
    t = 2.0 * torch.cross(u, v, dim=-1)
    return v + q[..., :1] * t + torch.cross(u, t, dim=-1)


class Pose:
    """structs/pose.py: a batch of rigid transforms as one ``raw_pose`` tensor (N, 7) = [p | q wxyz]."""

    def __init__(self, raw_pose: torch.Tensor):
        self.raw_pose = raw_pose if raw_pose.ndim == 2 else raw_pose[None]

    @classmethod
    def create_from_pq(cls, p=None, q=None, device=None) -> "Pose":
        p = torch.zeros(1, 3) if p is None else torch.as_tensor(p, dtype=torch.float32)
        q = torch.tensor([[1.0, 0.0, 0.0, 0.0]]) if q is None else torch.as_tensor(q, dtype=torch.float32)
        p, q = (p if p.ndim == 2 else p[None]), (q if q.ndim == 2 else q[None])
        n = max(len(p), len(q))
        raw = torch.cat([p.expand(n, 3), q.expand(n, 4).to(p.device)], dim=-1)
        return cls(raw.to(device) if device is not None else raw)

    @classmethod
    def create(cls, pose) -> "Pose":
        return pose if isinstance(pose, Pose) else cls(torch.as_tensor(pose, dtype=torch.float32))

    @property
    def p(self): return self.raw_pose[:, :3]
    @property
    def q(self): return self.raw_pose[:, 3:]
    @property
    def device(self): return self.raw_pose.device
    @property
    def shape(self): return self.raw_pose.shape
    def get_p(self): return self.p
    def get_q(self): return self.q
    def __len__(self): return len(self.raw_pose)
    def __getitem__(self, i): return Pose(self.raw_pose[i])
    def to(self, device): return Pose(self.raw_pose.to(device))

    def __mul__(self, other) -> "Pose":
        o = Pose.create(other).raw_pose.to(self.device)
        return Pose(torch.cat([self.p + _qrot(self.q, o[:, :3]), _qmul(self.q, o[:, 3:])], dim=-1))

    def inv(self) -> "Pose":
        qi = self.q * torch.tensor([1.0, -1.0, -1.0, -1.0], device=self.device)
        return Pose(torch.cat([-_qrot(qi, self.p), qi], dim=-1))

    def to_transformation_matrix(self) -> torch.Tensor:
        w, x, y, z = self.q.unbind(-1)
        R = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
                         2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
                         2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], dim=-1).view(-1, 3, 3)
        T = torch.eye(4, device=self.device).repeat(len(self), 1, 1)
        T[:, :3, :3] = R
        T[:, :3, 3] = self.p
        return T


# ------------------------------------------------------------------------------------------------ bodies
class _Body:
    """PhysxRigidBodyComponentStruct (structs/base.py:98-470) over one template body."""

    def __init__(self, view: "SceneView", body: int, name: str):
        self._view, self.body, self.name = view, int(body), name
        self._force_query = None

    def __repr__(self):
        return f"<{type(self).__name__} {self.name!r} body {self.body} x {self._view.num_envs} envs>"

    @property
    def px(self): return self._view.px
    @property
    def device(self): return self._view.px.device
    @property
    def _rows(self) -> torch.Tensor:
        self._view.fresh()
        return self._view.rbd[:, self.body]

    def _idx(self, env_idx):
        return slice(None) if env_idx is None else torch.as_tensor(env_idx, device=self.device, dtype=torch.long)

    # reads -------------------------------------------------------------------------------------------------------------
    @property
    def pose(self) -> Pose:
        raw = self._rows[:, :7].clone()
        raw[:, :3] -= self._view.offsets
        return Pose(raw)

    @property
    def linear_velocity(self) -> torch.Tensor: return self._rows[:, 7:10].clone()
    @property
    def angular_velocity(self) -> torch.Tensor: return self._rows[:, 10:13].clone()
    def get_linear_velocity(self): return self.linear_velocity
    def get_angular_velocity(self): return self.angular_velocity
    def get_pose(self): return self.pose

    @property
    def per_scene_id(self) -> torch.Tensor:
        """Segmentation id of the body in every sub-scene (structs/actor.py:305-314): body id + 1, 0 is the background."""
        return torch.full((self._view.num_envs,), self.body + 1, dtype=torch.int32, device=self.device)

    @property
    def mass(self) -> torch.Tensor:
        return torch.full((self._view.num_envs,), self._view.template.body_masses[self.body], device=self.device)

    def get_net_contact_impulses(self) -> torch.Tensor:
        """(N, 3) sum of the last step's contact impulses on this body (structs/base.py:116-136)."""
        if self._force_query is None:
            self._force_query = self.px.gpu_create_contact_body_impulse_query([self.body])
        self.px.gpu_query_contact_body_impulses(self._force_query)
        return self._force_query.cuda_impulses.torch().view(self._view.num_envs, 3).clone()

    def get_net_contact_forces(self) -> torch.Tensor:
        return self.get_net_contact_impulses() / self.px.timestep

    def is_static(self, lin_thresh=1e-2, ang_thresh=1e-1) -> torch.Tensor:
        """structs/actor.py:220-227."""
        r = self._rows
        return (torch.linalg.norm(r[:, 7:10], dim=1) <= lin_thresh) & (torch.linalg.norm(r[:, 10:13], dim=1) <= ang_thresh)


class Actor(_Body):
    """structs/actor.py: a dynamic or kinematic actor of every sub-scene."""

    @property
    def px_body_type(self) -> str:
        return "kinematic" if self._view.template.body_kind[self.body] == N.BODY_KINEMATIC else "dynamic"

    # writes (buffer only; commit with px.gpu_apply_rigid_dynamic_data()) -------------------------------------------------
    def set_pose(self, pose, env_idx=None):
        raw = Pose.create(pose).raw_pose.to(self.device)
        i = self._idx(env_idx)
        rows = self._rows
        off = self._view.offsets[i]
        rows[i, :3] = raw[:, :3] + off
        rows[i, 3:7] = raw[:, 3:]

    def set_linear_velocity(self, v, env_idx=None):
        self._rows[self._idx(env_idx), 7:10] = torch.as_tensor(v, dtype=torch.float32, device=self.device)

    def set_angular_velocity(self, w, env_idx=None):
        self._rows[self._idx(env_idx), 10:13] = torch.as_tensor(w, dtype=torch.float32, device=self.device)

    def get_state(self) -> torch.Tensor:
        """(N, 13) [pose | linear | angular velocity] (structs/actor.py:132-140)."""
        s = self._rows.clone()
        s[:, :3] -= self._view.offsets
        return s

    def set_state(self, state, env_idx=None):
        state = torch.as_tensor(state, dtype=torch.float32, device=self.device)
        state = state if state.ndim == 2 else state[None]
        self.set_pose(state[:, :7], env_idx)
        self.set_linear_velocity(state[:, 7:10], env_idx)
        self.set_angular_velocity(state[:, 10:13], env_idx)

    def apply_force(self, force):
        """structs/actor.py:316-322: committed at once, acts during the next px.step() only."""
        self.px.apply_force(self.body, force)


class Link(_Body):
    """structs/link.py: a link of the articulation of every sub-scene (pose and velocities come out of the kinematics)."""

    def __init__(self, view, body, name, articulation: "Articulation", index: int):
        super().__init__(view, body, name)
        self.articulation, self.index = articulation, index


class Articulation:
    """structs/articulation.py over articulation ``art`` of the template."""

    def __init__(self, view: "SceneView", art: int, name: str):
        tpl = view.template
        self._view, self.art, self.name = view, int(art), name
        self.links: List[Link] = [Link(view, b, tpl.body_names[b], self, k) for k, b in enumerate(tpl.art_links[art])]
        self.links_map: Dict[str, Link] = {l.name: l for l in self.links}
        self.root = self.links[0]
        self._active = list(tpl.art_active[art])                  # body ids of the links whose inbound joint is active, dof order
        self.active_joint_names = [tpl.joint_names[b] for b in self._active]
        self.dof_count = len(self._active)
        lim = np.array([tpl.joint_limits[b] for b in self._active], dtype=np.float32).reshape(-1, 2)
        self._qlimits = torch.from_numpy(lim).to(view.px.device)
        self._force_queries = {}

    def __repr__(self):
        return f"<Articulation {self.name!r}: {len(self.links)} links, {self.dof_count} dof x {self._view.num_envs} envs>"

    @property
    def px(self): return self._view.px
    @property
    def device(self): return self._view.px.device
    @property
    def max_dof(self) -> int: return self.px.max_dof
    @property
    def dof(self) -> torch.Tensor: return torch.full((self._view.num_envs,), self.dof_count, device=self.device)
    def get_dof(self): return self.dof_count
    def get_name(self): return self.name
    def get_links(self): return self.links
    def get_root(self): return self.root
    def get_active_joints(self): return list(self.active_joint_names)
    def find_link_by_name(self, name: str) -> Link: return self.links_map[name]

    def _buf(self, handle) -> torch.Tensor:
        self._view.fresh()
        n, na = self._view.num_envs, max(self.px.arts_per_env, 1)
        return handle.torch().view(n, na, -1)[:, self.art, :self.dof_count]

    def _idx(self, env_idx):
        return slice(None) if env_idx is None else torch.as_tensor(env_idx, device=self.device, dtype=torch.long)

    # joint state ---------------------------------------------------------------------------------------------------------
    @property
    def qpos(self): return self._buf(self.px.cuda_articulation_qpos).clone()
    @property
    def qvel(self): return self._buf(self.px.cuda_articulation_qvel).clone()
    @property
    def qacc(self): return self._buf(self.px.cuda_articulation_qacc).clone()
    @property
    def qf(self): return self._buf(self.px.cuda_articulation_qf).clone()
    @property
    def qlimits(self): return self._qlimits[None].expand(self._view.num_envs, -1, -1)
    @property
    def drive_targets(self): return self._buf(self.px.cuda_articulation_target_qpos).clone()
    @property
    def drive_velocities(self): return self._buf(self.px.cuda_articulation_target_qvel).clone()
    def get_qpos(self): return self.qpos
    def get_qvel(self): return self.qvel
    def get_qf(self): return self.qf
    def get_qlimits(self): return self.qlimits
    def get_drive_targets(self): return self.drive_targets
    def get_drive_velocities(self): return self.drive_velocities

    def _write(self, handle, value, env_idx=None, joint_indices=None):
        v = torch.as_tensor(value, dtype=torch.float32, device=self.device)
        cols = slice(None) if joint_indices is None else torch.as_tensor(joint_indices, device=self.device, dtype=torch.long)
        buf = self._buf(handle)
        if env_idx is None:
            buf[:, cols] = v
        else:
            rows = torch.as_tensor(env_idx, device=self.device, dtype=torch.long)
            if joint_indices is None:
                buf[rows] = v
            else:
                buf[rows[:, None], cols[None, :]] = v

    def set_qpos(self, qpos, env_idx=None): self._write(self.px.cuda_articulation_qpos, qpos, env_idx)       # then gpu_apply_articulation_qpos
    def set_qvel(self, qvel, env_idx=None): self._write(self.px.cuda_articulation_qvel, qvel, env_idx)
    def set_qf(self, qf, env_idx=None): self._write(self.px.cuda_articulation_qf, qf, env_idx)

    def set_joint_drive_targets(self, targets, joints=None, joint_indices=None, env_idx=None):
        """structs/articulation.py:873-896 (commit with px.gpu_apply_articulation_target_position())."""
        self._write(self.px.cuda_articulation_target_qpos, targets, env_idx, joint_indices)

    def set_joint_drive_velocity_targets(self, targets, joints=None, joint_indices=None, env_idx=None):
        self._write(self.px.cuda_articulation_target_qvel, targets, env_idx, joint_indices)

    # root / state --------------------------------------------------------------------------------------------------------
    @property
    def root_pose(self) -> Pose: return self.root.pose
    @property
    def pose(self) -> Pose: return self.root.pose
    def get_root_pose(self): return self.root_pose
    def get_root_linear_velocity(self): return self.root.linear_velocity
    def get_root_angular_velocity(self): return self.root.angular_velocity

    def set_root_pose(self, pose, env_idx=None):
        """Buffer write of the root link's row (commit with px.gpu_apply_articulation_root_pose())."""
        raw = Pose.create(pose).raw_pose.to(self.device)
        i = self._idx(env_idx)
        self._view.fresh()
        rows = self._view.rbd[:, self.root.body]
        rows[i, :3] = raw[:, :3] + self._view.offsets[i]
        rows[i, 3:7] = raw[:, 3:]

    def get_state(self) -> torch.Tensor:
        """(N, 13 + 2 dof) [root pose | root velocities | qpos | qvel] (structs/articulation.py:283-289)."""
        self._view.fresh()
        root = self._view.rbd[:, self.root.body].clone()
        root[:, :3] -= self._view.offsets
        return torch.cat([root, self.qpos, self.qvel], dim=1)

    def set_state(self, state, env_idx=None):
        state = torch.as_tensor(state, dtype=torch.float32, device=self.device)
        state = state if state.ndim == 2 else state[None]
        self.set_root_pose(state[:, :7], env_idx)
        self._view.rbd[:, self.root.body][self._idx(env_idx), 7:13] = state[:, 7:13]
        self.set_qpos(state[:, 13:13 + self.dof_count], env_idx)
        self.set_qvel(state[:, 13 + self.dof_count:13 + 2 * self.dof_count], env_idx)

    # contacts / forces ---------------------------------------------------------------------------------------------------
    def get_net_contact_impulses(self, link_names: Sequence[str]) -> torch.Tensor:
        """(N, len(link_names), 3) (structs/articulation.py:441-488)."""
        key = tuple(link_names)
        if key not in self._force_queries:
            self._force_queries[key] = self.px.gpu_create_contact_body_impulse_query([self.links_map[n].body for n in key])
        q = self._force_queries[key]
        self.px.gpu_query_contact_body_impulses(q)
        return q.cuda_impulses.torch().view(self._view.num_envs, len(key), 3).clone()

    def get_net_contact_forces(self, link_names: Sequence[str]) -> torch.Tensor:
        return self.get_net_contact_impulses(link_names) / self.px.timestep

    def get_link_incoming_joint_forces(self) -> torch.Tensor:
        """(N, links, 6) (structs/articulation.py:596-620)."""
        n, na = self._view.num_envs, max(self.px.arts_per_env, 1)
        return self.px.get_link_incoming_joint_forces().view(n, na, -1, 6)[:, self.art, :len(self.links)].clone()


# ------------------------------------------------------------------------------------------------ scene
class SceneView:
    """The actors and articulations of every sub-scene by name (ManiSkillScene.actors / .articulations, envs/scene.py:80-120)."""

    def __init__(self, px, fresh=None):
        self.px, self.template = px, px.template
        self.num_envs = px.num_envs
        self.rbd = px.cuda_rigid_body_data.torch().view(px.num_envs, px.bodies_per_env, 13)
        self._fresh = fresh
        tpl = self.template
        self.articulations: Dict[str, Articulation] = {name: Articulation(self, a, name) for a, name in enumerate(tpl.art_names)}
        self.actors: Dict[str, Actor] = {tpl.body_names[b]: Actor(self, b, tpl.body_names[b])
                                         for b in range(len(tpl.body_names)) if tpl.body_kind[b] != N.BODY_LINK}

    @property
    def offsets(self) -> torch.Tensor:
        return self.px.scene_offsets

    def fresh(self):
        if self._fresh is not None:
            self._fresh()

    def get_sim_state(self) -> dict:
        """envs/scene.py:852-874."""
        return {"actors": {n: a.get_state() for n, a in self.actors.items()},
                "articulations": {n: a.get_state() for n, a in self.articulations.items()}}
