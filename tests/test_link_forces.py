"""cuda_rigid_body_force / cuda_rigid_body_torque rows of ARTICULATION LINKS (the buffers have a row per rigid body component, links included:
mani_skill/utils/structs/actor.py:316-322 is the actors' use of them): a force at the link's centre of mass and a torque, for the next step only.
Known answers on the oracle (a slider: a = F / m; a pendulum arm: qacc = (r x F + tau) . axis / (I + m r^2)); HIP against the oracle under the
emulation of tests/hipemu and on hardware."""
import numpy as np
import pytest
import torch

from maniskill_amd import _native as N
from maniskill_amd.physx import SceneConfig, SceneTemplate, SimConfig


def _scene():
    """a fixed base with (a) a prismatic slider along x, 0.5 kg and (b) behind it a revolute arm about the joint's x axis whose centre of mass
    lies 0.2 m from the axis; no gravity, no drives"""
    tpl = SceneTemplate()
    art = tpl.add_articulation("rig")
    root = tpl.add_link(art, "base", -1, N.JOINT_FIXED, mass=1.0, inertia6=(1e-2,) * 3 + (0, 0, 0))
    slider = tpl.add_link(art, "slider", root, N.JOINT_PRISMATIC, "j_slide", limits=(-10, 10), mass=0.5, inertia6=(1e-3,) * 3 + (0, 0, 0), disable_gravity=True)
    arm = tpl.add_link(art, "arm", root, N.JOINT_REVOLUTE, "j_arm", mass=0.3, com=(0.0, 0.2, 0.0), inertia6=(2e-3, 1e-3, 2e-3, 0, 0, 0), disable_gravity=True)
    return tpl, slider, arm


def _step_with(factory, n, F_slider, F_arm, T_arm, steps=1):
    tpl, slider, arm = _scene()
    px = factory(tpl, n, SimConfig(scene_config=SceneConfig(gravity=(0.0, 0.0, 0.0)))); px.gpu_init()
    F = px.cuda_rigid_body_force.torch().view(n, px.bodies_per_env, 4)
    T = px.cuda_rigid_body_torque.torch().view(n, px.bodies_per_env, 4)
    out = []
    for t in range(steps):
        if t == 0:
            for e in range(n):
                F[e, slider, :3] = torch.tensor(F_slider, device=F.device) * (e + 1)
                F[e, arm, :3] = torch.tensor(F_arm, device=F.device) * (e + 1)
                T[e, arm, :3] = torch.tensor(T_arm, device=F.device) * (e + 1)
            px.gpu_apply_rigid_dynamic_force(); px.gpu_apply_rigid_dynamic_torque()
        px.step(); px.gpu_fetch_all()
        out.append(torch.cat([px.cuda_articulation_qvel.torch().view(n, -1)[:, :2].cpu().clone(), px.cuda_articulation_qpos.torch().view(n, -1)[:, :2].cpu().clone()], 1))
    return torch.stack(out), px


def test_force_and_torque_on_links_known_answers(oracle_factory):
    n = 2
    traj, px = _step_with(oracle_factory, n, (2.0, 7.0, 0.0), (0.0, 0.0, 1.5), (0.4, 9.0, 0.0), steps=3)
    dt = px.timestep
    for e in range(n):
        k = e + 1
        # slider along x: only the x component of the force moves it
        assert abs(traj[0, e, 0].item() - k * 2.0 * dt / 0.5) < 1e-6
        # arm about x with its centre of mass at r = (0, 0.2, 0): torque about the axis = (r x F).x + tau.x = 0.2 * Fz + tau_x; inertia about it = Ixx + m r^2
        want = k * (0.2 * 1.5 + 0.4) * dt / (2e-3 + 0.3 * 0.2 ** 2)
        assert abs(traj[0, e, 1].item() - want) < 1e-5 * max(1.0, abs(want)), (traj[0, e, 1].item(), want)
    # the wrench lasts one step: the slider keeps its speed afterwards (no drive, no damping); the arm's rate stays too
    assert torch.allclose(traj[2, :, 0], traj[0, :, 0], rtol=1e-6, atol=1e-8)
    assert torch.allclose(traj[2, :, 1], traj[0, :, 1], rtol=1e-4, atol=1e-7)


def test_hip_link_forces_match_the_oracle_under_emulation(oracle_factory):
    from emu_backend import EmuPhysxSystem
    a, _ = _step_with(lambda t, k, c: EmuPhysxSystem(t, k, c), 3, (2.0, 7.0, 0.0), (0.3, -0.2, 1.5), (0.4, 9.0, -1.0), steps=4)
    b, _ = _step_with(oracle_factory, 3, (2.0, 7.0, 0.0), (0.3, -0.2, 1.5), (0.4, 9.0, -1.0), steps=4)
    assert torch.equal(a, b), (a - b).abs().max().item()


@pytest.mark.gpu
def test_hip_link_forces_match_the_oracle(oracle_factory):
    from maniskill_amd.physx import PhysxGpuSystem
    a, _ = _step_with(lambda t, k, c: PhysxGpuSystem("cuda:0", t, k, c), 70, (2.0, 7.0, 0.0), (0.3, -0.2, 1.5), (0.4, 9.0, -1.0), steps=4)
    b, _ = _step_with(oracle_factory, 70, (2.0, 7.0, 0.0), (0.3, -0.2, 1.5), (0.4, 9.0, -1.0), steps=4)
    assert torch.equal(a, b), (a - b).abs().max().item()
