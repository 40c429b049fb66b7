"""Coupled multibody dynamics as known answers for the CPU oracle (oracle/orc_sim.c: composite-rigid-body mass matrix + bias forces):
the accelerations of a double pendulum and of a cart-pole released from rest against their Lagrangian equations of motion (written
down here, independently of the engine), and two seconds of a freely swinging double pendulum against those equations stepped here.  Together with
tests/test_oracle_mechanics.py (single pendulum, drives, tendon) this pins the articulation dynamics on closed forms; the HIP kernels are
then held to the oracle bit by bit (tests/test_gpu_parity.py)."""
import numpy as np
import pytest
import torch

from maniskill_amd import _native as N
from maniskill_amd.physx import SceneTemplate, SimConfig

G = 9.81
M1, M2, L1, R1, R2, RAD = 1.3, 0.7, 0.45, 0.45, 0.35, 0.04     # link masses, joint-2 offset along link 1, COM distances, bob radius
IC1, IC2 = 0.4 * M1 * RAD ** 2, 0.4 * M2 * RAD ** 2


def _start(factory, tpl, q0):
    px = factory(tpl, 1, SimConfig())
    px.gpu_init()
    px.set_scene_offsets(np.zeros((1, 3)))
    px.cuda_rigid_body_data.torch().view(1, px.bodies_per_env, 13)[:, 0, :7] = torch.tensor([0.0, 0.0, 2.0, 1, 0, 0, 0])
    px.cuda_articulation_qpos.torch()[0, : len(q0)] = torch.tensor(q0, dtype=torch.float32)
    px.gpu_apply_all()
    return px


def _double_pendulum(factory, q0):
    """Two links on revolute joints about the world x axis; angles from the downward vertical, q2 relative to link 1."""
    tpl = SceneTemplate()
    art = tpl.add_articulation("double", root_p=(0, 0, 2.0))
    base = tpl.add_link(art, "base", -1, N.JOINT_FIXED, mass=1.0, inertia6=(1e-2, 1e-2, 1e-2, 0, 0, 0))
    a = tpl.add_link(art, "upper", base, N.JOINT_REVOLUTE, joint_name="j1", mass=M1, com=(0, 0, -R1), inertia6=(IC1, IC1, IC1, 0, 0, 0))
    tpl.add_link(art, "lower", a, N.JOINT_REVOLUTE, joint_name="j2", pose_in_parent=[0, 0, -L1, 1, 0, 0, 0], mass=M2, com=(0, 0, -R2),
                 inertia6=(IC2, IC2, IC2, 0, 0, 0))
    return _start(factory, tpl, q0)


def _double_pendulum_mass_matrix(a1, a2):
    """in absolute angles (a1, a2 = q1, q1 + q2): T = 1/2 adot^T M adot"""
    c = M2 * L1 * R2 * np.cos(a1 - a2)
    return np.array([[M1 * R1 ** 2 + IC1 + M2 * L1 ** 2, c], [c, M2 * R2 ** 2 + IC2]])


def _double_pendulum_energy(q, qd):
    a1, a2 = q[0], q[0] + q[1]
    ad = np.array([qd[0], qd[0] + qd[1]])
    T = 0.5 * ad @ _double_pendulum_mass_matrix(a1, a2) @ ad
    V = -M1 * G * R1 * np.cos(a1) - M2 * G * (L1 * np.cos(a1) + R2 * np.cos(a2))
    return T + V


@pytest.mark.parametrize("q0", [(0.6, -0.9), (1.4, 0.5), (-2.2, 1.1)])
def test_double_pendulum_released_from_rest_accelerates_as_lagrange_says(oracle_factory, q0):
    """From rest the velocity terms vanish: M(a) addot = -dV/da; one semi-implicit Euler step gives qdot = dt * qddot(q0) exactly."""
    px = _double_pendulum(oracle_factory, q0)
    px.step()
    px.gpu_fetch_all()
    got = px.cuda_articulation_qvel.torch()[0, :2].numpy().astype(np.float64) / px.timestep
    a1, a2 = q0[0], q0[0] + q0[1]
    rhs = np.array([-(M1 * R1 + M2 * L1) * G * np.sin(a1), -M2 * G * R2 * np.sin(a2)])
    add = np.linalg.solve(_double_pendulum_mass_matrix(a1, a2), rhs)
    want = np.array([add[0], add[1] - add[0]])
    assert np.allclose(got, want, rtol=2e-4, atol=2e-4), (got, want)
    assert np.allclose(px.cuda_articulation_qacc.torch()[0, :2].numpy(), want, rtol=2e-4, atol=2e-4)


def test_double_pendulum_swing_follows_the_equations_of_motion(oracle_factory):
    """A frictionless double pendulum swinging through large angles: the engine's trajectory against the textbook equations of motion
    (absolute angles; the coupling terms +- m2 L1 r2 sin(a1 - a2) adot^2 are the centrifugal / Coriolis part) integrated here with the
    same semi-implicit Euler step.  The two are the same discrete system, so they agree to round-off growth over the first half
    second and stay close over two seconds of a chaotic swing; the energy stays within the oscillation symplectic Euler shows at 10 ms."""
    q0 = (2.0, -1.0)
    px = _double_pendulum(oracle_factory, q0)
    q, qd = px.cuda_articulation_qpos.torch(), px.cuda_articulation_qvel.torch()
    dt = px.timestep
    a, ad = np.array([q0[0], q0[0] + q0[1]]), np.zeros(2)
    k = M2 * L1 * R2
    err, es, speed = [], [], 0.0
    e0 = _double_pendulum_energy(np.array(q0), np.zeros(2))
    for step in range(200):
        s_ = np.sin(a[0] - a[1])
        rhs = np.array([-(M1 * R1 + M2 * L1) * G * np.sin(a[0]) - k * s_ * ad[1] ** 2, -M2 * G * R2 * np.sin(a[1]) + k * s_ * ad[0] ** 2])
        ad = ad + dt * np.linalg.solve(_double_pendulum_mass_matrix(a[0], a[1]), rhs)
        a = a + dt * ad
        px.step()
        px.gpu_fetch_all()
        got = q[0, :2].numpy().astype(np.float64)
        err.append(max(abs(got[0] - a[0]), abs(got[0] + got[1] - a[1])))
        es.append(_double_pendulum_energy(got, qd[0, :2].numpy().astype(np.float64)))
        speed = max(speed, float(qd[0, :2].abs().max()))
    assert speed > 4.0                                       # it does swing: the velocity-dependent terms matter
    assert max(err[:50]) < 2e-4 and max(err) < 2e-2, (max(err[:50]), max(err))
    span = (M1 * R1 + M2 * (L1 + R2)) * G * 2                # potential between hanging and inverted
    assert max(abs(e - e0) for e in es) < 0.08 * span and abs(np.mean(es[100:]) - e0) < 0.03 * span, (e0, min(es), max(es), np.mean(es[100:]))


def test_cart_pole_released_from_rest_accelerates_as_lagrange_says(oracle_factory):
    """A cart (prismatic along world y) carrying a pole (revolute about world x), no drives: from rest
    [[M + m, m l cos a], [m l cos a, m l^2 + Ic]] (sddot, addot) = (0, -m g l sin a), a from the downward vertical."""
    Mc, m, l = 2.0, 0.4, 0.5
    Ic = 0.4 * m * RAD ** 2
    turn = [0, 0, 0, np.cos(np.pi / 4), 0, 0, np.sin(np.pi / 4)]          # joint frame x = world y
    for a0 in (0.3, 2.4, -1.2):
        tpl = SceneTemplate()
        art = tpl.add_articulation("cartpole", root_p=(0, 0, 2.0))
        base = tpl.add_link(art, "rail", -1, N.JOINT_FIXED, mass=1.0, inertia6=(1e-2, 1e-2, 1e-2, 0, 0, 0))
        cart = tpl.add_link(art, "cart", base, N.JOINT_PRISMATIC, joint_name="slide", pose_in_parent=turn, pose_in_child=turn, mass=Mc,
                            inertia6=(1e-2, 1e-2, 1e-2, 0, 0, 0))
        tpl.add_link(art, "pole", cart, N.JOINT_REVOLUTE, joint_name="hinge", mass=m, com=(0, 0, -l), inertia6=(Ic, Ic, Ic, 0, 0, 0))
        px = _start(oracle_factory, tpl, (0.0, a0))
        px.step()
        px.gpu_fetch_all()
        got = px.cuda_articulation_qvel.torch()[0, :2].numpy().astype(np.float64) / px.timestep
        Mq = np.array([[Mc + m, m * l * np.cos(a0)], [m * l * np.cos(a0), m * l ** 2 + Ic]])
        want = np.linalg.solve(Mq, np.array([0.0, -m * G * l * np.sin(a0)]))
        assert np.allclose(got, want, rtol=2e-4, atol=2e-4), (a0, got, want)
        # the cart moves against the pole: the centre of mass has no horizontal acceleration (no horizontal force acts)
        assert abs((Mc + m) * got[0] + m * l * np.cos(a0) * got[1]) < 1e-3


def test_panda_arm_coasting_without_forces_keeps_its_kinetic_energy(oracle_factory):
    """Seven coupled revolute joints in 3D, no gravity, no drives, no contacts: the only thing that changes the joint velocities are the
    velocity-product (Coriolis / centrifugal) terms, and they do no work.  The kinetic energy is summed here from the link states the
    engine reports (COM velocities, angular velocities, poses) and the template's masses and inertias.  A first-order integrator keeps
    it to O(dt): the drift over the same 0.6 s shrinks with the step (4 % at 10 ms, 1.2 % at 2.5 ms) -- a wrong velocity-product
    term would leave a drift that does not."""
    from maniskill_amd.envs import scene_builders as sb
    tpl = SceneTemplate()
    art = sb.add_panda(tpl, root_p=(0, 0, 0), stiffness=0.0, damping=0.0, force_limit=0.0, disable_gravity=True)
    links = [op[1] for op in tpl.ops if op[0] == "add_link"]
    mass = np.array([l[7] for l in links])
    inertia = np.array([[[l[9][0], l[9][3], l[9][4]], [l[9][3], l[9][1], l[9][5]], [l[9][4], l[9][5], l[9][2]]] for l in links])
    assert all(l[11] == 0.0 for l in links)                  # no armature: the links' own inertia is all there is

    def drift(sim_freq):
        cfg = SimConfig()
        cfg.sim_freq = sim_freq
        px = oracle_factory(tpl, 1, cfg)
        px.gpu_init()
        px.set_scene_offsets(np.zeros((1, 3)))
        px.cuda_articulation_qpos.torch()[0, :9] = torch.tensor([0.0, 0.4, 0.0, -1.9, 0.0, 2.3, 0.8, 0.03, 0.03])
        px.cuda_articulation_qvel.torch()[0, :9] = torch.tensor([0.9, -0.7, 0.8, 0.6, -0.9, 0.7, 0.8, 0.0, 0.0])
        px.gpu_apply_all()
        rbd = px.cuda_rigid_body_data.torch().view(px.bodies_per_env, 13)
        qd = px.cuda_articulation_qvel.torch()
        es, vs = [], []
        for _ in range(int(round(0.6 * sim_freq))):
            px.step()
            px.gpu_fetch_all()
            d = rbd.numpy().astype(np.float64)
            T = 0.0
            for i in range(len(links)):
                w, x, y, z = d[i, 3:7]
                R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                              [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                              [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
                om = R.T @ d[i, 10:13]
                T += 0.5 * mass[i] * d[i, 7:10] @ d[i, 7:10] + 0.5 * om @ inertia[i] @ om
            es.append(T)
            vs.append(qd[0, :7].numpy().copy())
        assert len(px.get_contacts(0, 16)[0]) == 0
        assert np.abs(vs[-1] - vs[0]).max() > 0.15           # the joint velocities do change: the coupling terms are at work
        return max(abs(e - es[0]) for e in es) / es[0]

    coarse, fine = drift(100), drift(400)
    assert coarse < 0.06 and fine < 0.02 and coarse > 2.5 * fine, (coarse, fine)
