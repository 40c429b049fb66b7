"""Textbook mechanics as known answers for the CPU oracle (oracle/*.c).  The oracle cannot be pinned against PhysX here
(parity unpinned), so its physics is pinned against closed forms instead: the discrete free fall of semi-implicit
Euler, the Coulomb cone on an incline, the period of a physical pendulum, a saturated joint drive, the mimic tendon."""
import numpy as np
import pytest
import torch

from maniskill_amd import _native as N
from maniskill_amd.envs import scene_builders as sb
from maniskill_amd.physx import SceneTemplate, SimConfig


def _cube_world(factory, n, gravity=(0.0, 0.0, -9.81), z=0.02, with_table=True):
    tpl = SceneTemplate()
    if with_table:
        sb.add_table_scene(tpl)
    cube = sb.add_cube(tpl, "cube", 0.02, (0, 0, z))
    cfg = SimConfig()
    cfg.scene_config.gravity = tuple(gravity)
    px = factory(tpl, n, cfg)
    px.gpu_init()
    px.set_scene_offsets(np.zeros((n, 3)))
    rbd = px.cuda_rigid_body_data.torch().view(n, px.bodies_per_env, 13)
    if with_table:
        rbd[:, tpl.body_id("table-workspace"), :7] = torch.tensor([-0.12, 0.0, -sb.TABLE_HEIGHT, np.cos(np.pi / 4), 0, 0, np.sin(np.pi / 4)])
    rbd[:, cube, :3] = torch.tensor([0.0, 0.0, z])
    rbd[:, cube, 3:7] = torch.tensor([1.0, 0, 0, 0])
    rbd[:, cube, 7:13] = 0.0
    px.gpu_apply_all()
    return px, cube, rbd


def test_free_fall_is_the_semi_implicit_euler_sum(oracle_factory):
    """v_k = -g k dt,  z_k = z_0 - g dt^2 k (k + 1) / 2  (velocity first, then position: the integrator of §2)."""
    px, cube, rbd = _cube_world(oracle_factory, 1, z=5.0)
    dt, g = px.timestep, 9.81
    for k in (1, 10, 50):
        while getattr(px, "_k", 0) < k:
            px.step(); px._k = getattr(px, "_k", 0) + 1
        px.gpu_fetch_all()
        assert abs(rbd[0, cube, 9].item() + g * k * dt) < 1e-5 * k
        assert abs(rbd[0, cube, 2].item() - (5.0 - g * dt * dt * k * (k + 1) / 2)) < 2e-5
    assert rbd[0, cube, 7:9].abs().max() == 0 and rbd[0, cube, 10:13].abs().max() == 0


@pytest.mark.parametrize("tan_theta", [0.2, 0.5])
def test_coulomb_cone_on_an_incline(oracle_factory, tan_theta):
    """Gravity tilted by theta instead of the table: with mu = 0.3 the cube stays for tan(theta) = 0.2 and slides with
    a = g (sin(theta) - mu cos(theta)) for tan(theta) = 0.5."""
    th = np.arctan(tan_theta)
    g = 9.81
    px, cube, rbd = _cube_world(oracle_factory, 1, gravity=(g * np.sin(th), 0.0, -g * np.cos(th)))
    for _ in range(5):           # settle onto the contact
        px.step()
    px.gpu_fetch_all()
    x0, v0 = rbd[0, cube, 0].item(), rbd[0, cube, 7].item()
    steps = 35                   # 0.4 s in all: the sliding cube stays well inside the table top (x < 0.15 m)
    for _ in range(steps):
        px.step()
    px.gpu_fetch_all()
    v = rbd[0, cube, 7].item()
    if tan_theta < 0.3:
        assert abs(v) < 2e-3 and abs(rbd[0, cube, 0].item() - x0) < 2e-3
    else:
        a = g * (np.sin(th) - 0.3 * np.cos(th))
        assert abs((v - v0) - a * steps * px.timestep) < 0.03 * a * steps * px.timestep
    assert abs(rbd[0, cube, 2].item() - 0.02) < 1.5e-3 and abs(rbd[0, cube, 8].item()) < 1e-3      # stays on the table, no drift sideways


def _one_link(factory, length=0.5, mass=1.0, radius=0.05, gravity=True, drive=None, q0=0.0):
    """A fixed base and one link on a revolute joint about the world x axis; the link's mass sits `length` below the joint."""
    tpl = SceneTemplate()
    art = tpl.add_articulation("pendulum", root_p=(0, 0, 1.0))
    base = tpl.add_link(art, "base", -1, N.JOINT_FIXED, mass=1.0, inertia6=(1e-2, 1e-2, 1e-2, 0, 0, 0))
    Ic = 0.4 * mass * radius ** 2
    bob = tpl.add_link(art, "bob", base, N.JOINT_REVOLUTE, joint_name="hinge", mass=mass, com=(0, 0, -length),
                       inertia6=(Ic, Ic, Ic, 0, 0, 0), disable_gravity=not gravity)
    if drive is not None:
        tpl.set_drive(bob, *drive)
    cfg = SimConfig()
    px = factory(tpl, 1, cfg)
    px.gpu_init()
    px.set_scene_offsets(np.zeros((1, 3)))
    rbd = px.cuda_rigid_body_data.torch().view(1, px.bodies_per_env, 13)
    rbd[:, base, :7] = torch.tensor([0.0, 0.0, 1.0, 1, 0, 0, 0])
    px.cuda_articulation_qpos.torch()[0, 0] = q0
    px.gpu_apply_all()
    return px, Ic + mass * length ** 2


def test_physical_pendulum_period_and_amplitude(oracle_factory):
    """T = 2 pi sqrt(I / (m g l)) for small swings; a frictionless hinge keeps its amplitude."""
    length, mass, a0 = 0.5, 1.0, 0.1
    px, I = _one_link(oracle_factory, length, mass, q0=a0)
    q = px.cuda_articulation_qpos.torch()
    trace = []
    for _ in range(450):
        px.step(); px.gpu_fetch_articulation_qpos()
        trace.append(q[0, 0].item())
    trace = np.array(trace)
    down = np.where((trace[:-1] > 0) & (trace[1:] <= 0))[0]                      # downward zero crossings, one per period
    frac = trace[down] / (trace[down] - trace[down + 1])
    period = np.diff(down + frac).mean() * px.timestep
    want = 2 * np.pi * np.sqrt(I / (mass * 9.81 * length)) * (1 + a0 ** 2 / 16)   # first amplitude correction
    assert abs(period - want) < 0.01 * want
    peaks = [trace[a:b].max() for a, b in zip(down[:-1], down[1:])]
    assert all(abs(p - a0) < 0.03 * a0 for p in peaks)


def test_saturated_drive_is_a_constant_torque(oracle_factory):
    """A drive whose PD force exceeds its limit pushes with exactly the limit: qdd = f_max / I while it is saturated."""
    fmax = 2.0
    px, I = _one_link(oracle_factory, gravity=False, drive=(1e4, 0.0, fmax, "force"))
    px.cuda_articulation_target_qpos.torch()[0, 0] = 3.0
    px.gpu_apply_articulation_target_position()
    qd = px.cuda_articulation_qvel.torch()
    steps = 25
    for _ in range(steps):
        px.step()
    px.gpu_fetch_articulation_qvel()
    assert abs(qd[0, 0].item() - fmax / I * steps * px.timestep) < 0.01 * fmax / I * steps * px.timestep
    # far below the limit the same drive is an implicit spring: it settles on the target without overshoot growth
    px2, _ = _one_link(oracle_factory, gravity=False, drive=(1e3, 1e2, 100.0, "force"))
    px2.cuda_articulation_target_qpos.torch()[0, 0] = 0.2
    px2.gpu_apply_articulation_target_position()
    for _ in range(300):
        px2.step()
    px2.gpu_fetch_all()
    assert abs(px2.cuda_articulation_qpos.torch()[0, 0].item() - 0.2) < 1e-3 and abs(px2.cuda_articulation_qvel.torch()[0, 0].item()) < 1e-3


def test_mimic_tendon_keeps_the_fingers_together(oracle_factory):
    """URDF mimic -> fixed tendon (articulation_builder.py:161-200): fingers started apart are pulled to the same opening, and a
    target given to both is reached by both."""
    from maniskill_amd.envs.pick_cube import PickCubeEnv

    env = PickCubeEnv(num_envs=1, px_factory=oracle_factory)
    env.reset(seed=0)
    env._qpos[0, 7], env._qpos[0, 8] = 0.04, 0.0
    env.px.gpu_apply_all(); env.px.gpu_fetch_all()
    a = torch.zeros(1, 8); a[0, 7] = 1.0          # open: both finger targets 0.04
    for _ in range(15):
        env.step(a)
    q = env.qpos[0]
    assert abs(q[7].item() - q[8].item()) < 1e-3 and abs(q[7].item() - 0.04) < 3e-3


def test_torque_free_spin_decays_with_the_angular_damping_only(oracle_factory):
    """omega_k = omega_0 (1 - c dt)^k with the actor's angular damping c = 0.05 (SAPIEN's default); a cube's inertia is isotropic,
    so the axis does not move; the centre of mass falls as if it did not spin."""
    px, cube, rbd = _cube_world(oracle_factory, 1, z=4.0, with_table=False)
    w0 = torch.tensor([3.0, -2.0, 5.0])
    rbd[0, cube, 10:13] = w0
    px.gpu_apply_all()
    k = 40
    for _ in range(k):
        px.step()
    px.gpu_fetch_all()
    assert torch.allclose(rbd[0, cube, 10:13], w0 * (1 - 0.05 * px.timestep) ** k, rtol=1e-5)
    assert abs(rbd[0, cube, 2].item() - (4.0 - 9.81 * px.timestep ** 2 * k * (k + 1) / 2)) < 2e-5
    assert abs(torch.linalg.norm(rbd[0, cube, 3:7]).item() - 1.0) < 1e-6
    # the orientation is the rotation by the accumulated angle about the constant axis
    angle = sum(float(torch.linalg.norm(w0)) * (1 - 0.05 * px.timestep) ** (j + 1) * px.timestep for j in range(k))
    axis = w0 / torch.linalg.norm(w0)
    want = torch.cat([torch.tensor([np.cos(angle / 2)], dtype=torch.float32), axis * np.sin(angle / 2)])
    q = rbd[0, cube, 3:7]
    assert min(torch.linalg.norm(q - want).item(), torch.linalg.norm(q + want).item()) < 1e-4


def test_a_position_drive_sags_under_gravity_by_the_torque_over_its_stiffness(oracle_factory):
    """Static equilibrium of the implicit PD drive against gravity: K q = -m g r cos(q) (target 0, arm horizontal at q = 0)."""
    from scipy.optimize import brentq
    m, r, K, D = 2.0, 0.3, 200.0, 20.0
    tpl = SceneTemplate()
    art = tpl.add_articulation("lever", root_p=(0, 0, 1.0))
    base = tpl.add_link(art, "base", -1, N.JOINT_FIXED, mass=1.0, inertia6=(1e-2, 1e-2, 1e-2, 0, 0, 0))
    arm = tpl.add_link(art, "arm", base, N.JOINT_REVOLUTE, joint_name="hinge", mass=m, com=(0, r, 0), inertia6=(1e-3, 1e-3, 1e-3, 0, 0, 0))
    tpl.set_drive(arm, K, D, 1e6, "force")
    px = oracle_factory(tpl, 1, SimConfig())
    px.gpu_init()
    px.set_scene_offsets(np.zeros((1, 3)))
    px.cuda_rigid_body_data.torch().view(1, px.bodies_per_env, 13)[:, base, :7] = torch.tensor([0.0, 0.0, 1.0, 1, 0, 0, 0])
    px.gpu_apply_all()
    for _ in range(600):
        px.step()
    px.gpu_fetch_all()
    want = brentq(lambda x: K * x + m * 9.81 * r * np.cos(x), -1.5, 0.0)
    assert abs(px.cuda_articulation_qpos.torch()[0, 0].item() - want) < 1e-5 and abs(px.cuda_articulation_qvel.torch()[0, 0].item()) < 1e-5
