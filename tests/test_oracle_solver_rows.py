"""Known answers for the solver rows added in round 3 (CPU oracle; the HIP side is compared with the oracle in test_gpu_parity.py):
force-limited drives as clamped soft rows (PhysxArticulationJoint.set_drive_properties: force_limit, mode -- mani_skill/utils/structs/
articulation_joint.py:187-195, agents/controllers/pd_joint_pos.py:38-52), joint friction (pd_joint_pos.py:44-53), torsional patch friction
(agents/robots/panda/panda.py:20-32, utils/building/actor_builder.py:153-154), static != dynamic friction (agents/robots/dclaw/dclaw.py:23),
and stacks that come to rest by the reference's own criterion (utils/structs/actor.py:220-227 is_static: |v| < 1e-2, |w| < 0.5)."""
import numpy as np
import pytest
import torch

from maniskill_amd import _native as N
from maniskill_amd.envs import scene_builders as sb
from maniskill_amd.physx import SceneTemplate, SimConfig

G = 9.81


def _start(factory, tpl, cfg=None, n=1):
    px = factory(tpl, n, cfg or SimConfig())
    px.gpu_init()
    px.set_scene_offsets(np.zeros((n, 3)))
    return px


# ---- force-limited drives ----------------------------------------------------------------------------------------------------------
def test_a_saturated_slide_does_not_blow_up_the_damper_of_a_light_wrist(oracle_factory):
    """The round-2 failure of the floating hands: a 0.4 kg hand on a root slide whose drive saturates (15 m target, 100 N
    limit) accelerates at 250 m/s^2; the wrist's 100 N s / rad damper on a 1e-4 kg m^2 link was then replaced by +-100 N m of explicit
    bang-bang (1e9 rad/s after six steps).  As a clamped soft row the damper stays a damper."""
    tpl = SceneTemplate()
    art = tpl.add_articulation("hand", root_p=(0, 0, 1.0))
    base = tpl.add_link(art, "base", -1, N.JOINT_FIXED, mass=1.0, inertia6=(1e-2, 1e-2, 1e-2, 0, 0, 0))
    slide = tpl.add_link(art, "slide", base, N.JOINT_PRISMATIC, joint_name="x", mass=5e-4, inertia6=(1e-7, 1e-7, 1e-7, 0, 0, 0),
                         limits=(-20.0, 20.0), disable_gravity=True)
    wrist = tpl.add_link(art, "palm", slide, N.JOINT_REVOLUTE, joint_name="wrist", mass=0.4, com=(0.0, 0.02, 0.0),
                         inertia6=(1e-4, 1e-4, 1e-4, 0, 0, 0), limits=(-20.0, 20.0), disable_gravity=True,
                         pose_in_parent=[0, 0, 0, 0.70710678, 0, 0, 0.70710678])      # wrist axis = y: the slide's acceleration loads it
    tpl.set_drive(slide, 1000.0, 100.0, 100.0, "force")
    tpl.set_drive(wrist, 1000.0, 100.0, 100.0, "force")
    px = _start(oracle_factory, tpl)
    px.cuda_rigid_body_data.torch().view(1, px.bodies_per_env, 13)[:, base, :7] = torch.tensor([0.0, 0.0, 1.0, 1, 0, 0, 0])
    tq = px.cuda_articulation_target_qpos.torch()
    px.gpu_apply_all()
    gen = torch.Generator().manual_seed(0)
    qd = px.cuda_articulation_qvel.torch()
    worst = 0.0
    for t in range(100):
        if t % 5 == 0:
            tq[0, :2] = (2 * torch.rand(2, generator=gen) - 1) * 15.0
            px.gpu_apply_articulation_target_position()
        px.step()
        px.gpu_fetch_articulation_qvel()
        assert torch.isfinite(qd).all()
        worst = max(worst, qd[0, 1].abs().item())
    assert worst < 100.0 + 1e-3, worst        # and never faster than PhysX's maxJointVelocity
    # the slide itself is pushed with exactly its limit while saturated: a = f_max / m_total
    px2 = _start(oracle_factory, tpl)
    px2.cuda_rigid_body_data.torch().view(1, px2.bodies_per_env, 13)[:, base, :7] = torch.tensor([0.0, 0.0, 1.0, 1, 0, 0, 0])
    px2.cuda_articulation_target_qpos.torch()[0, 0] = 15.0
    px2.gpu_apply_all()
    px2.step(); px2.step()
    px2.gpu_fetch_articulation_qvel()
    assert abs(px2.cuda_articulation_qvel.torch()[0, 0].item() - 2 * px2.timestep * 100.0 / 0.4005) < 0.02 * 2 * px2.timestep * 100.0 / 0.4005


def _hinge(factory, inertia, drive, mode="force", friction=0.0, gravity=False, length=0.0, mass=1.0, q0=0.0):
    tpl = SceneTemplate()
    art = tpl.add_articulation("hinge", root_p=(0, 0, 1.0))
    base = tpl.add_link(art, "base", -1, N.JOINT_FIXED, mass=1.0, inertia6=(1e-2, 1e-2, 1e-2, 0, 0, 0))
    bob = tpl.add_link(art, "bob", base, N.JOINT_REVOLUTE, joint_name="hinge", mass=mass, com=(0, 0, -length),
                       inertia6=(inertia, inertia, inertia, 0, 0, 0), disable_gravity=not gravity, friction=friction)
    if drive is not None:
        tpl.set_drive(bob, drive[0], drive[1], drive[2], mode)
    px = _start(factory, tpl)
    px.cuda_rigid_body_data.torch().view(1, px.bodies_per_env, 13)[:, base, :7] = torch.tensor([0.0, 0.0, 1.0, 1, 0, 0, 0])
    px.cuda_articulation_qpos.torch()[0, 0] = q0
    px.gpu_apply_all()
    return px


def test_an_acceleration_drive_does_not_see_the_inertia(oracle_factory):
    """mode="acceleration": the gains are per unit inertia, so a hinge with ten times the inertia follows the same critically damped
    step response q(t) = 1 - (1 + w t) exp(-w t), w = sqrt(K) = 10 / s."""
    traces = []
    for inertia in (0.01, 0.1):
        px = _hinge(oracle_factory, inertia, (100.0, 20.0, 3.0e38), mode="acceleration")
        px.cuda_articulation_target_qpos.torch()[0, 0] = 1.0
        px.gpu_apply_articulation_target_position()
        q = px.cuda_articulation_qpos.torch()
        tr = []
        for _ in range(40):
            px.step(); px.gpu_fetch_articulation_qpos()
            tr.append(q[0, 0].item())
        traces.append(np.array(tr))
    assert np.abs(traces[0] - traces[1]).max() < 1e-4
    t = px.timestep * np.arange(1, 41)
    assert np.abs(traces[0] - (1 - (1 + 10 * t) * np.exp(-10 * t))).max() < 0.05      # implicit Euler at dt = 0.01 lags the closed form a little


# ---- joint friction ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("coef", [0.5, 0.1])
def test_joint_friction_holds_a_pendulum_while_the_gravity_torque_is_below_coef_times_the_hinge_force(oracle_factory, coef):
    """A point-like bob of 1 kg, 0.3 m from a horizontal hinge, released at 30 degrees: the hinge transmits m g (and no moment about the
    joint frame's other axes worth mentioning), so the friction row can hold coef x m g against the gravity torque m g l sin(q):
    it holds for l sin(q) = 0.15 < 0.5 and lets go for 0.15 > 0.1."""
    px = _hinge(oracle_factory, 1e-4, None, friction=coef, gravity=True, length=0.3, q0=np.pi / 6)
    q = px.cuda_articulation_qpos.torch()
    for _ in range(150):
        px.step()
    px.gpu_fetch_articulation_qpos()
    if coef > 0.15:
        assert abs(q[0, 0].item() - np.pi / 6) < 5e-3
    else:
        assert abs(q[0, 0].item() - np.pi / 6) > 0.2       # it swings ...
        for _ in range(1500):
            px.step()
        px.gpu_fetch_articulation_qpos(); px.gpu_fetch_articulation_qvel()
        # ... and comes to rest inside the stick zone l |sin q| <= coef (a frictionless hinge would swing for ever: test_oracle_mechanics)
        assert 0.3 * abs(np.sin(q[0, 0].item())) <= coef * 1.05 and abs(px.cuda_articulation_qvel.torch()[0, 0].item()) < 1e-3


# ---- torsional friction -------------------------------------------------------------------------------------------------------------
def _spinning_ball(factory, patch, min_patch, steps):
    r, w0 = 0.03, 20.0
    tpl = SceneTemplate()
    sb.add_table_scene(tpl)
    m = 1000.0 * 4.0 / 3.0 * np.pi * r ** 3
    inertia = 0.4 * m * r * r
    ball = tpl.add_actor("ball", N.BODY_DYNAMIC, p=(0, 0, r), mass=m, inertia6=(inertia,) * 3 + (0, 0, 0), angular_damping=0.0)
    tpl.add_shape(ball, N.SHAPE_SPHERE, params=(r, 0, 0), static_friction=1.0, dynamic_friction=1.0, patch_radius=patch, min_patch_radius=min_patch)
    px = _start(factory, tpl)
    rbd = px.cuda_rigid_body_data.torch().view(px.bodies_per_env, 13)
    rbd[tpl.body_id("table-workspace"), :7] = torch.tensor([-0.12, 0.0, -sb.TABLE_HEIGHT, np.cos(np.pi / 4), 0, 0, np.sin(np.pi / 4)])
    rbd[ball, :7] = torch.tensor([0.0, 0.0, r, 1, 0, 0, 0])
    rbd[ball, 7:13] = torch.tensor([0.0, 0, 0, 0, 0, w0])
    px.gpu_apply_all()
    for _ in range(steps):
        px.step()
    px.gpu_fetch_all()
    return rbd[ball].clone(), m, inertia, w0, px.timestep


def test_a_ball_spinning_on_the_spot_keeps_spinning_without_a_patch_and_brakes_with_one(oracle_factory):
    """A one-point contact carries no moment about its normal; with min_patch_radius = rho the torsional row brakes with mu rho m g:
    w(t) = w0 - (mu rho m g / I) t (table friction 0.3 averaged with the ball's 1.0: mu = 0.65)."""
    row, m, inertia, w0, dt = _spinning_ball(oracle_factory, 0.0, 0.0, 60)
    assert abs(row[12].item() - w0) < 1e-3 * w0
    rho, steps = 0.001, 40
    row, m, inertia, w0, dt = _spinning_ball(oracle_factory, 0.0, rho, steps)
    want = w0 - 0.65 * rho * m * G / inertia * steps * dt
    assert abs(row[12].item() - want) < 0.03 * (w0 - want), (row[12].item(), want)
    assert row[7:9].abs().max() < 1e-3 and abs(row[2].item() - 0.03) < 2e-4          # it does not wander off
    row, *_ = _spinning_ball(oracle_factory, 0.0, rho, 400)
    assert abs(row[12].item()) < 1e-3                                                 # and the brake holds it once it has stopped


# ---- static and dynamic friction ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("moving", [False, True])
def test_static_friction_holds_what_dynamic_friction_lets_slide(oracle_factory, moving):
    """mu_s = 0.7, mu_d = 0.3 on both surfaces, slope tan(theta) = 0.5: a cube at rest stays (0.5 < 0.7); the same cube given a push
    keeps sliding and accelerates with g (sin(theta) - mu_d cos(theta)) (0.5 > 0.3)."""
    th = np.arctan(0.5)
    tpl = SceneTemplate()
    sb.add_table_scene(tpl, material=(0.7, 0.3, 0.0))
    cube = sb.add_cube(tpl, "cube", 0.02, (0, 0, 0.02), material=(0.7, 0.3, 0.0))
    cfg = SimConfig()
    cfg.scene_config.gravity = (G * np.sin(th), 0.0, -G * np.cos(th))
    px = _start(oracle_factory, tpl, cfg)
    rbd = px.cuda_rigid_body_data.torch().view(px.bodies_per_env, 13)
    rbd[tpl.body_id("table-workspace"), :7] = torch.tensor([-0.12, 0.0, -sb.TABLE_HEIGHT, np.cos(np.pi / 4), 0, 0, np.sin(np.pi / 4)])
    rbd[cube, :7] = torch.tensor([-0.2, 0.0, 0.02, 1, 0, 0, 0])
    rbd[cube, 7:13] = 0.0
    if moving:
        rbd[cube, 7] = 0.2
    px.gpu_apply_all()
    for _ in range(5):
        px.step()
    px.gpu_fetch_all()
    v0, steps = rbd[cube, 7].item(), 25
    for _ in range(steps):
        px.step()
    px.gpu_fetch_all()
    if not moving:
        assert abs(rbd[cube, 7].item()) < 2e-3 and abs(rbd[cube, 0].item() + 0.2) < 2e-3
    else:
        a = G * (np.sin(th) - 0.3 * np.cos(th))
        assert abs((rbd[cube, 7].item() - v0) - a * steps * px.timestep) < 0.04 * a * steps * px.timestep


# ---- stacks --------------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n, top_density, offset", [(3, 1000.0, 0.0), (4, 1000.0, 0.0), (5, 1000.0, 0.0), (3, 10000.0, 0.0), (3, 1000.0, 0.005)])
def test_stacks_come_to_rest_by_the_references_is_static(oracle_factory, n, top_density, offset):
    """StackCube-v1 / StackPyramid-v1 succeed when the stacked cube `is_static(lin_thresh=1e-2, ang_thresh=0.5)` (envs/tasks/tabletop/
    stack_cube.py evaluate, utils/structs/actor.py:220-227).  Stacks of three to five 4 cm cubes, one with a ten times denser top cube,
    one leaning 5 mm per layer: every cube far inside both thresholds over the last half second, the stack where it was put."""
    h = 0.02
    tpl = SceneTemplate()
    sb.add_table_scene(tpl)
    cubes = [sb.add_cube(tpl, f"cube{k}", h, (0, 0, h + 2 * h * k), density=(top_density if k == n - 1 else 1000.0)) for k in range(n)]
    px = _start(oracle_factory, tpl)
    rbd = px.cuda_rigid_body_data.torch().view(px.bodies_per_env, 13)
    rbd[tpl.body_id("table-workspace"), :7] = torch.tensor([-0.12, 0.0, -sb.TABLE_HEIGHT, np.cos(np.pi / 4), 0, 0, np.sin(np.pi / 4)])
    for k, c in enumerate(cubes):
        rbd[c, :7] = torch.tensor([offset * k, 0.0, h + 2 * h * k, 1, 0, 0, 0])
        rbd[c, 7:13] = 0.0
    px.gpu_apply_all()
    lin = ang = 0.0
    for t in range(400):
        px.step()
        if t >= 350:
            px.gpu_fetch_all()
            lin = max(lin, rbd[cubes, 7:10].norm(dim=1).max().item())
            ang = max(ang, rbd[cubes, 10:13].norm(dim=1).max().item())
    assert lin < 0.8 * 1e-2 and ang < 0.5 / 10, (lin, ang)      # the five-cube stack sways most: 6 mm / s at its top
    for k, c in enumerate(cubes):
        assert abs(rbd[c, 2].item() - (h + 2 * h * k)) < 2.5e-3 and abs(rbd[c, 0].item() - offset * k) < 4e-3 and abs(rbd[c, 1].item()) < 4e-3, (k, rbd[c, :3])


# ---- the same scenes on the HIP backend -------------------------------------------------------------------------------------------
def _full_state(px):
    px.gpu_fetch_all()
    parts = [px.cuda_rigid_body_data.torch().detach().cpu().reshape(-1)]
    if px.arts_per_env > 0:
        parts += [px.cuda_articulation_qpos.torch().detach().cpu().reshape(-1), px.cuda_articulation_qvel.torch().detach().cpu().reshape(-1)]
    return torch.cat(parts)


def _scene_hand(factory):
    tpl = SceneTemplate()
    art = tpl.add_articulation("hand", root_p=(0, 0, 1.0))
    base = tpl.add_link(art, "base", -1, N.JOINT_FIXED, mass=1.0, inertia6=(1e-2, 1e-2, 1e-2, 0, 0, 0))
    slide = tpl.add_link(art, "slide", base, N.JOINT_PRISMATIC, joint_name="x", mass=5e-4, inertia6=(1e-7, 1e-7, 1e-7, 0, 0, 0), limits=(-20.0, 20.0), disable_gravity=True)
    wrist = tpl.add_link(art, "palm", slide, N.JOINT_REVOLUTE, joint_name="wrist", mass=0.4, com=(0.0, 0.02, 0.0), inertia6=(1e-4, 1e-4, 1e-4, 0, 0, 0),
                         limits=(-20.0, 20.0), disable_gravity=True, pose_in_parent=[0, 0, 0, 0.70710678, 0, 0, 0.70710678])
    tpl.set_drive(slide, 1000.0, 100.0, 100.0, "force")
    tpl.set_drive(wrist, 1000.0, 100.0, 100.0, "acceleration")
    px = _start(factory, tpl, n=4)
    dev = px.cuda_rigid_body_data.torch().device
    px.cuda_rigid_body_data.torch().view(4, px.bodies_per_env, 13)[:, base, :7] = torch.tensor([0.0, 0.0, 1.0, 1, 0, 0, 0], device=dev)
    px.gpu_apply_all()
    gen = torch.Generator().manual_seed(0)

    def drive(t):
        if t % 5 == 0:
            px.cuda_articulation_target_qpos.torch()[:, :2] = ((2 * torch.rand(4, 2, generator=gen) - 1) * 15.0).to(dev)
            px.gpu_apply_articulation_target_position()
    return px, 60, drive


def _scene_pendulum_with_joint_friction(factory):
    px = _hinge(factory, 1e-4, None, friction=0.1, gravity=True, length=0.3, q0=np.pi / 6)
    return px, 120, None


def _scene_ball(factory):
    r = 0.03
    tpl = SceneTemplate()
    sb.add_table_scene(tpl)
    m = 1000.0 * 4.0 / 3.0 * np.pi * r ** 3
    ball = tpl.add_actor("ball", N.BODY_DYNAMIC, p=(0, 0, r), mass=m, inertia6=(0.4 * m * r * r,) * 3 + (0, 0, 0), angular_damping=0.0)
    tpl.add_shape(ball, N.SHAPE_SPHERE, params=(r, 0, 0), static_friction=1.0, dynamic_friction=1.0, patch_radius=0.05, min_patch_radius=0.001)
    cube = sb.add_cube(tpl, "cube", 0.02, (0.1, 0, 0.02), material=(0.7, 0.3, 0.0))
    px = _start(factory, tpl, n=2)
    rbd = px.cuda_rigid_body_data.torch().view(2, px.bodies_per_env, 13)
    dev = rbd.device
    rbd[:, tpl.body_id("table-workspace"), :7] = torch.tensor([-0.12, 0.0, -sb.TABLE_HEIGHT, np.cos(np.pi / 4), 0, 0, np.sin(np.pi / 4)], device=dev)
    rbd[:, ball, :7] = torch.tensor([0.0, 0.0, r, 1, 0, 0, 0], device=dev)
    rbd[:, ball, 7:13] = torch.tensor([0.05, 0, 0, 0, 0, 20.0], device=dev)
    rbd[:, cube, :7] = torch.tensor([0.1, 0.0, 0.02, 1, 0, 0, 0], device=dev)
    rbd[:, cube, 7:13] = torch.tensor([0.3, 0.1, 0, 0, 0, 0.0], device=dev)      # slides on mu_d = 0.3, then sticks on mu_s = 0.7
    px.gpu_apply_all()
    return px, 100, None


def _scene_stack(factory):
    h = 0.02
    tpl = SceneTemplate()
    sb.add_table_scene(tpl)
    cubes = [sb.add_cube(tpl, f"cube{k}", h, (0, 0, h + 2 * h * k), density=(10000.0 if k == 4 else 1000.0)) for k in range(5)]
    px = _start(factory, tpl, n=2)
    rbd = px.cuda_rigid_body_data.torch().view(2, px.bodies_per_env, 13)
    dev = rbd.device
    rbd[:, tpl.body_id("table-workspace"), :7] = torch.tensor([-0.12, 0.0, -sb.TABLE_HEIGHT, np.cos(np.pi / 4), 0, 0, np.sin(np.pi / 4)], device=dev)
    for k, c in enumerate(cubes):
        rbd[:, c, :7] = torch.tensor([0.003 * k, 0.0, h + 2 * h * k + 0.002, 1, 0, 0, 0], device=dev)
        rbd[:, c, 7:13] = 0.0
    px.gpu_apply_all()
    return px, 150, None


@pytest.mark.gpu
@pytest.mark.parametrize("scene", [_scene_hand, _scene_pendulum_with_joint_friction, _scene_ball, _scene_stack])
def test_solver_rows_hip_equals_oracle(built, oracle_factory, scene):
    """drive rows (force and acceleration mode), joint friction, torsional and static / dynamic friction, a five-cube stack: the HIP kernels
    against the oracle, state by state (1e-4 relative, the north-star bar), and no solver scheduling error flags."""
    from maniskill_amd.physx import PhysxGpuSystem
    hip = lambda tpl, n, cfg: PhysxGpuSystem("cuda:0", tpl, n, cfg)   # noqa: E731
    a, steps, da = scene(hip)
    b, _, db = scene(oracle_factory)
    for t in range(steps):
        if da is not None:
            da(t); db(t)
        a.step(); b.step()
        if t % 10 == 9 or t == steps - 1:
            sa, sb_ = _full_state(a), _full_state(b)
            assert torch.isfinite(sa).all()
            assert torch.allclose(sa, sb_, rtol=1e-4, atol=2e-5), (scene.__name__, t, float((sa - sb_).abs().max()))
    assert a.get_overflow() & 6 == 0
