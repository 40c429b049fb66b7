"""Known-answer tests of the CPU oracle (tier T2 of SURVEY.md §8c): physical facts derived from the
reference's task definitions, since no PhysX trace exists to compare with (parity unpinned)."""
import json
import os

import numpy as np
import pytest
import torch

from maniskill_amd import _native as N
from maniskill_amd.agents.urdf import load_model
from maniskill_amd.envs import scene_builders as sb
from maniskill_amd.physx import SceneTemplate

HERE = os.path.dirname(os.path.abspath(__file__))


def _cube_scene(oracle_factory, n=2):
    tpl = SceneTemplate()
    table = sb.add_table_scene(tpl)
    cube = sb.add_cube(tpl, "cube", 0.02, (0, 0, 0.02))
    px = oracle_factory(tpl, n, None)
    px.gpu_init()
    return tpl, px, table, cube


def test_cube_rests_on_table(oracle_factory):
    """pick_cube.py:88-94,118: a cube placed at z = half_size stays there (Actor.is_static thresholds)."""
    tpl, px, table, cube = _cube_scene(oracle_factory)
    rbd = px.cuda_rigid_body_data.torch().view(2, -1, 13)
    for _ in range(200):
        px.step()
    px.gpu_fetch_all()
    assert abs(rbd[0, cube, 2].item() - 0.02) < 1e-3
    assert rbd[0, cube, 7:10].norm().item() < 1e-2 and rbd[0, cube, 10:13].norm().item() < 0.1
    ids, vals = px.get_contacts(0)
    assert len(ids) == 4 and np.allclose(vals[:, 5], -1.0, atol=1e-5)  # table is shape A: normal from cube to table
    m = 1000 * 0.04 ** 3
    assert abs(vals[:, 7].sum() - m * 9.81 * px.timestep) < 2e-5  # sum of normal impulses = m g dt


def test_dropped_tilted_cube_settles_flat(oracle_factory):
    tpl, px, table, cube = _cube_scene(oracle_factory)
    rbd = px.cuda_rigid_body_data.torch().view(2, -1, 13)
    rbd[1, cube, 2] = 0.15
    rbd[1, cube, 3:7] = torch.tensor([0.9659258, 0.2588190, 0.0, 0.0])  # 30 deg about x
    px.gpu_apply_all()
    for _ in range(300):
        px.step()
    px.gpu_fetch_all()
    assert abs(rbd[1, cube, 2].item() - 0.02) < 1.5e-3
    assert rbd[1, cube, 7:13].abs().max().item() < 5e-2
    assert torch.isfinite(rbd).all()


def test_cube_off_the_table_lands_on_ground_plane(oracle_factory):
    tpl, px, table, cube = _cube_scene(oracle_factory)
    rbd = px.cuda_rigid_body_data.torch().view(2, -1, 13)
    rbd[0, cube, :3] = torch.tensor([3.0, 0.0, 0.5])
    px.gpu_apply_all()
    for _ in range(400):
        px.step()
    px.gpu_fetch_all()
    assert abs(rbd[0, cube, 2].item() - (sb.GROUND_ALTITUDE + 0.02)) < 2e-3


def _fk_numpy(model, q, root_p):
    """Independent float64 forward kinematics straight from the cooked URDF data."""
    def qmat(qt):
        w, x, y, z = qt
        return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                         [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                         [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])
    T = []
    dof = 0
    for L in model["links"]:
        J = L["joint"]
        if L["parent"] < 0:
            M = np.eye(4)
            M[:3, 3] = root_p
        else:
            O = np.eye(4)
            O[:3, :3] = qmat(J["q"])
            O[:3, 3] = J["p"]
            Jm = np.eye(4)
            ax = np.asarray(J["axis"], dtype=float)
            if J["type"] == "revolute":
                a = q[dof]; dof += 1
                K = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
                Jm[:3, :3] = np.eye(3) + np.sin(a) * K + (1 - np.cos(a)) * K @ K
            elif J["type"] == "prismatic":
                Jm[:3, 3] = ax * q[dof]; dof += 1
            M = T[L["parent"]] @ O @ Jm
        T.append(M)
    return T


def test_forward_kinematics_matches_independent_fk(oracle_factory):
    tpl, ids = sb.build_pick_cube_template()
    px = oracle_factory(tpl, 4, None)
    px.gpu_init()
    model = load_model("panda_v2.json")
    rng = np.random.RandomState(0)
    qpos = px.cuda_articulation_qpos.torch()
    qs = sb.PANDA_REST_QPOS[None] + rng.uniform(-0.5, 0.5, (4, 9))
    qs[:, 7:] = rng.uniform(0, 0.04, (4, 2))
    qpos[:] = torch.tensor(qs, dtype=torch.float32)
    px.gpu_apply_articulation_qpos()
    px.gpu_update_articulation_kinematics()
    px.gpu_fetch_all()
    rbd = px.cuda_rigid_body_data.torch().view(4, -1, 13).numpy()
    for e in range(4):
        T = _fk_numpy(model, qs[e], np.array([-0.615, 0, 0]))
        for i, M in enumerate(T):
            assert np.allclose(rbd[e, i, :3], M[:3, 3], atol=2e-6), (e, i)
            w, x, y, z = rbd[e, i, 3:7]
            R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                          [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                          [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])
            assert np.allclose(R, M[:3, :3], atol=5e-6), (e, i)


def test_arm_holds_rest_pose_and_tracks_targets(oracle_factory):
    """pd_joint_pos drives (K=1e3, D=1e2) with gravity disabled on the links (base_agent.py:278-282)."""
    tpl, ids = sb.build_pick_cube_template()
    px = oracle_factory(tpl, 1, None)
    px.gpu_init()
    qpos, tq = px.cuda_articulation_qpos.torch(), px.cuda_articulation_target_qpos.torch()
    qpos[0] = torch.tensor(sb.PANDA_REST_QPOS, dtype=torch.float32)
    tq[0] = qpos[0]
    px.gpu_apply_all()
    for _ in range(100):
        px.step()
    px.gpu_fetch_all()
    assert np.allclose(qpos[0].numpy(), sb.PANDA_REST_QPOS, atol=1e-5)
    target = sb.PANDA_REST_QPOS.copy()
    target[:7] += np.array([0.2, -0.15, 0.1, 0.2, -0.2, -0.3, 0.3])
    tq[0] = torch.tensor(target, dtype=torch.float32)
    px.gpu_apply_articulation_target_position()
    for _ in range(300):
        px.step()
    px.gpu_fetch_all()
    assert np.allclose(qpos[0].numpy(), target, atol=2e-3)
    assert px.cuda_articulation_qvel.torch()[0].abs().max().item() < 1e-2


def test_joint_limits_hold(oracle_factory):
    tpl, ids = sb.build_pick_cube_template()
    px = oracle_factory(tpl, 1, None)
    px.gpu_init()
    qpos, tq = px.cuda_articulation_qpos.torch(), px.cuda_articulation_target_qpos.torch()
    qpos[0] = torch.tensor(sb.PANDA_REST_QPOS, dtype=torch.float32)
    tq[0] = qpos[0]
    tq[0, 3] = 1.0       # panda_joint4 upper limit is -0.0698
    tq[0, 7:] = 0.2      # fingers: upper limit 0.04
    px.gpu_apply_all()
    for _ in range(400):
        px.step()
    px.gpu_fetch_all()
    assert qpos[0, 3].item() < -0.0698 + 5e-3
    assert qpos[0, 7].item() < 0.04 + 1e-3 and qpos[0, 8].item() < 0.04 + 1e-3


def test_static_pair_filter(oracle_factory):
    """Collision filtering: adjacent links, SRDF-disabled pairs and static-static pairs are removed."""
    tpl, ids = sb.build_pick_cube_template()
    px = oracle_factory(tpl, 1, None)
    px.gpu_init()
    assert (px.bodies_per_env, px.arts_per_env, px.max_dof, px.nv, px.nshapes) == (18, 1, 9, 15, 20)
    assert px.npairs == 94


def test_scripted_pick_and_lift_keeps_the_cube_grasped(oracle_factory):
    """T2: closing the gripper on the cube gives is_grasping (>= 0.5 N, <= 85 deg: panda.py:237-265) and
    the lifted cube follows the TCP (cf. franka_pick_cube.py:26-36 in the reference's benchmark)."""
    from maniskill_amd.envs.pick_cube import PickCubeEnv

    wp = json.load(open(os.path.join(HERE, "golden", "grasp_waypoints.json")))
    env = PickCubeEnv(num_envs=1, px_factory=oracle_factory, robot_init_qpos_noise=0.0)
    env.reset(seed=0)
    # put the cube at the origin, unrotated
    st = env.get_state()
    st[0, 13:16] = torch.tensor([0.0, 0.0, 0.02]); st[0, 16:20] = torch.tensor([1.0, 0, 0, 0])
    env.set_state(st)
    tq = env._target_qpos_buf

    def go(qa, qb, grip, steps):
        for k in range(steps):
            a = (k + 1) / steps
            tq[0, :7] = torch.tensor(np.asarray(qa) * (1 - a) + np.asarray(qb) * a, dtype=torch.float32)
            tq[0, 7:9] = grip
            env.px.gpu_apply_articulation_target_position()
            for _ in range(5):
                env.px.step()
        env.px.gpu_fetch_all()

    go(wp["q_rest"], wp["q_pre"], 0.04, 20)
    go(wp["q_pre"], wp["q_grasp"], 0.04, 20)
    assert not env.is_grasping()[0]
    go(wp["q_grasp"], wp["q_grasp"], -0.01, 20)
    assert env.is_grasping()[0]
    go(wp["q_grasp"], wp["q_lift"], -0.01, 40)
    go(wp["q_lift"], wp["q_lift"], -0.01, 60)
    assert env.is_grasping()[0]
    cube, tcp = env.cube_pose[0], env.tcp_pose[0]
    assert cube[2].item() > 0.2 and (cube[:3] - tcp[:3]).norm().item() < 0.01
    lf = env.get_pairwise_contact_forces(env._q_lgrasp)[0]
    assert 15.0 < lf.norm().item() < 60.0  # K*(q - target) = 1e3 * 0.03 = 30 N nominal


def test_convex_vs_box_matches_box_vs_box(oracle_factory):
    """GJK/EPA + manifold on a hull with a box's vertices must agree with the SAT box path."""
    res = []
    for as_hull in (False, True):
        tpl = SceneTemplate()
        base = tpl.add_actor("base", N.BODY_KINEMATIC, p=(0, 0, 0))
        tpl.add_shape(base, N.SHAPE_BOX, params=(0.5, 0.5, 0.05))
        m, I = sb.box_mass_properties((0.03, 0.02, 0.01))
        b = tpl.add_actor("b", N.BODY_DYNAMIC, p=(0.1, 0.05, 0.0595), q=(0.9807853, 0, 0, 0.1950903), mass=m, inertia6=I)
        if as_hull:
            v = np.array([[sx * 0.03, sy * 0.02, sz * 0.01] for sx in (-1, 1) for sy in (-1, 1) for sz in (-1, 1)], dtype=np.float32)
            tpl.add_shape(b, N.SHAPE_CONVEX, verts=v)
        else:
            tpl.add_shape(b, N.SHAPE_BOX, params=(0.03, 0.02, 0.01))
        px = oracle_factory(tpl, 1, None)
        px.gpu_init()
        px.step()
        ids, vals = px.get_contacts(0)
        res.append(vals)
    a, b = res
    assert a.shape == b.shape == (4, 8)
    assert np.allclose(a[:, 3:6], b[:, 3:6], atol=1e-4)                       # normals
    assert np.allclose(np.sort(a[:, 6]), np.sort(b[:, 6]), atol=2e-5)          # separations (-0.5 mm)
    assert np.allclose(a[:, 6], -5e-4, atol=5e-5)
    pa = a[np.lexsort((a[:, 1], a[:, 0]))][:, :3]
    pb = b[np.lexsort((b[:, 1], b[:, 0]))][:, :3]
    assert np.allclose(pa, pb, atol=1e-4)


def test_oracle_matches_its_golden_rollout(oracle_factory):
    """Drift detector: the committed fixture (tests/golden/make_golden.py) is reproduced."""
    from maniskill_amd.envs.pick_cube import PickCubeEnv

    g = np.load(os.path.join(HERE, "golden", "pickcube_oracle_rollout.npz"))
    n = g["actions"].shape[1]
    env = PickCubeEnv(num_envs=n, px_factory=oracle_factory)
    obs, _ = env.reset(seed=2022)
    assert np.allclose(obs.numpy(), g["obs"][0], atol=1e-6)
    for t in range(g["actions"].shape[0]):
        obs, r, *_ = env.step(torch.from_numpy(g["actions"][t]))
        assert np.allclose(obs.numpy(), g["obs"][t + 1], rtol=1e-4, atol=1e-5), t
    assert np.allclose(env.get_state().numpy(), g["state"], rtol=1e-4, atol=1e-5)


def test_body_impulse_query_is_the_weight_of_the_resting_cube(oracle_factory):
    """gpu_create_contact_body_impulse_query (structs/base.py:116-136): net contact impulse on the cube at rest = m g dt upwards,
    and the table receives the opposite; equals the pair query cube<-table when nothing else touches the cube."""
    from maniskill_amd.envs.pick_cube import PickCubeEnv

    env = PickCubeEnv(num_envs=2, px_factory=oracle_factory)
    env.reset(seed=0)
    qb = env.px.gpu_create_contact_body_impulse_query([env._b_cube, env._b_table])
    qp = env.px.gpu_create_contact_pair_impulse_query([(env._b_cube, env._b_table)])
    for _ in range(20):
        env.step(None)
    env.px.gpu_query_contact_body_impulses(qb)
    env.px.gpu_query_contact_pair_impulses(qp)
    b = qb.cuda_impulses.torch().view(2, 2, 3)
    p = qp.cuda_impulses.torch().view(2, 1, 3)
    mgdt = 0.064 * 9.81 * env.px.timestep
    assert torch.allclose(b[:, 0, 2], torch.full((2,), mgdt), rtol=2e-2) and b[:, 0, :2].abs().max() < 1e-4
    assert torch.allclose(b[:, 1], -b[:, 0], atol=1e-6) and torch.allclose(b[:, 0], p[:, 0], atol=1e-7)


def test_per_env_box_sizes_and_masses(oracle_factory):
    """msk_declare_env_box / msk_set_env_boxes / msk_set_env_masses: every env instantiates the template's cube with its own
    half size and mass; each settles at its own height and presses on the table with its own weight."""
    from maniskill_amd.physx import SimConfig

    tpl = SceneTemplate()
    table = sb.add_table_scene(tpl)
    cube = sb.add_cube(tpl, "cube", 0.02, (0, 0, 0.1))
    cube_shape = tpl.nshapes - 1
    tpl.declare_env_box(cube_shape)
    tpl.declare_env_mass(cube)
    n = 4
    px = oracle_factory(tpl, n, SimConfig())
    px.gpu_init()
    px.set_scene_offsets(np.zeros((n, 3)))
    half = np.array([[0.02] * 3, [0.03] * 3, [0.05, 0.02, 0.01], [0.01] * 3], dtype=np.float32)
    mass = (8 * half.prod(1) * 1000.0).astype(np.float32)
    inertia = np.stack([mass / 3 * (half[:, 1] ** 2 + half[:, 2] ** 2), mass / 3 * (half[:, 0] ** 2 + half[:, 2] ** 2),
                        mass / 3 * (half[:, 0] ** 2 + half[:, 1] ** 2)], axis=1)
    px.set_env_boxes(cube_shape, half)
    px.set_env_masses(cube, mass, inertia)
    rbd = px.cuda_rigid_body_data.torch().view(n, px.bodies_per_env, 13)
    rbd[:, cube, :3] = torch.tensor([0.0, 0.0, 0.08])
    rbd[:, cube, 3:7] = torch.tensor([1.0, 0, 0, 0])
    rbd[:, table, :7] = torch.tensor([-0.12, 0.0, -sb.TABLE_HEIGHT, np.cos(np.pi / 4), 0, 0, np.sin(np.pi / 4)])
    px.gpu_apply_all()
    q = px.gpu_create_contact_body_impulse_query([cube])
    for _ in range(150):
        px.step()
    px.gpu_fetch_all()
    px.gpu_query_contact_body_impulses(q)
    z = rbd[:, cube, 2]
    assert torch.allclose(z, torch.tensor(half[:, 2]), atol=1.5e-3)                      # rests on its own half height
    imp = q.cuda_impulses.torch().view(n, 3)[:, 2]
    assert torch.allclose(imp, torch.tensor(mass) * 9.81 * px.timestep, rtol=3e-2)       # and weighs its own weight
    assert (rbd[:, cube, 7:13].abs() < 2e-2).all()


def _free_cube_scene(factory, n, gravity=True):
    """A cube floating far above the table (no contacts), optionally with gravity switched off through the config."""
    from maniskill_amd.physx import SimConfig

    tpl = SceneTemplate()
    sb.add_table_scene(tpl)
    cube = sb.add_cube(tpl, "cube", 0.02, (0, 0, 1.0))
    cfg = SimConfig()
    if not gravity:
        cfg.scene_config.gravity = (0.0, 0.0, 0.0)
    px = factory(tpl, n, cfg)
    px.gpu_init()
    px.set_scene_offsets(np.zeros((n, 3)))
    rbd = px.cuda_rigid_body_data.torch().view(n, px.bodies_per_env, 13)
    rbd[:, cube, :3] = torch.tensor([0.0, 0.0, 1.0], device=rbd.device)
    rbd[:, cube, 3:7] = torch.tensor([1.0, 0, 0, 0], device=rbd.device)
    rbd[:, cube, 7:13] = 0.0
    px.gpu_apply_all()
    return px, cube, rbd


def test_external_force_and_torque_act_for_one_step(oracle_factory):
    """cuda_rigid_body_force + gpu_apply_rigid_dynamic_force (Actor.apply_force, structs/actor.py:316-322) and the torque twin:
    dv = F dt / m and dw = I^-1 tau dt in the step after the apply, nothing afterwards (forces are cleared by the step); a
    second apply before the step replaces the wrench; kinematic rows are ignored."""
    n = 3
    px, cube, rbd = _free_cube_scene(oracle_factory, n, gravity=False)
    m, dt = 0.064, px.timestep
    I = m / 3 * (0.02 ** 2 + 0.02 ** 2)
    F = px.cuda_rigid_body_force.torch().view(n, px.bodies_per_env, 4)
    T = px.cuda_rigid_body_torque.torch().view(n, px.bodies_per_env, 4)
    assert F.shape[-1] == 4 and T.shape == F.shape
    F[:, cube, :3] = torch.tensor([[9.0, 9.0, 9.0], [0.0, 0.0, 0.0], [0.0, 0.0, 0.0]], device=F.device)
    px.gpu_apply_rigid_dynamic_force()
    F[:, cube, :3] = torch.tensor([[0.64, 0.0, 0.0], [0.0, -1.28, 0.0], [0.0, 0.0, 0.0]], device=F.device)   # replaces, does not add
    F[:, 0, :3] = 100.0                                                                   # the kinematic table: ignored
    px.gpu_apply_rigid_dynamic_force()
    T[2, cube, :3] = torch.tensor([0.0, 0.0, 2e-4], device=T.device)
    px.gpu_apply_rigid_dynamic_torque()
    px.step(); px.gpu_fetch_all()
    v, w = rbd[:, cube, 7:10].clone(), rbd[:, cube, 10:13].clone()
    want_v = torch.tensor([[0.64 * dt / m, 0, 0], [0, -1.28 * dt / m, 0], [0, 0, 0]], dtype=torch.float32)
    assert torch.allclose(v.cpu(), want_v, rtol=1e-5, atol=1e-7)
    assert torch.allclose(w[2].cpu(), torch.tensor([0.0, 0.0, 2e-4 * dt / I * (1 - 0.05 * dt)]), rtol=1e-5, atol=1e-7) and w[:2].abs().max() == 0
    assert rbd[:, 0, 7:13].abs().max() == 0
    px.step(); px.gpu_fetch_all()                          # no force any more: v keeps, w only sees the default angular damping 0.05
    assert torch.allclose(rbd[:, cube, 7:10], v, rtol=1e-6, atol=1e-8)
    assert torch.allclose(rbd[:, cube, 10:13], w * (1 - 0.05 * dt), rtol=1e-6, atol=1e-8)


def test_a_force_of_m_g_upwards_cancels_gravity(oracle_factory):
    px, cube, rbd = _free_cube_scene(oracle_factory, 2)
    F = px.cuda_rigid_body_force.torch().view(2, px.bodies_per_env, 4)
    F[0, cube, 2] = 0.064 * 9.81
    for _ in range(10):
        px.gpu_apply_rigid_dynamic_force()      # every step, like a task that holds an object up
        px.step()
    px.gpu_fetch_all()
    assert abs(float(rbd[0, cube, 9])) < 1e-5 and abs(float(rbd[0, cube, 2]) - 1.0) < 1e-5
    assert abs(float(rbd[1, cube, 9]) + 9.81 * 10 * px.timestep) < 1e-4                  # its neighbour falls freely


def _panda_under_gravity(factory, n, gravity=True, dev=None):
    """A Panda that has to carry itself (ManiSkill switches link gravity off -- balance_passive_force -- so the tasks' robots
    have nothing to carry), held at its rest pose by its drives."""
    from maniskill_amd.physx import SimConfig

    tpl = SceneTemplate()
    sb.add_panda(tpl, disable_gravity=not gravity)
    sb.add_table_scene(tpl)
    px = factory(tpl, n, SimConfig())
    px.gpu_init()
    px.set_scene_offsets(np.zeros((n, 3)))
    rest = torch.tensor(sb.PANDA_REST_QPOS, dtype=torch.float32)
    for buf in (px.cuda_articulation_qpos, px.cuda_articulation_target_qpos):
        t = buf.torch().view(n, -1)
        t[:, :9] = rest.to(t.device)
    rbd = px.cuda_rigid_body_data.torch().view(n, px.bodies_per_env, 13)
    rbd[:, tpl.body_id("panda_link0"), :7] = torch.tensor([-0.615, 0.0, 0.0, 1, 0, 0, 0], device=rbd.device)
    rbd[:, tpl.body_id("table-workspace"), :7] = torch.tensor([-0.12, 0.0, -sb.TABLE_HEIGHT, np.cos(np.pi / 4), 0, 0, np.sin(np.pi / 4)], device=rbd.device)
    px.gpu_apply_all()
    return px, tpl


def test_link_incoming_joint_forces_carry_the_weight_above_them(oracle_factory):
    """cuda_articulation_link_incoming_joint_forces + gpu_fetch_articulation_link_incoming_joint_forces
    (structs/articulation.py:596-620): for an arm held still by its drives the wrench through a joint is the weight of
    everything distal to it; a link with no children and no contacts carries its own weight; without gravity nothing."""
    px, tpl = _panda_under_gravity(oracle_factory, 2)
    for _ in range(150):
        px.step()
    px.gpu_fetch_all()
    w = px.get_link_incoming_joint_forces()
    names = tpl.body_names
    order = [n for n in names if n.startswith("panda_")]
    assert w.shape == (2, len(order), 6)
    mass = dict(zip(names, tpl.body_masses))
    g = 9.81
    # the Panda is a chain with a two-finger fork at the hand: every link after X in build order hangs on X's joint
    above = lambda first: sum(mass[n] for n in order[order.index(first):])
    f = w[:, :, :3].norm(dim=-1)
    for first in ("panda_link1", "panda_link4", "panda_hand"):
        assert torch.allclose(f[:, order.index(first)], torch.full((2,), above(first) * g), rtol=2e-2), first
    lf = order.index("panda_leftfinger")
    assert torch.allclose(f[:, lf], torch.full((2,), mass["panda_leftfinger"] * g), rtol=3e-2)
    # joint 1 turns about the vertical: the force along its axis (child-frame x) is the whole weight above it and its drive
    # carries (almost) no torque
    j1 = order.index("panda_link1")
    assert torch.allclose(w[:, j1, 0].abs(), torch.full((2,), above("panda_link1") * g), rtol=2e-2) and w[:, j1, 3].abs().max() < 0.1
    # joint 2 is horizontal: gravity loads its drive -- the torque about the joint axis is the drive force K (q_t - q) - D qd
    j2 = order.index("panda_link2")
    q = px.cuda_articulation_qpos.torch().view(2, -1)
    qd = px.cuda_articulation_qvel.torch().view(2, -1)
    qt = px.cuda_articulation_target_qpos.torch().view(2, -1)
    drive = 1e3 * (qt[:, 1] - q[:, 1]) - 1e2 * qd[:, 1]
    assert drive.abs().min() > 1.0 and torch.allclose(w[:, j2, 3].abs(), drive.abs(), rtol=5e-2)
    # no gravity: nothing to carry
    px0, _ = _panda_under_gravity(oracle_factory, 1, gravity=False)
    for _ in range(30):
        px0.step()
    px0.gpu_fetch_all()
    assert px0.get_link_incoming_joint_forces().abs().max() < 0.05
