"""Golden vectors computed by the REFERENCE'S OWN CODE (run here, where /root/reference exists; the .npz travels).

The physics of the reference lives in a binary wheel that cannot run here, but its task logic, controller action scaling, pose
algebra and rotation conversions are plain torch code (tests/golden/_reference_stubs.py makes them importable).  This script
  1. lets THIS package (CPU oracle backend) produce simulator states: random rollouts and the scripted pick-and-lift,
  2. hands those states to the reference's unbound ``evaluate`` / ``_get_obs_extra`` / ``compute_normalized_dense_reward`` /
     ``Panda.is_grasping`` / ``Panda.is_static`` / ``common.flatten_state_dict`` / controller ``_clip_and_scale_action`` / ``Pose`` /
     ``rotation_conversions`` functions,
  3. stores inputs and the reference's outputs in tests/golden/reference_vectors.npz.
tests/test_reference_vectors.py replays the inputs through this package's host code and compares.

Usage: python tests/golden/make_reference_vectors.py
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, HERE)

from _reference_stubs import Fake, install, ns  # noqa: E402

install()

from mani_skill.agents.controllers.pd_ee_pose import PDEEPoseController  # noqa: E402
from mani_skill.agents.robots.panda.panda import Panda as RefPanda  # noqa: E402
from mani_skill.envs.tasks import tabletop as ref_tasks  # noqa: E402
from mani_skill.envs.tasks.tabletop.push_t import PushTEnv as RefPushT  # noqa: E402
from mani_skill.utils import common as ref_common, gym_utils as ref_gym_utils  # noqa: E402
from mani_skill.utils.geometry import rotation_conversions as rc  # noqa: E402
from mani_skill.utils.structs.actor import Actor as RefActor  # noqa: E402
from mani_skill.utils.structs.pose import Pose as RefPose  # noqa: E402

from oracle_backend import OraclePhysxSystem  # noqa: E402
from maniskill_amd.envs import registered as _registry  # noqa: E402

FAC = lambda tpl, n, cfg: OraclePhysxSystem(tpl, n, cfg)   # noqa: E731
OUT = {}


def put(prefix, **arrays):
    for k, v in arrays.items():
        OUT[f"{prefix}/{k}"] = v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)


# ---------------------------------------------------------------------------------------------- pure functions
def pure_functions():
    g = torch.Generator().manual_seed(0)
    e = (torch.rand(64, 3, generator=g) * 2 - 1) * torch.tensor([3.0, 1.4, 3.0])
    q_from_e = rc.matrix_to_quaternion(rc.euler_angles_to_matrix(e, "XYZ"))
    q = torch.randn(64, 4, generator=g); q = q / q.norm(dim=1, keepdim=True)
    q2 = torch.randn(64, 4, generator=g); q2 = q2 / q2.norm(dim=1, keepdim=True)
    v = torch.randn(64, 3, generator=g)
    put("rot", euler=e, quat_from_euler=q_from_e, quat=q, euler_from_quat=rc.matrix_to_euler_angles(rc.quaternion_to_matrix(q), "XYZ"),
        quat2=q2, quat_mul=rc.quaternion_multiply(q, q2), vec=v, quat_apply=rc.quaternion_apply(q, v))
    p1, p2 = torch.randn(64, 3, generator=g), torch.randn(64, 3, generator=g)
    a, b = RefPose.create_from_pq(p1, q), RefPose.create_from_pq(p2, q2)
    put("pose", a=a.raw_pose, b=b.raw_pose, a_mul_b=(a * b).raw_pose, a_inv=a.inv().raw_pose, a_matrix=a.to_transformation_matrix())
    act = torch.rand(64, 8, generator=g) * 4 - 2
    put("clip_scale", action=act, arm=ref_gym_utils.clip_and_scale_action(act[:, :7], -0.1, 0.1),
        gripper=ref_gym_utils.clip_and_scale_action(act[:, 7:], -0.01, 0.04))
    # PDEEPoseController._clip_and_scale_action with Panda's pd_ee_delta_pose bounds (panda.py:111-123)
    ctl = Fake(PDEEPoseController, config=ns(rot_lower=-0.1, rot_upper=0.1),
               action_space_low=torch.tensor([-0.1] * 3 + [-0.1] * 3), action_space_high=torch.tensor([0.1] * 3 + [0.1] * 3))
    act6 = torch.rand(64, 6, generator=g) * 3 - 1.5
    put("ee_clip_scale", action=act6, out=PDEEPoseController._clip_and_scale_action(ctl, act6.clone()))
    # flatten_state_dict: order of the nested keys, bools as floats
    d = dict(agent=dict(qpos=torch.rand(5, 9, generator=g), qvel=torch.rand(5, 9, generator=g)),
             extra=dict(is_grasped=torch.tensor([True, False, True, False, False]), tcp_pose=torch.rand(5, 7, generator=g), goal_pos=torch.rand(5, 3, generator=g)))
    put("flatten", qpos=d["agent"]["qpos"], qvel=d["agent"]["qvel"], is_grasped=d["extra"]["is_grasped"], tcp_pose=d["extra"]["tcp_pose"],
        goal_pos=d["extra"]["goal_pos"], out=ref_common.flatten_state_dict(d, use_torch=True))


# ---------------------------------------------------------------------------------------------- task states from this package
def rollout_states(name, n=6, steps=24, every=6, seed=3, **kw):
    env = _registry()[name](num_envs=n, px_factory=FAC, **kw)
    env.reset(seed=seed)
    g = torch.Generator().manual_seed(seed)
    snaps = []
    for t in range(steps):
        env.step(0.8 * (2 * torch.rand(n, env.action_dim, generator=g) - 1))
        if (t + 1) % every == 0:
            snaps.append(env.get_state().clone())
    return env, snaps


def fake_agent(env, lforce, rforce):
    """The reference's Panda with this package's link poses / joint state behind it."""
    f1, f2 = ns(pose=RefPose.create(env._pose(env._b_f1))), ns(pose=RefPose.create(env._pose(env._b_f2)))
    scene = ns(get_pairwise_contact_forces=lambda link, obj: lforce if link is f1 else rforce)
    robot = ns(get_qvel=lambda: env.qvel, get_qpos=lambda: env.qpos,
               get_qlimits=lambda: env.robot.qlimits, get_links=lambda: [ns(pose=RefPose.create(env._pose(env._b_root)))])
    return Fake(RefPanda, scene=scene, finger1_link=f1, finger2_link=f2, tcp=ns(pose=RefPose.create(env.tcp_pose)), robot=robot)


def actor(env, body):
    rows = env._rbd[:, body]
    return Fake(RefActor, pose=RefPose.create(env._pose(body)), linear_velocity=rows[:, 7:10].clone(), angular_velocity=rows[:, 10:13].clone())


def reference_outputs(ref_cls, fake, env):
    info = ref_cls.evaluate(fake)
    extra = ref_cls._get_obs_extra(fake, info)
    obs = ref_common.flatten_state_dict(dict(agent=dict(qpos=env.qpos, qvel=env.qvel), extra=extra), use_torch=True)
    if "sparse" in getattr(ref_cls, "SUPPORTED_REWARD_MODES", ("normalized_dense",)) and "normalized_dense" not in getattr(ref_cls, "SUPPORTED_REWARD_MODES", ("normalized_dense",)):
        rew = info["success"].float()
    else:
        rew = ref_cls.compute_normalized_dense_reward(fake, obs=obs, action=None, info=info)
    return info, obs, rew


def synthetic_forces(n, g):
    """Finger contact forces that exercise both sides of is_grasping's thresholds: |f| around 0.5 N, angles around the limits."""
    mag = torch.rand(n, 1, generator=g) * 3.0 * (torch.rand(n, 1, generator=g) > 0.3)
    d = torch.randn(n, 3, generator=g)
    return mag * d / d.norm(dim=1, keepdim=True)


def task_vectors(name, ref_cls, build_fake, solved=None, zero_forces=False, **kw):
    env, snaps = rollout_states(name, **kw)
    if solved is not None:   # states of the success branch: the last snapshot with the task solved by hand in the even envs
        for variant in solved(env, snaps[-1].clone()):
            keep = torch.arange(env.num_envs) % 2 == 1
            variant[keep] = snaps[-1][keep]
            snaps.append(variant)
    g = torch.Generator().manual_seed(17)
    rec = dict(state=[], lforce=[], rforce=[], obs=[], reward=[], success=[], grasp=[])
    for st in snaps:
        env.set_state(st)
        lf, rf = synthetic_forces(env.num_envs, g), synthetic_forces(env.num_envs, g)
        # pull the synthetic forces towards the fingers' opening directions in half of the envs so that grasps do occur
        ydir = RefPose.create(env._pose(env._b_f1)).to_transformation_matrix()[:, :3, 1]
        half = torch.arange(env.num_envs) % 2 == 0
        lf[half] = ydir[half] * (0.3 + 2 * torch.rand(int(half.sum()), 1, generator=g))
        rf[half] = -RefPose.create(env._pose(env._b_f2)).to_transformation_matrix()[half, :3, 1] * (0.3 + 2 * torch.rand(int(half.sum()), 1, generator=g))
        if zero_forces:
            lf, rf = torch.zeros_like(lf), torch.zeros_like(rf)
        agent = fake_agent(env, lf, rf)
        fake = build_fake(env, agent)
        info, obs, rew = reference_outputs(ref_cls, fake, env)
        rec["state"].append(st); rec["lforce"].append(lf); rec["rforce"].append(rf)
        rec["obs"].append(obs); rec["reward"].append(rew); rec["success"].append(info["success"])
        rec["grasp"].append(RefPanda.is_grasping(agent, None, max_angle=kw.get("grasp_angle", 85)) if False else RefPanda.is_grasping(agent, None))
    put(name, num_envs=np.array(env.num_envs), **{k: torch.stack(v) for k, v in rec.items()})
    print(name, "snapshots", len(snaps), "obs", tuple(rec["obs"][0].shape), "successes", int(torch.stack(rec["success"]).sum()),
          "grasps", int(torch.stack(rec["grasp"]).sum()))


def camera_vectors():
    """RenderCamera.get_extrinsic_matrix / get_model_matrix (structs/render_camera.py:77-145) for random camera poses."""
    from mani_skill.utils.structs.render_camera import RenderCamera as RefCam

    g = torch.Generator().manual_seed(5)
    q = torch.randn(16, 4, generator=g); q = q / q.norm(dim=1, keepdim=True)
    pose = RefPose.create_from_pq(torch.randn(16, 3, generator=g), q)
    cam = Fake(RefCam, scene=ns(gpu_sim_enabled=True, device=torch.device("cpu")), mount=None, global_pose=pose,
               _cached_extrinsic_matrix=None, _cached_model_matrix=None)
    put("camera", pose=pose.raw_pose, extrinsic=RefCam.get_extrinsic_matrix(cam), model=RefCam.get_model_matrix(cam))


def vector_env_vectors():
    """ManiSkillVectorEnv.step / reset of the reference (vector/wrappers/gymnasium.py:104-184) on a scripted env: same-step auto
    reset with final_observation / final_info, ignore_terminations, episode metrics."""
    from mani_skill.vector.wrappers.gymnasium import ManiSkillVectorEnv as RefVec
    from scripted_env import ScriptedEnv

    for tag, (auto, ignore) in dict(auto=(True, False), ignore=(True, True), manual=(False, False)).items():
        env = ScriptedEnv()
        w = Fake(RefVec, _env=env, num_envs=env.num_envs, auto_reset=auto, ignore_terminations=ignore, record_metrics=True,
                 success_once=torch.zeros(4, dtype=torch.bool), fail_once=torch.zeros(4, dtype=torch.bool), returns=torch.zeros(4))
        RefVec.reset(w, seed=0)
        rec = {k: [] for k in ("obs", "rew", "term", "trunc", "success_once", "fail_once", "ret", "ep_len", "has_final", "final_obs", "final_mask",
                               "final_success_once", "final_ret")}
        g = torch.Generator().manual_seed(1)
        for t in range(12):
            a = torch.rand(4, 2, generator=g)
            obs, rew, term, trunc, infos = RefVec.step(w, a)
            ep = infos["episode"] if "episode" in infos else infos["final_info"]["episode"]
            rec["obs"].append(obs); rec["rew"].append(rew); rec["term"].append(term.clone()); rec["trunc"].append(trunc.clone())
            has = "final_info" in infos
            rec["has_final"].append(torch.tensor(has))
            fi = infos["final_info"]["episode"] if has else ep
            rec["final_obs"].append(infos["final_observation"] if has else torch.zeros_like(obs))
            rec["final_mask"].append(infos["_final_info"] if has else torch.zeros(4, dtype=torch.bool))
            rec["final_success_once"].append(fi["success_once"]); rec["final_ret"].append(fi["return"])
            rec["success_once"].append(w.success_once.clone()); rec["fail_once"].append(w.fail_once.clone()); rec["ret"].append(w.returns.clone())
            rec["ep_len"].append(env.elapsed_steps.clone())
        put(f"vector/{tag}", **{k: torch.stack(v) for k, v in rec.items()})


def ee_controller_vectors():
    """PDEEPos / PDEEPoseController.set_action (pd_ee_pose.py:104-129) + Kinematics.compute_ik's GPU branch (kinematics.py:185-245)
    for the five end-effector control modes of the Panda: the reference's code gets this package's Jacobian, joint state and link
    poses (its own come from pytorch_kinematics / sapien) and produces the arm's joint targets."""
    from mani_skill.agents.controllers.pd_ee_pose import PDEEPosController, PDEEPosControllerConfig, PDEEPoseControllerConfig
    from mani_skill.agents.controllers.utils.kinematics import Kinematics

    cfgs = {"pd_ee_delta_pos": (PDEEPosController, PDEEPosControllerConfig, dict(use_delta=True, use_target=False, normalize_action=True)),
            "pd_ee_delta_pose": (PDEEPoseController, PDEEPoseControllerConfig, dict(use_delta=True, use_target=False, normalize_action=True)),
            "pd_ee_target_delta_pos": (PDEEPosController, PDEEPosControllerConfig, dict(use_delta=True, use_target=True, normalize_action=True)),
            "pd_ee_target_delta_pose": (PDEEPoseController, PDEEPoseControllerConfig, dict(use_delta=True, use_target=True, normalize_action=True)),
            "pd_ee_pose": (PDEEPoseController, PDEEPoseControllerConfig, dict(use_delta=False, use_target=False, normalize_action=False))}
    g = torch.Generator().manual_seed(9)
    for mode, (ctl_cls, cfg_cls, flags) in cfgs.items():
        env = _registry()["PickCube-v1"](num_envs=5, px_factory=FAC, control_mode=mode)
        env.reset(seed=4)
        pos_only = ctl_cls is PDEEPosController
        adim = 3 if pos_only else 6
        b = 0.1 if mode != "pd_ee_pose" else 2.0
        extra = {} if pos_only else dict(rot_lower=-0.1, rot_upper=0.1)
        cfg = cfg_cls(joint_names=[f"panda_joint{k}" for k in range(1, 8)], pos_lower=-b, pos_upper=b, stiffness=1e3, damping=1e2, force_limit=100,
                      ee_link="panda_hand_tcp", urdf_path=None, **extra, **flags)      # the real config record (panda.py:101-137)
        rec = dict(action=[], state=[], target=[])
        target_pose = None
        for step in range(4):
            if mode == "pd_ee_pose":      # absolute targets near the current pose
                cur = env.ee_pose_at_base()
                arm = torch.cat([cur[:, :3] + 0.05 * (2 * torch.rand(5, 3, generator=g) - 1),
                                 env._quat_to_euler_xyz(cur[:, 3:7]) + 0.1 * (2 * torch.rand(5, 3, generator=g) - 1)], dim=1)
            else:
                arm = 1.6 * (2 * torch.rand(5, adim, generator=g) - 1)
            act = torch.cat([arm, 2 * torch.rand(5, 1, generator=g) - 1], dim=1)
            J = env.ee_jacobian()
            kin = Fake(Kinematics, use_gpu_ik=True, active_ancestor_joint_idxs=list(range(7)), qmask=torch.ones(7, dtype=torch.bool),
                       pk_chain=ns(jacobian=lambda q, J=J: J), device=torch.device("cpu"))
            sent = {}
            ctl = Fake(ctl_cls, config=cfg, scene=ns(gpu_sim_enabled=True, num_envs=5), kinematics=kin, device=torch.device("cpu"),
                       articulation=ns(get_qpos=lambda: env.qpos), active_joint_indices=torch.arange(7),
                       ee_link=ns(pose=RefPose.create(env._rbd[:, env._b_tcp, :7].clone())), root_link=ns(pose=RefPose.create(env._rbd[:, env._b_root, :7].clone())),
                       _target_pose=None if target_pose is None else RefPose.create(target_pose), _normalize_action=flags["normalize_action"],
                       action_space=ns(shape=(5, adim)), action_space_low=torch.tensor([-b] * 3 + [-0.1] * 3)[:adim], action_space_high=torch.tensor([b] * 3 + [0.1] * 3)[:adim],
                       set_drive_targets=lambda t: sent.__setitem__("t", t.clone()), _sim_steps=5)
            if flags["use_target"] and target_pose is None:
                ctl_cls.reset.__wrapped__(ctl) if hasattr(ctl_cls.reset, "__wrapped__") else object.__setattr__(ctl, "_target_pose", ctl.ee_pose_at_base)
            ctl_cls.set_action(ctl, arm.clone())
            if flags["use_target"]:
                target_pose = ctl._target_pose.raw_pose.clone()
            rec["action"].append(act); rec["state"].append(env.get_state().clone()); rec["target"].append(sent["t"])
            env.step(act)
        put(f"ee/{mode}", **{k: torch.stack(v) for k, v in rec.items()})
        print("ee", mode, "targets", tuple(rec["target"][0].shape))


def joint_controller_vectors():
    """PDJointPos / PDJointPosMimic / PDJointVel / PDJointPosVelController.set_action (agents/controllers/*.py) for the six joint-space
    control modes of the Panda, three consecutive control steps each (use_target modes carry their targets)."""
    from mani_skill.agents.controllers.pd_joint_pos import PDJointPosController, PDJointPosMimicController
    from mani_skill.agents.controllers.pd_joint_pos_vel import PDJointPosVelController
    from mani_skill.agents.controllers.pd_joint_vel import PDJointVelController

    dev = torch.device("cpu")
    arm = dict(pd_joint_delta_pos=(PDJointPosController, dict(use_delta=True, use_target=False), True, (-0.1, 0.1)),
               pd_joint_target_delta_pos=(PDJointPosController, dict(use_delta=True, use_target=True), True, (-0.1, 0.1)),
               pd_joint_pos=(PDJointPosController, dict(use_delta=False, use_target=False), False, None),
               pd_joint_vel=(PDJointVelController, dict(), True, (-1.0, 1.0)),
               pd_joint_pos_vel=(PDJointPosVelController, dict(use_delta=False, use_target=False), False, None),
               pd_joint_delta_pos_vel=(PDJointPosVelController, dict(use_delta=True, use_target=False), True, (-0.1, 0.1)))
    g = torch.Generator().manual_seed(12)
    for mode, (cls, flags, normalize, bound) in arm.items():
        env = _registry()["PickCube-v1"](num_envs=4, px_factory=FAC, control_mode=mode)
        env.reset(seed=6)
        n, adim = 4, env.action_dim - 1
        tq_arm, tq_grip = env.qpos[:, :7].clone(), env.qpos[:, 7:9].clone()
        rec = dict(action=[], qpos_target=[], qvel_target=[])
        for step in range(3):
            if mode in ("pd_joint_pos", "pd_joint_pos_vel"):
                a_arm = torch.cat([env.qpos[:, :7] + 0.1 * (2 * torch.rand(n, 7, generator=g) - 1)] +
                                  ([0.5 * (2 * torch.rand(n, 7, generator=g) - 1)] if adim == 14 else []), dim=1)
            else:
                a_arm = 1.5 * (2 * torch.rand(n, adim, generator=g) - 1)
            a_grip = 1.5 * (2 * torch.rand(n, 1, generator=g) - 1)
            sent = dict(vel=torch.zeros(n, 7))
            art = ns(get_qpos=lambda: env.qpos, set_joint_drive_targets=lambda t, joints, idx: sent.__setitem__("pos" if len(idx) == 7 else "grip", t.clone()),
                     set_joint_drive_velocity_targets=lambda t, joints, idx: sent.__setitem__("vel", t.clone()))
            lo = None if bound is None else torch.tensor(([bound[0]] * 7 + ([-1.0] * 7 if adim == 14 else []))[:adim])
            hi = None if bound is None else torch.tensor(([bound[1]] * 7 + ([1.0] * 7 if adim == 14 else []))[:adim])
            ctl = Fake(cls, config=ns(interpolate=False, **flags), scene=ns(num_envs=n), articulation=art, joints=None, active_joint_indices=torch.arange(7),
                       _target_qpos=tq_arm, _target_qvel=None, _normalize_action=normalize, action_space=ns(shape=(n, adim)), action_space_low=lo, action_space_high=hi, device=dev, _sim_steps=5)
            cls.set_action(ctl, a_arm.clone())
            if cls is not PDJointVelController:
                tq_arm = ctl._target_qpos.clone()
            grip = Fake(PDJointPosMimicController, config=ns(interpolate=False, use_delta=False, use_target=False), scene=ns(num_envs=n), articulation=art, joints=None,
                        active_joint_indices=torch.tensor([7, 8]), control_joint_indices=torch.tensor([0]), mimic_joint_indices=torch.tensor([1]),
                        mimic_control_joint_indices=torch.tensor([0]), _multiplier=torch.ones(1), _offset=torch.zeros(1), _target_qpos=tq_grip.clone(),
                        _normalize_action=True, action_space=ns(shape=(n, 1)), action_space_low=torch.tensor([-0.01]), action_space_high=torch.tensor([0.04]), device=dev, _sim_steps=5)
            PDJointPosMimicController.set_action(grip, a_grip.clone())
            tq_grip = grip._target_qpos.clone()
            act = torch.cat([a_arm, a_grip], dim=1)
            rec["action"].append(act)
            rec["qpos_target"].append(torch.cat([sent.get("pos", torch.full((n, 7), float("nan"))), sent["grip"]], dim=1))
            rec["qvel_target"].append(sent["vel"])
            env.step(act)
        put(f"joint/{mode}", **{k: torch.stack(v) for k, v in rec.items()})
        print("joint", mode, "action dim", adim + 1)


def shader_vectors():
    """render/shaders.py:66-84: the minimal pack's texture transforms on this package's rasterised textures."""
    from mani_skill.render.shaders import PREBUILT_SHADER_CONFIGS

    env = _registry()["PickCube-v1"](num_envs=2, px_factory=FAC, obs_mode="rgb+depth+segmentation")
    env.reset(seed=0)
    env.camera.take_picture()
    pos = env.camera.get_picture_cuda().torch().clone()
    col = env.camera.get_picture_cuda("Color").torch().clone()
    tr = PREBUILT_SHADER_CONFIGS["minimal"].texture_transforms
    out = dict(tr["PositionSegmentation"](pos)); out.update(tr["Color"](col))
    put("shader", position_segmentation=pos, color=col, **out)


def constants_vectors():
    """Numbers the reference fixes in Python: registered episode lengths, task constants, and what TableSceneBuilder.initialize (with
    the noise switched off) hands to the robot and the table."""
    from unittest.mock import MagicMock

    from mani_skill.envs.tasks.tabletop.pick_cube_cfgs import PICK_CUBE_CONFIGS
    from mani_skill.utils.registration import REGISTERED_ENVS
    from mani_skill.utils.scene_builder.table import TableSceneBuilder

    names = ["PickCube-v1", "PushCube-v1", "PullCube-v1", "StackCube-v1", "LiftPegUpright-v1", "PokeCube-v1", "PegInsertionSide-v1", "StackPyramid-v1", "PushT-v1",
             "PullCubeTool-v1"]
    put("const", max_episode_steps=np.array([REGISTERED_ENVS[n].max_episode_steps for n in names]))
    OUT["const/names"] = np.array(names)
    c = PICK_CUBE_CONFIGS["panda"]
    put("const/pickcube", values=np.array([c["cube_half_size"], c["goal_thresh"], c["cube_spawn_half_size"], c["max_goal_height"], *c["cube_spawn_center"],
                                           *c["sensor_cam_eye_pos"], *c["sensor_cam_target_pos"]], dtype=np.float64))
    T = ref_tasks
    put("const/tasks", push_goal_radius=np.array(T.PushCubeEnv.goal_radius), pull_goal_radius=np.array(T.PullCubeEnv.goal_radius),
        poke=np.array([T.PokeCubeEnv.cube_half_size, T.PokeCubeEnv.peg_half_width, T.PokeCubeEnv.peg_half_length, T.PokeCubeEnv.goal_radius]),
        liftpeg=np.array([T.LiftPegUprightEnv.peg_half_width, T.LiftPegUprightEnv.peg_half_length]),
        pusht=np.array([RefPushT.intersection_thresh, RefPushT.goal_z_rot, *RefPushT.goal_offset.tolist(), RefPushT.tee_spawnbox_xlength,
                        RefPushT.tee_spawnbox_ylength, RefPushT.tee_spawnbox_xoffset, RefPushT.tee_spawnbox_yoffset], dtype=np.float64))
    zero_rng = ns(normal=lambda mean, std, shape: np.zeros(shape))
    for uid in ("panda", "panda_wristcam"):
        agent = MagicMock()
        env = ns(robot_uids=uid, _enhanced_determinism=False, _episode_rng=zero_rng, agent=agent)
        b = Fake(TableSceneBuilder, env=env, table=MagicMock(), robot_init_qpos_noise=0.02)
        TableSceneBuilder.initialize(b, torch.arange(3))
        pose = agent.robot.set_pose.call_args[0][0]
        table_pose = b.table.set_pose.call_args[0][0]
        put(f"const/{uid}", rest_qpos=np.asarray(agent.reset.call_args[0][0])[0], root_p=np.asarray(pose.p), table_p=np.asarray(table_pose.p))


def episode_init_vectors():
    """The reference's _initialize_episode of every task, 4000 sub-scenes at once on a fake env whose actors only record the poses
    they are given: support (min / max) and mean of every placed actor's position and yaw -- the layout statistics this package's own
    samplers (different RNG) have to reproduce."""
    class Rec:
        def __init__(self):
            self.raw = None

        def set_pose(self, pose):
            self.raw = RefPose.create(pose).raw_pose.clone()

    B = 4000
    T = ref_tasks
    base = dict(device=torch.device("cpu"), num_envs=B, table_scene=ns(initialize=lambda idx: None), scene_builder=ns(initialize=lambda idx: None))
    jobs = {
        "PickCube-v1": (T.PickCubeEnv, dict(cube="cube", goal_site="goal_site"), dict(cube_half_size=0.02, cube_spawn_half_size=0.1, cube_spawn_center=(0, 0), max_goal_height=0.3)),
        "PushCube-v1": (T.PushCubeEnv, dict(obj="cube", goal_region="goal_region"), {}),
        "PullCube-v1": (T.PullCubeEnv, dict(obj="cube", goal_region="goal_region"), {}),
        "StackCube-v1": (T.StackCubeEnv, dict(cubeA="cubeA", cubeB="cubeB"), dict(cube_half_size=torch.tensor([0.02] * 3))),
        "LiftPegUpright-v1": (T.LiftPegUprightEnv, dict(peg="peg"), {}),
        "PokeCube-v1": (T.PokeCubeEnv, dict(peg="peg", cube="cube", goal_region="goal_region"), {}),
        "PullCubeTool-v1": (T.PullCubeToolEnv, dict(l_shape_tool="l_shape_tool", cube="cube"), {}),
        "StackPyramid-v1": (T.StackPyramidEnv, dict(cubeA="cubeA", cubeB="cubeB", cubeC="cubeC"), dict(cube_half_size=torch.tensor([0.02] * 3))),
        "PushT-v1": (RefPushT, dict(tee="Tee", goal_tee="goal_Tee", ee_goal_pos="goal_ee"), {}),
        # per-env peg sizes drawn from the reference's ranges (peg_insertion_side.py:97-98): lengths U(0.085, 0.125), radii U(0.015, 0.025)
        "PegInsertionSide-v1": (T.PegInsertionSideEnv, dict(peg="peg", box="box_with_hole"),
                                dict(peg_half_sizes=torch.stack([torch.rand(B, generator=torch.Generator().manual_seed(1)) * 0.04 + 0.085,
                                                                 torch.rand(B, generator=torch.Generator().manual_seed(2)) * 0.01 + 0.015,
                                                                 torch.rand(B, generator=torch.Generator().manual_seed(2)) * 0.01 + 0.015], dim=1),
                                     agent=__import__("unittest.mock").mock.MagicMock(), _enhanced_determinism=False, robot_init_qpos_noise=0.02,
                                     _episode_rng=ns(normal=lambda m, sd, shape: np.zeros(shape)))),
    }
    for name, (cls, actors, extra) in jobs.items():
        recs = {attr: Rec() for attr in actors}
        fake = Fake(cls, **base, **recs, **extra)
        torch.manual_seed(0)
        cls._initialize_episode(fake, torch.arange(B), {})
        for attr, mine in actors.items():
            raw = recs[attr].raw
            if raw.shape[1] < 7:   # the orientation came from transforms3d (a stand-in here): position statistics only
                raw = torch.cat([raw[:, :3], torch.tensor([[1.0, 0, 0, 0]]).expand(len(raw), 4)], dim=1)
            raw = raw.expand(B, 7) if raw.shape[0] == 1 else raw
            yaw = 2 * torch.atan2(raw[:, 6], raw[:, 3])
            yaw = torch.remainder(yaw + np.pi, 2 * np.pi) - np.pi
            put(f"init/{name}/{mine}", pmin=raw[:, :3].min(0)[0], pmax=raw[:, :3].max(0)[0], pmean=raw[:, :3].mean(0),
                yaw_min=yaw.min(), yaw_max=yaw.max(), qxy_absmax=raw[:, 4:6].abs().max())
        print("init", name, {a: tuple(np.round(recs[a].raw[:, :3].mean(0).numpy(), 3)) for a in actors})


def reward_mode_vectors():
    """BaseEnv.get_reward / compute_sparse_reward (sapien_env.py:648-697) for the four reward modes, with and without a fail flag."""
    from mani_skill.envs.sapien_env import BaseEnv as RefBase

    g = torch.Generator().manual_seed(3)
    dense = torch.rand(8, generator=g) * 5
    success = torch.tensor([True, False, False, True, False, False, True, False])
    fail = torch.tensor([False, True, False, False, False, True, False, False])
    put("reward_mode", dense=dense, success=success, fail=fail)
    for mode in ("sparse", "dense", "normalized_dense", "none"):
        for tag, info in (("s", dict(success=success)), ("sf", dict(success=success, fail=fail))):
            fake = Fake(RefBase, _reward_mode=mode, num_envs=8, device=torch.device("cpu"), compute_dense_reward=lambda obs, action, info: dense.clone(),
                        compute_normalized_dense_reward=lambda obs, action, info: dense / 5.0)
            put(f"reward_mode/{mode}_{tag}", out=RefBase.get_reward(fake, obs=None, action=None, info=info).to(torch.float32))


def sensor_config_vectors():
    """Every task's _default_sensor_configs (base_camera): eye / target handed to sapien_utils.look_at, resolution, fov, near, far."""
    import importlib
    su = importlib.import_module("mani_skill.utils.sapien_utils")

    names = ["PickCube-v1", "PushCube-v1", "PullCube-v1", "StackCube-v1", "LiftPegUpright-v1", "PokeCube-v1", "PegInsertionSide-v1", "StackPyramid-v1", "PushT-v1",
             "PullCubeTool-v1"]
    from mani_skill.utils.registration import REGISTERED_ENVS
    real = su.look_at
    rows = []
    for n in names:
        cls = REGISTERED_ENVS[n].cls
        mod = sys.modules[cls.__module__]
        def rec(eye, target, up=(0, 0, 1)):   # stands in for look_at (transforms3d is not here): a pose that remembers what it was built from
            return RefPose.create_from_pq(torch.tensor([float(x) for x in eye]), torch.tensor([7.0] + [float(x) for x in target]))   # q = (tag, target)
        su.look_at = rec
        had = getattr(mod, "look_at", None)
        if had is not None:
            mod.look_at = rec
        extra = dict(sensor_cam_eye_pos=[0.3, 0, 0.6], sensor_cam_target_pos=[-0.1, 0, 0.1]) if n == "PickCube-v1" else {}
        cfgs = cls._default_sensor_configs.fget(Fake(cls, **extra))
        if had is not None:
            mod.look_at = had
        cam = [c for c in cfgs if c.uid == "base_camera"][0]
        raw = cam.pose.raw_pose[0]
        assert float(raw[3]) == 7.0, (n, "builds its camera pose without look_at")
        eye, target = raw[:3].tolist(), raw[4:7].tolist()
        rows.append([*eye, *target, cam.width, cam.height, cam.fov, cam.near, cam.far])
    su.look_at = real
    OUT["sensor/names"] = np.array(names)
    put("sensor", rows=np.array(rows, dtype=np.float64))


def pusht_vectors():
    """PushT-v1: the reference's own _load_scene builds the pseudo-render tables (sapien calls land in mocks), then evaluate
    (pseudo_render_intersection), _get_obs_extra and the pose-based reward run on this package's states; the T is also put on and
    near its goal for the success branch."""
    from unittest.mock import MagicMock

    env, snaps = rollout_states("PushT-v1", n=6, steps=24, every=6)
    ref = Fake(RefPushT, device=torch.device("cpu"), robot_init_qpos_noise=0.02, scene=MagicMock(), agent=MagicMock(),
               obs_mode="state", obs_mode_struct=ns(use_state=True))
    RefPushT._load_scene(ref, {})
    put("PushT-tables", tee_render=ref.tee_render, world_to_goal_trans=ref.world_to_goal_trans, uv_grid=ref.uv_grid)
    # [table | Tee | goal_Tee | goal_ee | root | qpos 7 | qvel 7]: the T on its goal, 4 mm / 1.5 deg off, 2 cm off
    goal = snaps[-1][:, 26:33]
    for dxy, dang in ((0.0, 0.0), (0.004, 0.026), (0.02, 0.1)):
        st = snaps[-1].clone()
        ang = 2 * torch.atan2(goal[:, 6], goal[:, 3]) + dang
        st[:, 13:15] = goal[:, :2] + dxy; st[:, 15] = 0.02
        st[:, 16:20] = torch.stack([(ang / 2).cos(), torch.zeros(6), torch.zeros(6), (ang / 2).sin()], dim=1)
        st[:, 20:26] = 0.0
        snaps.append(st)
    rec = dict(state=[], obs=[], reward=[], success=[], intersection=[])
    for st in snaps:
        env.set_state(st)
        ref.tee, ref.goal_tee = actor(env, env._b_tee), actor(env, env._b_goal)
        ref.agent = ns(tcp=ns(pose=RefPose.create(env.tcp_pose)))
        info = RefPushT.evaluate(ref)
        extra = RefPushT._get_obs_extra(ref, info)
        obs = ref_common.flatten_state_dict(dict(agent=dict(qpos=env.qpos, qvel=env.qvel), extra=extra), use_torch=True)
        rew = RefPushT.compute_normalized_dense_reward(ref, obs=obs, action=None, info=info)
        rec["state"].append(st); rec["obs"].append(obs); rec["reward"].append(rew); rec["success"].append(info["success"])
        rec["intersection"].append(RefPushT.pseudo_render_intersection(ref))
    put("PushT-v1", num_envs=np.array(env.num_envs), **{k: torch.stack(v) for k, v in rec.items()})
    print("PushT-v1 snapshots", len(snaps), "successes", int(torch.stack(rec["success"]).sum()), "max intersection",
          float(torch.stack(rec["intersection"]).max()))


def main():
    pure_functions()
    T = ref_tasks
    common_kw = dict(obs_mode="state", obs_mode_struct=ns(use_state=True), robot_uids="panda", device=torch.device("cpu"))
    # flat state of the PickCube family: [table 13 | object 13 | goal 13 | root 13 | qpos 9 | qvel 9]
    def on_goal(env, st, dz=0.0):
        st[:, 13:16] = st[:, 26:29]; st[:, 15] += dz; st[:, 20:26] = 0.0; st[:, -9:] = 0.0
        moving = st.clone(); moving[:, -9] = 0.5                       # object placed but the robot still moving
        return [st, moving]

    def on_region(env, st):
        st[:, 13:15] = st[:, 26:28]; st[:, 15] = 0.02; st[:, 16:20] = torch.tensor([1.0, 0, 0, 0]); st[:, 20:26] = 0.0; st[:, -9:] = 0.0
        return [st]

    def upright(env, st):   # [table 13 | peg 13 | root 13 | qpos | qvel]; tilted up about the world y axis, keeping its roll
        q = env._qmul(torch.tensor([[np.cos(-np.pi / 4), 0.0, np.sin(-np.pi / 4), 0.0]], dtype=torch.float32),
                      torch.tensor([[np.cos(np.pi / 4), np.sin(np.pi / 4), 0.0, 0.0]], dtype=torch.float32))
        st[:, 13:16] = torch.tensor([0.3, 0.3, 0.12]); st[:, 16:20] = q; st[:, 20:26] = 0.0
        tilted = st.clone(); tilted[:, 15] = 0.13                     # upright but 1 cm too high
        return [st, tilted]

    task_vectors("PickCube-v1", T.PickCubeEnv, lambda env, agent: Fake(T.PickCubeEnv, agent=agent, cube=actor(env, env._b_cube),
                 goal_site=actor(env, env._b_goal), goal_thresh=0.025, **common_kw), solved=on_goal)
    task_vectors("PushCube-v1", T.PushCubeEnv, lambda env, agent: Fake(T.PushCubeEnv, agent=agent, obj=actor(env, env._b_cube),
                 goal_region=actor(env, env._b_goal), **common_kw), solved=on_region)
    task_vectors("PullCube-v1", T.PullCubeEnv, lambda env, agent: Fake(T.PullCubeEnv, agent=agent, obj=actor(env, env._b_cube),
                 goal_region=actor(env, env._b_goal), **common_kw), solved=on_region)
    task_vectors("LiftPegUpright-v1", T.LiftPegUprightEnv, lambda env, agent: Fake(T.LiftPegUprightEnv, agent=agent, peg=actor(env, env._b_cube), **common_kw),
                 solved=upright)
    # StackCube: [table | cubeA | cubeB | root | qpos | qvel]
    def stacked(env, st):
        st[:, 13:16] = st[:, 26:29]; st[:, 15] += 0.04; st[:, 16:20] = st[:, 29:33]; st[:, 20:26] = 0.0; st[:, 33:39] = 0.0; st[:, -9:] = 0.0
        off = st.clone(); off[:, 13] += 0.03                         # 3 cm off centre: not "on"
        return [st, off]
    task_vectors("StackCube-v1", T.StackCubeEnv, lambda env, agent: Fake(T.StackCubeEnv, agent=agent, cubeA=actor(env, env._b_cube),
                 cubeB=actor(env, env._b_goal), cube_half_size=torch.tensor([0.02] * 3), **common_kw), solved=stacked)

    # PokeCube: [table | cube | peg | goal | root | qpos | qvel]
    def poked(env, st):
        st[:, 13:15] = st[:, 39:41]; st[:, 15] = 0.02; st[:, 20:26] = 0.0; st[:, -9:] = 0.0
        # peg head against the cube, aligned with it: the "fit" branch
        fit = st.clone(); fit[:, 29:33] = fit[:, 16:20]; fit[:, 26] = fit[:, 13] - 0.12 - 0.02; fit[:, 27] = fit[:, 14]
        return [st, fit]
    task_vectors("PokeCube-v1", T.PokeCubeEnv, lambda env, agent: Fake(T.PokeCubeEnv, agent=agent, cube=actor(env, env._b_poked),
                 peg=actor(env, env._b_cube), goal_region=actor(env, env._b_goal),
                 peg_head_offsets=RefPose.create_from_pq(p=[0.12, 0, 0]), **common_kw), solved=poked)

    # PegInsertionSide: [table | peg | box | root | qpos | qvel]; per-env sizes are a function of the env index
    def inserted(env, st):
        goal = env.goal_pose
        st[:, 13:20] = goal; st[:, 20:26] = 0.0
        near = st.clone(); near[:, 13:16] -= env._qrot(goal[:, 3:7], torch.tensor([[0.03, 0.0, 0.0]]).repeat(len(st), 1))   # 3 cm short
        return [st, near]

    def peg_fake(env, agent):
        head = torch.zeros(env.num_envs, 3); head[:, 0] = env.peg_half_sizes[:, 0]
        return Fake(T.PegInsertionSideEnv, agent=agent, peg=actor(env, env._b_cube), box=actor(env, env._b_goal), peg_half_sizes=env.peg_half_sizes,
                    box_hole_radii=env.box_hole_radii, box_hole_offsets=RefPose.create_from_pq(p=env._hole_offset),
                    peg_head_offsets=RefPose.create_from_pq(p=head), **common_kw)
    task_vectors("PegInsertionSide-v1", T.PegInsertionSideEnv, peg_fake, solved=inserted)

    # StackPyramid: [table | A | B | C | root | qpos | qvel]; sparse reward
    def pyramid(env, st):
        base = torch.tensor([0.25, 0.3, 0.02]); ident = torch.tensor([1.0, 0, 0, 0])
        for k, dp in enumerate(((-0.0225, 0.0, 0.0), (0.0225, 0.0, 0.0), (0.0, 0.0, 0.0405))):
            o = 13 * (k + 1)
            st[:, o:o + 3] = base + torch.tensor(dp); st[:, o + 3:o + 7] = ident; st[:, o + 7:o + 13] = 0.0
        moving = st.clone(); moving[:, 39 + 7] = 0.05                 # the top cube still sliding
        return [st, moving]
    task_vectors("StackPyramid-v1", T.StackPyramidEnv, lambda env, agent: Fake(T.StackPyramidEnv, agent=agent, cubeA=actor(env, env._b_cube),
                 cubeB=actor(env, env._b_cubeB), cubeC=actor(env, env._b_cubeC), cube_half_size=torch.tensor([0.02] * 3), **common_kw),
                 solved=pyramid, zero_forces=True)
    # PullCubeTool: [table | cube | tool | root | qpos | qvel]
    def pulled(env, st):
        near = st.clone(); near[:, 13:15] = torch.tensor([-0.2, 0.05]); near[:, 15] = 0.02; near[:, 20:26] = 0.0          # within 0.6 m of the base
        away = st.clone(); away[:, 13] = 0.55                                                                            # pushed out of reach: -2
        hooked = st.clone(); hooked[:, 26:29] = hooked[:, 13:16] + torch.tensor([-(0.05 + 0.02), -0.067, 0.005]); hooked[:, 29:33] = torch.tensor([1.0, 0, 0, 0])
        return [near, away, hooked]
    task_vectors("PullCubeTool-v1", T.PullCubeToolEnv, lambda env, agent: Fake(T.PullCubeToolEnv, agent=agent, cube=actor(env, env._b_pulled),
                 l_shape_tool=actor(env, env._b_cube), **common_kw), solved=pulled)
    pusht_vectors()
    camera_vectors()
    vector_env_vectors()
    ee_controller_vectors()
    joint_controller_vectors()
    shader_vectors()
    constants_vectors()
    episode_init_vectors()
    reward_mode_vectors()
    sensor_config_vectors()
    out_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(HERE, "reference_vectors.npz")
    np.savez_compressed(out_path, **OUT)
    print("wrote", os.path.basename(out_path) + ":", len(OUT), "arrays")


if __name__ == "__main__":
    main()
