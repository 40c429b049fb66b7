"""Template-level scene builders (host only, cold path).

Mirror of the pieces of the reference that populate a PickCube sub-scene:
  * TableSceneBuilder.build      mani_skill/utils/scene_builder/table/scene_builder.py:20-66
  * build_ground                 mani_skill/utils/building/ground.py:20-44
  * actors.build_cube / build_sphere   mani_skill/utils/building/actors/common.py:70-91
  * Panda agent (URDF, urdf_config, drives)   mani_skill/agents/robots/panda/panda.py:16-98
"""
from __future__ import annotations

import numpy as np

from .. import _native as N
from ..agents.urdf import add_urdf_articulation, load_model
from ..physx import SceneTemplate

TABLE_HEIGHT = 0.9196429
# value of table_scene.table_height (aabb z extent, scene_builder.py:49-57)
GROUND_ALTITUDE = -(3.5762787e-07 + 0.91964257)

PANDA_URDF_CONFIG = dict(
    _materials=dict(gripper=dict(static_friction=2.0, dynamic_friction=2.0, restitution=0.0)),
    link=dict(
        panda_leftfinger=dict(material="gripper", patch_radius=0.1, min_patch_radius=0.1),
        panda_rightfinger=dict(material="gripper", patch_radius=0.1, min_patch_radius=0.1),
    ),
)
PANDA_ARM_JOINTS = [f"panda_joint{i}" for i in range(1, 8)]
PANDA_GRIPPER_JOINTS = ["panda_finger_joint1", "panda_finger_joint2"]
PANDA_REST_QPOS = np.array([0.0, np.pi / 8, 0, -np.pi * 5 / 8, 0, np.pi * 3 / 4, np.pi / 4, 0.04, 0.04])


def box_mass_properties(half_size, density=1000.0):
    hx, hy, hz = [float(h) for h in half_size]
    m = density * 8.0 * hx * hy * hz
    I = (m / 3.0 * (hy * hy + hz * hz), m / 3.0 * (hx * hx + hz * hz), m / 3.0 * (hx * hx + hy * hy), 0.0, 0.0, 0.0)
    return m, I


def add_table_scene(tpl: SceneTemplate, material=(0.3, 0.3, 0.0)):
    """Kinematic table box with its top at z = 0 and the ground plane below it."""
    q = (float(np.cos(np.pi / 4)), 0.0, 0.0, float(np.sin(np.pi / 4)))  # euler2quat(0, 0, pi/2)
    table = tpl.add_actor("table-workspace", N.BODY_KINEMATIC, p=(-0.12, 0.0, -TABLE_HEIGHT), q=q)
    tpl.add_shape(table, N.SHAPE_BOX, p=(0, 0, TABLE_HEIGHT / 2), params=(2.418 / 2, 1.209 / 2, TABLE_HEIGHT / 2),
                  static_friction=material[0], dynamic_friction=material[1], restitution=material[2])
    # ground: static plane, normal = +x of the shape frame rotated onto +z (ground.py:38-40)
    tpl.add_shape(-1, N.SHAPE_PLANE, p=(0, 0, GROUND_ALTITUDE), q=(0.7071068, 0, -0.7071068, 0),
                  static_friction=material[0], dynamic_friction=material[1], restitution=material[2])
    return table


def add_cube(tpl: SceneTemplate, name, half_size, p, material=(0.3, 0.3, 0.0), density=1000.0):
    m, I = box_mass_properties([half_size] * 3, density)
    cube = tpl.add_actor(name, N.BODY_DYNAMIC, p=p, mass=m, inertia6=I)
    tpl.add_shape(cube, N.SHAPE_BOX, params=(half_size,) * 3, static_friction=material[0],
                  dynamic_friction=material[1], restitution=material[2])
    return cube


def add_site(tpl: SceneTemplate, name, p=(0, 0, 0)):
    """Kinematic marker without collision (build_sphere(..., add_collision=False))."""
    return tpl.add_actor(name, N.BODY_KINEMATIC, p=p)


def add_panda(tpl: SceneTemplate, root_p=(-0.615, 0.0, 0.0), stiffness=1e3, damping=1e2, force_limit=100.0, arm_stiffness=None,
              asset="panda_v2.json", disable_gravity=True):
    """Panda with the drive properties of its controllers (panda.py:68-98,177-190); arm_stiffness = 0 is what
    PDJointVelController.set_drive_property sets on the arm joints (pd_joint_vel.py:25-38)."""
    model = load_model(asset)   # panda_v2.json; panda_v3.json = PandaWristCam (panda_wristcam.py:16-17)
    art = add_urdf_articulation(tpl, model, "panda", root_p=root_p, urdf_config=PANDA_URDF_CONFIG,
                                disable_gravity=disable_gravity)  # True = balance_passive_force (base_agent.py:263-282)
    for k, bid in enumerate(tpl.art_active[art]):
        ks = stiffness if (arm_stiffness is None or k >= 7) else arm_stiffness
        tpl.set_drive(bid, ks, damping, force_limit, "force")
    return art


def build_pick_cube_template(cube_half_size=0.02, arm_stiffness=None):
    """Body order: 15 panda links (ids 0..14), table-workspace, cube, goal_site = 18 rows per env
    (_load_agent runs before _load_scene, sapien_env.py:725-759)."""
    tpl = SceneTemplate()
    art = add_panda(tpl, arm_stiffness=arm_stiffness)
    table = add_table_scene(tpl)
    cube = add_cube(tpl, "cube", cube_half_size, (0, 0, cube_half_size))
    goal = add_site(tpl, "goal_site")
    # base colours of the Color texture: cube red, goal site green (pick_cube.py:91,98); robot light grey, table / ground: default
    tpl.set_body_color(cube, (1.0, 0.0, 0.0, 1.0))
    tpl.set_body_color(goal, (0.0, 1.0, 0.0, 1.0))
    for b in range(len(tpl.body_names)):
        if tpl.body_names[b].startswith("panda_"):
            tpl.set_body_color(b, (0.9, 0.9, 0.9, 1.0))
    return tpl, dict(art=art, table=table, cube=cube, goal_site=goal)


# ---- PushT-v1 (mani_skill/envs/tasks/tabletop/push_t.py) ------------------------------------------------------
PANDA_STICK_REST_QPOS = np.array([0.662, 0.212, 0.086, -2.685, -0.115, 2.898, 1.673])  # WhiteTableSceneBuilder.initialize
TEE_COM_Y = 0.0375
TEE_BOXES = [((0.0, 0.0 - TEE_COM_Y, 0.0), (0.1, 0.025, 0.02)),            # horizontal bar (push_t.py:205-213)
             ((0.0, 4 * 0.025 - TEE_COM_Y, 0.0), (0.025, 0.075, 0.02))]     # vertical bar   (push_t.py:238-241)


def add_panda_stick(tpl: SceneTemplate, root_p=(-0.615, 0.0, 0.0), stiffness=1e3, damping=1e2, force_limit=100.0):
    """PandaStick (agents/robots/panda/panda_stick.py): 7 dof arm, a 10 cm stick (16-sided prism hull) on the hand."""
    model = load_model("panda_stick.json")
    art = add_urdf_articulation(tpl, model, "panda_stick", root_p=root_p, disable_gravity=True)
    for bid in tpl.art_active[art]:
        tpl.set_drive(bid, stiffness, damping, force_limit, "force")
    return art


def add_tee(tpl: SceneTemplate, name="Tee", mass=0.8, friction=3.0):
    """The T block (push_t.py:193-251): two boxes about the common centre of mass, mass 0.8, friction 3."""
    vols = [8 * h[0] * h[1] * h[2] for _, h in TEE_BOXES]
    dens = mass / sum(vols)
    I = np.zeros(3)
    for (c, h), v in zip(TEE_BOXES, vols):
        m = dens * v
        I += m / 3.0 * np.array([h[1] ** 2 + h[2] ** 2, h[0] ** 2 + h[2] ** 2, h[0] ** 2 + h[1] ** 2])
        I += m * np.array([c[1] ** 2 + c[2] ** 2, c[0] ** 2 + c[2] ** 2, c[0] ** 2 + c[1] ** 2])   # parallel axis (products vanish: x = 0)
    tee = tpl.add_actor(name, N.BODY_DYNAMIC, p=(0, 0, 0.1), mass=mass, inertia6=(I[0], I[1], I[2], 0, 0, 0))
    for c, h in TEE_BOXES:
        tpl.add_shape(tee, N.SHAPE_BOX, p=c, params=h, static_friction=friction, dynamic_friction=friction, restitution=0.0)
    return tee


def build_push_t_template():
    """Body order: 11 panda_stick links, table-workspace, Tee, goal_Tee, goal_ee (push_t.py:159-262)."""
    tpl = SceneTemplate()
    art = add_panda_stick(tpl)
    table = add_table_scene(tpl)
    tee = add_tee(tpl)
    goal_tee = tpl.add_actor("goal_Tee", N.BODY_KINEMATIC, p=(0, 0, 0.1))
    for c, h in TEE_BOXES:                                  # visual only, 0.2 mm thick
        tpl.add_visual(goal_tee, N.SHAPE_BOX, p=c, params=(h[0], h[1], 1e-4))
    goal_ee = tpl.add_actor("goal_ee", N.BODY_KINEMATIC, p=(0, 0, 0.1))
    ang = np.arange(16) * (2 * np.pi / 16)                 # cylinder visual r = 0.02, half length 1e-4, axis = local x
    disc = np.concatenate([np.c_[np.full(16, -1e-4), 0.02 * np.cos(ang), 0.02 * np.sin(ang)],
                           np.c_[np.full(16, 1e-4), 0.02 * np.cos(ang), 0.02 * np.sin(ang)]])
    tpl.add_visual(goal_ee, N.SHAPE_CONVEX, verts=disc)
    # base colours (push_t.py:170-252): the T red (TARGET_RED), its goal outline and the end-effector goal disc grey, white table top
    tpl.set_body_color(tee, (194 / 255, 19 / 255, 22 / 255, 1.0))
    tpl.set_body_color(goal_tee, (128 / 255, 128 / 255, 128 / 255, 1.0))
    tpl.set_body_color(goal_ee, (128 / 255, 128 / 255, 128 / 255, 1.0))
    tpl.set_body_color(table, (1.0, 1.0, 1.0, 1.0))
    for b in range(len(tpl.body_names)):
        if tpl.body_names[b].startswith("panda_"):
            tpl.set_body_color(b, (0.9, 0.9, 0.9, 1.0))
    return tpl, dict(art=art, table=table, tee=tee, goal_tee=goal_tee, goal_ee=goal_ee)
