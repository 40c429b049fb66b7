"""maniskill_amd.fused_step: the control step of an env built by the reference's own code, taken over -- against the reference's unmodified BaseEnv.step on a
twin env (same seeds, same actions; tests/ref_fused_step.py in a fresh interpreter).  The bar: the simulation state and everything the step returns have the
reference's bits (the fused step restates the same arithmetic with fewer launches: ~230 -> ~85 kernels and 28 -> 10 boundary calls per OpenCabinetDrawer step)."""
import json
import os
import subprocess
import sys

import pytest

import ref_harness

HERE = os.path.dirname(os.path.abspath(__file__))
needs_ref = pytest.mark.skipif(ref_harness.find_reference() is None, reason="no ManiSkill checkout (reference) available")


def _run(backend, case, *args, timeout=1800):
    r = subprocess.run([sys.executable, os.path.join(HERE, "ref_fused_step.py"), backend, case, *[str(a) for a in args]], cwd=HERE, capture_output=True, text=True,
                       timeout=timeout)
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("FUSED ")]
    assert r.returncode == 0 and line, r.stdout[-2000:] + r.stderr[-3000:]
    return json.loads(line[-1][6:])


def _same_bits(res, level):
    assert res["level"] == level and res["reset_equal"] and res["flags"] and res["finite"] and res["restored"], res
    assert res["worst_state"] == 0.0 and res["worst_obs"] == 0.0 and res["worst_rew"] == 0.0, res


@needs_ref
def test_open_cabinet_drawer_task_plugin_has_the_references_bits_on_cpu_checker(built):
    res = _run("oracle", "cabinet", 6, 12)
    _same_bits(res, "task")
    assert res["groups"] > 1          # structurally different cabinets: several contexts behind one px


@needs_ref
def test_envs_that_stepped_before_their_first_reset_still_agree(built):
    """A capture runs throw-away steps before the env's first reset, and the reference's reset does not undo all of a step (drive targets stay in the simulation;
    OpenCabinetDrawer's _initialize_episode steps the physics once under them): with the SAME history the plugin and the reference agree bit for bit -- the
    comparison the -m gpu graph tests make (round 5's first hardware run compared a captured env with a fresh one: 0.29 apart at the first step)."""
    res = _run("oracle", "cabinet_history", 6, 8)
    assert res["reset_equal"] and res["worst_state"] == 0.0 and res["worst_obs"] == 0.0 and res["worst_rew"] == 0.0 and res["flags"], res


def _kernel_bar(res, graph=False):
    """the plugins on the fused task kernels: the simulation state has the reference's bits (eager; a replayed graph orders its launches differently: 1e-6);
    observation and reward are the kernels' fp32 restatement of the task code -- libm's tanhf / sqrtf / acosf against torch's: within 2e-6 --, flags equal"""
    assert res["level"] == "task-kernel" and res["graph"] == graph and res["reset_equal"] and res["flags"] and res["finite"] and res["restored"], res
    assert res["worst_state"] <= (1e-6 if graph else 0.0) and res["worst_obs"] <= 2e-6 and res["worst_rew"] <= 2e-6, res


@needs_ref
@pytest.mark.parametrize("case", ["kernel:PickCube-v1", "kernel:PickCube-v1:state_dict", "kernel:PegInsertionSide-v1", "kernel:PushT-v1", "kernel:PushT-v1:depth+segmentation",
                                  "kernel:PushT-v1:rgb+depth+segmentation"])
def test_task_plugins_on_the_fused_task_kernels_against_the_references_step_on_the_emulated_library(built, case):
    """BASELINE configs 2, 3 and 4 over the drop-in path: the env is the reference's, its control step the library's fused kernels (controller, substeps, one
    fetch, observe) -- here with the HIP sources compiled against tests/hipemu, against the reference's unmodified BaseEnv.step on the same library"""
    _kernel_bar(_run("emu", case, 4, 10))


@needs_ref
def test_the_kernel_plugin_steps_aside_while_the_reference_hides_an_actor(built):
    res = _run("emu", "kernel_hidden", 3, 3)
    assert res["level"] == "task-kernel" and res["usable"] == [True, False, True] and res["worst"] <= 2e-6, res


@needs_ref
@pytest.mark.parametrize("case", ["pickcube", "pickcube:dense", "pickcube:sparse"])
def test_pick_cube_task_plugin_has_the_references_bits_on_cpu_checker(built, case):
    _same_bits(_run("oracle", case, 5, 25), "task")


@needs_ref
def test_pick_cube_task_plugin_through_a_grasp(built):
    """is_grasping's force / angle branch: the cube between the fingers, the gripper closed, the hand lifted -- every env grasps, info and reward as the reference's"""
    res = _run("oracle", "pickcube_grasp", 3, 12)
    assert res["level"] == "task" and res["worst"] == 0.0 and res["grasped"] == 3, res


@needs_ref
@pytest.mark.parametrize("mode", ["pd_joint_delta_pos", "pd_joint_pos", "pd_joint_target_delta_pos", "pd_joint_vel"])
def test_fused_control_of_the_panda_modes_has_the_references_bits_on_cpu_checker(built, mode):
    _same_bits(_run("oracle", "panda:" + mode, 4, 10), "control")


@needs_ref
def test_a_plugin_that_does_not_cover_the_observation_mode_steps_aside(built):
    res = _run("oracle", "plugin_refused")
    assert res["level"] == "control" and "state observations" in res["refused"] and res["same"], res


@needs_ref
def test_unsupported_controllers_leave_the_env_untouched(built):
    res = _run("oracle", "unsupported")
    assert res["raised"] and res["untouched"] and "PDEEPose" in res["message"], res


@needs_ref
@pytest.mark.parametrize("env_id", ["OpenCabinetDrawer-v1", "PickCube-v1", "RollBall-v1", "PushCube-v1", "PegInsertionSide-v1", "PushT-v1", "StackCube-v1",
                                    "MS-AntWalk-v1", "MS-HumanoidStand-v1"])      # (round 6: pose setters through the all-true reset mask; a task's _before_control_step hook)
def test_steps_that_are_replayed_as_hip_graphs_are_graph_safe(built, env_id):
    """What a stream capture forbids (.item(), nonzero, boolean-mask indexing, host constants uploaded inside the step) and what a replay gets wrong (state
    handed from one step to the next through a tensor the earlier step allocated), watched in the op stream of two consecutive steps: OpenCabinetDrawer-v1
    through its task plugin, the others through the reference's OWN evaluate / observation / reward code behind the fused controller, with the host
    constants that code makes inside the step served from the device (fused_step.DeviceConstants)"""
    res = _run("oracle", "graph_safe:" + env_id, 3)
    assert res["sync"] == [] and res["flow"] == [], res


@needs_ref
@pytest.mark.parametrize("env_id", ["PushCube-v1", "PegInsertionSide-v1", "StackCube-v1", "PlaceSphere-v1", "MS-HumanoidWalk-v1"])      # (the Ant tasks: the reference's own partial reset
                                                                                                                              # raises -- ant.py:178 sets ALL envs' torso poses through the reset mask)
def test_the_capture_path_run_eagerly_has_the_references_bits(built, env_id):
    """accelerate(graph="dry"): the reference's own step under DeviceConstants behind the fused controller -- what a capture would run -- against the twin"""
    res = _run("oracle", "dry:" + env_id, 4, 10)
    assert res["level"] == "graph-dry" and res["reset_equal"] and res["flags"] and res["finite"] and res["restored"], res
    assert res["worst_state"] == 0.0 and res["worst_obs"] == 0.0 and res["worst_rew"] == 0.0, res


@needs_ref
def test_auto_accelerate_wraps_gym_make_and_leaves_unsupported_envs_alone(built):
    res = _run("oracle", "auto")
    assert res == dict(accelerated=True, left_alone=True, warned=True, undone=True), res


@needs_ref
def test_a_reconfigured_env_keeps_its_fused_step(built):
    res = _run("oracle", "reconfigure", 3)
    assert res == dict(worst=0.0, rebuilds=2, level="task", same_scene=True), res      # once for the new scene, once for the new control mode


@needs_ref
def test_the_capture_path_with_camera_observations_has_the_references_pictures(built):
    """PickCube-v1, obs_mode rgbd: the task plugin steps aside (state observations only), the reference's own step -- take_picture, the shader's texture
    transforms with their list indices served from the device -- behind the fused controller"""
    res = _run("oracle", "dry_rgbd", 2, 6)
    assert res["level"] == "graph-dry" and res["reset_equal"] and res["flags"] and res["finite"], res
    assert res["worst_state"] == 0.0 and res["worst_obs"] == 0.0 and res["worst_rew"] == 0.0, res


@needs_ref
@pytest.mark.parametrize("case", ["dry_pusht", "dry_pusht_cam"])
def test_push_t_with_its_intersection_renderer_patched_has_the_references_bits(built, case):
    """BASELINE config 3's task: pseudo_render_intersection restated without the boolean mask (fused_step._METHOD_PATCHES), the rest of the task's own step --
    with state and with rgb + depth + segmentation observations -- as the capture would run it"""
    res = _run("oracle", case, 3, 10)
    assert res["level"] == "graph-dry" and res["reset_equal"] and res["flags"] and res["finite"] and res["restored"], res
    assert res["worst_state"] == 0.0 and res["worst_obs"] == 0.0 and res["worst_rew"] == 0.0, res


@needs_ref
def test_a_task_that_keeps_state_in_python_is_not_captured(built):
    """accelerate(graph=True) watches the task's own step first (fused_step.graph_safety + the env's Python scalars before and after): the Draw tasks count their
    dots in ``self.draw_step`` and move THIS step's dot actor (drawing/draw.py:185-188) -- a replay runs no Python and would move the captured dot for ever.  That
    captures without an error, so it is refused by the watch (round 4 kept a list of task ids instead)"""
    res = _run("oracle", "not_verified")
    assert res["raised"] and res["untouched"] and "not safe to replay" in res["message"] and "Python attribute `draw_step`" in res["message"], res


@needs_ref
def test_state_handed_over_through_fresh_tensors_moves_into_persistent_ones(built):
    """RotateSingleObjectInHand rebinds ``self.prev_unit_vector`` to a tensor each step makes (rotate_single_object_in_hand.py:261): replayed as it is, a graph
    would re-read the capture's memory (round 5 refused the task for it).  The capture path copies such an attribute's value into ONE persistent tensor at the end
    of the step and binds the attribute to it: the watch then finds nothing, and the same path run eagerly has the reference's bits, through a partial reset"""
    res = _run("oracle", "graph_safe:RotateSingleObjectInHandLevel0-v1", 3)
    assert res["sync"] == [] and res["flow"] == [], res
    res = _run("oracle", "dry:RotateSingleObjectInHandLevel0-v1", 2, 12)
    assert res["worst_state"] == 0.0 and res["worst_obs"] == 0.0 and res["worst_rew"] == 0.0 and res["flags"], res


@needs_ref
def test_the_look_at_and_matrix_to_quaternion_idioms_become_selects_and_a_gather(built):
    """SO100GraspCube-v1 re-aims its camera through sapien_utils.look_at (``x[zero] = torch.zeros(3); x[~zero] /= norm[~zero].view(-1, 1)``, :349-355) and
    matrix_to_quaternion (``cand[F.one_hot(q_abs.argmax(-1), 4) > 0.5, :]``, rotation_conversions.py:161-163): boolean indexing sized by data in the reference,
    selects and a gather under DeviceConstants -- the watch finds nothing, the eager run of the same path has the reference's bits (round 5 refused the task)"""
    res = _run("oracle", "graph_safe:SO100GraspCube-v1", 3)
    assert res["sync"] == [] and res["flow"] == [] and res["rewritten"] >= 8, res
    res = _run("oracle", "dry:SO100GraspCube-v1", 2, 12)
    assert res["worst_state"] == 0.0 and res["worst_rew"] == 0.0 and res["flags"] and res["finite"], res


@needs_ref
def test_host_data_that_changes_between_steps_cannot_be_baked_into_a_graph(built):
    res = _run("oracle", "changing_constant")
    assert res["raised"] and res["served"] == 4 and res["clones"] and res["equal"], res


@needs_ref
@pytest.mark.parametrize("env_id,idioms", [("StackCube-v1", 8), ("PokeCube-v1", 12), ("PlugCharger-v1", 8), ("PullCubeTool-v1", 16)])
def test_masked_assignments_of_the_reference_become_selects(built, env_id, idioms):
    """``x[mask] = y[mask]`` (stack_cube.py:161), ``x[mask] += y[mask]`` (poke_cube.py:221), ``x[~m] = sin(h[~m]) / a[~m]`` (rotation_conversions.py:549-552): each
    a ``nonzero()`` -- a host synchronisation no capture allows.  DeviceConstants rewrites them into ``where`` while the step is warmed up, watched and captured:
    the watch then finds nothing, and the eager run of the same path has the reference's bits.  No per-task code (round 4 restated two reward functions)."""
    res = _run("oracle", "graph_safe:" + env_id, 3)
    assert res["sync"] == [] and res["flow"] == [] and res["rewritten"] >= idioms, res
    res = _run("oracle", "dry:" + env_id, 4, 12)
    assert res["worst_state"] == 0.0 and res["worst_obs"] == 0.0 and res["worst_rew"] == 0.0 and res["flags"], res


@needs_ref
def test_the_watch_does_flag_a_step_that_cannot_be_captured(built):
    res = _run("oracle", "graph_safe:DrawTriangle-v1", 3)           # `torch.all(self.dots_dist[mask], dim=-1)` (draw_triangle.py:382): a selection sized by the data
    assert res["sync"], res
    res = _run("oracle", "graph_safe:MS-HopperHop-v1", 3)           # `link.mass[0].item()` (control/hopper.py:196): host data the step has just made, read on the host
    assert res["sync"] == [] and res["flow"] == [], res


@pytest.mark.gpu
def test_open_cabinet_drawer_task_plugin_on_hip(built):
    _same_bits(_run("hip", "cabinet", 32, 20), "task")


@needs_ref
@pytest.mark.gpu
def test_open_cabinet_drawer_step_as_one_hip_graph(built):
    """the replayed graph against the reference's eager step: same launches, same arithmetic"""
    res = _run("hip", "cabinet_graph", 32, 20)
    assert res["graph"] and res["level"] == "task" and res["flags"] and res["finite"], json.dumps(res)
    assert res["worst_state"] <= 1e-6 and res["worst_obs"] <= 1e-6 and res["worst_rew"] <= 1e-6, json.dumps(res)


@needs_ref
@pytest.mark.gpu
@pytest.mark.parametrize("case", ["kernel_graph:PickCube-v1", "kernel_graph:PegInsertionSide-v1", "kernel_graph:PushT-v1:depth+segmentation"])
def test_task_plugins_on_the_fused_task_kernels_as_one_hip_graph(built, case):
    """BASELINE configs 2, 3, 4 over the drop-in path, replayed: the reference's env, the library's kernels, against the reference's eager step"""
    _kernel_bar(_run("hip", case, 64, 20), graph=True)


@needs_ref
@pytest.mark.gpu
def test_pick_cube_torch_plugin_as_one_hip_graph(built):
    res = _run("hip", "pickcube_graph_torch", 64, 20)
    assert res["graph"] and res["level"] == "task" and res["flags"] and res["finite"], res
    assert res["worst_state"] <= 1e-6 and res["worst_obs"] <= 1e-6 and res["worst_rew"] <= 1e-6, res


@needs_ref
@pytest.mark.gpu
def test_push_cube_reference_task_code_behind_the_fused_controller_as_one_hip_graph(built):
    res = _run("hip", "graph:PushCube-v1", 64, 20)
    assert res["graph"] and res["level"] == "graph" and res["flags"] and res["finite"], res
    assert res["worst_state"] <= 1e-6 and res["worst_obs"] <= 1e-6 and res["worst_rew"] <= 1e-6, res


@needs_ref
@pytest.mark.gpu
def test_peg_insertion_side_reference_task_code_as_one_hip_graph(built):
    """BASELINE config 4's task over the drop-in path: host constants made inside the step come from the device (DeviceConstants)"""
    res = _run("hip", "graph:PegInsertionSide-v1", 64, 20)
    assert res["graph"] and res["level"] == "graph" and res["flags"] and res["finite"], res
    assert res["worst_state"] <= 1e-6 and res["worst_obs"] <= 1e-6 and res["worst_rew"] <= 1e-6, res


@needs_ref
@pytest.mark.gpu
def test_pick_cube_with_camera_observations_as_one_hip_graph(built):
    """the reference's own step incl. take_picture / get_picture_cuda over the shim, captured: pictures and state against the eager twin"""
    res = _run("hip", "graph_rgbd", 16, 10)
    assert res["graph"] and res["level"] == "graph" and res["flags"] and res["finite"], res
    assert res["worst_state"] <= 1e-6 and res["worst_obs"] <= 1e-6 and res["worst_rew"] <= 1e-6, res


@needs_ref
@pytest.mark.gpu
def test_push_t_with_cameras_as_one_hip_graph(built):
    """BASELINE config 3 over the drop-in path: the reference's own step (patched intersection, take_picture, texture transforms) captured"""
    res = _run("hip", "graph_pusht", 16, 10)
    assert res["graph"] and res["level"] == "graph" and res["flags"] and res["finite"], res
    assert res["worst_state"] <= 1e-6 and res["worst_obs"] <= 1e-6 and res["worst_rew"] <= 1e-6, res
