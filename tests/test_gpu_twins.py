"""The -m gpu tests' BODIES in the CPU suite: the ones small enough run here, unchanged, on the product's HIP sources under the emulation of tests/hipemu
(tests/gpu_twin_plugin.py swaps the library and sends "cuda:0" to the CPU).  Round 4 ended with a red GPU suite because one -m gpu test still sized an array
for four solver classes after the ABI had grown to five -- a bug in a test body that no CPU run ever executed.  Now a test body that no longer matches the ABI
fails here first.

TWINS: every -m gpu node that passed within ~12 s under the plugin when the whole GPU suite was surveyed that way (round 5; `pytest -p gpu_twin_plugin -m gpu
--timeout 45`: 53 of 140 pass; the others need HIP graphs / events / streams, thousands of envs, or the reference build's GPU device), plus the node that broke
round 4.  A new -m gpu test small enough belongs on the list."""
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))

TWINS = [
    "test_bound_buffers.py::test_bound_contexts_equal_standalone_on_hip",
    "test_floating_base.py::test_floating_base_hip_equals_oracle",
    "test_gpu_parity.py::test_env_counts_that_are_not_powers_of_two[100]",
    "test_gpu_parity.py::test_env_counts_that_are_not_powers_of_two[1]",
    "test_gpu_parity.py::test_env_counts_that_are_not_powers_of_two[37]",
    "test_gpu_parity.py::test_external_wrench_matches_oracle",
    "test_gpu_parity.py::test_matches_committed_golden_rollout",
    "test_gpu_parity.py::test_per_env_box_instances_match_oracle",
    "test_gpu_parity.py::test_ragged_env_counts_match_the_oracle[17]",
    "test_gpu_parity.py::test_ragged_env_counts_match_the_oracle[1]",
    "test_gpu_parity.py::test_ragged_env_counts_match_the_oracle[3]",
    "test_gpu_parity.py::test_ragged_env_counts_match_the_oracle[5]",
    "test_gpu_parity.py::test_scripted_grasp_matches_oracle",
    "test_ik_generic.py::test_hip_ik_matches_the_torch_mirror",
    "test_link_forces.py::test_hip_link_forces_match_the_oracle",
    "test_locked_axes.py::test_locked_axes_hip_equals_oracle",
    "test_many_coordinates.py::test_hip_joint_friction_beyond_the_32nd_coordinate_matches_oracle",
    "test_more_tabletop_tasks.py::test_hip_matches_oracle_rollout[LiftPegUpright-v1]",
    "test_more_tabletop_tasks.py::test_hip_matches_oracle_rollout[PullCube-v1]",
    "test_oracle_contact_known_answers.py::test_contact_known_answer_scenes_hip_equals_oracle",
    "test_oracle_friction_kinematics.py::test_sliding_in_any_direction_hip_equals_oracle",
    "test_oracle_grasp.py::test_grasp_hip_equals_oracle",
    "test_oracle_press.py::test_pressing_hip_equals_oracle",
    "test_oracle_solver_rows.py::test_solver_rows_hip_equals_oracle[_scene_ball]",
    "test_oracle_solver_rows.py::test_solver_rows_hip_equals_oracle[_scene_hand]",
    "test_oracle_solver_rows.py::test_solver_rows_hip_equals_oracle[_scene_pendulum_with_joint_friction]",
    "test_oracle_solver_rows.py::test_solver_rows_hip_equals_oracle[_scene_stack]",
    "test_peg_insertion_side.py::test_ragged_env_counts_with_per_env_sizes_match_the_oracle[1]",
    "test_peg_insertion_side.py::test_ragged_env_counts_with_per_env_sizes_match_the_oracle[5]",
    "test_reference_conformance.py::test_reference_suite_on_hip[tests/structs/test_obs_mode_struct.py]",
    "test_reference_conformance.py::test_reference_suite_on_hip[tests/structs/test_pose.py]",
    "test_render.py::test_hip_local_lights_match_oracle",
    "test_render.py::test_hip_textures_match_oracle[-0.6-size2]",
    "test_render.py::test_hip_textures_match_oracle[0.0-size0]",
    "test_render.py::test_hip_textures_match_oracle[0.9-size1]",
    "test_rounded_shapes.py::test_bounce_hip_equals_oracle",
    "test_rounded_shapes.py::test_rounded_shapes_hip_equals_oracle",
    "test_trajectory.py::test_hip_replays_a_trace_recorded_on_the_oracle",
    "test_trajectory.py::test_hip_replays_the_committed_trace",
    "test_gpu_parity.py::test_every_solver_class_computes_the_same_bits[caps3]",
]


def test_gpu_test_bodies_run_on_the_emulated_library(built):
    env = dict(os.environ, MSK_FHR_CHILD="1", PYTHONPATH=HERE + os.pathsep + os.environ.get("PYTHONPATH", ""))
    r = subprocess.run([sys.executable, "-m", "pytest", "-p", "gpu_twin_plugin", "-m", "gpu", "-q", "-p", "no:cacheprovider", "-n", "4", "--timeout", "300", *TWINS],
                       cwd=HERE, env=env, capture_output=True, text=True, timeout=1500)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, tail
    assert f"{len(TWINS)} passed" in r.stdout, tail


def test_every_gpu_test_module_imports_and_every_twin_still_exists(built):
    """collection alone: a -m gpu module that no longer imports, or a twin that was renamed away, shows here"""
    r = subprocess.run([sys.executable, "-m", "pytest", "-m", "gpu", "--collect-only", "-q", "-p", "no:cacheprovider", "."], cwd=HERE, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    nodes = {ln.strip() for ln in r.stdout.splitlines() if "::" in ln}
    missing = [t for t in TWINS if t not in nodes]
    assert not missing, missing
    assert len(nodes) >= 140, len(nodes)
