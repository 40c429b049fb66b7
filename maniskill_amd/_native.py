"""ctypes binding of the C ABI declared in include/msk_physx.h.

The product path loads ``maniskill_amd/csrc/libmsk_physx.so`` (hand-written HIP kernels for
gfx950) and FAILS LOUDLY when it is missing: there is no CPU fallback in this package.
``NativeLib`` itself is prefix-generic so that the test-suite can bind its CPU checker
behind the same Python classes; nothing under ``maniskill_amd/`` ever does that.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# MSK_LIB: another BUILD of the same HIP library (A/B measurements of two kernel versions on one GPU box); never a CPU library
DEFAULT_LIB = os.path.abspath(os.environ["MSK_LIB"]) if os.environ.get("MSK_LIB") else os.path.join(_HERE, "csrc", "libmsk_physx.so")

# enums of include/msk_physx.h
JOINT_FIXED, JOINT_REVOLUTE, JOINT_PRISMATIC = 0, 1, 2
BODY_KINEMATIC, BODY_DYNAMIC, BODY_LINK = 1, 2, 3
SHAPE_PLANE, SHAPE_BOX, SHAPE_SPHERE, SHAPE_CONVEX, SHAPE_CAPSULE, SHAPE_CYLINDER = 0, 1, 2, 3, 4, 5
(BUF_RIGID_BODY_DATA, BUF_ART_QPOS, BUF_ART_QVEL, BUF_ART_QACC, BUF_ART_QF, BUF_ART_TARGET_QPOS,
 BUF_ART_TARGET_QVEL, BUF_RIGID_BODY_FORCE, BUF_RIGID_BODY_TORQUE, BUF_ART_LINK_JOINT_FORCES) = range(10)
APPLY_RIGID_DATA, APPLY_ART_QPOS, APPLY_ART_QVEL, APPLY_ART_QF = 1, 2, 4, 8
APPLY_ART_TARGET_QPOS, APPLY_ART_TARGET_QVEL, APPLY_ART_ROOT_POSE = 16, 32, 64
APPLY_RIGID_FORCE, APPLY_RIGID_TORQUE, APPLY_ART_ROOT_VELOCITY = 128, 256, 512
FETCH_RIGID_DATA, FETCH_ART_QPOS, FETCH_ART_QVEL, FETCH_ART_QACC, FETCH_ART_TARGETS, FETCH_ART_LINK_FORCES = 1, 2, 4, 8, 16, 32

EXPORTS = [
    "create", "destroy", "last_error", "add_articulation", "add_link", "set_drive", "add_tendon",
    "add_actor", "add_shape", "disable_collision", "finalize", "set_scene_offsets", "buffer", "apply",
    "fetch", "update_kinematics", "step", "step_n", "get_step_parts", "set_step_parts", "query_create_pairs", "query_create_bodies", "query_buffer", "query_run",
    "get_sizes", "get_contacts", "get_env_contact_counts", "timing_enable", "timing_read",
    "set_solver_classes", "get_solver_class_counts", "declare_env_box", "declare_env_mass", "set_env_boxes", "set_env_masses",
    "bind_buffers", "batch", "set_articulation_floating", "warnings", "set_locked_axes", "reset_masked", "episode_book_step",
]
# include/msk_render.h — camera pipeline (both libraries)
RENDER_EXPORTS = ["render_add_mesh", "render_set_base_color", "render_set_texture", "render_bind_env_box", "render_set_lights", "render_set_local_lights", "render_finalize", "camera_create", "camera_buffer",
                  "camera_obs_buffer", "camera_set_outputs", "camera_take_picture"]
# include/msk_task.h — fused task kernels (HIP library only; the test-suite's CPU checker has no counterpart)
TASK_EXPORTS = ["task_pickcube_init", "task_pickcube_set_action", "task_pickcube_set_action_ee", "compute_ik_delta", "control_step", "task_pickcube_observe",
                "task_pusht_init", "task_pusht_set_action", "task_pusht_observe", "task_peg_init", "task_peg_observe"]
BATCH_STEP, BATCH_APPLY, BATCH_FETCH, BATCH_UPDATE_KINEMATICS = 0, 1, 2, 3
K_DYNAMICS, K_COLLIDE, K_SOLVE, K_SUBSTEP = 0, 1, 2, 3
# the kernels' names as rocprofv3 prints them (template arguments dropped); "substep" = begin of the first to end of the last
KERNEL_SLOTS = {"k_dynamics": K_DYNAMICS, "k_narrowphase": K_COLLIDE, "k_csolve": K_SOLVE, "substep": K_SUBSTEP}


class MskEpisodeBook(C.Structure):
    """include/msk_physx.h: msk_episode_book"""
    _fields_ = [("reward", C.c_void_p), ("elapsed", C.c_void_p), ("success", C.c_void_p), ("fail", C.c_void_p), ("terminated", C.c_void_p), ("truncated", C.c_void_p),
                ("success_stride", C.c_int32), ("fail_stride", C.c_int32), ("terminated_stride", C.c_int32), ("truncated_stride", C.c_int32),
                ("record_metrics", C.c_int32), ("ignore_terminations", C.c_int32), ("clear_done", C.c_int32),
                ("returns", C.c_void_p), ("success_once", C.c_void_p), ("fail_once", C.c_void_p),
                ("out_return", C.c_void_p), ("out_episode_len", C.c_void_p), ("out_reward", C.c_void_p), ("out_success_once", C.c_void_p), ("out_fail_once", C.c_void_p),
                ("out_success_at_end", C.c_void_p), ("out_fail_at_end", C.c_void_p), ("out_terminated", C.c_void_p), ("out_done", C.c_void_p), ("any_done", C.c_void_p)]


class MskConfig(C.Structure):
    _fields_ = [
        ("timestep", C.c_float),
        ("gravity", C.c_float * 3),
        ("solver_position_iterations", C.c_int32),
        ("solver_velocity_iterations", C.c_int32),
        ("contact_offset", C.c_float),
        ("rest_offset", C.c_float),
        ("bounce_threshold", C.c_float),
        ("sleep_threshold", C.c_float),
        ("enable_tgs", C.c_int32),
        ("enable_pcm", C.c_int32),
        ("contact_capacity", C.c_int32),
        ("reserved", C.c_int32 * 5),
    ]


IK_MAX_JOINTS = 8


class MskIkDesc(C.Structure):
    """msk_ik_desc (include/msk_task.h)"""
    _fields_ = [("ee_body", C.c_int32), ("root_body", C.c_int32), ("njoints", C.c_int32), ("joint_links", C.c_int32 * IK_MAX_JOINTS),
                ("damping", C.c_float), ("alpha", C.c_float)]


class PickCubeDesc(C.Structure):
    _fields_ = [
        ("cube", C.c_int32), ("goal", C.c_int32), ("tcp", C.c_int32), ("left_finger", C.c_int32), ("right_finger", C.c_int32),
        ("arm_dofs", C.c_int32), ("arm_delta", C.c_float), ("gripper_mid", C.c_float), ("gripper_half", C.c_float),
        ("goal_thresh", C.c_float), ("min_force", C.c_float), ("max_angle_deg", C.c_float), ("static_thresh", C.c_float),
        ("max_episode_steps", C.c_int32),
    ]


class PushTDesc(C.Structure):
    _fields_ = [
        ("tee", C.c_int32), ("goal", C.c_int32), ("tcp", C.c_int32), ("arm_dofs", C.c_int32), ("arm_delta", C.c_float),
        ("goal_xy", C.c_float * 2), ("goal_z_rot", C.c_float), ("world_to_goal", C.c_float * 6), ("uv_scale", C.c_float),
        ("intersection_thresh", C.c_float), ("max_episode_steps", C.c_int32),
    ]


def _fa(values, n=None):
    vals = [float(v) for v in values]
    if n is not None and len(vals) != n:
        raise ValueError(f"expected {n} floats, got {len(vals)}")
    return (C.c_float * len(vals))(*vals)


class NativeLib:
    """One loaded shared library exporting ``<prefix><name>`` for every name in EXPORTS."""

    def __init__(self, path: str = DEFAULT_LIB, prefix: str = "msk_"):
        if not os.path.exists(path):
            raise RuntimeError(
                f"native backend library not found: {path}. Build it with "
                "`python -c 'import __graft_entry__ as g; g.build()'` (hipcc --offload-arch=gfx950); "
                "maniskill_amd has no CPU fallback."
            )
        self.path, self.prefix = path, prefix
        if prefix == "msk_":
            # the HIP library shares the process's HIP runtime with torch: torch's own libamdhip64 has to be the one that is loaded
            # (a library opened first pulls in /opt/rocm's copy, and a second runtime in the process sees no device)
            import torch  # noqa: F401
        self.dll = C.CDLL(path)
        f = lambda name: getattr(self.dll, prefix + name)  # noqa: E731
        vp, i32, f32, u32 = C.c_void_p, C.c_int, C.c_float, C.c_uint32
        fp = C.POINTER(C.c_float)
        sig = {
            "create": (vp, [i32, C.POINTER(MskConfig)]),
            "destroy": (None, [vp]),
            "last_error": (C.c_char_p, [vp]),
            "warnings": (C.c_char_p, [vp]),
            "add_articulation": (i32, [vp, fp]),
            "add_link": (i32, [vp, i32, i32, i32, fp, fp, f32, f32, f32, fp, fp, i32, f32, f32]),
            "set_drive": (i32, [vp, i32, f32, f32, f32, i32]),
            "add_tendon": (i32, [vp, i32, i32, f32, f32, f32, f32, f32]),
            "add_actor": (i32, [vp, i32, fp, f32, fp, fp, f32, f32, i32]),
            "add_shape": (i32, [vp, i32, i32, fp, fp, fp, i32, f32, f32, f32, C.POINTER(u32), f32, f32]),
            "disable_collision": (i32, [vp, i32, i32]),
            "finalize": (i32, [vp, i32]),
            "set_scene_offsets": (i32, [vp, fp]),
            "buffer": (vp, [vp, i32, C.POINTER(C.c_int64)]),
            "apply": (i32, [vp, u32, vp]),
            "fetch": (i32, [vp, u32, vp]),
            "update_kinematics": (i32, [vp, vp]),
            "reset_masked": (i32, [vp, vp, vp, i32, vp, i32, vp, vp, vp]),
            "episode_book_step": (i32, [vp, i32, C.POINTER(MskEpisodeBook), vp]),
            "step": (i32, [vp, vp]),
            "step_n": (i32, [vp, i32, vp]),
            "get_step_parts": (i32, [vp]),
            "set_step_parts": (i32, [vp, i32]),
            "query_create_pairs": (i32, [vp, C.POINTER(C.c_int32), i32]),
            "query_create_bodies": (i32, [vp, C.POINTER(C.c_int32), i32]),
            "query_buffer": (vp, [vp, i32, C.POINTER(C.c_int64)]),
            "query_run": (i32, [vp, i32, vp]),
            "get_sizes": (i32, [vp, C.POINTER(C.c_int32)]),
            "get_contacts": (i32, [vp, i32, C.POINTER(C.c_int32), fp, i32]),
            "get_env_contact_counts": (i32, [vp, C.POINTER(C.c_int32)]),
            "declare_env_box": (i32, [vp, i32]),
            "declare_env_mass": (i32, [vp, i32]),
            "set_env_boxes": (i32, [vp, i32, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
            "set_env_masses": (i32, [vp, i32, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
            "set_solver_classes": (i32, [vp, C.POINTER(C.c_int32)]),
            "get_solver_class_counts": (i32, [vp, C.POINTER(C.c_int32)]),
            "set_articulation_floating": (i32, [vp, i32]),
            "set_locked_axes": (i32, [vp, i32, u32]),
            "bind_buffers": (i32, [vp, C.POINTER(C.c_void_p), C.c_int64]),
            "batch": (i32, [C.POINTER(C.c_void_p), i32, i32, u32, vp]),
            "timing_enable": (i32, [vp, i32]),
            "timing_read": (i32, [vp, i32, C.POINTER(C.c_double), C.POINTER(C.c_int32)]),
        }
        for name, (res, args) in sig.items():
            fn = f(name)
            fn.restype, fn.argtypes = res, args
            setattr(self, name, fn)
        render_sig = {
            "render_add_mesh": (i32, [vp, i32, fp, fp, i32, C.POINTER(C.c_int32), i32, i32]),
            "render_finalize": (i32, [vp]),
            "camera_create": (i32, [vp, i32, i32, f32, f32, f32, i32, fp]),
            "render_set_base_color": (i32, [vp, i32, C.POINTER(C.c_float)]),
            "render_bind_env_box": (i32, [vp, i32, i32]),
            "render_set_lights": (i32, [vp, C.POINTER(C.c_float), i32, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
            "render_set_local_lights": (i32, [vp, i32, C.POINTER(C.c_float)]),
            "render_set_texture": (i32, [vp, i32, C.POINTER(C.c_uint8), i32, i32, C.POINTER(C.c_float)]),
            "camera_buffer": (vp, [vp, i32, C.POINTER(C.c_int64)]),
            "camera_obs_buffer": (vp, [vp, i32, i32, C.POINTER(C.c_int64)]),
            "camera_set_outputs": (i32, [vp, i32, i32]),
            "camera_take_picture": (i32, [vp, i32, vp]),
        }
        for name, (res, args) in render_sig.items():
            fn = f(name)
            fn.restype, fn.argtypes = res, args
            setattr(self, name, fn)
        task_sig = {
            "task_pickcube_init": (i32, [vp, C.POINTER(PickCubeDesc)]),
            "task_pickcube_set_action": (i32, [vp, vp, vp]),
            "control_step": (i32, [vp, i32, vp]),
            "task_pickcube_observe": (i32, [vp, vp, vp, vp, vp, i32, vp]),
            "task_pickcube_set_action_ee": (i32, [vp, vp, i32, i32, C.c_float, C.c_float, C.c_float, vp]),
            "compute_ik_delta": (i32, [vp, C.POINTER(MskIkDesc), vp, vp, i32, vp]),
            "task_pusht_init": (i32, [vp, C.POINTER(PushTDesc), C.POINTER(C.c_uint8)]),
            "task_pusht_set_action": (i32, [vp, vp, vp]),
            "task_pusht_observe": (i32, [vp, vp, i32, vp, vp, vp, i32, vp]),
            "task_peg_init": (i32, [vp, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float)]),
            "task_peg_observe": (i32, [vp, vp, vp, vp, vp, vp, i32, vp]),
        }
        self.has_task_kernels = all(hasattr(self.dll, prefix + n) for n in task_sig)
        if self.has_task_kernels:
            for name, (res, args) in task_sig.items():
                fn = f(name)
                fn.restype, fn.argtypes = res, args
                setattr(self, name, fn)

    def check(self, ctx, code: int, what: str) -> int:
        if code < 0:
            msg = self.last_error(ctx)
            raise RuntimeError(f"{self.prefix}{what} failed ({code}): {msg.decode() if msg else ''}")
        return code


_default = None


def default_lib() -> NativeLib:
    """The HIP backend.  Raises RuntimeError if the extension has not been built."""
    global _default
    if _default is None:
        _default = NativeLib(DEFAULT_LIB, "msk_")
    return _default
