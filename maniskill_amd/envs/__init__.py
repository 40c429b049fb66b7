"""Native fast-path task hosts (torch over the backend's buffers; fused task kernels where include/msk_task.h has them).

The general way to run a ManiSkill task on this backend is the reference's own Python over the ``sapien`` shim
(``maniskill_amd.shim.install()``; every registered task, the reference's ``ManiSkillVectorEnv`` / wrappers included).  The classes
here are hand-written hosts for the benchmark configurations: one fused kernel for the controller, one for evaluate / observations /
reward, a whole control step captured as one HIP graph.  ``registered()`` maps the reference's env ids to them; importing this
package does not load the HIP library -- constructing an env does.
"""

_ENVS = {}


def registered():
    """{env id: class}, ids as registered in the reference (mani_skill/utils/registration.py REGISTERED_ENVS)."""
    if not _ENVS:
        from .pick_cube import PickCubeEnv
        from .peg_insertion_side import PegInsertionSideEnv
        from .push_cube import PushCubeEnv
        from .push_t import PushTEnv
        from .stack_cube import StackCubeEnv
        from .pull_cube import PullCubeEnv
        from .lift_peg_upright import LiftPegUprightEnv
        from .poke_cube import PokeCubeEnv
        from .stack_pyramid import StackPyramidEnv
        from .pull_cube_tool import PullCubeToolEnv
        _ENVS.update({"PickCube-v1": PickCubeEnv, "PushCube-v1": PushCubeEnv, "StackCube-v1": StackCubeEnv, "PushT-v1": PushTEnv,
                      "PegInsertionSide-v1": PegInsertionSideEnv, "PullCube-v1": PullCubeEnv, "LiftPegUpright-v1": LiftPegUprightEnv,
                      "PokeCube-v1": PokeCubeEnv, "StackPyramid-v1": StackPyramidEnv, "PullCubeTool-v1": PullCubeToolEnv})
    return dict(_ENVS)
