"""Task host mirrors (torch over the backend's buffers; fused task kernels where include/msk_task.h has them).

``REGISTERED`` maps the reference's env ids to the classes here (the same table ``maniskill_amd.vector.ManiSkillVectorEnv("<id>", ...)``
resolves); importing this package does not load the HIP library -- constructing an env does.
"""


def registered():
    """{env id: class}, as registered in the reference (mani_skill/utils/registration.py REGISTERED_ENVS)."""
    from ..vector import _registry
    return dict(_registry())
