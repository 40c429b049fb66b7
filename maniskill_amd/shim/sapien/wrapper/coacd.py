"""sapien.wrapper.coacd.do_coacd: approximate convex decomposition needs the external ``coacd`` package, which is not in this
image; ManiSkill only reaches it for ``decomposition="coacd"`` collision records (utils/building/actor_builder.py:124-131)."""


def do_coacd(filename, **params):
    raise RuntimeError("convex decomposition (coacd) is not available in this backend; provide pre-decomposed collision meshes")
