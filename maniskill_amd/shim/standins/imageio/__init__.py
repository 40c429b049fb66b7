"""Stand-in for ``imageio`` (mani_skill/utils/visualization/misc.py:4 imports it at module level; writing videos needs the
real package)."""


def _missing(*a, **k):
    raise ImportError("imageio is not installed (stand-in module): video / image writing is unavailable")


get_writer = imwrite = imread = mimsave = mimwrite = _missing
