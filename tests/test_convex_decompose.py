"""Non-convex static scenery (add_nonconvex_collision_from_file: PhysX collides the triangles themselves) is cut into convex pieces for
this engine (maniskill_amd/shim/sapien/_mesh.py: convex_decompose).  Known answers on meshes built here: a convex mesh stays one piece,
the notch of an L-shaped prism and the hole of a ring come out free, the material stays covered."""
import numpy as np
import pytest

from maniskill_amd.shim.sapien import _mesh


def _prism(poly, z0, z1):
    """closed prism over a simple polygon (counter-clockwise, may be non-convex): triangles by ear clipping"""
    poly = [tuple(p) for p in poly]
    idx, ears = list(range(len(poly))), []

    def area(a, b, c):
        return (b[0] - a[0]) * (c[1] - a[1]) - (b[1] - a[1]) * (c[0] - a[0])

    def inside(p, a, b, c):
        return area(a, b, p) >= -1e-12 and area(b, c, p) >= -1e-12 and area(c, a, p) >= -1e-12

    while len(idx) > 3:
        for k in range(len(idx)):
            i, j, l = idx[k - 1], idx[k], idx[(k + 1) % len(idx)]
            if area(poly[i], poly[j], poly[l]) <= 1e-12:
                continue
            if any(inside(poly[m], poly[i], poly[j], poly[l]) for m in idx if m not in (i, j, l)):
                continue
            ears.append((i, j, l))
            idx.pop(k)
            break
        else:
            raise RuntimeError("not a simple polygon")
    ears.append(tuple(idx))
    n = len(poly)
    v = np.array([(x, y, z0) for x, y in poly] + [(x, y, z1) for x, y in poly], dtype=np.float64)
    f = [(a, c, b) for a, b, c in ears] + [(a + n, b + n, c + n) for a, b, c in ears]
    for k in range(n):
        a, b = k, (k + 1) % n
        f += [(a, b, b + n), (a, b + n, a + n)]
    return v, np.array(f)


def _covered(hulls, pts):
    from scipy.spatial import ConvexHull
    out = np.zeros(len(pts), dtype=bool)
    for h in hulls:
        e = ConvexHull(h).equations
        out |= ((pts @ e[:, :3].T + e[:, 3]) <= 1e-9).all(axis=1)
    return out


def test_a_convex_mesh_stays_one_piece():
    v, f = _prism([(0, 0), (2, 0), (2, 1), (0, 1)], 0.0, 0.5)
    hulls, worst = _mesh.convex_decompose(v, f, max_parts=8)
    assert len(hulls) == 1 and worst <= 0.01 * np.linalg.norm([2, 1, 0.5])      # within the sampling resolution of the measure
    assert sorted(map(tuple, np.round(hulls[0], 6))) == sorted(map(tuple, v))


def test_the_notch_of_an_l_shaped_prism_comes_out_free():
    v, f = _prism([(0, 0), (2, 0), (2, 1), (1, 1), (1, 2), (0, 2)], 0.0, 0.5)
    one = _mesh.reduce_hull(v)
    notch = np.array([[1.5, 1.5, 0.25], [1.2, 1.2, 0.1], [1.8, 1.1, 0.4], [1.1, 1.8, 0.4]])
    solid = np.array([[0.5, 0.5, 0.25], [1.5, 0.5, 0.25], [0.5, 1.5, 0.25], [0.99, 0.99, 0.25], [1.9, 0.9, 0.05], [0.9, 1.9, 0.45]])
    assert _covered([one], notch)[:2].all()                 # what a single hull does: the notch is filled (its outer corner region is not)
    hulls, worst = _mesh.convex_decompose(v, f, max_parts=8)
    assert 2 <= len(hulls) <= 8 and worst <= 0.01 * np.linalg.norm([2, 2, 0.5]) + 1e-9
    assert not _covered(hulls, notch).any() and _covered(hulls, solid).all()


def test_the_hole_of_a_ring_comes_out_free():
    n, r0, r1 = 48, 0.2, 0.28
    ang = np.arange(n) * 2 * np.pi / n
    v = np.concatenate([np.c_[r * np.cos(ang), r * np.sin(ang), np.full(n, z)] for r in (r0, r1) for z in (0.0, 0.1)])   # in0 in1 out0 out1
    f = []
    for k in range(n):
        l = (k + 1) % n
        i0, i1, o0, o1 = k, k + n, k + 2 * n, k + 3 * n
        j0, j1, p0, p1 = l, l + n, l + 2 * n, l + 3 * n
        f += [(i0, i1, j1), (i0, j1, j0), (o0, p0, p1), (o0, p1, o1), (i0, j0, p0), (i0, p0, o0), (i1, o1, p1), (i1, p1, j1)]
    f = np.array(f)
    hulls, worst = _mesh.convex_decompose(v, f, max_parts=16)
    assert len(hulls) == 16
    rng = np.random.default_rng(0)
    a, z = rng.uniform(0, 2 * np.pi, 4000), rng.uniform(0.005, 0.095, 4000)
    hole = np.c_[0.15 * rng.uniform(0, 1, 4000) ** 0.5 * np.cos(a), 0.15 * rng.uniform(0, 1, 4000) ** 0.5 * np.sin(a), z]
    near = np.c_[(r0 - 0.012) * np.cos(a), (r0 - 0.012) * np.sin(a), z]          # 12 mm inside the inner wall
    body = np.c_[0.24 * np.cos(a), 0.24 * np.sin(a), z]
    assert not _covered(hulls, hole).any()
    assert _covered(hulls, near).mean() < 0.05              # sagitta of a 1/16 arc: 0.2 (1 - cos(pi/16)) = 3.8 mm
    assert _covered(hulls, body).mean() > 0.99
    assert _covered([_mesh.reduce_hull(v)], hole).all()     # a single hull: the whole hole is solid


def test_every_piece_fits_the_engines_hull_capacity():
    v, f = _prism([(0, 0), (2, 0), (2, 1), (1, 1), (1, 2), (0, 2)], 0.0, 0.5)
    for h in _mesh.convex_decompose(v, f, max_parts=4)[0]:
        assert 4 <= len(_mesh.reduce_hull(h)) <= _mesh.MAX_HULL_VERTS
