"""Mesh files, convex hulls and mass properties for the shim's builders (host only, cold path).

What SAPIEN does natively behind ``PhysxCollisionShapeConvexMesh(filename, scale, material)`` /
``add_visual_from_file`` (mani_skill/utils/building/actor_builder.py:114-143; SURVEY §7.2 "Meshes without trimesh"):
binary + ASCII STL, OBJ and GLB (positions + indices) readers, convex hull cooking capped at 64 vertices per hull (the cap
GPU PhysX cooks with [ext]), mass / centre of mass / inertia of boxes, spheres, capsules, cylinders and hulls from a density
(``shape.set_density``, actor_builder.py:152).
"""
from __future__ import annotations

import json
import os
import struct

import numpy as np

MAX_HULL_VERTS = 64
_cache: dict = {}


# ------------------------------------------------------------------------------------------------ readers
def read_stl(path):
    """-> (vertices [n,3] float64, faces [m,3] int64); vertices are not merged."""
    with open(path, "rb") as f:
        data = f.read()
    if data[:5] == b"solid" and b"facet" in data[:400]:
        verts = []
        for line in data.decode("ascii", "ignore").splitlines():
            t = line.split()
            if len(t) == 4 and t[0] == "vertex":
                verts.append([float(t[1]), float(t[2]), float(t[3])])
        v = np.asarray(verts, dtype=np.float64).reshape(-1, 3)
    else:
        (ntri,) = struct.unpack_from("<I", data, 80)
        rec = np.dtype([("n", "<f4", 3), ("v", "<f4", (3, 3)), ("attr", "<u2")])
        tris = np.frombuffer(data, dtype=rec, count=ntri, offset=84)
        v = tris["v"].reshape(-1, 3).astype(np.float64)
    return v, np.arange(len(v), dtype=np.int64).reshape(-1, 3)


_PLY_TYPES = {"char": "i1", "int8": "i1", "uchar": "u1", "uint8": "u1", "short": "i2", "int16": "i2", "ushort": "u2", "uint16": "u2",
              "int": "i4", "int32": "i4", "uint": "u4", "uint32": "u4", "float": "f4", "float32": "f4", "double": "f8", "float64": "f8"}


def read_ply(path):
    """Stanford PLY (ascii, binary little / big endian): vertex x y z and the face lists, polygons fanned into triangles."""
    with open(path, "rb") as f:
        data = f.read()
    end = data.find(b"end_header")
    if not data.startswith(b"ply") or end < 0:
        raise RuntimeError(f"not a PLY file: {path}")
    body = end + len(b"end_header")
    body = data.index(b"\n", body) + 1
    fmt, elements = None, []
    for line in data[:end].decode("ascii", "replace").splitlines():
        t = line.split()
        if not t:
            continue
        if t[0] == "format":
            fmt = t[1]
        elif t[0] == "element":
            elements.append(dict(name=t[1], count=int(t[2]), props=[]))
        elif t[0] == "property" and elements:
            if t[1] == "list":
                elements[-1]["props"].append(("list", t[2], t[3], t[4]))
            else:
                elements[-1]["props"].append(("scalar", t[1], t[2]))
    verts, faces = None, []
    if fmt == "ascii":
        tok = data[body:].split()
        pos = 0
        for el in elements:
            rows = []
            for _ in range(el["count"]):
                row = {}
                for pr in el["props"]:
                    if pr[0] == "scalar":
                        row[pr[2]] = float(tok[pos]); pos += 1
                    else:
                        n = int(tok[pos]); pos += 1
                        row[pr[3]] = [int(x) for x in tok[pos:pos + n]]; pos += n
                rows.append(row)
            if el["name"] == "vertex":
                verts = np.array([[r["x"], r["y"], r["z"]] for r in rows], dtype=np.float64)
            elif el["name"] == "face":
                key = "vertex_indices" if rows and "vertex_indices" in rows[0] else "vertex_index"
                for r in rows:
                    idx = r[key]
                    faces.extend([idx[0], idx[k], idx[k + 1]] for k in range(1, len(idx) - 1))
    else:
        en = "<" if fmt == "binary_little_endian" else ">"
        pos = body
        for el in elements:
            if all(pr[0] == "scalar" for pr in el["props"]):
                dt = np.dtype([(pr[2], en + _PLY_TYPES[pr[1]]) for pr in el["props"]])
                arr = np.frombuffer(data, dtype=dt, count=el["count"], offset=pos)
                pos += dt.itemsize * el["count"]
                if el["name"] == "vertex":
                    verts = np.stack([arr["x"], arr["y"], arr["z"]], axis=1).astype(np.float64)
            else:
                for _ in range(el["count"]):
                    for pr in el["props"]:
                        if pr[0] == "scalar":
                            pos += np.dtype(_PLY_TYPES[pr[1]]).itemsize
                        else:
                            ct, it = np.dtype(en + _PLY_TYPES[pr[1]]), np.dtype(en + _PLY_TYPES[pr[2]])
                            n = int(np.frombuffer(data, dtype=ct, count=1, offset=pos)[0]); pos += ct.itemsize
                            idx = np.frombuffer(data, dtype=it, count=n, offset=pos).astype(np.int64); pos += it.itemsize * n
                            if el["name"] == "face" and pr[3] in ("vertex_indices", "vertex_index"):
                                faces.extend([idx[0], idx[k], idx[k + 1]] for k in range(1, n - 1))
    if verts is None:
        raise RuntimeError(f"no vertex element in {path}")
    return verts, np.array(faces, dtype=np.int64).reshape(-1, 3)


def read_obj(path):
    """-> list of (vertices, faces), one per ``o`` / ``g`` group that has faces (polygons are fan-triangulated)."""
    verts, groups, cur = [], [], []
    with open(path, "r", errors="ignore") as f:
        for line in f:
            t = line.split()
            if not t:
                continue
            if t[0] == "v" and len(t) >= 4:
                verts.append([float(t[1]), float(t[2]), float(t[3])])
            elif t[0] == "f":
                idx = [int(w.split("/")[0]) for w in t[1:]]
                idx = [i - 1 if i > 0 else len(verts) + i for i in idx]
                for k in range(1, len(idx) - 1):
                    cur.append([idx[0], idx[k], idx[k + 1]])
            elif t[0] in ("o", "g") and cur:
                groups.append(cur)
                cur = []
    if cur:
        groups.append(cur)
    V = np.asarray(verts, dtype=np.float64).reshape(-1, 3)
    out = []
    for g in groups:
        F = np.asarray(g, dtype=np.int64)
        used = np.unique(F)
        remap = -np.ones(len(V), dtype=np.int64)
        remap[used] = np.arange(len(used))
        out.append((V[used], remap[F]))
    return out


_GLTF_DTYPE = {5120: np.int8, 5121: np.uint8, 5122: np.int16, 5123: np.uint16, 5125: np.uint32, 5126: np.float32}
_GLTF_NCOMP = {"SCALAR": 1, "VEC2": 2, "VEC3": 3, "VEC4": 4, "MAT4": 16}


def read_glb(path):
    """Binary glTF 2.0 -> list of parts dict(vertices [n,3], faces [m,3], base_color rgba, name), node transforms applied.
    Positions, indices and the material's baseColorFactor only (no textures, normals, skins)."""
    with open(path, "rb") as f:
        data = f.read()
    magic, version, _ = struct.unpack_from("<III", data, 0)
    if magic != 0x46546C67:
        raise RuntimeError(f"{path}: not a GLB file")
    off, doc, blob = 12, None, b""
    while off < len(data):
        clen, ctype = struct.unpack_from("<II", data, off)
        chunk = data[off + 8: off + 8 + clen]
        if ctype == 0x4E4F534A:
            doc = json.loads(chunk.decode("utf-8"))
        elif ctype == 0x004E4942:
            blob = chunk
        off += 8 + clen
    return _gltf_parts(doc, [blob], os.path.dirname(path))


def _gltf_parts(doc, blobs, base_dir):
    def accessor(i):
        a = doc["accessors"][i]
        bv = doc["bufferViews"][a["bufferView"]]
        dt, nc = np.dtype(_GLTF_DTYPE[a["componentType"]]), _GLTF_NCOMP[a["type"]]
        buf = blobs[bv.get("buffer", 0)]
        start = bv.get("byteOffset", 0) + a.get("byteOffset", 0)
        stride = bv.get("byteStride", 0) or dt.itemsize * nc
        if stride == dt.itemsize * nc:
            arr = np.frombuffer(buf, dtype=dt, count=a["count"] * nc, offset=start).reshape(a["count"], nc)
        else:
            raw = np.frombuffer(buf, dtype=np.uint8, count=stride * (a["count"] - 1) + dt.itemsize * nc, offset=start)
            arr = np.lib.stride_tricks.as_strided(raw, (a["count"], dt.itemsize * nc), (stride, 1)).copy().view(dt).reshape(a["count"], nc)
        return arr

    def node_matrix(n):
        if "matrix" in n:
            return np.asarray(n["matrix"], dtype=np.float64).reshape(4, 4).T
        M = np.eye(4)
        if "scale" in n:
            M = np.diag(list(n["scale"]) + [1.0]) @ M
        if "rotation" in n:
            x, y, z, w = n["rotation"]
            from ._pose import _quat2mat
            R = np.eye(4)
            R[:3, :3] = _quat2mat([w, x, y, z])
            M = R @ M
        if "translation" in n:
            T = np.eye(4)
            T[:3, 3] = n["translation"]
            M = T @ M
        return M

    parts = []

    def visit(ni, parent):
        n = doc["nodes"][ni]
        M = parent @ node_matrix(n)
        if "mesh" in n:
            mesh = doc["meshes"][n["mesh"]]
            for prim in mesh["primitives"]:
                if prim.get("mode", 4) != 4 or "POSITION" not in prim["attributes"]:
                    continue
                v = accessor(prim["attributes"]["POSITION"]).astype(np.float64)
                f = (accessor(prim["indices"]).astype(np.int64).reshape(-1, 3) if "indices" in prim
                     else np.arange(len(v), dtype=np.int64).reshape(-1, 3))
                v = v @ M[:3, :3].T + M[:3, 3]
                if np.linalg.det(M[:3, :3]) < 0:
                    f = f[:, ::-1]
                color = [0.8, 0.8, 0.8, 1.0]
                if "material" in prim:
                    mat = doc["materials"][prim["material"]]
                    color = list(mat.get("pbrMetallicRoughness", {}).get("baseColorFactor", color))
                parts.append(dict(vertices=v, faces=f, base_color=color, name=mesh.get("name", n.get("name", ""))))
        for c in n.get("children", []):
            visit(c, M)

    scenes = doc.get("scenes") or [dict(nodes=list(range(len(doc.get("nodes", [])))))]
    for ni in scenes[doc.get("scene", 0)]["nodes"]:
        visit(ni, np.eye(4))
    return parts


def load_mesh_parts(path):
    """Any supported mesh file -> list of dict(vertices, faces, base_color, name) in the file's own frame and units."""
    key = ("parts", os.path.abspath(path), os.path.getmtime(path))
    if key in _cache:
        return _cache[key]
    ext = os.path.splitext(path)[1].lower()
    if ext == ".stl":
        v, f = read_stl(path)
        parts = [dict(vertices=v, faces=f, base_color=[0.8, 0.8, 0.8, 1.0], name=os.path.basename(path))]
    elif ext == ".obj":
        parts = [dict(vertices=v, faces=f, base_color=[0.8, 0.8, 0.8, 1.0], name=os.path.basename(path)) for v, f in read_obj(path)]
    elif ext == ".glb":
        # node transforms applied, nothing else: ManiSkill's GLB assets come out in their link frames that way (checked against the
        # collision STLs of the same links: identical bounds), i.e. the files carry their own y-up -> z-up root rotation
        parts = read_glb(path)
    elif ext == ".dae":
        parts = read_dae(path)
    elif ext == ".ply":
        v, f = read_ply(path)
        parts = [dict(vertices=v, faces=f, base_color=[0.8, 0.8, 0.8, 1.0], name=os.path.basename(path))]
    else:
        raise RuntimeError(f"unsupported mesh format: {path}")
    if not parts:
        raise RuntimeError(f"no triangles in {path}")
    _cache[key] = parts
    return parts


def read_dae(path):
    """COLLADA: <triangles>/<polylist> of every geometry, positions only, <up_axis> honoured, node transforms ignored
    except the unit scale."""
    import xml.etree.ElementTree as ET
    root = ET.parse(path).getroot()
    ns = {"c": root.tag[1:root.tag.index("}")]} if root.tag.startswith("{") else {}
    q = (lambda s: "/".join("c:" + t if t not in (".", "..", "") else t for t in s.split("/"))) if ns else (lambda s: s)
    unit = root.find(q("asset/unit"), ns)
    scale = float(unit.get("meter", "1")) if unit is not None else 1.0
    up = root.find(q("asset/up_axis"), ns)
    up = up.text.strip() if up is not None and up.text else "Y_UP"
    parts = []
    for geom in root.iter(("{%s}geometry" % ns["c"]) if ns else "geometry"):
        mesh = geom.find(q("mesh"), ns)
        if mesh is None:
            continue
        sources = {}
        for src in mesh.findall(q("source"), ns):
            fa = src.find(q("float_array"), ns)
            if fa is not None and fa.text:
                sources["#" + src.get("id")] = np.array(fa.text.split(), dtype=np.float64)
        vmap = {}
        for vs in mesh.findall(q("vertices"), ns):
            for inp in vs.findall(q("input"), ns):
                if inp.get("semantic") == "POSITION":
                    vmap["#" + vs.get("id")] = inp.get("source")
        for tag in ("triangles", "polylist"):
            for prim in mesh.findall(q(tag), ns):
                inputs = prim.findall(q("input"), ns)
                stride = max(int(i.get("offset", "0")) for i in inputs) + 1
                vin = next((i for i in inputs if i.get("semantic") == "VERTEX"), None)
                pel = prim.find(q("p"), ns)
                if vin is None or pel is None or not pel.text:
                    continue
                V = sources[vmap[vin.get("source")]].reshape(-1, 3) * scale
                idx = np.array(pel.text.split(), dtype=np.int64).reshape(-1, stride)[:, int(vin.get("offset", "0"))]
                if tag == "polylist":
                    vc = prim.find(q("vcount"), ns)
                    counts = np.array(vc.text.split(), dtype=np.int64)
                    faces, o = [], 0
                    for n in counts:
                        for k in range(1, n - 1):
                            faces.append([idx[o], idx[o + k], idx[o + k + 1]])
                        o += n
                    F = np.asarray(faces, dtype=np.int64).reshape(-1, 3)
                else:
                    F = idx.reshape(-1, 3)
                if up == "Y_UP":
                    V = np.stack([V[:, 0], -V[:, 2], V[:, 1]], axis=1)
                parts.append(dict(vertices=V, faces=F, base_color=[0.8, 0.8, 0.8, 1.0], name=geom.get("name", "")))
    return parts


# ------------------------------------------------------------------------------------------------ hulls
def reduce_hull(points, max_verts=MAX_HULL_VERTS):
    """Convex hull of `points`, greedily limited to `max_verts` vertices: start from the axis-extreme points and repeatedly
    add the input vertex that lies farthest outside the current hull (progressive hull).  Deterministic."""
    from scipy.spatial import ConvexHull
    pts = np.unique(np.round(np.asarray(points, dtype=np.float64), 7), axis=0)
    full = ConvexHull(pts)
    hv = pts[full.vertices]
    if len(hv) <= max_verts:
        return hv
    sel = set()
    for ax in range(3):
        sel.add(int(np.argmin(hv[:, ax])))
        sel.add(int(np.argmax(hv[:, ax])))
    sel = sorted(sel)
    k = 0
    while True:   # make sure the seed is full-dimensional
        try:
            ConvexHull(hv[sel])
            break
        except Exception:
            if k not in sel:
                sel.append(k)
            k += 1
    while len(sel) < max_verts:
        h = ConvexHull(hv[sel])
        d = hv @ h.equations[:, :3].T + h.equations[:, 3]   # signed distance of every vertex to every facet
        out = d.max(axis=1)
        out[sel] = -1.0
        j = int(np.argmax(out))
        if out[j] < 1e-5:
            break
        sel.append(j)
    h = ConvexHull(hv[sel])
    return hv[sel][h.vertices]


def hull_faces(verts):
    """Triangles of the convex hull of `verts` (indices into verts), counter-clockwise seen from outside."""
    from scipy.spatial import ConvexHull
    v = np.asarray(verts, dtype=np.float64)
    h = ConvexHull(v)
    tris = h.simplices.copy()
    c = v.mean(axis=0)
    n = np.cross(v[tris[:, 1]] - v[tris[:, 0]], v[tris[:, 2]] - v[tris[:, 0]])
    flip = np.einsum("ij,ij->i", n, v[tris[:, 0]] - c) < 0
    tris[flip] = tris[flip][:, ::-1]
    return tris.astype(np.int64)


def cook_convex(path, scale=(1, 1, 1)):
    """One hull (<= 64 vertices) of everything in the file, scaled.  -> (vertices float32 [n,3], faces)"""
    sc = np.asarray(scale, dtype=np.float64).reshape(-1)
    sc = np.full(3, sc[0]) if sc.size == 1 else sc
    key = ("hull", os.path.abspath(path), os.path.getmtime(path))
    if key not in _cache:
        pts = np.concatenate([p["vertices"] for p in load_mesh_parts(path)])
        _cache[key] = np.round(reduce_hull(pts), 6)
    kf = key + (tuple(float(x) for x in sc),)      # every sub-scene asks for the same file and scale again: cooked once
    if kf not in _cache:
        v = (_cache[key] * sc).astype(np.float32)
        v.setflags(write=False)
        _cache[kf] = (v, hull_faces(v))
    return _cache[kf]


def cook_multiple_convex(path, scale=(1, 1, 1)):
    """PhysxCollisionShapeConvexMesh.load_multiple: one hull per part / connected group of the file."""
    sc = np.asarray(scale, dtype=np.float64).reshape(-1)
    sc = np.full(3, sc[0]) if sc.size == 1 else sc
    out = []
    for p in load_mesh_parts(path):
        try:
            v = (np.round(reduce_hull(p["vertices"]), 6) * sc).astype(np.float32)
        except Exception:
            continue
        out.append((v, hull_faces(v)))
    if not out:
        raise RuntimeError(f"failed to cook any convex mesh from {path}")
    return out


def _clip_triangles(tri, axis, value):
    """Split triangles (n, 3, 3) by the plane x[axis] = value -> (below, above), each (m, 3, 3); cut triangles are re-triangulated as fans."""
    d = tri[:, :, axis] - value
    lo_all, hi_all = (d <= 0).all(axis=1), (d >= 0).all(axis=1)
    below, above = [tri[lo_all]], [tri[hi_all & ~lo_all]]
    for t, dd in zip(tri[~lo_all & ~hi_all], d[~lo_all & ~hi_all]):
        polys = ([], [])
        for i in range(3):
            a, b, da, db = t[i], t[(i + 1) % 3], dd[i], dd[(i + 1) % 3]
            if da <= 0:
                polys[0].append(a)
            if da >= 0:
                polys[1].append(a)
            if (da < 0 < db) or (db < 0 < da):
                x = a + (b - a) * (da / (da - db))
                x[axis] = value
                polys[0].append(x)
                polys[1].append(x)
        for poly, out in zip(polys, (below, above)):
            if len(poly) >= 3:
                out.append(np.array([[poly[0], poly[k], poly[k + 1]] for k in range(1, len(poly) - 1)]))
    return np.concatenate(below), np.concatenate(above)


def _surface_samples(tri, spacing):
    """Points on the triangles no farther apart than ~`spacing` (corners, and a barycentric lattice on the large ones)."""
    pts = [tri.reshape(-1, 3)]
    e = np.linalg.norm(tri - np.roll(tri, 1, axis=1), axis=2).max(axis=1)
    n = np.minimum(np.ceil(e / spacing).astype(int), 64)
    for k in np.unique(n[n > 1]):
        t = tri[n == k]
        i, j = np.meshgrid(np.arange(k + 1), np.arange(k + 1), indexing="ij")
        m = (i + j) <= k
        w = np.stack([i[m], j[m], k - i[m] - j[m]], axis=1) / float(k)          # (q, 3) barycentric weights
        pts.append(np.einsum("qc,ncd->nqd", w, t).reshape(-1, 3))
    return np.concatenate(pts)


def _inside_solid(points, tri):
    """Ray parity (+z) of `points` against the triangles: True where a point lies inside the closed surface.  Points whose ray grazes
    an edge are ambiguous; the caller jitters.  Open surfaces give parity of whatever is above the point, which is what a thin
    sheet should give (nothing is inside)."""
    out = np.zeros(len(points), dtype=bool)
    if not len(points):
        return out
    lo, hi = points[:, :2].min(0), points[:, :2].max(0)
    t = tri[(tri[:, :, 0].max(1) >= lo[0]) & (tri[:, :, 0].min(1) <= hi[0]) & (tri[:, :, 1].max(1) >= lo[1]) & (tri[:, :, 1].min(1) <= hi[1])]
    if not len(t):
        return out
    a, b, c = t[:, 0], t[:, 1], t[:, 2]
    det = (b[:, 0] - a[:, 0]) * (c[:, 1] - a[:, 1]) - (b[:, 1] - a[:, 1]) * (c[:, 0] - a[:, 0])
    ok = np.abs(det) > 1e-18
    a, b, c, det = a[ok], b[ok], c[ok], det[ok]
    for i0 in range(0, len(points), 256):
        p = points[i0:i0 + 256]
        px, py = p[:, 0:1] - a[None, :, 0], p[:, 1:2] - a[None, :, 1]
        u = (px * (c[:, 1] - a[:, 1]) - py * (c[:, 0] - a[:, 0])) / det
        v = (py * (b[:, 0] - a[:, 0]) - px * (b[:, 1] - a[:, 1])) / det
        hit = (u >= 0) & (v >= 0) & (u + v <= 1)
        z = a[None, :, 2] + u * (b[:, 2] - a[:, 2]) + v * (c[:, 2] - a[:, 2])
        out[i0:i0 + 256] = ((hit & (z > p[:, 2:3])).sum(axis=1) % 2) == 1
    return out


def _concavity(tri, cuts, spacing, whole):
    """How far the convex hull of a surface piece reaches into free space.  Sample points of the hull's faces -> their distances d to
    the surface of the WHOLE mesh; samples of faces that lie in one of the piece's cutting planes and fall inside the solid count as
    0 (the material goes on behind the cut).  `whole` = (KD-tree of surface samples, triangles) of the whole mesh.
    -> (hull vertices, max d, excess) with excess = sum d x area, a stand-in for the volume the hull adds."""
    from scipy.spatial import ConvexHull
    pts = np.unique(np.round(tri.reshape(-1, 3), 7), axis=0)
    try:
        h = ConvexHull(pts)
    except Exception:                                        # flat or degenerate piece: its own hull, nothing to refine
        return pts, 0.0, 0.0
    hv = pts[h.vertices]
    faces = pts[h.simplices]                                 # (m, 3, 3)
    on_cut = np.zeros(len(faces), dtype=bool)
    for axis, value in cuts:
        on_cut |= (np.abs(faces[:, :, axis] - value) < 1e-6).all(axis=1)
    area = 0.5 * np.linalg.norm(np.cross(faces[:, 1] - faces[:, 0], faces[:, 2] - faces[:, 0]), axis=1)
    # probes: face centroids + the centroids of the four sub-triangles of every face larger than a probe cell
    mid = 0.5 * (faces + np.roll(faces, -1, axis=1))
    big = area > (4.0 * spacing) ** 2
    sub = [np.stack([faces[big][:, k], mid[big][:, k], mid[big][:, (k + 2) % 3]], axis=1).mean(axis=1) for k in range(3)] + [mid[big].mean(axis=1)]
    probe = np.concatenate([faces.mean(axis=1)] + sub)
    w = np.concatenate([np.where(big, 0.2, 1.0) * area, np.tile(0.2 * area[big], 4)])
    cutp = np.concatenate([on_cut, np.tile(on_cut[big], 4)])
    tree, wtri = whole
    cap = 40.0 * spacing                                     # "far" is all the ranking needs to know
    d, _ = tree.query(probe, distance_upper_bound=cap, workers=-1)
    d = np.maximum(np.minimum(d, cap) - 0.5 * spacing, 0.0)  # the sampling's own resolution
    if cutp.any():
        jit = probe[cutp] + np.array([1.37e-5, 2.11e-5, 0.0])    # off the mesh's own edge / vertex coordinates
        d[np.flatnonzero(cutp)[_inside_solid(jit, wtri)]] = 0.0
    return hv, float(d.max()), float((d * w).sum())


def convex_decompose(verts, faces, tol=None, max_parts=24, groups=None):
    """Approximate convex decomposition of a triangle mesh that is NOT convex (static scenery added with
    add_nonconvex_collision_from_file: PhysX collides its triangles as they are; this engine collides convex shapes).  The mesh is
    bisected recursively -- always the piece whose hull adds the most volume, across the axis whose halves add the least, triangles
    clipped at the cut -- until every piece's hull stays within `tol` of the surface (default: 1 % of the bounding-box diagonal, at
    least 2 mm) or `max_parts` pieces exist.  `groups` (one id per face) names parts of the mesh that start out as pieces of their
    own.  -> (list of hull vertex arrays, largest remaining hull-to-surface distance).  Deterministic."""
    import heapq
    from scipy.spatial import cKDTree
    v = np.asarray(verts, dtype=np.float64)
    tri = v[np.asarray(faces, dtype=np.int64)]
    ok = np.linalg.norm(np.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0]), axis=1) > 1e-14
    tri = tri[ok]
    gid = np.zeros(len(tri), dtype=np.int64) if groups is None else np.asarray(groups, dtype=np.int64)[ok]
    diag = float(np.linalg.norm(v.max(0) - v.min(0)))
    tol = max(0.002, 0.01 * diag) if tol is None else float(tol)
    spacing = 0.5 * tol
    surf = _surface_samples(tri, spacing)
    surf = np.unique(np.round(surf / (0.5 * spacing)).astype(np.int64), axis=0) * (0.5 * spacing)     # one sample per half-spacing cell
    whole = (cKDTree(surf), tri)
    heap, done, tick = [], [], 0
    for g in np.unique(gid):
        hv, dmax, ex = _concavity(tri[gid == g], (), spacing, whole)
        heapq.heappush(heap, (-ex, tick, dmax, tri[gid == g], (), hv))
        tick += 1
    while heap and len(heap) + len(done) < max_parts:
        ex, _, dmax, t, cuts, hv = heapq.heappop(heap)
        if dmax <= tol:
            done.append((dmax, hv))
            continue
        p = t.reshape(-1, 3)
        lo, hi = p.min(0), p.max(0)
        best = None
        for axis in range(3):
            if hi[axis] - lo[axis] < 2.0 * tol:
                continue
            value = float(0.5 * (lo[axis] + hi[axis]))
            halves = _clip_triangles(t, axis, value)
            if min(len(halves[0]), len(halves[1])) == 0:
                continue
            kids = [(half, cuts + ((axis, value),)) + _concavity(half, cuts + ((axis, value),), spacing, whole) for half in halves]
            score = sum(k[4] for k in kids)
            if best is None or score < best[0] - 1e-12:
                best = (score, kids)
        if best is None:
            done.append((dmax, hv))
            continue
        for half, hcuts, hhv, hd, hex_ in best[1]:
            heapq.heappush(heap, (-hex_, tick, hd, half, hcuts, hhv))
            tick += 1
    pieces = done + [(dmax, hv) for _, _, dmax, _, _, hv in heap]
    worst = max(d for d, _ in pieces)
    return [hv for _, hv in pieces if len(hv) >= 4], worst


def cook_nonconvex(path, max_parts=16):
    """add_nonconvex_collision_from_file: the file as at most max(`max_parts`, number of parts of the file) convex pieces in the file's
    own frame and units -- every part starts as its hull; the piece whose hull adds the most volume is cut until all hulls stay
    within 1 % of the mesh's size of the surface or the budget is spent (convex_decompose).
    -> (list of hull vertex arrays, largest hull-to-surface distance left, as a fraction of the mesh's bounding-box diagonal)"""
    key = ("nonconvex", os.path.abspath(path), os.path.getmtime(path), max_parts)
    if key in _cache:
        return _cache[key]
    vs, fs, gs, o = [], [], [], 0
    for k, p in enumerate(load_mesh_parts(path)):
        if len(p["faces"]) == 0:
            continue
        vs.append(np.asarray(p["vertices"], dtype=np.float64))
        fs.append(np.asarray(p["faces"], dtype=np.int64) + o)
        gs.append(np.full(len(p["faces"]), k))
        o += len(vs[-1])
    if not vs:
        raise RuntimeError(f"no triangles in {path}")
    v, f, g = np.concatenate(vs), np.concatenate(fs), np.concatenate(gs)
    diag = float(np.linalg.norm(v.max(0) - v.min(0)))
    if len(f) > 40000:        # scanned scenes: the cuts and the inside tests are linear in the triangle count; 0.2 % of the size is below what 16 pieces resolve
        cell = 0.002 * diag
        key_ = np.floor((v - v.min(0)) / cell).astype(np.int64)
        _, inv = np.unique(key_, axis=0, return_inverse=True)
        inv = inv.reshape(-1)
        rep = np.zeros((inv.max() + 1, 3))
        np.add.at(rep, inv, v)
        rep /= np.bincount(inv)[:, None]
        f2 = inv[f]
        ok = (f2[:, 0] != f2[:, 1]) & (f2[:, 1] != f2[:, 2]) & (f2[:, 0] != f2[:, 2])
        v, f, g = rep, f2[ok], g[ok]
    try:
        hulls, worst = convex_decompose(v, f, tol=0.01 * diag, max_parts=max(max_parts, len(vs)), groups=g)
    except Exception:                                        # degenerate input (flat sheets ...): the parts' hulls as they are
        hulls, worst = vs, 0.0
    _cache[key] = (hulls, worst / diag if diag > 0 else 0.0)
    return _cache[key]


def prism(radius, half_length, sides=16, axis=0):
    """Cylinder stand-in: `sides`-gon prism along `axis` (SAPIEN / PhysX cylinders and capsules lie along local x)."""
    ang = np.arange(sides) * (2 * np.pi / sides)
    ring = np.stack([radius * np.cos(ang), radius * np.sin(ang)], axis=1)
    v = np.concatenate([np.c_[np.full(sides, -half_length), ring], np.c_[np.full(sides, half_length), ring]])
    if axis != 0:
        v = np.roll(v, axis, axis=1)
    return np.round(v, 6).astype(np.float32)


def cluster_simplify(verts, faces, max_tris):
    """Vertex-clustering simplification of a triangle mesh to at most `max_tris` triangles: vertices are merged per cell of a
    uniform grid (representative = the cell's quadric-error minimiser), collapsed triangles dropped; the cell size is bisected to the finest grid that
    meets the budget.  -> (vertices, faces, moved) where `moved` is the largest distance between an input vertex and the vertex
    that replaces it: the measured error quoted for visual meshes that are too dense for the rasteriser's scene template
    (SURVEY.md §7.2)."""
    v = np.asarray(verts, dtype=np.float64)
    f = np.asarray(faces, dtype=np.int64)
    if len(f) <= max_tris:
        return v, f, 0.0
    lo, hi = v.min(0), v.max(0)
    ext = float(np.max(hi - lo))

    def at(cell):
        key = np.floor((v - lo) / cell).astype(np.int64)
        uniq, inv = np.unique(key, axis=0, return_inverse=True)
        inv = inv.reshape(-1)
        nf = inv[f]
        keep = (nf[:, 0] != nf[:, 1]) & (nf[:, 1] != nf[:, 2]) & (nf[:, 0] != nf[:, 2])
        nf = nf[keep]
        if len(nf):
            srt = np.sort(nf, axis=1)
            _, first = np.unique(srt, axis=0, return_index=True)
            nf = nf[np.sort(first)]
        return uniq, inv, nf

    a, b = ext / 512.0, ext            # too fine .. certainly coarse enough
    for _ in range(18):
        mid = 0.5 * (a + b)
        if len(at(mid)[2]) <= max_tris:
            b = mid
        else:
            a = mid
    uniq, inv, nf = at(b)
    cnt = np.bincount(inv, minlength=len(uniq)).astype(np.float64)
    mean = np.stack([np.bincount(inv, weights=v[:, k], minlength=len(uniq)) / np.maximum(cnt, 1) for k in range(3)], axis=1)
    # the representative of a cell: the point that minimises the summed squared distances to the planes of the triangles that touch the cell's vertices (quadric
    # error, area weighted; Lindstrom's clustering) instead of the cell's mean -- a mean pulls convex surfaces inwards (silhouettes shrink: tools/camera_fidelity.py
    # measured segmentation IoU 0.82-0.94 per Panda link at 128 x 128 with means).  Solved about the mean, along the well-determined directions only, and kept inside
    # the cell: a flat or degenerate neighbourhood stays at its mean.
    tri = v[f]
    nrm = np.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0])
    area = np.linalg.norm(nrm, axis=1)
    okt = area > 1e-18
    nrm = np.where(okt[:, None], nrm / np.maximum(area, 1e-300)[:, None], 0.0)
    dpl = -(nrm * tri[:, 0]).sum(1)
    A = np.zeros((len(uniq), 3, 3))
    rhs = np.zeros((len(uniq), 3))
    wA = area[:, None, None] * nrm[:, :, None] * nrm[:, None, :]
    wb = -(area * dpl)[:, None] * nrm
    for k in range(3):
        cells = inv[f[:, k]]
        np.add.at(A, cells, wA)
        np.add.at(rhs, cells, wb)
    r = rhs - np.einsum("nij,nj->ni", A, mean)
    U, S, Vt = np.linalg.svd(A)
    Sinv = np.where(S > 1e-3 * np.maximum(S[:, :1], 1e-300), 1.0 / np.maximum(S, 1e-300), 0.0)
    dx = np.einsum("nji,nj->ni", Vt, Sinv * np.einsum("nji,nj->ni", U, r))
    clo = lo + uniq * b
    nv = np.clip(mean + dx, clo - 0.25 * b, clo + 1.25 * b)
    nv = np.where(np.isfinite(nv).all(axis=1, keepdims=True), nv, mean)
    used = np.unique(nf)
    remap = -np.ones(len(uniq), dtype=np.int64)
    remap[used] = np.arange(len(used))
    moved = float(np.linalg.norm(v - nv[inv], axis=1).max())      # measured: the farthest any input vertex moved (<= cell diagonal)
    return nv[used], remap[nf], moved


# ------------------------------------------------------------------------------------------------ mass properties
def _sym(ixx, iyy, izz, ixy=0.0, ixz=0.0, iyz=0.0):
    return np.array([[ixx, ixy, ixz], [ixy, iyy, iyz], [ixz, iyz, izz]], dtype=np.float64)


def box_mass(half, density):
    hx, hy, hz = [float(h) for h in half]
    m = density * 8 * hx * hy * hz
    return m, np.zeros(3), _sym(m / 3 * (hy * hy + hz * hz), m / 3 * (hx * hx + hz * hz), m / 3 * (hx * hx + hy * hy))


def sphere_mass(r, density):
    m = density * 4.0 / 3.0 * np.pi * r ** 3
    i = 0.4 * m * r * r
    return m, np.zeros(3), _sym(i, i, i)


def cylinder_mass(r, half_length, density):
    """axis = local x"""
    h = 2 * half_length
    m = density * np.pi * r * r * h
    return m, np.zeros(3), _sym(0.5 * m * r * r, m * (3 * r * r + h * h) / 12, m * (3 * r * r + h * h) / 12)


def capsule_mass(r, half_length, density):
    """axis = local x: cylinder of length 2*half_length + two hemispheres"""
    h = 2 * half_length
    mc = density * np.pi * r * r * h
    ms = density * 4.0 / 3.0 * np.pi * r ** 3
    ixx = 0.5 * mc * r * r + 0.4 * ms * r * r
    # hemisphere pair about a transverse axis through the capsule centre
    iyy = mc * (3 * r * r + h * h) / 12 + ms * (0.4 * r * r + 0.375 * r * h + 0.25 * h * h)
    return mc + ms, np.zeros(3), _sym(ixx, iyy, iyy)


def mesh_mass(verts, faces, density):
    """Closed triangle mesh (outward faces): signed tetrahedra against the origin."""
    v = np.asarray(verts, dtype=np.float64)
    t = v[np.asarray(faces, dtype=np.int64)]
    a, b, c = t[:, 0], t[:, 1], t[:, 2]
    vol6 = np.einsum("ij,ij->i", a, np.cross(b, c))
    vol = vol6.sum() / 6.0
    if abs(vol) < 1e-18:
        return 0.0, np.zeros(3), np.zeros((3, 3))
    com = ((a + b + c) * vol6[:, None]).sum(0) / (24.0 * vol)
    # second moments: integral of x_i x_j over each tetrahedron
    S = np.zeros((3, 3))
    for P, Q in ((a, a), (b, b), (c, c)):
        S += np.einsum("n,ni,nj->ij", vol6, P, Q) * 2
    for P, Q in ((a, b), (a, c), (b, c)):
        S += np.einsum("n,ni,nj->ij", vol6, P, Q) + np.einsum("n,ni,nj->ij", vol6, Q, P)
    S /= 120.0
    if vol < 0:
        vol, S = -vol, -S
    m = density * vol
    C = density * S - m * np.outer(com, com)          # covariance about the centre of mass
    I = np.trace(C) * np.eye(3) - C
    return m, com, I


def combine(parts):
    """parts: list of (mass, com_in_body [3], inertia_about_com_in_body_axes [3,3]) -> total (mass, com, inertia about com)."""
    M = sum(p[0] for p in parts)
    if M <= 0:
        return 0.0, np.zeros(3), np.zeros((3, 3))
    com = sum(p[0] * np.asarray(p[1]) for p in parts) / M
    I = np.zeros((3, 3))
    for m, c, Ic in parts:
        d = np.asarray(c) - com
        I += Ic + m * (np.dot(d, d) * np.eye(3) - np.outer(d, d))
    return M, com, I


def principal(I):
    """Symmetric inertia -> (principal moments [3], quaternion wxyz of the principal frame in the original axes)."""
    from ._pose import _mat2quat
    w, V = np.linalg.eigh(np.asarray(I, dtype=np.float64))
    if np.linalg.det(V) < 0:
        V[:, 2] = -V[:, 2]
    return w, _mat2quat(V)
