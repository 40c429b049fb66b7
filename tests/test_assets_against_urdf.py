"""The cooked robot models (maniskill_amd/assets/*.json, tools/cook_assets.py) against an independent reading of the reference's URDF /
SRDF files: link inertials, joint types / frames / axes / limits / mimic, box collision primitives, disabled collision pairs.
Runs where /root/reference exists (this container); skipped elsewhere -- the JSON files are what travels."""
import json
import os
import xml.etree.ElementTree as ET

import numpy as np
import pytest

REF = "/root/reference/mani_skill/assets/robots/panda"
HERE = os.path.dirname(os.path.abspath(__file__))
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="the reference tree is not present on this machine")


def _floats(s, n=3):
    v = [float(x) for x in s.split()] if s else [0.0] * n
    return np.array(v)


def _rpy_matrix(rpy):
    r, p, y = rpy
    Rx = np.array([[1, 0, 0], [0, np.cos(r), -np.sin(r)], [0, np.sin(r), np.cos(r)]])
    Ry = np.array([[np.cos(p), 0, np.sin(p)], [0, 1, 0], [-np.sin(p), 0, np.cos(p)]])
    Rz = np.array([[np.cos(y), -np.sin(y), 0], [np.sin(y), np.cos(y), 0], [0, 0, 1]])
    return Rz @ Ry @ Rx                     # URDF: fixed-axis roll, pitch, yaw


def _quat_matrix(q):
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


@pytest.mark.parametrize("name", ["panda_v2", "panda_v3", "panda_stick"])
def test_cooked_model_matches_the_urdf(name):
    model = json.load(open(os.path.join(HERE, "..", "maniskill_amd", "assets", f"{name}.json")))
    root = ET.parse(os.path.join(REF, f"{name}.urdf")).getroot()
    links = {l.get("name"): l for l in root.findall("link")}
    joints = {j.find("child").get("link"): j for j in root.findall("joint")}
    cooked = {l["name"]: l for l in model["links"]}
    assert set(cooked) == set(links)
    n_boxes = 0
    for lname, c in cooked.items():
        inert = links[lname].find("inertial")
        if inert is not None:
            assert abs(c["mass"] - float(inert.find("mass").get("value"))) < 1e-6 * max(1.0, c["mass"]), lname
            o = inert.find("origin")
            assert np.allclose(c["com"], _floats(o.get("xyz") if o is not None else ""), atol=1e-6), lname
            I = inert.find("inertia")
            want = [float(I.get(k)) for k in ("ixx", "iyy", "izz", "ixy", "ixz", "iyz")]
            # the cooked inertia is about the centre of mass in the frame given by com_q: rotate it back before comparing
            R = _quat_matrix(c["com_q"])
            xx, yy, zz, xy, xz, yz = c["inertia"]
            Ic = R @ np.array([[xx, xy, xz], [xy, yy, yz], [xz, yz, zz]]) @ R.T
            Ro = _rpy_matrix(_floats(o.get("rpy") if o is not None else ""))
            Iw = Ro @ np.array([[want[0], want[3], want[4]], [want[3], want[1], want[5]], [want[4], want[5], want[2]]]) @ Ro.T
            assert np.allclose(Ic, Iw, atol=1e-6 + 1e-5 * np.abs(Iw).max()), lname
        boxes = [col for col in links[lname].findall("collision") if col.find("geometry").find("box") is not None]
        mine = [col for col in c["collisions"] if col["type"] == "box"]
        assert len(boxes) == len(mine), lname
        for col, m in zip(boxes, mine):
            n_boxes += 1
            assert np.allclose(np.array(m["half_size"] if "half_size" in m else m["params"]) * 2, _floats(col.find("geometry").find("box").get("size")), atol=1e-6), lname
            o = col.find("origin")
            assert np.allclose(m["p"], _floats(o.get("xyz") if o is not None else ""), atol=1e-6), lname
        if lname not in joints:
            assert c["parent"] < 0 or c.get("joint") is None or c["parent"] == -1
            continue
        j, cj = joints[lname], c["joint"]
        assert cj["name"] == j.get("name") and cj["type"] == {"continuous": "revolute"}.get(j.get("type"), j.get("type")), lname
        assert model["links"][c["parent"]]["name"] == j.find("parent").get("link"), lname
        o = j.find("origin")
        assert np.allclose(cj["p"], _floats(o.get("xyz") if o is not None else ""), atol=1e-6), lname
        assert np.allclose(_quat_matrix(cj["q"]), _rpy_matrix(_floats(o.get("rpy") if o is not None else "")), atol=1e-6), lname
        if j.get("type") in ("revolute", "prismatic"):
            assert np.allclose(cj["axis"], _floats(j.find("axis").get("xyz")), atol=1e-9), lname
            lim = j.find("limit")
            assert np.allclose(cj["limit"], [float(lim.get("lower")), float(lim.get("upper"))], atol=1e-6), lname
        mimic = j.find("mimic")
        assert (cj["mimic"] is None) == (mimic is None), lname
        if mimic is not None:
            assert cj["mimic"]["joint"] == mimic.get("joint") and abs(cj["mimic"].get("multiplier", 1.0) - float(mimic.get("multiplier", 1.0))) < 1e-9
    srdf = ET.parse(os.path.join(REF, f"{name}.srdf")).getroot()
    pairs = {frozenset((d.get("link1"), d.get("link2"))) for d in srdf.findall("disable_collisions")}
    names = [l["name"] for l in model["links"]]
    mine = {frozenset((names[a], names[b]) if isinstance(a, int) else (a, b)) for a, b in model["disable_collisions"]}
    assert mine == pairs
    assert n_boxes > 0 or name == "panda_stick"


def _read_stl(path):
    import struct
    data = open(path, "rb").read()
    if data[:5] == b"solid" and b"facet" in data[:400]:
        return np.array([[float(x) for x in line.split()[1:4]] for line in data.decode(errors="ignore").splitlines() if line.strip().startswith("vertex")])
    n = struct.unpack("<I", data[80:84])[0]
    tri = np.frombuffer(data, dtype=np.dtype([("n", "<f4", 3), ("v", "<f4", (3, 3)), ("a", "<u2")]), count=n, offset=84)
    return tri["v"].reshape(-1, 3).astype(np.float64)


@pytest.mark.parametrize("name", ["panda_v2", "panda_stick"])
def test_cooked_hulls_stay_inside_and_close_to_the_collision_meshes(name):
    """Hull vertices are capped at 64 per shape (GPU PhysX cooks convex meshes to <= 64 vertices): the capped hull must be a subset of
    the mesh's vertices, lie inside its convex hull, and keep >= 93 % of its volume."""
    from scipy.spatial import ConvexHull, Delaunay

    model = json.load(open(os.path.join(HERE, "..", "maniskill_amd", "assets", f"{name}.json")))
    checked = 0
    for link in model["links"]:
        for col in link["collisions"]:
            if col["type"] != "convex" or "source" not in col or not col["source"].endswith(".stl"):
                continue
            path = os.path.join(REF, col["source"])
            if not os.path.exists(path):
                continue
            mesh = _read_stl(path) * np.asarray(col.get("scale", [1.0, 1.0, 1.0]))
            verts = np.asarray(col["verts"])
            assert len(verts) <= 64
            assert Delaunay(mesh[ConvexHull(mesh).vertices]).find_simplex(verts * 0.999 + mesh.mean(0) * 0.001).min() >= 0, (link["name"], "outside")
            ratio = ConvexHull(verts).volume / ConvexHull(mesh).volume
            assert 0.93 <= ratio <= 1.0 + 1e-6, (link["name"], ratio)
            checked += 1
    assert checked >= 1
