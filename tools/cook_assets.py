#!/usr/bin/env python3
"""Cook robot assets into the compact model files shipped under maniskill_amd/assets/.

The GPU box has no /root/reference, so the URDF/SRDF/STL inputs that ManiSkill keeps
under mani_skill/assets/robots are converted ONCE, here, into a small JSON model
(link tree, joint frames, inertials, collision primitives, convex hull vertices capped
at 64 per hull the way GPU PhysX cooking caps them [ext], SRDF disabled pairs).

Input data (read-only, only when present):
  mani_skill/assets/robots/panda/panda_v2.urdf            (reference asset, data)
  mani_skill/assets/robots/panda/panda_v2.srdf
  mani_skill/assets/robots/panda/franka_description/meshes/collision/*.stl

Usage: python tools/cook_assets.py [--ref /root/reference]
"""
import argparse
import json
import os
import struct
import xml.etree.ElementTree as ET

import numpy as np
from scipy.spatial import ConvexHull

MAX_HULL_VERTS = 64


def rpy_to_quat(r, p, y):
    cr, sr = np.cos(r / 2), np.sin(r / 2)
    cp, sp = np.cos(p / 2), np.sin(p / 2)
    cy, sy = np.cos(y / 2), np.sin(y / 2)
    return [
        cr * cp * cy + sr * sp * sy,
        sr * cp * cy - cr * sp * sy,
        cr * sp * cy + sr * cp * sy,
        cr * cp * sy - sr * sp * cy,
    ]


def read_stl(path):
    with open(path, "rb") as f:
        data = f.read()
    if data[:5] == b"solid" and b"facet" in data[:400]:
        verts = []
        for line in data.decode("ascii", "ignore").splitlines():
            t = line.split()
            if len(t) == 4 and t[0] == "vertex":
                verts.append([float(t[1]), float(t[2]), float(t[3])])
        return np.asarray(verts, dtype=np.float64)
    (ntri,) = struct.unpack_from("<I", data, 80)
    rec = np.dtype([("n", "<f4", 3), ("v", "<f4", (3, 3)), ("attr", "<u2")])
    tris = np.frombuffer(data, dtype=rec, count=ntri, offset=84)
    return tris["v"].reshape(-1, 3).astype(np.float64)


def reduce_hull(points, max_verts=MAX_HULL_VERTS):
    """Convex hull of `points`, greedily limited to `max_verts` vertices.

    Start from the axis-extreme points and repeatedly add the input vertex that lies
    farthest outside the current hull (progressive hull). Deterministic.
    """
    pts = np.unique(np.round(points, 7), axis=0)
    full = ConvexHull(pts)
    hv = pts[full.vertices]
    if len(hv) <= max_verts:
        return hv
    sel = set()
    for ax in range(3):
        sel.add(int(np.argmin(hv[:, ax])))
        sel.add(int(np.argmax(hv[:, ax])))
    sel = sorted(sel)
    # make sure the seed is full-dimensional
    k = 0
    while True:
        try:
            ConvexHull(hv[sel])
            break
        except Exception:
            if k not in sel:
                sel.append(k)
            k += 1
    while len(sel) < max_verts:
        h = ConvexHull(hv[sel])
        # signed distance of every vertex to every facet; outside if > 0
        d = hv @ h.equations[:, :3].T + h.equations[:, 3]
        out = d.max(axis=1)
        out[sel] = -1.0
        j = int(np.argmax(out))
        if out[j] < 1e-5:
            break
        sel.append(j)
    h = ConvexHull(hv[sel])
    return hv[sel][h.vertices]


def cook_urdf(urdf_path, srdf_path, mesh_root):
    root = ET.parse(urdf_path).getroot()
    links = {}
    for le in root.findall("link"):
        name = le.get("name")
        L = dict(name=name, mass=0.0, com=[0, 0, 0], com_q=[1, 0, 0, 0], inertia=[0] * 6, collisions=[])
        ine = le.find("inertial")
        if ine is not None:
            o = ine.find("origin")
            xyz = [float(x) for x in (o.get("xyz", "0 0 0") if o is not None else "0 0 0").split()]
            rpy = [float(x) for x in (o.get("rpy", "0 0 0") if o is not None else "0 0 0").split()]
            L["com"] = xyz
            L["com_q"] = rpy_to_quat(*rpy)
            L["mass"] = float(ine.find("mass").get("value"))
            I = ine.find("inertia")
            # order: ixx iyy izz ixy ixz iyz (tensor about the COM, in the inertial frame)
            L["inertia"] = [float(I.get(k)) for k in ("ixx", "iyy", "izz", "ixy", "ixz", "iyz")]
        for ce in le.findall("collision"):
            o = ce.find("origin")
            xyz = [float(x) for x in (o.get("xyz", "0 0 0") if o is not None else "0 0 0").split()]
            rpy = [float(x) for x in (o.get("rpy", "0 0 0") if o is not None else "0 0 0").split()]
            g = ce.find("geometry")
            C = dict(p=xyz, q=rpy_to_quat(*rpy))
            if g.find("box") is not None:
                size = [float(x) for x in g.find("box").get("size").split()]
                C.update(type="box", half_size=[s / 2 for s in size])
            elif g.find("mesh") is not None:
                fn = g.find("mesh").get("filename")
                verts = reduce_hull(read_stl(os.path.join(mesh_root, fn)))
                C.update(type="convex", source=fn, verts=np.round(verts, 6).tolist())
            elif g.find("cylinder") is not None:
                # cylinders are cooked as 16-sided prisms (PhysX has no cylinder primitive either: SAPIEN hands it a
                # convex mesh [ext]); URDF cylinders are along the local z axis
                r, clen = float(g.find("cylinder").get("radius")), float(g.find("cylinder").get("length"))
                ang = np.arange(16) * (2 * np.pi / 16)
                ring = np.stack([r * np.cos(ang), r * np.sin(ang)], axis=1)
                verts = np.concatenate([np.c_[ring, np.full(16, -clen / 2)], np.c_[ring, np.full(16, clen / 2)]])
                C.update(type="convex", source=f"cylinder r={r} l={clen}", verts=np.round(verts, 6).tolist())
            elif g.find("sphere") is not None:
                C.update(type="sphere", radius=float(g.find("sphere").get("radius")))
            else:
                raise NotImplementedError(ET.tostring(g))
            L["collisions"].append(C)
        links[name] = L
    joints = []
    child_of = {}
    for je in root.findall("joint"):
        o = je.find("origin")
        xyz = [float(x) for x in (o.get("xyz", "0 0 0") if o is not None else "0 0 0").split()]
        rpy = [float(x) for x in (o.get("rpy", "0 0 0") if o is not None else "0 0 0").split()]
        ax = je.find("axis")
        axis = [float(x) for x in ax.get("xyz").split()] if ax is not None else [1, 0, 0]
        lim = je.find("limit")
        dyn = je.find("dynamics")
        mim = je.find("mimic")
        J = dict(
            name=je.get("name"),
            type=je.get("type"),
            parent=je.find("parent").get("link"),
            child=je.find("child").get("link"),
            p=xyz,
            q=rpy_to_quat(*rpy),
            axis=axis,
            limit=[float(lim.get("lower", "-inf")), float(lim.get("upper", "inf"))] if lim is not None else None,
            effort=float(lim.get("effort", "0")) if lim is not None else 0.0,
            damping=float(dyn.get("damping", "0")) if dyn is not None else 0.0,
            friction=float(dyn.get("friction", "0")) if dyn is not None else 0.0,
            mimic=dict(joint=mim.get("joint"), multiplier=float(mim.get("multiplier", "1")), offset=float(mim.get("offset", "0")))
            if mim is not None
            else None,
        )
        joints.append(J)
        child_of[J["child"]] = J
    roots = [n for n in links if n not in child_of]
    assert len(roots) == 1
    # depth-first order following the joint order of the file (parents before children)
    order = []

    def visit(n):
        order.append(n)
        for J in joints:
            if J["parent"] == n:
                visit(J["child"])

    visit(roots[0])
    out_links = []
    for n in order:
        L = links[n]
        J = child_of.get(n)
        L["parent"] = order.index(J["parent"]) if J else -1
        L["joint"] = (
            {k: J[k] for k in ("name", "type", "p", "q", "axis", "limit", "effort", "damping", "friction", "mimic")}
            if J
            else dict(name="", type="fixed", p=[0, 0, 0], q=[1, 0, 0, 0], axis=[1, 0, 0], limit=None, effort=0, damping=0, friction=0, mimic=None)
        )
        out_links.append(L)
    disabled = []
    if srdf_path and os.path.exists(srdf_path):
        for de in ET.parse(srdf_path).getroot().findall("disable_collisions"):
            disabled.append([de.get("link1"), de.get("link2")])
    return dict(name=root.get("name"), links=out_links, disable_collisions=disabled)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    args = ap.parse_args()
    here = os.path.dirname(os.path.abspath(__file__))
    out_dir = os.path.join(here, "..", "maniskill_amd", "assets")
    os.makedirs(out_dir, exist_ok=True)
    base = os.path.join(args.ref, "mani_skill", "assets", "robots", "panda")
    model = cook_urdf(os.path.join(base, "panda_v2.urdf"), os.path.join(base, "panda_v2.srdf"), base)
    model["source"] = "mani_skill/assets/robots/panda/panda_v2.urdf (+ .srdf, collision STLs); cooked by tools/cook_assets.py"
    with open(os.path.join(out_dir, "panda_v2.json"), "w") as f:
        json.dump(model, f, separators=(",", ":"))
    nv = [len(c["verts"]) for L in model["links"] for c in L["collisions"] if c["type"] == "convex"]
    print("links", len(model["links"]), "hull verts", nv)
    stick = cook_urdf(os.path.join(base, "panda_stick.urdf"), os.path.join(base, "panda_stick.srdf"), base)
    stick["source"] = "mani_skill/assets/robots/panda/panda_stick.urdf (+ .srdf, collision STLs); cooked by tools/cook_assets.py"
    with open(os.path.join(out_dir, "panda_stick.json"), "w") as f:
        json.dump(stick, f, separators=(",", ":"))
    print("panda_stick links", len(stick["links"]), [L["name"] for L in stick["links"]])
    wrist = cook_urdf(os.path.join(base, "panda_v3.urdf"), os.path.join(base, "panda_v3.srdf"), base)
    wrist["source"] = "mani_skill/assets/robots/panda/panda_v3.urdf (+ .srdf, collision STLs); cooked by tools/cook_assets.py"
    with open(os.path.join(out_dir, "panda_v3.json"), "w") as f:
        json.dump(wrist, f, separators=(",", ":"))
    print("panda_v3 (wrist camera) links", len(wrist["links"]), [L["name"] for L in wrist["links"]][-4:])


if __name__ == "__main__":
    main()
