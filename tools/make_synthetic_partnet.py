#!/usr/bin/env python3
"""Synthetic stand-in for the PartNet-Mobility cabinets that OpenCabinetDrawer-v1 loads (BASELINE.json config 5).

The reference reads ``$MS_ASSET_DIR/data/partnet_mobility/dataset/<id>/mobility_cvx.urdf`` for the 25 model ids of
``mani_skill/assets/partnet_mobility/meta/info_cabinet_drawer_train.json`` (utils/building/articulations/partnet_mobility.py:9-39);
the dataset is a download and absent here.  SURVEY.md §8(d) prescribes the substitute this script writes: per model id (seeded by
the id) a cabinet with k in {1..max_drawers} prismatic drawers, every link 3-8 convex hulls of 16-64 vertices, in the dataset's file
layout (URDF + OBJ meshes, one ``o`` group per hull, a visual named "handle" on every drawer), so that the reference's own task
code loads it unmodified.  Heterogeneous on purpose: link counts, hull counts, vertex counts, sizes and drawer travel differ per id.

    python tools/make_synthetic_partnet.py --out /tmp/ms_assets [--max-drawers 4] [--ids-from <meta json>]
    MS_ASSET_DIR=/tmp/ms_assets python ... gym.make("OpenCabinetDrawer-v1", num_envs=...)
"""
import argparse
import json
import os

import numpy as np

DEFAULT_IDS = [1000, 1004, 1005, 1013, 1016, 1021, 1024, 1027, 1032, 1033, 1035, 1038, 1040, 1044, 1045, 1052, 1054, 1056, 1061,
               1063, 1066, 1067, 1076, 1079, 1082]


def _hull_cloud(rng, half, nverts):
    """Vertices of a slightly rounded slab: the 8 corners of the box `half` plus points on an inscribed ellipsoid-ish shell."""
    half = np.asarray(half, dtype=np.float64)
    corners = np.array([[x, y, z] for x in (-1, 1) for y in (-1, 1) for z in (-1, 1)], dtype=np.float64) * half
    extra = rng.normal(size=(max(nverts - 8, 0), 3))
    extra /= np.linalg.norm(extra, axis=1, keepdims=True)
    extra = extra * half * rng.uniform(1.0, 1.12, size=(len(extra), 1))     # bulges a little beyond the faces: all are hull vertices
    extra = np.clip(extra, -half * 1.12, half * 1.12)
    return np.concatenate([corners, extra])


def _write_obj(path, groups):
    """groups: list of (name, vertices); every group is written as the triangulated convex hull of its vertices."""
    from scipy.spatial import ConvexHull
    with open(path, "w") as f:
        base = 1
        for name, v in groups:
            h = ConvexHull(v)
            used = h.vertices
            remap = {int(u): i for i, u in enumerate(used)}
            c = v[used].mean(0)
            f.write(f"o {name}\n")
            for p in v[used]:
                f.write(f"v {p[0]:.6f} {p[1]:.6f} {p[2]:.6f}\n")
            for tri in h.simplices:
                a, b, cc = (v[tri[0]], v[tri[1]], v[tri[2]])
                if np.dot(np.cross(b - a, cc - a), a - c) < 0:
                    tri = tri[::-1]
                f.write("f %d %d %d\n" % tuple(base + remap[int(t)] for t in tri))
            base += len(used)


def make_cabinet(out_dir, model_id, max_drawers=4):
    rng = np.random.default_rng(int(model_id))
    os.makedirs(out_dir, exist_ok=True)
    k = int(rng.integers(1, max_drawers + 1))
    # dataset units: the loader multiplies by the model's "scale" (~0.5-0.65); the carcass is ~1.2-1.6 units wide / tall
    W, D, H = rng.uniform(1.1, 1.5), rng.uniform(0.9, 1.2), rng.uniform(0.45, 0.6) * k + 0.25
    t = 0.05
    def slab(center, half):
        return np.asarray(center) + _hull_cloud(rng, half, int(rng.integers(16, 65)))
    body = [("back", slab([D / 2 - t / 2, 0, 0], [t / 2, W / 2, H / 2])),
            ("left", slab([0, -W / 2 + t / 2, 0], [D / 2, t / 2, H / 2])),
            ("right", slab([0, W / 2 - t / 2, 0], [D / 2, t / 2, H / 2])),
            ("top", slab([0, 0, H / 2 - t / 2], [D / 2, W / 2, t / 2])),
            ("bottom", slab([0, 0, -H / 2 + t / 2], [D / 2, W / 2, t / 2]))]
    cell = (H - 2 * t) / k
    for j in range(k - 1):                                          # shelves between the drawers: 5 + (k - 1) <= 8 hulls
        body.append((f"shelf{j}", slab([0, 0, -H / 2 + t + cell * (j + 1)], [D / 2, W / 2 - t, t / 4])))
    _write_obj(os.path.join(out_dir, "body_cvx.obj"), body)
    links = ['  <link name="base">\n    <visual name="body"><origin xyz="0 0 0"/><geometry><mesh filename="body_cvx.obj"/></geometry></visual>\n'
             '    <collision><origin xyz="0 0 0"/><geometry><mesh filename="body_cvx.obj"/></geometry></collision>\n  </link>\n']
    joints = []
    for j in range(k):
        zc = -H / 2 + t + cell * (j + 0.5)
        dw, dd, dh = W / 2 - 1.5 * t, D / 2 - t, cell / 2 - 0.03
        tw = 0.03
        parts = [("floor", slab([0, 0, -dh + tw / 2], [dd, dw, tw / 2])),
                 ("front", slab([-dd + tw / 2, 0, 0], [tw / 2, dw, dh])),
                 ("rear", slab([dd - tw / 2, 0, 0], [tw / 2, dw, dh * 0.8])),
                 ("wall_l", slab([0, -dw + tw / 2, 0], [dd, tw / 2, dh * 0.8])),
                 ("wall_r", slab([0, dw - tw / 2, 0], [dd, tw / 2, dh * 0.8]))]
        n_extra = int(rng.integers(0, 3))                           # 5-7 hulls + the handle: 6-8 per drawer
        for e in range(n_extra):
            parts.append((f"divider{e}", slab([0, rng.uniform(-dw / 2, dw / 2), -dh / 3], [dd * 0.9, tw / 3, dh / 2])))
        handle = ("handle", slab([-dd - 0.06, 0, 0], [0.03, rng.uniform(0.12, 0.25), 0.025]))
        _write_obj(os.path.join(out_dir, f"drawer{j}_cvx.obj"), parts + [handle])
        _write_obj(os.path.join(out_dir, f"drawer{j}_handle.obj"), [handle])
        _write_obj(os.path.join(out_dir, f"drawer{j}_shell.obj"), parts)
        links.append(f'  <link name="link_{j}">\n'
                     f'    <visual name="drawer-{j}"><origin xyz="0 0 0"/><geometry><mesh filename="drawer{j}_shell.obj"/></geometry></visual>\n'
                     f'    <visual name="handle-{j}"><origin xyz="0 0 0"/><geometry><mesh filename="drawer{j}_handle.obj"/></geometry></visual>\n'
                     f'    <collision><origin xyz="0 0 0"/><geometry><mesh filename="drawer{j}_cvx.obj"/></geometry></collision>\n'
                     f'    <inertial><origin xyz="0 0 0"/><mass value="{rng.uniform(3.0, 8.0):.3f}"/>'
                     f'<inertia ixx="0.2" iyy="0.2" izz="0.3" ixy="0" ixz="0" iyz="0"/></inertial>\n  </link>\n')
        travel = rng.uniform(0.5, 0.8) * D
        joints.append(f'  <joint name="joint_{j}" type="prismatic">\n    <origin xyz="0 0 {zc:.5f}"/>\n    <axis xyz="-1 0 0"/>\n'
                      f'    <parent link="base"/>\n    <child link="link_{j}"/>\n    <limit lower="0" upper="{travel:.4f}" effort="100" velocity="1"/>\n'
                      f'    <dynamics damping="{rng.uniform(2.0, 8.0):.2f}" friction="0"/>\n  </joint>\n')
    with open(os.path.join(out_dir, "mobility_cvx.urdf"), "w") as f:
        f.write(f'<?xml version="1.0"?>\n<robot name="synthetic_cabinet_{model_id}">\n' + "".join(links) + "".join(joints) + "</robot>\n")
    return k


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True, help="asset root (what MS_ASSET_DIR points to)")
    ap.add_argument("--max-drawers", type=int, default=4)
    ap.add_argument("--ids-from", default=None, help="info_cabinet_drawer_train.json of a ManiSkill checkout (default: the 25 train ids)")
    ap.add_argument("--placeholder-ids-from", nargs="*", default=[],
                    help="meta files whose model ids only need to EXIST (the task's asset check covers the whole partnet_mobility_cabinet "
                         "group, utils/registration.py:42-76, but OpenCabinetDrawer-v1 loads the drawer models only)")
    a = ap.parse_args()
    ids = DEFAULT_IDS
    if a.ids_from:
        with open(a.ids_from) as f:
            ids = [int(k) for k in json.load(f).keys()]
    summary = {}
    for i in ids:
        summary[i] = make_cabinet(os.path.join(a.out, "data", "partnet_mobility", "dataset", str(i)), i, a.max_drawers)
    for fn in a.placeholder_ids_from:
        with open(fn) as f:
            for k in json.load(f).keys():
                os.makedirs(os.path.join(a.out, "data", "partnet_mobility", "dataset", str(k)), exist_ok=True)
    print(json.dumps({"models": len(ids), "drawers": summary}))


if __name__ == "__main__":
    main()
