#!/usr/bin/env python3
"""How far are the pictures this backend draws from pictures of the reference's own visual geometry?

The rasteriser's scene template is small (include/msk_render.h: 8192 triangles, 4096 vertices), so the shim welds and simplifies the reference's visual meshes
when it compiles a scene (shim/sapien/render.py `_compile`: the Panda's visual GLBs are 134 300 triangles).  The parity tests compare the HIP
rasteriser with `oracle/orc_render.c` on the SAME simplified template; this tool measures the simplification itself, without a GPU:

  * picture A -- what ships: the env's own `depth+segmentation` observation, drawn by the oracle's rasteriser from the compiled template (bit-equal to the HIP
    rasteriser: tests/test_render.py, tests/test_gpu_parity.py);
  * picture B -- ground truth: every triangle of every render shape the reference attached (`RenderShapeTriangleMesh` parts as loaded from the GLB / box /
    cylinder primitives, UN-simplified), placed by the bodies' current poses, drawn with the same pinhole camera by a float64 z-buffer at the pixel centres.

Printed per task: pixel agreement of the segmentation ids, intersection-over-union per id (with the entity's name), depth error percentiles over the pixels whose ids
agree (the pictures' unit: millimetres).  profiles/r06_camera_fidelity.log is this script's output.

    python tools/camera_fidelity.py [PushT-v1 PickCube-v1 ...] [--steps 5] [--frames 3]
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def quat_to_mat(q):
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]], dtype=np.float64)


def rasterise(tris_cam, ids, K, w, h, near, far):
    """tris_cam: [T, 3, 3] vertices in the OpenCV camera frame (x right, y down, z forward); -> (seg [h, w] int32, depth [h, w] float64, 0 = background).
    Nearest surface at each pixel centre; a triangle crossing the near plane is clipped by dropping it when any vertex is behind it (none of the scenes here has geometry
    through the camera)."""
    z = tris_cam[:, :, 2]
    ok = (z > near).all(axis=1)
    tris_cam, ids = tris_cam[ok], ids[ok]
    z = tris_cam[:, :, 2]
    px = K[0, 0] * tris_cam[:, :, 0] / z + K[0, 2]
    py = K[1, 1] * tris_cam[:, :, 1] / z + K[1, 2]
    seg = np.zeros((h, w), dtype=np.int32)
    zbuf = np.full((h, w), np.inf)
    x0 = np.clip(np.floor(px.min(axis=1) - 0.5).astype(np.int64) + 0, 0, w - 1)
    x1 = np.clip(np.ceil(px.max(axis=1) - 0.5).astype(np.int64), 0, w - 1)
    y0 = np.clip(np.floor(py.min(axis=1) - 0.5).astype(np.int64) + 0, 0, h - 1)
    y1 = np.clip(np.ceil(py.max(axis=1) - 0.5).astype(np.int64), 0, h - 1)
    vis = (px.max(axis=1) >= 0) & (px.min(axis=1) <= w) & (py.max(axis=1) >= 0) & (py.min(axis=1) <= h)
    # edge functions and the plane of 1/z (affine on the screen)
    ax, ay, bx, by, cx, cy = px[:, 0], py[:, 0], px[:, 1], py[:, 1], px[:, 2], py[:, 2]
    area = (bx - ax) * (cy - ay) - (by - ay) * (cx - ax)
    vis &= np.abs(area) > 1e-14
    iz = 1.0 / z
    order = np.nonzero(vis)[0]
    spans = (x1 - x0 + 1) * (y1 - y0 + 1)
    small = order[spans[order] <= 9]
    large = order[spans[order] > 9]
    # small triangles: up to 3 x 3 pixel centres each, vectorised over the triangles
    for dy in range(3):
        for dx in range(3):
            t = small
            xs, ys = x0[t] + dx, y0[t] + dy
            m = (xs <= x1[t]) & (ys <= y1[t])
            t, xs, ys = t[m], xs[m], ys[m]
            cxp, cyp = xs + 0.5, ys + 0.5
            w0 = ((bx[t] - cxp) * (cy[t] - cyp) - (by[t] - cyp) * (cx[t] - cxp)) / area[t]
            w1 = ((cx[t] - cxp) * (ay[t] - cyp) - (cy[t] - cyp) * (ax[t] - cxp)) / area[t]
            w2 = 1.0 - w0 - w1
            inside = (w0 >= 0) & (w1 >= 0) & (w2 >= 0)
            t, xs, ys = t[inside], xs[inside], ys[inside]
            zz = 1.0 / (w0[inside] * iz[t, 0] + w1[inside] * iz[t, 1] + w2[inside] * iz[t, 2])
            # nearest wins: sort far-to-near so that the last write per pixel is the nearest
            o = np.argsort(-zz)
            t, xs, ys, zz = t[o], xs[o], ys[o], zz[o]
            better = zz < zbuf[ys, xs]
            t, xs, ys, zz = t[better], xs[better], ys[better], zz[better]
            zbuf[ys, xs] = zz          # (duplicates: the last = nearest of this batch)
            seg[ys, xs] = ids[t]
            # a batch may hold several candidates for one pixel; the far-to-near order makes the survivor the nearest
    for t in large:
        xs = np.arange(x0[t], x1[t] + 1) + 0.5
        ys = np.arange(y0[t], y1[t] + 1) + 0.5
        gx, gy = np.meshgrid(xs, ys)
        w0 = ((bx[t] - gx) * (cy[t] - gy) - (by[t] - gy) * (cx[t] - gx)) / area[t]
        w1 = ((cx[t] - gx) * (ay[t] - gy) - (cy[t] - gy) * (ax[t] - gx)) / area[t]
        w2 = 1.0 - w0 - w1
        inside = (w0 >= 0) & (w1 >= 0) & (w2 >= 0)
        zz = 1.0 / (w0 * iz[t, 0] + w1 * iz[t, 1] + w2 * iz[t, 2])
        sub = zbuf[y0[t]:y1[t] + 1, x0[t]:x1[t] + 1]
        ssub = seg[y0[t]:y1[t] + 1, x0[t]:x1[t] + 1]
        win = inside & (zz < sub) & (zz < far)
        sub[win] = zz[win]
        ssub[win] = ids[t]
    depth = np.where(np.isfinite(zbuf) & (zbuf < far), zbuf, 0.0)
    seg = np.where(depth > 0, seg, 0)
    return seg, depth


def scene_triangles(base):
    """every triangle the reference attached to sub-scene 0, in the world frame, with the entity's per_scene_id"""
    from sapien import Pose
    scene = base.scene
    px = scene.px
    group = scene.render_system_group
    if group is None:      # (set up lazily, with the first picture)
        scene.update_render(update_sensors=True, update_human_render_cameras=False)
        group = scene.render_system_group
    rs0 = group.systems[0]
    rows = px.cuda_rigid_body_data.torch().detach().cpu().numpy()
    tris, ids, names, counts = [], [], {}, {}
    for rb in rs0.render_bodies:
        if rb.visibility <= 0:
            continue
        ent = rb.entity
        pb = ent._physx_body() if ent is not None else None
        if pb is not None and pb._body_id >= 0:
            r = rows[px._pose_index(pb)]
            bp = Pose(r[:3], r[3:7])
        else:
            bp = ent._pose if ent is not None else Pose()
        seg = int(ent.per_scene_id)
        names[seg] = ent.name
        for shape in rb.render_shapes:
            wp = bp * shape.local_pose
            R, t = quat_to_mat(np.asarray(wp._q, dtype=np.float64)), np.asarray(wp._p, dtype=np.float64)
            for v, f, _ in shape._triangles():
                v = np.asarray(v, dtype=np.float64) @ R.T + t
                f = np.asarray(f, dtype=np.int64).reshape(-1, 3)
                tris.append(v[f])
                ids.append(np.full(len(f), seg, dtype=np.int32))
                counts[seg] = counts.get(seg, 0) + len(f)
    return np.concatenate(tris), np.concatenate(ids), names, counts


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("envs", nargs="*", default=["PushT-v1", "PickCube-v1"])
    ap.add_argument("--steps", type=int, default=5, help="random-action control steps between frames")
    ap.add_argument("--frames", type=int, default=3)
    a = ap.parse_args()
    import ref_harness
    gym = ref_harness.setup("oracle")
    if gym is None:
        raise SystemExit("needs the reference checkout (or oracle/_ref/maniskill)")
    import torch
    for eid in a.envs:
        env = gym.make(eid, num_envs=2, obs_mode="depth+segmentation")      # (two sub-scenes: one would take the reference to its CPU-simulation path)
        base = env.unwrapped
        obs, _ = env.reset(seed=0)
        torch.manual_seed(0)
        agree_all, depth_err, inter, union = [], [], {}, {}
        for frame in range(a.frames):
            for _ in range(a.steps if frame else 0):
                obs, *_ = env.step(torch.as_tensor(env.action_space.sample()))
            name, sensor = next(iter(base.scene.sensors.items()))
            pic = obs["sensor_data"][name]
            segA = pic["segmentation"][0, ..., 0].cpu().numpy().astype(np.int32)
            depA = pic["depth"][0, ..., 0].cpu().numpy().astype(np.float64)      # millimetres
            h, w = segA.shape
            params = obs["sensor_param"][name]
            E = params["extrinsic_cv"][0].cpu().numpy().astype(np.float64)       # [3, 4] world -> OpenCV camera
            K = params["intrinsic_cv"][0].cpu().numpy().astype(np.float64)
            tris, ids, names, counts = scene_triangles(base)
            cam = tris @ E[:, :3].T + E[:, 3]
            cfg = sensor.config
            segB, depB = rasterise(cam, ids, K, w, h, float(cfg.near), float(cfg.far))
            depB = np.round(depB * 1000.0)
            same = segA == segB
            agree_all.append(same.mean())
            both = same & (segA > 0)
            depth_err.append(np.abs(depA[both] - depB[both]))
            for s in np.union1d(np.unique(segA), np.unique(segB)):
                inter[s] = inter.get(s, 0) + int(((segA == s) & (segB == s)).sum())
                union[s] = union.get(s, 0) + int(((segA == s) | (segB == s)).sum())
        de = np.concatenate(depth_err)
        sm = base.scene.render_system_group.simplification
        print(f"== {eid}: {a.frames} frames of {w} x {h}; reference geometry {sum(counts.values())} triangles; template: {sm['tris_before']} -> {sm['tris_after']} triangles in "
              f"{sm['parts']} simplified parts (largest surface displacement {sm['max_surface_error'] * 1e3:.1f} mm)")
        print(f"   segmentation ids agree on {100 * np.mean(agree_all):.2f} % of the pixels; depth where they agree: median |error| {np.median(de):.1f} mm, 90 % {np.percentile(de, 90):.1f} mm, "
              f"99 % {np.percentile(de, 99):.1f} mm, max {de.max():.0f} mm")
        for s in sorted(union):
            if union[s] == 0:
                continue
            print(f"   id {s:3d} {names.get(s, 'background') if s else 'background':28s} IoU {inter[s] / union[s]:.3f}  ({union[s] // a.frames} pixels per frame, {counts.get(s, 0)} reference triangles)")
        env.close()


if __name__ == "__main__":
    main()
