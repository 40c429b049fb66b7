"""Load a cooked robot model (maniskill_amd/assets/*.json, produced by tools/cook_assets.py
from the reference's URDF/SRDF/STL assets) into a SceneTemplate.

Mirrors what ``URDFLoader.parse`` + ``ArticulationBuilder.build`` do in the reference
(mani_skill/agents/base_agent.py:153-200, utils/building/articulation_builder.py:65-213):
joint frames use SAPIEN's convention (joint axis = +x of the joint frame), URDF ``mimic``
joints become fixed tendons with stiffness 1e5, SRDF ``disable_collisions`` pairs are
filtered, per-link materials / patch radii come from the agent's ``urdf_config``.
"""
from __future__ import annotations

import json
import os

import numpy as np

from .. import PACKAGE_ASSET_DIR
from .. import _native as N
from ..physx import SceneTemplate


def _qmul(a, b):
    aw, ax, ay, az = a
    bw, bx, by, bz = b
    return np.array([aw * bw - ax * bx - ay * by - az * bz, aw * bx + ax * bw + ay * bz - az * by,
                     aw * by - ax * bz + ay * bw + az * bx, aw * bz + ax * by - ay * bx + az * bw])


def _qmat(q):
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def _shortest_rotation_from_x(axis):
    """Quaternion rotating +x onto `axis` (sapien.math.shortest_rotation([1,0,0], axis))."""
    a = np.asarray(axis, dtype=np.float64)
    a = a / np.linalg.norm(a)
    d = a[0]
    if d > 1 - 1e-12:
        return np.array([1.0, 0, 0, 0])
    if d < -1 + 1e-12:
        return np.array([0.0, 0, 0, 1.0])
    c = np.cross([1.0, 0, 0], a)
    q = np.array([1 + d, c[0], c[1], c[2]])
    return q / np.linalg.norm(q)


def load_model(name: str) -> dict:
    with open(os.path.join(PACKAGE_ASSET_DIR, name)) as f:
        return json.load(f)


def add_urdf_articulation(tpl: SceneTemplate, model: dict, name: str, root_p=(0, 0, 0), root_q=(1, 0, 0, 0),
                          urdf_config: dict | None = None, default_material=(0.3, 0.3, 0.0),
                          disable_gravity: bool = True) -> int:
    """Returns the articulation index; link body ids are ``tpl.art_links[art]`` in link order."""
    urdf_config = urdf_config or {}
    mats = urdf_config.get("_materials", {})
    link_cfg = urdf_config.get("link", {})
    art = tpl.add_articulation(name, root_p, root_q)
    ids = []
    jmap = {"fixed": N.JOINT_FIXED, "revolute": N.JOINT_REVOLUTE, "continuous": N.JOINT_REVOLUTE,
            "prismatic": N.JOINT_PRISMATIC}
    for L in model["links"]:
        J = L["joint"]
        qa = _shortest_rotation_from_x(J["axis"])
        pin_parent = list(J["p"]) + list(_qmul(J["q"], qa))
        pin_child = [0, 0, 0] + list(qa)
        # inertia tensor about the COM expressed in link axes
        ixx, iyy, izz, ixy, ixz, iyz = L["inertia"]
        I = np.array([[ixx, ixy, ixz], [ixy, iyy, iyz], [ixz, iyz, izz]])
        Rc = _qmat(L["com_q"])
        I = Rc @ I @ Rc.T
        inertia6 = [I[0, 0], I[1, 1], I[2, 2], I[0, 1], I[0, 2], I[1, 2]]
        lim = J["limit"] if J["limit"] is not None else (-np.inf, np.inf)
        if J["type"] == "continuous":
            lim = (-np.inf, np.inf)
        parent = ids[L["parent"]] if L["parent"] >= 0 else -1
        bid = tpl.add_link(art, L["name"], parent, jmap[J["type"]], J["name"], pin_parent, pin_child, lim,
                           L["mass"], L["com"], inertia6, disable_gravity)
        ids.append(bid)
        cfg = link_cfg.get(L["name"], {})
        mat = mats.get(cfg.get("material"), None)
        sf, df, rest = (mat["static_friction"], mat["dynamic_friction"], mat["restitution"]) if mat else default_material
        for c in L["collisions"]:
            if c["type"] == "box":
                tpl.add_shape(bid, N.SHAPE_BOX, c["p"], c["q"], c["half_size"], None, sf, df, rest,
                              patch_radius=cfg.get("patch_radius", 0.0), min_patch_radius=cfg.get("min_patch_radius", 0.0))
            elif c["type"] == "convex":
                tpl.add_shape(bid, N.SHAPE_CONVEX, c["p"], c["q"], (0, 0, 0), np.asarray(c["verts"], dtype=np.float32),
                              sf, df, rest, patch_radius=cfg.get("patch_radius", 0.0),
                              min_patch_radius=cfg.get("min_patch_radius", 0.0))
            else:
                raise NotImplementedError(c["type"])
    names = [L["name"] for L in model["links"]]
    jnames = {L["joint"]["name"]: ids[i] for i, L in enumerate(model["links"])}
    # URDF mimic -> fixed tendon (articulation_builder.py:161-200): coefficients [0, -multiplier, 1]
    for i, L in enumerate(model["links"]):
        m = L["joint"].get("mimic")
        if m:
            tpl.add_tendon(jnames[m["joint"]], ids[i], -m["multiplier"], 1.0, m["offset"], 1e5, 0.0)
    for a, b in model.get("disable_collisions", []):
        if a in names and b in names:
            tpl.disable_collision(ids[names.index(a)], ids[names.index(b)])
    return art
