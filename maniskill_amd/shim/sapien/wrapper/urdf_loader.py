"""sapien.wrapper.urdf_loader.URDFLoader: URDF (+ SRDF) -> ArticulationBuilder, the base class of ManiSkill's URDFLoader
(mani_skill/utils/building/urdf_loader.py:23-123; configured through mani_skill/agents/base_agent.py:160-178 and
utils/sapien_utils.py:147-169: ``fix_root_link``, ``load_multiple_collisions_from_file``, ``set_link_material`` ...).

Conventions: the joint axis is +x of the joint frame (``pose_in_parent = origin * R(x -> axis)``, ``pose_in_child = R(x -> axis)``);
URDF cylinders / capsules lie along z, SAPIEN's along x; ``<mimic>`` becomes a MimicJointRecord; links appear depth-first in the
joint order of the file, parents before children.
"""
from __future__ import annotations

import os
import xml.etree.ElementTree as ET

import numpy as np

from .._pose import Pose, shortest_rotation
from .. import physx
from ..render import RenderMaterial
from .articulation_builder import ArticulationBuilder, MimicJointRecord


def _floats(s, default):
    return [float(x) for x in s.split()] if s else list(default)


def _origin(elem) -> Pose:
    o = elem.find("origin") if elem is not None else None
    if o is None:
        return Pose()
    xyz = _floats(o.get("xyz"), (0, 0, 0))
    r, p, y = _floats(o.get("rpy"), (0, 0, 0))
    cr, sr, cp, sp, cy, sy = np.cos(r / 2), np.sin(r / 2), np.cos(p / 2), np.sin(p / 2), np.cos(y / 2), np.sin(y / 2)
    q = [cr * cp * cy + sr * sp * sy, sr * cp * cy - cr * sp * sy, cr * sp * cy + sr * cp * sy, cr * cp * sy - sr * sp * cy]
    return Pose(xyz, q)


_Z_TO_X = Pose([0, 0, 0], [0.7071067811865476, 0, -0.7071067811865476, 0])    # rotates local x onto z


class URDFLoader:
    def __init__(self):
        self.scene = None
        self.fix_root_link = True
        self.load_multiple_collisions_from_file = False
        self.load_nonconvex_collisions_from_file = False
        self.multiple_collisions_decomposition = "none"
        self.multiple_collisions_decomposition_params = dict()
        self.collision_is_visual = False
        self.revolute_unwrapped = False
        self.scale = 1.0
        self.package_dir = None
        self._material = None
        self._patch_radius = 0.0
        self._min_patch_radius = 0.0
        self._density = 1000.0
        self._link_material = dict()
        self._link_patch_radius = dict()
        self._link_min_patch_radius = dict()
        self._link_density = dict()

    def set_scene(self, scene):
        self.scene = scene
        return self

    # -- per-link physical properties (sapien_utils.apply_urdf_config) ----------------------------------------------------
    def set_material(self, static_friction, dynamic_friction, restitution):
        self._material = physx.PhysxMaterial(static_friction, dynamic_friction, restitution)

    def set_link_material(self, link_name, static_friction, dynamic_friction, restitution):
        self._link_material[link_name] = physx.PhysxMaterial(static_friction, dynamic_friction, restitution)

    def set_patch_radius(self, r):
        self._patch_radius = r

    def set_link_patch_radius(self, link_name, r):
        self._link_patch_radius[link_name] = r

    def set_min_patch_radius(self, r):
        self._min_patch_radius = r

    def set_link_min_patch_radius(self, link_name, r):
        self._link_min_patch_radius[link_name] = r

    def set_density(self, d):
        self._density = d

    def set_link_density(self, link_name, d):
        self._link_density[link_name] = d

    # -- parsing ----------------------------------------------------------------------------------------------------------
    def _resolve(self, filename, urdf_dir):
        if filename.startswith("package://"):
            filename = filename[len("package://"):]
            base = self.package_dir if self.package_dir else urdf_dir
            path = os.path.join(base, filename)
            if not os.path.exists(path) and not self.package_dir:
                # ROS reading: "package://<name>/..." names a directory <name> that holds the files -- usually an ancestor of the URDF
                # (koch: robots/koch/follower_arm_v1.1.urdf refers to package://koch/meshes/*.stl)
                up = urdf_dir
                for _ in range(4):
                    up = os.path.dirname(up)
                    if os.path.exists(os.path.join(up, filename)):
                        return os.path.join(up, filename)
            return path
        if os.path.isabs(filename):
            return filename
        return os.path.join(urdf_dir, filename)

    def _create_builder(self):
        """The scene decides which ArticulationBuilder class is used (ManiSkillScene.create_articulation_builder)."""
        if self.scene is not None and hasattr(self.scene, "create_articulation_builder"):
            return self.scene.create_articulation_builder()
        return ArticulationBuilder().set_scene(self.scene)

    def _visual_material(self, vis, named):
        m = vis.find("material")
        if m is None:
            return None
        c = m.find("color")
        if c is None and m.get("name") in named:
            return named[m.get("name")]
        if c is None:
            return None
        return RenderMaterial(base_color=_floats(c.get("rgba"), (0.8, 0.8, 0.8, 1)))

    def parse(self, urdf_file, srdf_file=None, package_dir=None):
        if package_dir:
            self.package_dir = package_dir
        urdf_file = str(urdf_file)
        urdf_dir = os.path.dirname(os.path.abspath(urdf_file))
        root = ET.parse(urdf_file).getroot()
        if srdf_file is None:
            cand = urdf_file[:-4] + "srdf" if urdf_file.endswith(".urdf") else None
            srdf_file = cand if cand and os.path.exists(cand) else None
        named = {}
        for m in root.findall("material"):
            c = m.find("color")
            if c is not None:
                named[m.get("name")] = RenderMaterial(base_color=_floats(c.get("rgba"), (0.8, 0.8, 0.8, 1)))
        links = {le.get("name"): le for le in root.findall("link")}
        joints = root.findall("joint")
        child_joint = {je.find("child").get("link"): je for je in joints}
        roots = [n for n in links if n not in child_joint]
        if len(roots) != 1:
            raise RuntimeError(f"{urdf_file}: expected exactly one root link, found {roots}")
        order = []

        def visit(n):
            order.append(n)
            for je in joints:
                if je.find("parent").get("link") == n:
                    visit(je.find("child").get("link"))
        visit(roots[0])

        S = float(self.scale)
        builder = self._create_builder()
        lbs = {}
        for name in order:
            le, je = links[name], child_joint.get(name)
            lb = builder.create_link_builder(lbs[je.find("parent").get("link")] if je is not None else None)
            lbs[name] = lb
            lb.set_name(name)
            # ---- joint ----
            if je is None:
                lb.set_joint_name("")
                lb.set_joint_properties("fixed" if self.fix_root_link else "undefined", [])
            else:
                jt = je.get("type")
                ax = je.find("axis")
                axis = _floats(ax.get("xyz"), (1, 0, 0)) if ax is not None else [1.0, 0, 0]
                jo = _origin(je)
                jo = Pose(jo._p * S, jo._q)
                axis_pose = Pose([0, 0, 0], shortest_rotation([1, 0, 0], axis))
                lim = je.find("limit")
                dyn = je.find("dynamics")
                friction = float(dyn.get("friction", "0")) if dyn is not None else 0.0
                damping = float(dyn.get("damping", "0")) if dyn is not None else 0.0
                lb.set_joint_name(je.get("name"))
                if jt in ("revolute", "continuous"):
                    if jt == "continuous" or lim is None:
                        lo, hi = -np.inf, np.inf
                        typ = "revolute_unwrapped"
                    else:
                        lo, hi = float(lim.get("lower", "0")), float(lim.get("upper", "0"))
                        typ = "revolute_unwrapped" if self.revolute_unwrapped else "revolute"
                    lb.set_joint_properties(typ, [[lo, hi]], jo * axis_pose, axis_pose, friction, damping)
                elif jt == "prismatic":
                    lo, hi = (float(lim.get("lower", "0")) * S, float(lim.get("upper", "0")) * S) if lim is not None else (-np.inf, np.inf)
                    lb.set_joint_properties("prismatic", [[lo, hi]], jo * axis_pose, axis_pose, friction, damping)
                elif jt == "fixed":
                    lb.set_joint_properties("fixed", [], jo, Pose(), friction, damping)
                else:
                    raise RuntimeError(f"{urdf_file}: joint type {jt!r} of {je.get('name')!r} is not supported")
                mim = je.find("mimic")
                if mim is not None:
                    # q(this joint) = multiplier * q(mimicked joint) + offset.  ManiSkill turns a record into the tendon
                    # -multiplier * q(record.joint) + q(record.mimic) = offset (articulation_builder.py:161-200), so record.joint is
                    # the mimicked joint and record.mimic the one carrying the <mimic> tag
                    builder.mimic_joint_records.append(MimicJointRecord(mim.get("joint"), je.get("name"), float(mim.get("multiplier", "1")),
                                                                        float(mim.get("offset", "0"))))
            # ---- inertial ----
            ine = le.find("inertial")
            mass = float(ine.find("mass").get("value")) if ine is not None and ine.find("mass") is not None else 0.0
            if ine is not None and mass > 0:
                io = _origin(ine)
                I = ine.find("inertia")
                ixx, iyy, izz, ixy, ixz, iyz = [float(I.get(k, "0")) for k in ("ixx", "iyy", "izz", "ixy", "ixz", "iyz")] if I is not None else [0] * 6
                T = np.array([[ixx, ixy, ixz], [ixy, iyy, iyz], [ixz, iyz, izz]], dtype=np.float64)
                from .._mesh import principal
                w, q = principal(T)
                lb.set_mass_and_inertia(mass * S ** 3, Pose(io._p * S, io._q) * Pose([0, 0, 0], q), np.maximum(w, 0) * S ** 5)
                R = io._matrix64()[:3, :3]
                lb._exact_inertial = (mass * S ** 3, io._p * S, (R @ T @ R.T) * S ** 5)
            # ---- collisions ----
            mat = self._link_material.get(name, self._material)
            density = self._link_density.get(name, self._density)
            pr = self._link_patch_radius.get(name, self._patch_radius)
            mpr = self._link_min_patch_radius.get(name, self._min_patch_radius)
            kw = dict(material=mat, density=density, patch_radius=pr, min_patch_radius=mpr)
            for ce in le.findall("collision"):
                g = ce.find("geometry")
                co = _origin(ce)
                co = Pose(co._p * S, co._q)
                if g.find("box") is not None:
                    size = _floats(g.find("box").get("size"), (1, 1, 1))
                    lb.add_box_collision(co, [s * S / 2 for s in size], **kw)
                elif g.find("sphere") is not None:
                    lb.add_sphere_collision(co, float(g.find("sphere").get("radius")) * S, **kw)
                elif g.find("cylinder") is not None:
                    c = g.find("cylinder")
                    lb.add_cylinder_collision(co * _Z_TO_X, float(c.get("radius")) * S, float(c.get("length")) * S / 2, **kw)
                elif g.find("capsule") is not None:
                    c = g.find("capsule")
                    lb.add_capsule_collision(co * _Z_TO_X, float(c.get("radius")) * S, float(c.get("length")) * S / 2, **kw)
                elif g.find("mesh") is not None:
                    me = g.find("mesh")
                    fn = self._resolve(me.get("filename"), urdf_dir)
                    sc = np.asarray(_floats(me.get("scale"), (1, 1, 1))) * S
                    if self.load_multiple_collisions_from_file:
                        lb.add_multiple_convex_collisions_from_file(fn, co, sc, decomposition=self.multiple_collisions_decomposition,
                                                                    decomposition_params=self.multiple_collisions_decomposition_params, **kw)
                    elif self.load_nonconvex_collisions_from_file:
                        lb.add_nonconvex_collision_from_file(fn, co, sc, material=mat, patch_radius=pr, min_patch_radius=mpr)
                    else:
                        lb.add_convex_collision_from_file(fn, co, sc, **kw)
            # ---- visuals ----
            for ve in le.findall("visual"):
                g = ve.find("geometry")
                vo = _origin(ve)
                vo = Pose(vo._p * S, vo._q)
                vm = self._visual_material(ve, named)
                vname = ve.get("name", "")
                if g is None:
                    continue
                if g.find("box") is not None:
                    lb.add_box_visual(vo, [s * S / 2 for s in _floats(g.find("box").get("size"), (1, 1, 1))], vm, vname)
                elif g.find("sphere") is not None:
                    lb.add_sphere_visual(vo, float(g.find("sphere").get("radius")) * S, vm, vname)
                elif g.find("cylinder") is not None:
                    c = g.find("cylinder")
                    lb.add_cylinder_visual(vo * _Z_TO_X, float(c.get("radius")) * S, float(c.get("length")) * S / 2, vm, vname)
                elif g.find("capsule") is not None:
                    c = g.find("capsule")
                    lb.add_capsule_visual(vo * _Z_TO_X, float(c.get("radius")) * S, float(c.get("length")) * S / 2, vm, vname)
                elif g.find("mesh") is not None:
                    me = g.find("mesh")
                    lb.add_visual_from_file(self._resolve(me.get("filename"), urdf_dir), vo, np.asarray(_floats(me.get("scale"), (1, 1, 1))) * S,
                                            vm, vname)
        # ---- SRDF: pairs of links that never collide ----
        if srdf_file is not None and os.path.exists(srdf_file):
            for de in ET.parse(srdf_file).getroot().findall("disable_collisions"):
                a, b = de.get("link1"), de.get("link2")
                if a in lbs and b in lbs:
                    lbs[a]._srdf_disabled = getattr(lbs[a], "_srdf_disabled", set()) | {b}
        return [builder], [], []

    def load_file_as_articulation_builder(self, urdf_file, srdf_file=None, package_dir=None):
        arts, actors, cams = URDFLoader.parse(self, urdf_file, srdf_file, package_dir)
        assert len(arts) == 1 and not actors
        return arts[0]

    def load(self, urdf_file, srdf_file=None, package_dir=None):
        builder = self.load_file_as_articulation_builder(urdf_file, srdf_file, package_dir)
        return builder.build(fix_root_link=self.fix_root_link)
