"""sapien.wrapper.pinocchio_model.PinocchioModel: CPU kinematics helper of SAPIEN, built from a URDF string.  ManiSkill uses it for the
end-effector control modes of the CPU simulation backend (agents/controllers/utils/kinematics.py:96-140, 260-274:
``compute_inverse_kinematics(link, pose, initial_qpos, active_qmask, max_iterations)``) and when converting recorded trajectories.  Here:
forward kinematics and the geometric Jacobian of the serial chain root -> link (the URDF reader of the pytorch_kinematics stand-in), and
damped least squares iterated to convergence, as SAPIEN's closed-loop IK does -- float64 numpy, one configuration at a time."""
import numpy as np
import xml.etree.ElementTree as ET


def _rpy(r, p, y):
    cr, sr, cp, sp, cy, sy = np.cos(r), np.sin(r), np.cos(p), np.sin(p), np.cos(y), np.sin(y)
    return np.array([[cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr],
                     [sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr],
                     [-sp, cp * sr, cp * cr]])


def _rot(axis, q):
    K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    return np.eye(3) + np.sin(q) * K + (1 - np.cos(q)) * (K @ K)


def _quat_to_mat(q):
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def _mat_to_quat(R):
    t = np.trace(R)
    if t > 0:
        s = np.sqrt(t + 1.0) * 2
        q = np.array([0.25 * s, (R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s])
    else:
        i = int(np.argmax(np.diag(R)))
        j, k = (i + 1) % 3, (i + 2) % 3
        s = np.sqrt(1.0 + R[i, i] - R[j, j] - R[k, k]) * 2
        q = np.zeros(4)
        q[0] = (R[k, j] - R[j, k]) / s
        q[1 + i] = 0.25 * s
        q[1 + j] = (R[j, i] + R[i, j]) / s
        q[1 + k] = (R[k, i] + R[i, k]) / s
    return q / np.linalg.norm(q)


def _rotvec(R):
    """rotation vector of a rotation matrix"""
    q = _mat_to_quat(R)
    if q[0] < 0:
        q = -q
    n = np.linalg.norm(q[1:])
    if n < 1e-12:
        return np.zeros(3)
    return q[1:] / n * (2.0 * np.arctan2(n, q[0]))


class PinocchioModel:
    def __init__(self, urdf_string, gravity=(0, 0, -9.81)):
        data = urdf_string.encode("utf-8") if isinstance(urdf_string, str) else bytes(urdf_string)
        root = ET.fromstring(data)
        self._joint_of_child = {}
        self._links = [le.get("name") for le in root.findall("link")]
        self._movable = []
        for je in root.findall("joint"):
            o = je.find("origin")
            xyz = [float(x) for x in (o.get("xyz", "0 0 0") if o is not None else "0 0 0").split()]
            rpy = [float(x) for x in (o.get("rpy", "0 0 0") if o is not None else "0 0 0").split()]
            M = np.eye(4)
            M[:3, :3] = _rpy(*rpy)
            M[:3, 3] = xyz
            ax = je.find("axis")
            axis = np.array([float(x) for x in ax.get("xyz").split()]) if ax is not None else np.array([1.0, 0.0, 0.0])
            n = np.linalg.norm(axis)
            axis = axis / n if n > 0 else axis
            jt = je.get("type")
            lim = je.find("limit")
            lo = float(lim.get("lower", "-inf")) if lim is not None else -np.inf
            hi = float(lim.get("upper", "inf")) if lim is not None else np.inf
            if jt == "continuous":
                jt, lo, hi = "revolute", -np.inf, np.inf
            if jt not in ("revolute", "prismatic"):
                jt = "fixed"
            rec = dict(name=je.get("name"), type=jt, origin=M, axis=axis, lower=lo, upper=hi, parent=je.find("parent").get("link"),
                       child=je.find("child").get("link"))
            self._joint_of_child[rec["child"]] = rec
            if jt != "fixed":
                self._movable.append(rec["name"])
        self._joint_order = list(self._movable)
        self._link_order = list(self._links)
        self._q = np.zeros(len(self._joint_order))

    @classmethod
    def _from_articulation(cls, art):
        raise NotImplementedError("create_pinocchio_model() needs the robot's URDF: build PinocchioModel(urdf_string, gravity) instead")

    # -- orders (SAPIEN lets the caller fix them: kinematics.py:117-118) ---------------------------------------------------
    def set_joint_order(self, names):
        self._joint_order = list(names)
        self._q = np.zeros(len(self._joint_order))

    def set_link_order(self, names):
        self._link_order = list(names)

    def _chain(self, link_index):
        chain, link = [], self._link_order[link_index]
        while link in self._joint_of_child:
            j = self._joint_of_child[link]
            chain.append(j)
            link = j["parent"]
        chain.reverse()
        return chain

    def _fk(self, chain, q):
        """-> (T end (4,4), per movable joint of the chain: (index in joint order, type, axis world, origin world))"""
        T = np.eye(4)
        info = []
        idx = {n: i for i, n in enumerate(self._joint_order)}
        for j in chain:
            T = T @ j["origin"]
            if j["type"] == "fixed" or j["name"] not in idx:
                continue
            k = idx[j["name"]]
            info.append((k, j["type"], T[:3, :3] @ j["axis"], T[:3, 3].copy()))
            J = np.eye(4)
            if j["type"] == "prismatic":
                J[:3, 3] = q[k] * j["axis"]
            else:
                J[:3, :3] = _rot(j["axis"], q[k])
            T = T @ J
        return T, info

    # -- SAPIEN's surface ---------------------------------------------------------------------------------------------------
    def compute_forward_kinematics(self, qpos):
        self._q = np.asarray(qpos, dtype=np.float64).reshape(-1).copy()

    def get_link_pose(self, link_index):
        from .._pose import Pose
        T, _ = self._fk(self._chain(link_index), self._q)
        return Pose(T[:3, 3], _mat_to_quat(T[:3, :3]))

    def compute_single_link_local_jacobian(self, qpos, link_index):
        q = np.asarray(qpos, dtype=np.float64).reshape(-1)
        T, info = self._fk(self._chain(link_index), q)
        J = self._jacobian(T, info, len(q))
        R = T[:3, :3]
        return np.vstack([R.T @ J[:3], R.T @ J[3:]])

    def compute_full_jacobian(self, qpos):
        self.compute_forward_kinematics(qpos)

    def get_link_jacobian(self, link_index, local=False):
        T, info = self._fk(self._chain(link_index), self._q)
        J = self._jacobian(T, info, len(self._q))
        if local:
            R = T[:3, :3]
            return np.vstack([R.T @ J[:3], R.T @ J[3:]])
        return J

    @staticmethod
    def _jacobian(T, info, n):
        """geometric Jacobian in the base frame: rows [linear; angular]"""
        J = np.zeros((6, n))
        pe = T[:3, 3]
        for k, typ, axis, origin in info:
            if typ == "prismatic":
                J[:3, k] = axis
            else:
                J[:3, k] = np.cross(axis, pe - origin)
                J[3:, k] = axis
        return J

    def compute_inverse_kinematics(self, link_index, pose, initial_qpos=None, active_qmask=None, eps=1e-4, max_iterations=1000, dt=0.1,
                                   damp=1e-6):
        """closed-loop IK for one link: -> (qpos in joint order, success, error 6-vector)"""
        n = len(self._joint_order)
        q = np.zeros(n) if initial_qpos is None else np.asarray(initial_qpos, dtype=np.float64).reshape(-1).copy()
        mask = np.ones(n, dtype=bool) if active_qmask is None else np.asarray(
            active_qmask.cpu().numpy() if hasattr(active_qmask, "cpu") else active_qmask).astype(bool).reshape(-1)
        Rt, pt = _quat_to_mat(np.asarray(pose.q, dtype=np.float64)), np.asarray(pose.p, dtype=np.float64)
        chain = self._chain(link_index)
        lo = np.array([self._limit(nm, "lower") for nm in self._joint_order])
        hi = np.array([self._limit(nm, "upper") for nm in self._joint_order])
        err = np.zeros(6)
        for _ in range(int(max_iterations)):
            T, info = self._fk(chain, q)
            err = np.concatenate([pt - T[:3, 3], _rotvec(Rt @ T[:3, :3].T)])
            if np.linalg.norm(err) < eps:
                return q, True, err
            J = self._jacobian(T, info, n)[:, mask]
            dq = J.T @ np.linalg.solve(J @ J.T + damp * np.eye(6), err)
            q[mask] = q[mask] + dq          # full Newton step on the damped system (SAPIEN integrates v * dt with dt = 0.1: slower, same fixed point)
            q = np.minimum(np.maximum(q, lo), hi)
        return q, bool(np.linalg.norm(err) < eps), err

    def _limit(self, joint_name, which):
        for j in self._joint_of_child.values():
            if j["name"] == joint_name:
                return j[which]
        return -np.inf if which == "lower" else np.inf
