"""sapien.Pose: a rigid transform (p: xyz, q: wxyz), float32, immutable-by-convention value type.

Reference behaviour pinned by /root/reference/tests/structs/test_pose.py and the uses in
mani_skill/utils/structs/pose.py:60-120 (``Pose.create`` reads ``.p``/``.q``; ``to_sapien_pose`` builds ``sapien.Pose(p, q)``)."""
from __future__ import annotations

import numpy as np


def _qmul(a, b):
    w1, x1, y1, z1 = a
    w2, x2, y2, z2 = b
    return np.array([w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2, w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2,
                     w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2, w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2], dtype=np.float64)


def _qrot(q, v):
    w, x, y, z = q
    u = np.array([x, y, z], dtype=np.float64)
    v = np.asarray(v, dtype=np.float64)
    t = 2.0 * np.cross(u, v)
    return v + w * t + np.cross(u, t)


def _mat2quat(R):
    R = np.asarray(R, dtype=np.float64)
    t = np.trace(R)
    if t > 0:
        s = np.sqrt(t + 1.0) * 2
        q = np.array([0.25 * s, (R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s])
    else:
        i = int(np.argmax(np.diag(R)))
        j, k = (i + 1) % 3, (i + 2) % 3
        s = np.sqrt(max(1.0 + R[i, i] - R[j, j] - R[k, k], 1e-30)) * 2
        q = np.zeros(4)
        q[0] = (R[k, j] - R[j, k]) / s
        q[1 + i] = 0.25 * s
        q[1 + j] = (R[j, i] + R[i, j]) / s
        q[1 + k] = (R[k, i] + R[i, k]) / s
    return q / np.linalg.norm(q)


def _quat2mat(q):
    w, x, y, z = [float(v) for v in q]
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]], dtype=np.float64)


_P0 = np.zeros(3, dtype=np.float64)
_Q0 = np.array([1, 0, 0, 0], dtype=np.float64)
_P0.setflags(write=False)
_Q0.setflags(write=False)


class Pose:
    __slots__ = ("_p", "_q")

    def __init__(self, p=None, q=None):
        if p is not None and q is None and np.ndim(p) == 2:      # Pose(4x4 matrix)
            M = np.asarray(p, dtype=np.float64)
            self._p = M[:3, 3].copy()
            self._q = _mat2quat(M[:3, :3])
            return
        # kept in float64 (compositions of build-time frames stay exact to the last fp32 bit); .p / .q hand out float32 like SAPIEN.
        # The arrays are never written in place (setters replace them), so the defaults and copies of another Pose's arrays are shared.
        if p is None:
            self._p = _P0
        elif type(p) is np.ndarray and p.dtype == np.float64 and p.shape == (3,):
            self._p = p.copy()
        else:
            self._p = np.array(p, dtype=np.float64).reshape(3)
        if q is None:
            self._q = _Q0
        elif type(q) is np.ndarray and q.dtype == np.float64 and q.shape == (4,):
            self._q = q.copy()
        else:
            self._q = np.array(q, dtype=np.float64).reshape(4)

    @classmethod
    def _like(cls, other: "Pose"):
        """A new Pose object on the same (immutable by convention) arrays."""
        new = cls.__new__(cls)
        new._p, new._q = other._p, other._q
        return new

    # -- accessors ----------------------------------------------------------------------------------------------
    @property
    def p(self):
        return self._p.astype(np.float32)

    @p.setter
    def p(self, v):
        self._p = np.array(v, dtype=np.float64).reshape(3)

    @property
    def q(self):
        return self._q.astype(np.float32)

    @q.setter
    def q(self, v):
        self._q = np.array(v, dtype=np.float64).reshape(4)

    def get_p(self):
        return self.p

    def get_q(self):
        return self.q

    def set_p(self, v):
        self.p = v
        return self

    def set_q(self, v):
        self.q = v
        return self

    @property
    def rpy(self):
        from scipy.spatial.transform import Rotation as R
        w, x, y, z = self._q
        return R.from_quat([x, y, z, w]).as_euler("xyz").astype(np.float32)

    def get_rpy(self):
        return self.rpy

    def set_rpy(self, rpy):
        from scipy.spatial.transform import Rotation as R
        x, y, z, w = R.from_euler("xyz", np.asarray(rpy, dtype=np.float64)).as_quat()
        self._q = np.array([w, x, y, z], dtype=np.float64)
        return self

    # -- algebra ------------------------------------------------------------------------------------------------
    def __mul__(self, other):
        if isinstance(other, Pose):
            return Pose(_qrot(self._q, other._p) + self._p, _qmul(self._q, other._q))
        return NotImplemented

    def inv(self):
        qi = np.array([self._q[0], -self._q[1], -self._q[2], -self._q[3]], dtype=np.float64)
        return Pose(-_qrot(qi, self._p), qi)

    def to_transformation_matrix(self):
        M = np.eye(4, dtype=np.float64)
        M[:3, :3] = _quat2mat(self._q)
        M[:3, 3] = self._p
        return M.astype(np.float32)

    def _matrix64(self):
        M = np.eye(4, dtype=np.float64)
        M[:3, :3] = _quat2mat(self._q)
        M[:3, 3] = self._p
        return M

    def __repr__(self):
        return f"Pose({self._p.tolist()}, {self._q.tolist()})"

    def __getstate__(self):
        return (self._p.tolist(), self._q.tolist())

    def __setstate__(self, s):
        self._p, self._q = np.array(s[0], dtype=np.float64), np.array(s[1], dtype=np.float64)

    def __eq__(self, other):
        return isinstance(other, Pose) and np.array_equal(self._p, other._p) and np.array_equal(self._q, other._q)

    def __hash__(self):
        return hash((self._p.tobytes(), self._q.tobytes()))


def shortest_rotation(source, target):
    """sapien.math.shortest_rotation: quaternion (wxyz) of the smallest rotation taking direction `source` to `target`."""
    a = np.asarray(source, dtype=np.float64)
    b = np.asarray(target, dtype=np.float64)
    na, nb = np.linalg.norm(a), np.linalg.norm(b)
    if na < 1e-12 or nb < 1e-12:          # no direction (the "0 0 0" axis of a fixed URDF joint): identity
        return np.array([1.0, 0, 0, 0], dtype=np.float32)
    a, b = a / na, b / nb
    d = float(np.dot(a, b))
    if d < -1.0 + 1e-9:       # opposite: any axis perpendicular to a
        axis = np.cross(a, [1.0, 0, 0])
        if np.linalg.norm(axis) < 1e-6:
            axis = np.cross(a, [0, 1.0, 0])
        axis /= np.linalg.norm(axis)
        return np.array([0.0, *axis], dtype=np.float32)
    q = np.array([1.0 + d, *np.cross(a, b)])
    return (q / np.linalg.norm(q)).astype(np.float32)
