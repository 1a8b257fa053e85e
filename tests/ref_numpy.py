"""Independent numpy/scipy restatement of the hot path, used ONLY to pin the C++ oracle.

Written from the reference sources and SURVEY.md App. A/B without looking at oracle/kicp_oracle.cpp's code paths:
  * rigid-body algebra through scipy.spatial.transform.Rotation (not the oracle's hand-written quaternion code);
  * the voxel map as a python dict;
  * GetClosestNeighbor as a brute-force masked arg-min over ALL map points whose voxel is within +-1 of the query's
    voxel on every axis (equivalent to the 27-probe loop up to exact ties, which seeded random data never produces);
  * ComputeRobotMotion following registration/Registration.cpp:151-190 literally.
Pure-python loops: small cases only."""
import numpy as np
from scipy.spatial.transform import Rotation as R

EPS = np.finfo(np.float64).tiny  # std::numeric_limits<double>::min(), Registration.cpp:46


# ---- SE3 as (Rotation, translation) ----------------------------------------------------------------------------
def from_qt(p):
    return R.from_quat(p[:4]), np.asarray(p[4:], dtype=np.float64)


def to_qt(T):
    q = T[0].as_quat()
    return np.concatenate([q, T[1]])


def mul(A, B):
    return A[0] * B[0], A[1] + A[0].apply(B[1])


def inv(A):
    ri = A[0].inv()
    return ri, -ri.apply(A[1])


def act(A, pts):
    return A[0].apply(pts) + A[1]


def se3_exp(xi):
    """Sophus SE3::exp (tangent = (upsilon, omega)) via the closed-form V matrix."""
    ups, om = np.asarray(xi[:3], float), np.asarray(xi[3:], float)
    th = np.linalg.norm(om)
    W = np.array([[0, -om[2], om[1]], [om[2], 0, -om[0]], [-om[1], om[0], 0]])
    if th < 1e-10:
        V = np.eye(3) + 0.5 * W
    else:
        V = np.eye(3) + (1 - np.cos(th)) / th**2 * W + (th - np.sin(th)) / th**3 * W @ W
    return R.from_rotvec(om), V @ ups


def se3_log(T):
    om = T[0].as_rotvec()
    th = np.linalg.norm(om)
    W = np.array([[0, -om[2], om[1]], [om[2], 0, -om[0]], [-om[1], om[0], 0]])
    if th < 1e-10:
        Vinv = np.eye(3) - 0.5 * W + W @ W / 12.0
    else:
        Vinv = np.eye(3) - 0.5 * W + (1 - th * np.cos(th / 2) / (2 * np.sin(th / 2))) / th**2 * W @ W
    return np.concatenate([Vinv @ T[1], om])


# ---- kiss-icp v1.2.0 VoxelHashMap --------------------------------------------------------------
class VoxelHashMap:
    def __init__(self, voxel_size, max_distance, max_points_per_voxel):
        self.vs, self.max_distance, self.cap = voxel_size, max_distance, max_points_per_voxel
        self.map = {}

    def voxel(self, p):
        return tuple(np.floor(np.asarray(p) / self.vs).astype(np.int64))

    def AddPoints(self, pts):
        res = np.sqrt(self.vs * self.vs / self.cap)
        for p in np.asarray(pts, dtype=np.float64):
            b = self.map.get(self.voxel(p))
            if b is None:
                self.map[self.voxel(p)] = [p]
            elif len(b) < self.cap and not any(np.linalg.norm(q - p) < res for q in b):
                b.append(p)

    def RemovePointsFarFromLocation(self, origin):
        d2 = self.max_distance**2
        for k in [k for k, b in self.map.items() if np.sum((b[0] - origin) ** 2) >= d2]:
            del self.map[k]

    def Update(self, pts, pose_qt):
        T = from_qt(pose_qt)
        self.AddPoints(act(T, np.asarray(pts)))
        self.RemovePointsFarFromLocation(T[1])

    def Pointcloud(self):
        return np.array([p for b in self.map.values() for p in b]).reshape(-1, 3)

    def closest(self, queries):
        """brute force over voxel-adjacent points -> (nn (N,3), dist (N,)); no candidate: (0, DBL_MAX)."""
        pts = self.Pointcloud()
        q = np.asarray(queries, dtype=np.float64).reshape(-1, 3)
        nn = np.zeros_like(q)
        dist = np.full(len(q), np.finfo(np.float64).max)
        if len(pts) == 0:
            return nn, dist
        pv = np.floor(pts / self.vs).astype(np.int64)
        qv = np.floor(q / self.vs).astype(np.int64)
        for i in range(len(q)):
            m = np.all(np.abs(pv - qv[i]) <= 1, axis=1)
            if m.any():
                c = pts[m]
                d = np.linalg.norm(c - q[i], axis=1)
                j = int(np.argmin(d))
                nn[i], dist[i] = c[j], d[j]
        return nn, dist


# ---- registration/Registration.cpp ---------------------------------------------------------------------------------
def pass_sums(vmap, frame, T, tau):
    """DataAssociation + the reduction of ComputePerturbation at pose T -> [JTJ00,JTJ01,JTJ11,JTr0,JTr1,ssq,N]."""
    frame = np.asarray(frame, dtype=np.float64)
    q = act(T, frame)
    nn, d = vmap.closest(q)
    keep = d < tau
    s, t = frame[keep], nn[keep]
    r = act(T, s) - t
    J0 = T[0].apply(np.array([1.0, 0.0, 0.0]))
    J1 = T[0].apply(np.stack([-s[:, 1], s[:, 0], np.zeros(len(s))], axis=1))
    return np.array([len(s) * float(J0 @ J0), float(np.sum(J1 @ J0)), float(np.sum(J1 * J1)), float(np.sum(r @ J0)),
                     float(np.sum(J1 * r)), float(np.sum(r * r)), float(len(s))])


def solve(sums, beta):
    n = sums[6]
    A = np.array([[sums[0], sums[1]], [sums[1], sums[2]]]) / n + np.diag([beta, 0.0])
    b = np.array([sums[3], sums[4]]) / n
    return -np.linalg.solve(A, b)


def motion_model(dx):
    d, th = dx
    return se3_exp([d * np.sin(th) / (th + EPS), d * (1 - np.cos(th)) / (th + EPS), 0, 0, 0, th])


def compute_robot_motion(frame, vmap, last_pose_qt, rel_odom_qt, tau, max_iter=10, conv=1e-3, adaptive=True, fixed_reg=0.0):
    T = mul(from_qt(last_pose_qt), from_qt(rel_odom_qt))
    if not vmap.map:
        return to_qt(T), 0
    sums = pass_sums(vmap, frame, T, tau)
    beta = 1.0 / (sums[5] / sums[6] + EPS) if adaptive else fixed_reg
    it = 0
    for _ in range(max_iter):
        dx = solve(sums, beta)
        T = mul(T, motion_model(dx))
        it += 1
        if np.linalg.norm(dx) < conv:
            break
        sums = pass_sums(vmap, frame, T, tau)
    return to_qt(T), it


def compute_threshold(map_res, odom_sse, num_samples):
    return 3.0 * (map_res + np.sqrt(odom_sse / num_samples))


def odometry_error_in_point_space(pose_qt, max_range):
    T = from_qt(pose_qt)
    theta = np.linalg.norm(T[0].as_rotvec())
    return np.linalg.norm(T[1]) + 2.0 * max_range * np.sin(theta / 2.0)
