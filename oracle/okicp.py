"""ctypes binding of the CPU oracle (oracle/kicp_oracle.cpp).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package (kinematic_icp_amd) never imports this module.
PARITY UNPINNED - see the header of kicp_oracle.cpp.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libkicp_oracle.so")
MAX_PASSES = 64


def build(force=False):
    src = os.path.join(_HERE, "kicp_oracle.cpp")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s", "libkicp_oracle.so"])
    return _LIB_PATH


class Stats(C.Structure):
    _fields_ = [
        ("iterations", C.c_int32),
        ("associations", C.c_int32),
        ("converged", C.c_int32),
        ("empty_map", C.c_int32),
        ("beta", C.c_double),
        ("n_corr", C.c_double * MAX_PASSES),
        ("sums", (C.c_double * 6) * MAX_PASSES),
        ("dx", (C.c_double * 2) * MAX_PASSES),
        ("probes", C.c_uint64 * MAX_PASSES),
        ("occupied", C.c_uint64 * MAX_PASSES),
        ("points_scanned", C.c_uint64 * MAX_PASSES),
        ("seconds", C.c_double),
    ]


class Threshold(C.Structure):
    _fields_ = [
        ("map_discretization_error_", C.c_double),
        ("max_range_", C.c_double),
        ("use_adaptive_threshold_", C.c_int),
        ("fixed_threshold_", C.c_double),
        ("odom_sse_", C.c_double),
        ("num_samples_", C.c_double),
    ]


_lib = None
_dp = C.POINTER(C.c_double)


def lib():
    global _lib
    if _lib is None:
        build()
        # OKICP_LIB: another build of the same sources (`make -C oracle asan`: the restatement under ASan + UBSan)
        L = C.CDLL(os.environ.get("OKICP_LIB") or _LIB_PATH)
        L.okicp_map_create.restype = C.c_void_p
        L.okicp_map_create.argtypes = [C.c_double, C.c_double, C.c_uint]
        L.okicp_map_destroy.argtypes = [C.c_void_p]
        L.okicp_map_clear.argtypes = [C.c_void_p]
        L.okicp_map_empty.argtypes = [C.c_void_p]
        L.okicp_map_add_points.argtypes = [C.c_void_p, _dp, C.c_size_t]
        L.okicp_map_remove_far.argtypes = [C.c_void_p, _dp]
        L.okicp_map_update_origin.argtypes = [C.c_void_p, _dp, C.c_size_t, _dp]
        L.okicp_map_update_pose.argtypes = [C.c_void_p, _dp, C.c_size_t, _dp]
        L.okicp_map_num_voxels.restype = C.c_size_t
        L.okicp_map_num_voxels.argtypes = [C.c_void_p]
        L.okicp_map_num_points.restype = C.c_size_t
        L.okicp_map_num_points.argtypes = [C.c_void_p]
        L.okicp_map_pointcloud.restype = C.c_size_t
        L.okicp_map_pointcloud.argtypes = [C.c_void_p, _dp, C.c_size_t]
        L.okicp_map_closest.argtypes = [C.c_void_p, _dp, C.c_size_t, _dp, _dp]
        L.okicp_pass.argtypes = [C.c_void_p, _dp, C.c_size_t, _dp, C.c_double, C.c_int, _dp, C.POINTER(C.c_uint64)]
        L.okicp_associate.argtypes = [C.c_void_p, _dp, C.c_size_t, _dp, C.c_double, C.POINTER(C.c_int32), _dp, _dp]
        L.okicp_register.restype = C.c_int
        L.okicp_register.argtypes = [C.c_void_p, _dp, C.c_size_t, _dp, _dp, C.c_double, C.c_int, C.c_double, C.c_int,
                                     C.c_int, C.c_double, C.c_int, _dp, C.POINTER(Stats)]
        L.okicp_max_threads.restype = C.c_int
        L.okicp_se3_exp.argtypes = [_dp, _dp]
        L.okicp_se3_log.argtypes = [_dp, _dp]
        L.okicp_se3_mul.argtypes = [_dp, _dp, _dp]
        L.okicp_se3_inverse.argtypes = [_dp, _dp]
        L.okicp_se3_act.argtypes = [_dp, _dp, C.c_size_t, _dp]
        L.okicp_motion_model.argtypes = [_dp, _dp]
        L.okicp_solve.argtypes = [_dp, C.c_double, C.c_double, _dp]
        L.okicp_threshold_init.argtypes = [C.POINTER(Threshold), C.c_double, C.c_double, C.c_int, C.c_double]
        L.okicp_threshold_compute.restype = C.c_double
        L.okicp_threshold_compute.argtypes = [C.POINTER(Threshold)]
        L.okicp_threshold_update.argtypes = [C.POINTER(Threshold), _dp]
        L.okicp_threshold_reset.argtypes = [C.POINTER(Threshold)]
        L.okicp_voxel_downsample.restype = C.c_size_t
        L.okicp_voxel_downsample.argtypes = [_dp, C.c_size_t, C.c_double, _dp]
        L.okicp_ingest.restype = C.c_int
        L.okicp_ingest.argtypes = [C.c_void_p, C.c_size_t, C.c_uint, C.c_uint, C.c_uint, C.c_uint, C.c_int, C.c_uint, _dp, _dp, _dp, _dp]
        L.okicp_preprocess.restype = C.c_size_t
        L.okicp_preprocess.argtypes = [_dp, C.c_size_t, _dp, C.c_size_t, _dp, C.c_double, C.c_double, C.c_int, _dp]
        _lib = L
    return _lib


def _d(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(_dp)


IDENTITY = np.array([0, 0, 0, 1, 0, 0, 0], dtype=np.float64)


class VoxelHashMap:
    """Oracle twin of kiss_icp::VoxelHashMap (kiss-icp v1.2.0)."""

    def __init__(self, voxel_size, max_distance, max_points_per_voxel):
        self.voxel_size_, self.max_distance_, self.max_points_per_voxel_ = voxel_size, max_distance, max_points_per_voxel
        self._h = lib().okicp_map_create(voxel_size, max_distance, max_points_per_voxel)

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.okicp_map_destroy(self._h)
            self._h = None

    def Clear(self):
        lib().okicp_map_clear(self._h)

    def Empty(self):
        return bool(lib().okicp_map_empty(self._h))

    def AddPoints(self, points):
        a, p = _d(points)
        lib().okicp_map_add_points(self._h, p, a.size // 3)

    def RemovePointsFarFromLocation(self, origin):
        _, p = _d(origin)
        lib().okicp_map_remove_far(self._h, p)

    def Update(self, points, pose_or_origin):
        a, p = _d(points)
        b, q = _d(pose_or_origin)
        if b.size == 7:
            lib().okicp_map_update_pose(self._h, p, a.size // 3, q)
        else:
            lib().okicp_map_update_origin(self._h, p, a.size // 3, q)

    def num_voxels(self):
        return lib().okicp_map_num_voxels(self._h)

    def num_points(self):
        return lib().okicp_map_num_points(self._h)

    def Pointcloud(self):
        n = self.num_points()
        out = np.empty((n, 3), dtype=np.float64)
        lib().okicp_map_pointcloud(self._h, out.ctypes.data_as(_dp), n)
        return out

    def GetClosestNeighbor(self, queries):
        a, p = _d(queries)
        n = a.size // 3
        nn = np.empty((n, 3), dtype=np.float64)
        d = np.empty(n, dtype=np.float64)
        lib().okicp_map_closest(self._h, p, n, nn.ctypes.data_as(_dp), d.ctypes.data_as(_dp))
        return nn, d


def icp_pass(vmap, frame, pose_qt, tau, num_threads=1):
    """One association + accumulation pass at a fixed pose -> (sums[7], counters[3])."""
    a, p = _d(frame)
    _, q = _d(pose_qt)
    sums = np.zeros(7, dtype=np.float64)
    cnt = (C.c_uint64 * 3)()
    lib().okicp_pass(vmap._h, p, a.size // 3, q, tau, num_threads, sums.ctypes.data_as(_dp), cnt)
    return sums, np.array(list(cnt), dtype=np.uint64)


def associate(vmap, frame, pose_qt, tau):
    """DataAssociation per query (Registration.cpp:73-77): (accepted[n] bool, nn[n, 3], distance[n]) - nn / distance are
    GetClosestNeighbor(pose * frame[i]); accepted = distance < tau."""
    a, p = _d(frame)
    _, q = _d(pose_qt)
    n = a.size // 3
    acc = np.zeros(n, dtype=np.int32)
    nn = np.empty((n, 3), dtype=np.float64)
    d = np.empty(n, dtype=np.float64)
    lib().okicp_associate(vmap._h, p, n, q, tau, acc.ctypes.data_as(C.POINTER(C.c_int32)), nn.ctypes.data_as(_dp), d.ctypes.data_as(_dp))
    return acc.astype(bool), nn, d


class KinematicRegistration:
    """Oracle twin of kinematic_icp::KinematicRegistration (registration/Registration.hpp:32-50)."""

    def __init__(self, max_num_iteration=10, convergence_criterion=1e-3, max_num_threads=1,
                 use_adaptive_odometry_regularization=True, fixed_regularization=0.0):
        self.max_num_iterations_ = max_num_iteration
        self.convergence_criterion_ = convergence_criterion
        self.max_num_threads_ = max_num_threads
        self.use_adaptive_odometry_regularization_ = use_adaptive_odometry_regularization
        self.fixed_regularization_ = fixed_regularization
        self.last_stats = None
        self.last_status = 0

    def ComputeRobotMotion(self, frame, voxel_map, last_robot_pose, relative_wheel_odometry, max_correspondence_distance,
                           count_work=False):
        a, p = _d(frame)
        _, lp = _d(last_robot_pose)
        _, ro = _d(relative_wheel_odometry)
        out = np.zeros(7, dtype=np.float64)
        st = Stats()
        self.last_status = lib().okicp_register(
            voxel_map._h, p, a.size // 3, lp, ro, max_correspondence_distance, self.max_num_iterations_,
            self.convergence_criterion_, self.max_num_threads_, int(self.use_adaptive_odometry_regularization_),
            self.fixed_regularization_, int(count_work), out.ctypes.data_as(_dp), C.byref(st))
        self.last_stats = st
        return out


def se3_exp(xi):
    _, p = _d(xi)
    out = np.zeros(7)
    lib().okicp_se3_exp(p, out.ctypes.data_as(_dp))
    return out


def se3_log(qt):
    _, p = _d(qt)
    out = np.zeros(6)
    lib().okicp_se3_log(p, out.ctypes.data_as(_dp))
    return out


def se3_mul(a, b):
    _, p = _d(a)
    _, q = _d(b)
    out = np.zeros(7)
    lib().okicp_se3_mul(p, q, out.ctypes.data_as(_dp))
    return out


def se3_inverse(a):
    _, p = _d(a)
    out = np.zeros(7)
    lib().okicp_se3_inverse(p, out.ctypes.data_as(_dp))
    return out


def se3_act(a, xyz):
    _, p = _d(a)
    x, q = _d(xyz)
    out = np.empty_like(x)
    lib().okicp_se3_act(p, q, x.size // 3, out.ctypes.data_as(_dp))
    return out


def motion_model(controls):
    _, p = _d(controls)
    out = np.zeros(7)
    lib().okicp_motion_model(p, out.ctypes.data_as(_dp))
    return out


def solve(sums5, n_corr, beta):
    _, p = _d(sums5)
    out = np.zeros(2)
    lib().okicp_solve(p, float(n_corr), float(beta), out.ctypes.data_as(_dp))
    return out


class CorrespondenceThreshold:
    """Oracle twin of kinematic_icp::CorrespondenceThreshold (CorrespondenceThreshold.hpp:30-55)."""

    def __init__(self, map_discretization_error, max_range, use_adaptive_threshold, fixed_threshold):
        self._t = Threshold()
        lib().okicp_threshold_init(C.byref(self._t), map_discretization_error, max_range, int(use_adaptive_threshold),
                                   fixed_threshold)

    def ComputeThreshold(self):
        return lib().okicp_threshold_compute(C.byref(self._t))

    def UpdateOdometryError(self, odometry_error_qt):
        _, p = _d(odometry_error_qt)
        lib().okicp_threshold_update(C.byref(self._t), p)

    def Reset(self):
        lib().okicp_threshold_reset(C.byref(self._t))


def voxel_downsample(points, voxel_size):
    a, p = _d(points)
    n = a.size // 3
    out = np.empty((max(n, 1), 3))
    k = lib().okicp_voxel_downsample(p, n, voxel_size, out.ctypes.data_as(_dp))
    return out[:k].copy()


def preprocess(points, timestamps, relative_motion_qt, max_range, min_range, deskew):
    a, p = _d(points)
    t, tp = _d(timestamps if timestamps is not None else [])
    _, r = _d(relative_motion_qt)
    n = a.size // 3
    out = np.empty((max(n, 1), 3))
    k = lib().okicp_preprocess(p, n, tp, t.size, r, max_range, min_range, int(deskew), out.ctypes.data_as(_dp))
    return out[:k].copy()


def ingest(raw, n, point_step, off_x, off_y, off_z, stamp_type=0, off_t=0, sensor_pose_qt=None):
    """PointCloud2ToEigen + ExtractTimestampsFromMsg + normalisation on raw message bytes (RosUtils.cpp:30-39,
    TimeStampHandler.cpp:57-104,106,121-128) -> (xyz (n,3), normalised stamps (n,) or None, (min, max) seconds)."""
    buf = np.ascontiguousarray(np.frombuffer(raw, dtype=np.uint8))
    xyz = np.empty((max(n, 1), 3))
    st = np.empty(max(n, 1))
    mm = np.zeros(2)
    q = None if sensor_pose_qt is None else _d(sensor_pose_qt)[1]
    rc = lib().okicp_ingest(buf.ctypes.data, n, point_step, off_x, off_y, off_z, stamp_type, off_t, q, xyz.ctypes.data_as(_dp),
                            st.ctypes.data_as(_dp), mm.ctypes.data_as(_dp))
    if rc < 0:
        raise RuntimeError("timestamp field type not supported")
    return xyz[:n].copy(), (st[:n].copy() if rc == 1 else None), (float(mm[0]), float(mm[1]))
