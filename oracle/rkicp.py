"""ctypes binding of oracle/_ref/libkicp_ref.so: the REFERENCE'S OWN hot-path sources (Registration.cpp,
CorrespondenceThreshold.cpp, KinematicICP.cpp), compiled unmodified against the stand-in headers of oracle/ref_shim/
(oracle/Makefile target `ref`; C entry points in oracle/ref_capi.cpp).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.  The
product package (kinematic_icp_amd) never imports this module.  Same Python surface as oracle/okicp.py so that a test
can run either checker through the same code.

/root/reference exists only in the build container: there `build()` (re)compiles the library; on the GPU box the
prebuilt file that travelled with the snapshot is used, and `available()` says whether it is there.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_ref", "libkicp_ref.so")
REFERENCE = os.environ.get("KICP_REFERENCE", "/root/reference")


def reference_present():
    return os.path.exists(os.path.join(REFERENCE, "cpp", "kinematic_icp", "registration", "Registration.cpp"))


def build():
    """make -C oracle ref (a no-op when up to date; skipped when the reference's sources are absent)."""
    if reference_present():
        subprocess.check_call(["make", "-s", "-C", _HERE, "ref", "REFERENCE=" + REFERENCE])
    return _LIB_PATH


def available():
    if reference_present():
        build()
    return os.path.exists(_LIB_PATH)


class Config(C.Structure):
    """pipeline::Config (pipeline/KinematicICP.hpp:38-60), defaults included."""
    _fields_ = [("max_range", C.c_double), ("min_range", C.c_double), ("voxel_size", C.c_double),
                ("max_points_per_voxel", C.c_uint), ("use_adaptive_threshold", C.c_int), ("fixed_threshold", C.c_double),
                ("max_num_iterations", C.c_int), ("convergence_criterion", C.c_double), ("max_num_threads", C.c_int),
                ("use_adaptive_odometry_regularization", C.c_int), ("fixed_regularization", C.c_double), ("deskew", C.c_int)]

    def __init__(self, **kw):
        super().__init__(max_range=100.0, min_range=0.0, voxel_size=1.0, max_points_per_voxel=20, use_adaptive_threshold=1,
                         fixed_threshold=1.0, max_num_iterations=10, convergence_criterion=0.001, max_num_threads=1,
                         use_adaptive_odometry_regularization=1, fixed_regularization=0.0, deskew=0)
        for k, v in kw.items():
            setattr(self, k, type(getattr(self, k))(v))


_lib = None
_dp = C.POINTER(C.c_double)
_sp = C.POINTER(C.c_size_t)


def lib():
    global _lib
    if _lib is None:
        build()
        if not os.path.exists(_LIB_PATH):
            raise RuntimeError("oracle/_ref/libkicp_ref.so is missing and %s is not present to build it from" % REFERENCE)
        L = C.CDLL(_LIB_PATH)
        L.rkicp_sources.restype = C.c_char_p
        L.rkicp_hardware_threads.restype = C.c_int
        L.rkicp_map_create.restype = C.c_void_p
        L.rkicp_map_create.argtypes = [C.c_double, C.c_double, C.c_uint]
        L.rkicp_map_destroy.argtypes = [C.c_void_p]
        L.rkicp_map_clear.argtypes = [C.c_void_p]
        L.rkicp_map_empty.argtypes = [C.c_void_p]
        L.rkicp_map_add_points.argtypes = [C.c_void_p, _dp, C.c_size_t]
        L.rkicp_map_remove_far.argtypes = [C.c_void_p, _dp]
        L.rkicp_map_update_origin.argtypes = [C.c_void_p, _dp, C.c_size_t, _dp]
        L.rkicp_map_update_pose.argtypes = [C.c_void_p, _dp, C.c_size_t, _dp]
        for f in (L.rkicp_map_num_voxels, L.rkicp_map_num_points):
            f.restype = C.c_size_t
            f.argtypes = [C.c_void_p]
        L.rkicp_map_pointcloud.restype = C.c_size_t
        L.rkicp_map_pointcloud.argtypes = [C.c_void_p, _dp, C.c_size_t]
        L.rkicp_map_closest.argtypes = [C.c_void_p, _dp, C.c_size_t, _dp, _dp]
        L.rkicp_register.restype = C.c_int
        L.rkicp_register.argtypes = [C.c_void_p, _dp, C.c_size_t, _dp, _dp, C.c_double, C.c_int, C.c_double, C.c_int, C.c_int,
                                     C.c_double, _dp, _dp]
        L.rkicp_register_timed.restype = C.c_double
        L.rkicp_register_timed.argtypes = [C.c_void_p, _dp, C.c_size_t, _dp, _dp, C.c_double, C.c_int, C.c_double, C.c_int, C.c_int,
                                           C.c_double, C.c_int, _dp]
        L.rkicp_threshold_create.restype = C.c_void_p
        L.rkicp_threshold_create.argtypes = [C.c_double, C.c_double, C.c_int, C.c_double]
        L.rkicp_threshold_destroy.argtypes = [C.c_void_p]
        L.rkicp_threshold_compute.restype = C.c_double
        L.rkicp_threshold_compute.argtypes = [C.c_void_p]
        L.rkicp_threshold_update.argtypes = [C.c_void_p, _dp]
        L.rkicp_threshold_reset.argtypes = [C.c_void_p]
        L.rkicp_pipeline_create.restype = C.c_void_p
        L.rkicp_pipeline_create.argtypes = [C.POINTER(Config)]
        L.rkicp_pipeline_destroy.argtypes = [C.c_void_p]
        L.rkicp_pipeline_set_pose.argtypes = [C.c_void_p, _dp]
        L.rkicp_pipeline_pose.argtypes = [C.c_void_p, _dp]
        L.rkicp_pipeline_tau.restype = C.c_double
        L.rkicp_pipeline_tau.argtypes = [C.c_void_p]
        L.rkicp_pipeline_register_frame.argtypes = [C.c_void_p, _dp, C.c_size_t, _dp, C.c_size_t, _dp, _dp, C.c_int, _dp, _sp, _dp, _sp]
        if hasattr(L, "rkicp_register_throughput"):
            L.rkicp_register_throughput.restype = C.c_double
            L.rkicp_register_throughput.argtypes = [C.c_void_p, _dp, _sp, C.c_size_t, _dp, _dp, C.c_double, C.c_int, C.c_double, C.c_int, C.c_double, C.c_int,
                                                    C.c_double, _sp]
        if hasattr(L, "rkicp_pipeline_register_frame_timed"):  # (a prebuilt library of an earlier round lacks it)
            L.rkicp_pipeline_register_frame_timed.restype = C.c_double
            L.rkicp_pipeline_register_frame_timed.argtypes = [C.c_void_p, _dp, C.c_size_t, _dp, C.c_size_t, _dp, _dp, C.c_int, _dp, _sp, _dp, _sp]
        L.rkicp_pipeline_local_map.restype = C.c_size_t
        L.rkicp_pipeline_local_map.argtypes = [C.c_void_p, _dp, C.c_size_t]
        L.rkicp_pipeline_map_num_points.restype = C.c_size_t
        L.rkicp_pipeline_map_num_points.argtypes = [C.c_void_p]
        L.rkicp_voxel_downsample.restype = C.c_size_t
        L.rkicp_voxel_downsample.argtypes = [_dp, C.c_size_t, C.c_double, _dp]
        L.rkicp_preprocess.restype = C.c_size_t
        L.rkicp_preprocess.argtypes = [_dp, C.c_size_t, _dp, C.c_size_t, _dp, C.c_double, C.c_double, C.c_int, _dp]
        L.rkicp_se3_exp.argtypes = [_dp, _dp]
        L.rkicp_se3_log.argtypes = [_dp, _dp]
        L.rkicp_se3_mul.argtypes = [_dp, _dp, _dp]
        L.rkicp_se3_inverse.argtypes = [_dp, _dp]
        L.rkicp_se3_act.argtypes = [_dp, _dp, C.c_size_t, _dp]
        _lib = L
    return _lib


def _d(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(_dp)


class VoxelHashMap:
    """kiss_icp::VoxelHashMap: the kiss-icp v1.2.0 stand-in of oracle/ref_shim (on tsl::robin_map's stand-in)."""

    def __init__(self, voxel_size, max_distance, max_points_per_voxel):
        self.voxel_size_, self.max_distance_, self.max_points_per_voxel_ = voxel_size, max_distance, max_points_per_voxel
        self._h = lib().rkicp_map_create(voxel_size, max_distance, max_points_per_voxel)

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.rkicp_map_destroy(self._h)
            self._h = None

    def Clear(self):
        lib().rkicp_map_clear(self._h)

    def Empty(self):
        return bool(lib().rkicp_map_empty(self._h))

    def AddPoints(self, points):
        a, p = _d(points)
        lib().rkicp_map_add_points(self._h, p, a.size // 3)

    def RemovePointsFarFromLocation(self, origin):
        _, p = _d(origin)
        lib().rkicp_map_remove_far(self._h, p)

    def Update(self, points, pose_or_origin):
        a, p = _d(points)
        b, q = _d(pose_or_origin)
        if b.size == 7:
            lib().rkicp_map_update_pose(self._h, p, a.size // 3, q)
        else:
            lib().rkicp_map_update_origin(self._h, p, a.size // 3, q)

    def num_voxels(self):
        return lib().rkicp_map_num_voxels(self._h)

    def num_points(self):
        return lib().rkicp_map_num_points(self._h)

    def Pointcloud(self):
        n = self.num_points()
        out = np.empty((n, 3), dtype=np.float64)
        lib().rkicp_map_pointcloud(self._h, out.ctypes.data_as(_dp), n)
        return out

    def GetClosestNeighbor(self, queries):
        a, p = _d(queries)
        n = a.size // 3
        nn = np.empty((n, 3), dtype=np.float64)
        d = np.empty(n, dtype=np.float64)
        lib().rkicp_map_closest(self._h, p, n, nn.ctypes.data_as(_dp), d.ctypes.data_as(_dp))
        return nn, d


class KinematicRegistration:
    """kinematic_icp::KinematicRegistration - the reference's own Registration.cpp."""

    def __init__(self, max_num_iteration=10, convergence_criterion=1e-3, max_num_threads=1,
                 use_adaptive_odometry_regularization=True, fixed_regularization=0.0):
        self.max_num_iterations_ = max_num_iteration
        self.convergence_criterion_ = convergence_criterion
        self.max_num_threads_ = max_num_threads
        self.use_adaptive_odometry_regularization_ = use_adaptive_odometry_regularization
        self.fixed_regularization_ = fixed_regularization
        self.last_status = 0
        self.last_seconds = 0.0

    def ComputeRobotMotion(self, frame, voxel_map, last_robot_pose, relative_wheel_odometry, max_correspondence_distance):
        a, p = _d(frame)
        _, lp = _d(last_robot_pose)
        _, ro = _d(relative_wheel_odometry)
        out = np.zeros(7, dtype=np.float64)
        sec = C.c_double(0.0)
        self.last_status = lib().rkicp_register(
            voxel_map._h, p, a.size // 3, lp, ro, max_correspondence_distance, self.max_num_iterations_, self.convergence_criterion_,
            self.max_num_threads_, int(self.use_adaptive_odometry_regularization_), self.fixed_regularization_,
            out.ctypes.data_as(_dp), C.cast(C.byref(sec), _dp))
        self.last_seconds = sec.value
        return out

    def timed(self, frame, voxel_map, last_robot_pose, relative_wheel_odometry, max_correspondence_distance, repeats):
        """`repeats` ComputeRobotMotion calls on a frame converted once -> (pose, seconds of the calls alone)."""
        a, p = _d(frame)
        _, lp = _d(last_robot_pose)
        _, ro = _d(relative_wheel_odometry)
        out = np.zeros(7, dtype=np.float64)
        sec = lib().rkicp_register_timed(
            voxel_map._h, p, a.size // 3, lp, ro, max_correspondence_distance, self.max_num_iterations_, self.convergence_criterion_,
            self.max_num_threads_, int(self.use_adaptive_odometry_regularization_), self.fixed_regularization_, int(repeats),
            out.ctypes.data_as(_dp))
        return out, sec


def register_throughput(voxel_map, frames, last_poses, rel_odoms, tau, threads, seconds, max_num_iteration=10, convergence_criterion=1e-3,
                        use_adaptive_odometry_regularization=True, fixed_regularization=0.0):
    """`threads` independent one-thread registrations at a time for ~`seconds` (rkicp_register_throughput) -> (scans, wall seconds)"""
    flat = np.ascontiguousarray(np.concatenate([np.asarray(f, dtype=np.float64).reshape(-1, 3) for f in frames]))
    n = (C.c_size_t * len(frames))(*[len(f) for f in frames])
    _, lp = _d(np.stack(last_poses))
    _, ro = _d(np.stack(rel_odoms))
    scans = C.c_size_t(0)
    wall = lib().rkicp_register_throughput(voxel_map._h, flat.ctypes.data_as(_dp), n, len(frames), lp, ro, tau, max_num_iteration, convergence_criterion,
                                           int(use_adaptive_odometry_regularization), fixed_regularization, threads, seconds, C.byref(scans))
    return scans.value, wall


class CorrespondenceThreshold:
    """kinematic_icp::CorrespondenceThreshold - the reference's own CorrespondenceThreshold.cpp."""

    def __init__(self, map_discretization_error, max_range, use_adaptive_threshold, fixed_threshold):
        self._h = lib().rkicp_threshold_create(map_discretization_error, max_range, int(use_adaptive_threshold), fixed_threshold)

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.rkicp_threshold_destroy(self._h)
            self._h = None

    def ComputeThreshold(self):
        return lib().rkicp_threshold_compute(self._h)

    def UpdateOdometryError(self, odometry_error_qt):
        _, p = _d(odometry_error_qt)
        lib().rkicp_threshold_update(self._h, p)

    def Reset(self):
        lib().rkicp_threshold_reset(self._h)


class KinematicICP:
    """kinematic_icp::pipeline::KinematicICP - the reference's own KinematicICP.cpp over the kiss-icp stand-ins."""

    def __init__(self, config=None, **kw):
        self.config = config if config is not None else Config(**kw)
        self._h = lib().rkicp_pipeline_create(C.byref(self.config))

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.rkicp_pipeline_destroy(self._h)
            self._h = None

    def SetPose(self, pose_qt):
        _, p = _d(pose_qt)
        lib().rkicp_pipeline_set_pose(self._h, p)

    def pose(self):
        out = np.zeros(7)
        lib().rkicp_pipeline_pose(self._h, out.ctypes.data_as(_dp))
        return out

    def tau(self):
        return lib().rkicp_pipeline_tau(self._h)

    def RegisterFrame(self, frame, timestamps, lidar_to_base, relative_odometry, num_threads=1):
        a, p = _d(frame)
        t, tp = _d(timestamps if timestamps is not None else [])
        _, lb = _d(lidar_to_base)
        _, ro = _d(relative_odometry)
        n = a.size // 3
        out_f, out_s = np.empty((max(n, 1), 3)), np.empty((max(n, 1), 3))
        nf, ns = C.c_size_t(0), C.c_size_t(0)
        lib().rkicp_pipeline_register_frame(self._h, p, n, tp, t.size, lb, ro, num_threads, out_f.ctypes.data_as(_dp), C.byref(nf),
                                            out_s.ctypes.data_as(_dp), C.byref(ns))
        return out_f[:nf.value].copy(), out_s[:ns.value].copy()

    def RegisterFrameTimed(self, frame, timestamps, lidar_to_base, relative_odometry, num_threads=1):
        """RegisterFrame -> (frame, source, seconds of the reference's RegisterFrame alone)"""
        a, p = _d(frame)
        t, tp = _d(timestamps if timestamps is not None else [])
        _, lb = _d(lidar_to_base)
        _, ro = _d(relative_odometry)
        n = a.size // 3
        out_f, out_s = np.empty((max(n, 1), 3)), np.empty((max(n, 1), 3))
        nf, ns = C.c_size_t(0), C.c_size_t(0)
        sec = lib().rkicp_pipeline_register_frame_timed(self._h, p, n, tp, t.size, lb, ro, num_threads, out_f.ctypes.data_as(_dp), C.byref(nf),
                                                        out_s.ctypes.data_as(_dp), C.byref(ns))
        return out_f[:nf.value].copy(), out_s[:ns.value].copy(), sec

    def LocalMap(self):
        n = lib().rkicp_pipeline_map_num_points(self._h)
        out = np.empty((max(n, 1), 3))
        lib().rkicp_pipeline_local_map(self._h, out.ctypes.data_as(_dp), n)
        return out[:n].copy()


def voxel_downsample(points, voxel_size):
    a, p = _d(points)
    n = a.size // 3
    out = np.empty((max(n, 1), 3))
    k = lib().rkicp_voxel_downsample(p, n, voxel_size, out.ctypes.data_as(_dp))
    return out[:k].copy()


def preprocess(points, timestamps, relative_motion_qt, max_range, min_range, deskew):
    a, p = _d(points)
    t, tp = _d(timestamps if timestamps is not None else [])
    _, r = _d(relative_motion_qt)
    n = a.size // 3
    out = np.empty((max(n, 1), 3))
    k = lib().rkicp_preprocess(p, n, tp, t.size, r, max_range, min_range, int(deskew), out.ctypes.data_as(_dp))
    return out[:k].copy()


def se3_exp(xi):
    _, p = _d(xi)
    out = np.zeros(7)
    lib().rkicp_se3_exp(p, out.ctypes.data_as(_dp))
    return out


def se3_log(qt):
    _, p = _d(qt)
    out = np.zeros(6)
    lib().rkicp_se3_log(p, out.ctypes.data_as(_dp))
    return out


def se3_mul(a, b):
    _, p = _d(a)
    _, q = _d(b)
    out = np.zeros(7)
    lib().rkicp_se3_mul(p, q, out.ctypes.data_as(_dp))
    return out


def se3_inverse(a):
    _, p = _d(a)
    out = np.zeros(7)
    lib().rkicp_se3_inverse(p, out.ctypes.data_as(_dp))
    return out


def se3_act(a, xyz):
    _, p = _d(a)
    x, q = _d(xyz)
    out = np.empty_like(x)
    lib().rkicp_se3_act(p, q, x.size // 3, out.ctypes.data_as(_dp))
    return out
