"""kinematic_icp_amd -- MI355X-native backend of the kinematic-ICP registration hot path.

Python mirror of the reference's operator interface for this path, bound with ctypes to the C-ABI of
include/kicp.h (libkicp_amd.so, hand-written HIP for gfx950):

    kinematic_icp::KinematicRegistration   /root/reference/cpp/kinematic_icp/registration/Registration.hpp:32-50
    kiss_icp::VoxelHashMap (v1.2.0)        call sites Registration.cpp:63,74,157; KinematicICP.hpp:79,88,92; KinematicICP.cpp:79

Names, argument order and argument meaning follow the reference.  Poses are 7-vectors
[qx, qy, qz, qw, tx, ty, tz] (Sophus::SE3d parameter order); point clouds are (N, 3) float64 arrays.

There is NO CPU fallback: importing works anywhere, but creating a KinematicRegistration (or running any
device operation) without the built extension or without a visible gfx950 device raises KicpError.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libkicp_amd.so")
MAX_LOG_PASSES = 32
P2P_HANDLE_BYTES = 64
COMM_ID_BYTES = 128

KICP_OK = 0
KICP_WARN_NO_CORRESPONDENCES = 1
KICP_WARN_TABLE_ORDER = 2
KICP_ERR_HIP, KICP_ERR_ARG, KICP_ERR_CAPACITY, KICP_ERR_COMM = -1, -2, -3, -4


class KicpError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("kicp error %d: %s" % (code, message))
        self.code = code


class RegConfig(C.Structure):
    _fields_ = [("max_num_iterations", C.c_int32), ("convergence_criterion", C.c_double), ("max_num_threads", C.c_int32),
                ("use_adaptive_odometry_regularization", C.c_int32), ("fixed_regularization", C.c_double)]


class Stats(C.Structure):
    _fields_ = [("iterations", C.c_int32), ("converged", C.c_int32), ("empty_map", C.c_int32), ("reserved", C.c_int32),
                ("beta", C.c_double), ("n_corr", C.c_double * MAX_LOG_PASSES), ("sums", (C.c_double * 6) * MAX_LOG_PASSES),
                ("dx", (C.c_double * 2) * MAX_LOG_PASSES), ("gpu_ms", C.c_double), ("pass_ms", C.c_double * MAX_LOG_PASSES)]


ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p)

_dp = C.POINTER(C.c_double)
_lib = None

# every symbol include/kicp.h declares: (restype, argtypes)
_SIGNATURES = {
    "kicp_last_error": (C.c_char_p, []),
    "kicp_version": (C.c_int, []),
    "kicp_device_count": (C.c_int, []),
    "kicp_device_locality": (C.c_int, [C.c_int, C.POINTER(C.c_int), C.c_char_p, C.c_size_t]),
    "kicp_map_create": (C.c_int, [C.c_double, C.c_double, C.c_uint, C.POINTER(C.c_void_p)]),
    "kicp_map_destroy": (None, [C.c_void_p]),
    "kicp_map_clone": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "kicp_map_clear": (C.c_int, [C.c_void_p]),
    "kicp_map_empty": (C.c_int, [C.c_void_p]),
    "kicp_map_add_points": (C.c_int, [C.c_void_p, _dp, C.c_size_t]),
    "kicp_map_remove_far": (C.c_int, [C.c_void_p, _dp]),
    "kicp_map_update_origin": (C.c_int, [C.c_void_p, _dp, C.c_size_t, _dp]),
    "kicp_map_update_pose": (C.c_int, [C.c_void_p, _dp, C.c_size_t, _dp]),
    "kicp_map_update_pose_device": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, _dp]),
    "kicp_map_update_pose_device_begin": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, _dp]),
    "kicp_map_update_finish": (C.c_int, [C.c_void_p]),
    "kicp_map_last_update_on_device": (C.c_int, [C.c_void_p]),
    "kicp_map_device_updates": (C.c_ulonglong, [C.c_void_p]),
    "kicp_map_set_device": (C.c_int, [C.c_void_p, C.c_int]),
    "kicp_map_num_points": (C.c_size_t, [C.c_void_p]),
    "kicp_map_num_voxels": (C.c_size_t, [C.c_void_p]),
    "kicp_map_pointcloud": (C.c_size_t, [C.c_void_p, _dp, C.c_size_t]),
    "kicp_map_closest": (C.c_int, [C.c_void_p, C.c_int, _dp, C.c_size_t, _dp, _dp]),
    "kicp_map_check": (C.c_size_t, [C.c_void_p]),
    "kicp_map_sync": (C.c_int, [C.c_void_p, C.c_int]),
    "kicp_map_last_upload": (C.c_int, [C.c_void_p, C.POINTER(C.c_size_t), C.POINTER(C.c_int)]),
    "kicp_reg_create": (C.c_int, [C.POINTER(RegConfig), C.c_int, C.POINTER(C.c_void_p)]),
    "kicp_reg_destroy": (None, [C.c_void_p]),
    "kicp_reg_clone": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "kicp_reg_get_config": (C.c_int, [C.c_void_p, C.POINTER(RegConfig)]),
    "kicp_reg_set_config": (C.c_int, [C.c_void_p, C.POINTER(RegConfig)]),
    "kicp_reg_set_option": (C.c_int, [C.c_void_p, C.c_char_p, C.c_double]),
    "kicp_reg_get_option": (C.c_double, [C.c_void_p, C.c_char_p]),
    "kicp_register": (C.c_int, [C.c_void_p, C.c_void_p, _dp, C.c_size_t, _dp, _dp, C.c_double, _dp, C.POINTER(Stats)]),
    "kicp_register_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, _dp, _dp, C.c_double, _dp, C.POINTER(Stats)]),
    "kicp_register_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, _dp, _dp, C.c_double, _dp, C.POINTER(Stats)]),
    "kicp_register_device_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), _dp, _dp,
                                             C.c_double, _dp, C.POINTER(C.c_int)]),
    "kicp_register_device_concurrent": (C.c_int, [C.POINTER(C.c_void_p), C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t),
                                                  _dp, _dp, C.c_double, _dp, C.POINTER(C.c_int)]),
    "kicp_pass_sums": (C.c_int, [C.c_void_p, C.c_void_p, _dp, C.c_size_t, _dp, C.c_double, _dp]),
    "kicp_pass_correspondences": (C.c_int, [C.c_void_p, C.c_void_p, _dp, C.c_size_t, _dp, C.c_double, C.POINTER(C.c_int32), _dp, _dp]),
    "kicp_pass_words": (C.c_int, [C.c_void_p, C.c_void_p, _dp, C.c_size_t, _dp, C.c_double, C.POINTER(C.c_longlong)]),
    "kicp_pre_create": (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    "kicp_pre_destroy": (None, [C.c_void_p]),
    "kicp_pre_preprocess": (C.c_int, [C.c_void_p, _dp, C.c_size_t, _dp, C.c_size_t, _dp, _dp, C.c_double, C.c_double, C.c_int, C.c_int,
                                      C.POINTER(C.c_size_t)]),
    "kicp_pre_ingest": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, _dp, _dp, _dp]),
    "kicp_pre_ingest_ahead": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, _dp]),
    "kicp_pre_ahead_hits": (C.c_ulonglong, [C.c_void_p]),
    "kicp_pre_set_option": (C.c_int, [C.c_void_p, C.c_char_p, C.c_double]),
    "kicp_pre_get_option": (C.c_double, [C.c_void_p, C.c_char_p]),
    "kicp_pre_preprocess_ingested": (C.c_int, [C.c_void_p, _dp, _dp, C.c_double, C.c_double, C.c_int, C.c_int, C.POINTER(C.c_size_t)]),
    "kicp_pre_ingested": (C.c_int, [C.c_void_p, _dp, _dp, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(C.c_int)]),
    "kicp_pre_frame": (C.c_int, [C.c_void_p, _dp, C.c_size_t, _dp, C.c_size_t, _dp, _dp, C.c_double, C.c_double, C.c_int, C.c_double, C.c_double, _dp, C.c_size_t,
                                 C.POINTER(C.c_size_t)]),
    "kicp_pre_frame_ingested": (C.c_int, [C.c_void_p, _dp, _dp, C.c_double, C.c_double, C.c_int, C.c_double, C.c_double, _dp, C.c_size_t, C.POINTER(C.c_size_t)]),
    "kicp_pre_ingested_count": (C.c_size_t, [C.c_void_p]),
    "kicp_pre_voxel_downsample": (C.c_int, [C.c_void_p, C.c_int, C.c_double, C.c_int, C.POINTER(C.c_size_t)]),
    "kicp_pre_last_max_probe": (C.c_uint, [C.c_void_p]),
    "kicp_pre_set_probe_limit": (C.c_int, [C.c_void_p, C.c_uint]),
    "kicp_pre_upload": (C.c_int, [C.c_void_p, C.c_int, _dp, C.c_size_t]),
    "kicp_pre_download": (C.c_int, [C.c_void_p, C.c_int, _dp, C.c_size_t, C.POINTER(C.c_size_t)]),
    "kicp_pre_download_begin": (C.c_int, [C.c_void_p, C.c_int]),
    "kicp_pre_download_begin_into": (C.c_int, [C.c_void_p, C.c_int, _dp, C.c_size_t]),
    "kicp_pre_download_finish": (C.c_int, [C.c_void_p, C.c_int, _dp, C.c_size_t, C.POINTER(C.c_size_t)]),
    "kicp_pre_device_ptr": (C.c_void_p, [C.c_void_p, C.c_int, C.POINTER(C.c_size_t)]),
    "kicp_device_malloc": (C.c_int, [C.c_int, C.c_size_t, C.POINTER(C.c_void_p)]),
    "kicp_device_free": (C.c_int, [C.c_int, C.c_void_p]),
    "kicp_device_upload": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_size_t]),
    "kicp_device_synchronize": (C.c_int, [C.c_int]),
    "kicp_comm_unique_id": (C.c_int, [C.c_char_p]),
    "kicp_reg_comm_init": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_char_p]),
    "kicp_reg_comm_destroy": (C.c_int, [C.c_void_p]),
    "kicp_reg_shm_init": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_char_p]),
    "kicp_reg_shm_destroy": (C.c_int, [C.c_void_p]),
    "kicp_reg_p2p_export": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_char_p]),
    "kicp_reg_p2p_connect": (C.c_int, [C.c_void_p, C.c_char_p]),
    "kicp_reg_p2p_destroy": (C.c_int, [C.c_void_p]),
    "kicp_reg_set_allreduce": (C.c_int, [C.c_void_p, ALLREDUCE_FN, C.c_void_p]),
    "kicp_aql_kernel_names": (C.c_size_t, [C.c_char_p, C.c_size_t]),
    "kicp_probe_dependent_load": (C.c_int, [C.c_int, C.c_size_t, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]),
    "kicp_map_device_bytes": (C.c_size_t, [C.c_void_p]),
}


def lib():
    """Load libkicp_amd.so (once).  Fails loudly if the HIP extension has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise KicpError(KICP_ERR_HIP, "HIP extension %s not built (run `python -c 'import __graft_entry__ as g; g.build()'` "
                                          "or `make -C kinematic_icp_amd/csrc`); there is no CPU fallback" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError here = header/library mismatch
            fn.restype, fn.argtypes = res, args
        _lib = L
    return _lib


def _check(rc):
    if rc < 0:
        raise KicpError(rc, lib().kicp_last_error().decode(errors="replace"))
    return rc


def _d(a):
    if not (isinstance(a, np.ndarray) and a.dtype == np.float64 and a.flags.c_contiguous):
        a = np.ascontiguousarray(a, dtype=np.float64)
    return a, C.cast(a.ctypes.data, _dp)


# Small arrays that come back call after call (poses): their ctypes pointer is built once.  The cache holds a reference
# to the array, so its address - and the id() used as the key - stays valid; it is only used for arrays of at most 16
# doubles and is bounded.
_PTR_CACHE = {}


def _pose_ptr(a):
    hit = _PTR_CACHE.get(id(a))
    if hit is not None and hit[0] is a:
        return hit[1]
    arr, ptr = _d(a)
    if arr is a and arr.size <= 16:
        if len(_PTR_CACHE) >= 256:
            _PTR_CACHE.clear()
        _PTR_CACHE[id(a)] = (a, ptr)
    elif arr is not a:
        ptr._keep = arr  # a converted temporary must outlive the call
    return ptr


def probe_dependent_load(working_set_bytes, workgroups=512, block=256, steps=64, device=0):
    """ns per dependent load step of `workgroups` x `block` lanes walking a random chain through `working_set_bytes` (kicp.h)"""
    out = C.c_double(0.0)
    _check(lib().kicp_probe_dependent_load(device, int(working_set_bytes), workgroups, block, steps, C.byref(out)))
    return out.value


def device_count():
    return lib().kicp_device_count()


def device_locality(device=0):
    """(NUMA node the GPU is attached to | -1, its CPUs as a set | empty) - kicp_device_locality"""
    node, buf = C.c_int(-1), C.create_string_buffer(4096)
    _check(lib().kicp_device_locality(device, C.byref(node), buf, len(buf)))
    cpus = set()
    for part in buf.value.decode().split(","):
        if part.strip():
            lo, _, hi = part.partition("-")
            cpus.update(range(int(lo), int(hi or lo) + 1))
    return node.value, cpus


def cpus_near_gpu(device=0, one_l3_domain=True):
    """CPUs to bind a process that drives `device` to: the GPU's NUMA node, narrowed to what the process may use and (by default) to
    ONE L3 domain of it - the calling thread and the library's helper threads then share a last-level cache.  Empty: unknown."""
    import os
    _, cpus = device_locality(device)
    cpus &= os.sched_getaffinity(0)
    if not cpus or not one_l3_domain:
        return cpus
    try:
        with open("/sys/devices/system/cpu/cpu%d/cache/index3/shared_cpu_list" % min(cpus)) as f:
            l3 = set()
            for part in f.read().strip().split(","):
                lo, _, hi = part.partition("-")
                l3.update(range(int(lo), int(hi or lo) + 1))
        return (cpus & l3) or cpus
    except OSError:
        return cpus


class VoxelHashMap:
    """kiss_icp::VoxelHashMap: host-authoritative voxel map with an HBM mirror."""

    def __init__(self, voxel_size, max_distance, max_points_per_voxel, device=None):
        """device: preferred GPU for bulk insertions (AddPoints / Update with >= 4096 points run there); None = on the host."""
        self.voxel_size_, self.max_distance_, self.max_points_per_voxel_ = voxel_size, max_distance, max_points_per_voxel
        h = C.c_void_p()
        _check(lib().kicp_map_create(voxel_size, max_distance, max_points_per_voxel, C.byref(h)))
        self._h = h
        if device is not None:
            self.set_device(device)

    def set_device(self, device):
        _check(lib().kicp_map_set_device(self._h, -1 if device is None else int(device)))

    def copy(self):
        """VoxelHashMap(const VoxelHashMap&): a deep copy of the newest state."""
        c = VoxelHashMap.__new__(VoxelHashMap)
        c.voxel_size_, c.max_distance_, c.max_points_per_voxel_ = self.voxel_size_, self.max_distance_, self.max_points_per_voxel_
        h = C.c_void_p()
        _check(lib().kicp_map_clone(self._h, C.byref(h)))
        c._h = h
        return c

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.kicp_map_destroy(self._h)
            self._h = None

    def Clear(self):
        _check(lib().kicp_map_clear(self._h))

    def Empty(self):
        return bool(lib().kicp_map_empty(self._h))

    def AddPoints(self, points):
        a, p = _d(points)
        _check(lib().kicp_map_add_points(self._h, p, a.size // 3))

    def RemovePointsFarFromLocation(self, origin):
        _, p = _d(origin)
        _check(lib().kicp_map_remove_far(self._h, p))

    def Update(self, points, pose_or_origin):
        a, p = _d(points)
        b, q = _d(pose_or_origin)
        if b.size == 7:
            _check(lib().kicp_map_update_pose(self._h, p, a.size // 3, q))
        elif b.size == 3:
            _check(lib().kicp_map_update_origin(self._h, p, a.size // 3, q))
        else:
            raise ValueError("Update expects a 7-vector pose or a 3-vector origin")

    def UpdateDevice(self, device_frame, pose):
        """Update(points, pose) with the points already in HBM (DeviceFrame / PreSteps.frame); returns True if it ran on
        the GPU, False if the host fallback (table or pool growth) was taken."""
        _, q = _d(pose)
        _check(lib().kicp_map_update_pose_device(self._h, device_frame.device, device_frame.ptr, device_frame.n, q))
        return bool(lib().kicp_map_last_update_on_device(self._h))

    def UpdateDeviceBegin(self, device_frame, pose):
        """kicp_map_update_pose_device_begin: the update's kernels are queued, nothing is waited for (UpdateFinish, or any other
        call on the map, collects it); the frame must stay alive and unchanged until then"""
        _, q = _d(pose)
        _check(lib().kicp_map_update_pose_device_begin(self._h, device_frame.device, device_frame.ptr, device_frame.n, q))

    def UpdateFinish(self):
        _check(lib().kicp_map_update_finish(self._h))
        return bool(lib().kicp_map_last_update_on_device(self._h))

    def num_points(self):
        return lib().kicp_map_num_points(self._h)

    def num_voxels(self):
        return lib().kicp_map_num_voxels(self._h)

    def Pointcloud(self):
        n = self.num_points()
        out = np.empty((n, 3), dtype=np.float64)
        lib().kicp_map_pointcloud(self._h, out.ctypes.data_as(_dp), n)
        return out

    def GetClosestNeighbor(self, queries, device=0):
        """Batch form of GetClosestNeighbor: (N,3) queries -> ((N,3) neighbours, (N,) distances)."""
        a, p = _d(queries)
        n = a.size // 3
        nn = np.empty((n, 3), dtype=np.float64)
        d = np.empty(n, dtype=np.float64)
        _check(lib().kicp_map_closest(self._h, device, p, n, nn.ctypes.data_as(_dp), d.ctypes.data_as(_dp)))
        return nn, d

    def check(self):
        """Number of violated table invariants on the host copy (0 = consistent); debug aid."""
        return lib().kicp_map_check(self._h)

    def sync(self, device=0):
        _check(lib().kicp_map_sync(self._h, device))

    def device_bytes(self):
        return lib().kicp_map_device_bytes(self._h)

    def last_upload(self):
        """(bytes sent by the last mirror upload, True if it was a full re-send rather than a delta)."""
        b, f = C.c_size_t(), C.c_int()
        _check(lib().kicp_map_last_upload(self._h, C.byref(b), C.byref(f)))
        return b.value, bool(f.value)


class DeviceFrame:
    """A scan resident in HBM (what an on-device pre-step would hand to the registration)."""

    def __init__(self, points, device=0):
        a, _ = _d(points)
        self.n, self.device = a.size // 3, device
        self.ptr = C.c_void_p()
        _check(lib().kicp_device_malloc(device, max(a.nbytes, 8), C.byref(self.ptr)))
        if a.nbytes:
            _check(lib().kicp_device_upload(device, self.ptr, a.ctypes.data_as(C.c_void_p), a.nbytes))

    def __del__(self):
        if getattr(self, "ptr", None) and _lib is not None and not getattr(self, "_borrowed", False):
            _lib.kicp_device_free(self.device, self.ptr)
            self.ptr = None


class _Batch:
    pass


class KinematicRegistration:
    """kinematic_icp::KinematicRegistration (registration/Registration.hpp:32-50) on one MI355X."""

    def __init__(self, max_num_iteration=10, convergence_criterion=1e-3, max_num_threads=1,
                 use_adaptive_odometry_regularization=True, fixed_regularization=0.0, device=0):
        cfg = RegConfig(max_num_iteration, convergence_criterion, max_num_threads, int(use_adaptive_odometry_regularization),
                        fixed_regularization)
        h = C.c_void_p()
        _check(lib().kicp_reg_create(C.byref(cfg), device, C.byref(h)))
        self._h, self.device = h, device
        self.last_stats, self.last_status = Stats(), 0
        self._stats_ref = C.byref(self.last_stats)
        self._out = np.zeros(7, dtype=np.float64)
        self._out_p = C.cast(self._out.ctypes.data, _dp)
        self._cb = None

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.kicp_reg_destroy(self._h)
            self._h = None

    def copy(self):
        """KinematicRegistration(const KinematicRegistration &): the reference's struct is copyable (Registration.hpp:32-50);
        same parameters and tuning options, workspaces of its own (kicp_reg_clone)."""
        h = C.c_void_p()
        _check(lib().kicp_reg_clone(self._h, C.byref(h)))
        other = object.__new__(KinematicRegistration)
        other._h, other.device = h, self.device
        other.last_stats, other.last_status = Stats(), 0
        other._stats_ref = C.byref(other.last_stats)
        other._out = np.zeros(7, dtype=np.float64)
        other._out_p = C.cast(other._out.ctypes.data, _dp)
        other._cb = None
        return other

    # the reference exposes its five parameters as public mutable fields (Registration.hpp:45-49)
    def _cfg(self):
        c = RegConfig()
        _check(lib().kicp_reg_get_config(self._h, C.byref(c)))
        return c

    def _set(self, **kw):
        c = self._cfg()
        for k, v in kw.items():
            setattr(c, k, v)
        _check(lib().kicp_reg_set_config(self._h, C.byref(c)))

    max_num_iterations_ = property(lambda s: s._cfg().max_num_iterations, lambda s, v: s._set(max_num_iterations=int(v)))
    convergence_criterion_ = property(lambda s: s._cfg().convergence_criterion, lambda s, v: s._set(convergence_criterion=float(v)))
    max_num_threads_ = property(lambda s: s._cfg().max_num_threads, lambda s, v: s._set(max_num_threads=int(v)))
    use_adaptive_odometry_regularization_ = property(lambda s: bool(s._cfg().use_adaptive_odometry_regularization),
                                                     lambda s, v: s._set(use_adaptive_odometry_regularization=int(bool(v))))
    fixed_regularization_ = property(lambda s: s._cfg().fixed_regularization, lambda s, v: s._set(fixed_regularization=float(v)))

    def set_option(self, name, value):
        _check(lib().kicp_reg_set_option(self._h, name.encode(), float(value)))

    def get_option(self, name):
        return lib().kicp_reg_get_option(self._h, name.encode())

    def ComputeRobotMotion(self, frame, voxel_map, last_robot_pose, relative_wheel_odometry, max_correspondence_distance):
        """frame: (N,3) float64 host array, or a DeviceFrame already resident in HBM."""
        lp, ro = _pose_ptr(last_robot_pose), _pose_ptr(relative_wheel_odometry)
        if isinstance(frame, DeviceFrame):
            rc = _lib.kicp_register_device(self._h, voxel_map._h, frame.ptr, frame.n, lp, ro, max_correspondence_distance,
                                           self._out_p, self._stats_ref)
        elif isinstance(frame, np.ndarray) and frame.dtype == np.float32:
            # float32 xyz as a PointCloud2 carries it: widened on the device (kicp_register_f32), half the bytes over PCIe
            a = np.ascontiguousarray(frame)
            rc = _lib.kicp_register_f32(self._h, voxel_map._h, a.ctypes.data, a.size // 3, lp, ro, max_correspondence_distance,
                                        self._out_p, self._stats_ref)
        else:
            a, p = _d(frame)
            rc = _lib.kicp_register(self._h, voxel_map._h, p, a.size // 3, lp, ro, max_correspondence_distance,
                                    self._out_p, self._stats_ref)
        self.last_status = rc if rc >= 0 else _check(rc)
        return self._out.copy()

    def prepare_batch(self, frames, last_robot_poses, relative_wheel_odometries):
        """Marshal a queue of independent scans (DeviceFrames + their poses) once; run it with ComputeRobotMotionBatch."""
        k = len(frames)
        b = _Batch()
        b.frames = list(frames)  # keep the device buffers alive
        b.ptrs = (C.c_void_p * k)(*[f.ptr.value for f in frames])
        b.ns = (C.c_size_t * k)(*[f.n for f in frames])
        b.last = np.ascontiguousarray(np.asarray(last_robot_poses, dtype=np.float64).reshape(k, 7))
        b.rel = np.ascontiguousarray(np.asarray(relative_wheel_odometries, dtype=np.float64).reshape(k, 7))
        b.out = np.zeros((k, 7), dtype=np.float64)
        b.iterations = np.zeros(k, dtype=np.int32)
        b.count = k
        return b

    def ComputeRobotMotionBatch(self, batch, voxel_map, max_correspondence_distance):
        """kicp_register_device_batch: a queue of INDEPENDENT scans against one map; the library keeps several of them in flight
        (options "batch_queues" / "batch_depth"; 0 and 1: strictly one after the other), every pose bit-equal to
        ComputeRobotMotionDevice on that scan alone.  Returns the (count, 7) poses; batch.iterations holds the iteration counts."""
        rc = _lib.kicp_register_device_batch(self._h, voxel_map._h, batch.count, batch.ptrs, batch.ns, batch.last.ctypes.data_as(_dp),
                                             batch.rel.ctypes.data_as(_dp), max_correspondence_distance, batch.out.ctypes.data_as(_dp),
                                             batch.iterations.ctypes.data_as(C.POINTER(C.c_int)))
        self.last_status = rc if rc >= 0 else _check(rc)
        return batch.out

    def ComputeRobotMotionConcurrent(self, others, batch, voxel_map, max_correspondence_distance):
        """kicp_register_device_concurrent: the batch's INDEPENDENT scans with 1 + len(others) of them in flight, one lane per
        handle (this one and `others`).  Same poses as ComputeRobotMotionBatch, bit for bit."""
        handles = (C.c_void_p * (1 + len(others)))(self._h, *[o._h for o in others])
        rc = _lib.kicp_register_device_concurrent(handles, len(handles), voxel_map._h, batch.count, batch.ptrs, batch.ns, batch.last.ctypes.data_as(_dp),
                                                  batch.rel.ctypes.data_as(_dp), max_correspondence_distance, batch.out.ctypes.data_as(_dp),
                                                  batch.iterations.ctypes.data_as(C.POINTER(C.c_int)))
        self.last_status = rc if rc >= 0 else _check(rc)
        return batch.out

    def pass_sums(self, frame, voxel_map, pose, max_correspondence_distance):
        """One fused association+accumulation pass at a fixed pose -> [JTJ00,JTJ01,JTJ11,JTr0,JTr1,ssq,N]."""
        a, p = _d(frame)
        _, q = _d(pose)
        out = np.zeros(7, dtype=np.float64)
        _check(lib().kicp_pass_sums(self._h, voxel_map._h, p, a.size // 3, q, max_correspondence_distance, out.ctypes.data_as(_dp)))
        return out

    def pass_correspondences(self, frame, voxel_map, pose, max_correspondence_distance):
        """kicp_pass_correspondences: DataAssociation's output per source point from the pass kernel this handle registers a scan of this
        size with -> (index[n] int32, -1 = none; d2[n]; nn[n, 3])."""
        a, p = _d(frame)
        _, q = _d(pose)
        n = a.size // 3
        idx = np.empty(n, dtype=np.int32)
        d2 = np.empty(n, dtype=np.float64)
        nn = np.empty((n, 3), dtype=np.float64)
        _check(lib().kicp_pass_correspondences(self._h, voxel_map._h, p, n, q, max_correspondence_distance, idx.ctypes.data_as(C.POINTER(C.c_int32)),
                                               d2.ctypes.data_as(_dp), nn.ctypes.data_as(_dp)))
        return idx, d2, nn

    def pass_words(self, frame, voxel_map, pose, max_correspondence_distance):
        """The same pass as raw int64[24] limb words (the multi-GPU all-reduce payload; see sharding.py)."""
        a, p = _d(frame)
        _, q = _d(pose)
        out = np.zeros(24, dtype=np.int64)
        _check(lib().kicp_pass_words(self._h, voxel_map._h, p, a.size // 3, q, max_correspondence_distance,
                                     out.ctypes.data_as(C.POINTER(C.c_longlong))))
        return out

    # ---- multi-GPU ----
    def comm_init(self, nranks, rank, unique_id):
        _check(lib().kicp_reg_comm_init(self._h, nranks, rank, unique_id))

    def comm_destroy(self):
        _check(lib().kicp_reg_comm_destroy(self._h))

    def shm_init(self, nranks, rank, name):
        """Single-node multi-process mode without a device collective (see kicp.h); rank 0 first, then the others."""
        _check(lib().kicp_reg_shm_init(self._h, nranks, rank, name.encode()))

    def shm_destroy(self):
        _check(lib().kicp_reg_shm_destroy(self._h))

    def p2p_export(self, nranks, rank):
        """Peer-mailbox exchange (kicp.h): allocate this rank's mailbox, return its IPC handle (bytes) for the all-gather."""
        buf = C.create_string_buffer(P2P_HANDLE_BYTES)
        _check(lib().kicp_reg_p2p_export(self._h, nranks, rank, buf))
        return buf.raw

    def p2p_connect(self, handles):
        """handles: every rank's handle in rank order (list of bytes, or their concatenation)."""
        blob = handles if isinstance(handles, (bytes, bytearray)) else b"".join(handles)
        _check(lib().kicp_reg_p2p_connect(self._h, bytes(blob)))

    def p2p_destroy(self):
        _check(lib().kicp_reg_p2p_destroy(self._h))

    def set_allreduce(self, fn):
        """fn(device_ptr:int, count:int, stream:int) -> None: sum-all-reduce `count` doubles in place on `stream`."""
        if fn is None:
            self._cb = None
            _check(lib().kicp_reg_set_allreduce(self._h, C.cast(None, ALLREDUCE_FN), None))
            return

        def tramp(user, buf, count, stream):
            try:
                fn(buf or 0, count, stream or 0)
                return 0
            except Exception:  # noqa: BLE001 - must not unwind through C
                import traceback
                traceback.print_exc()
                return 1

        self._cb = ALLREDUCE_FN(tramp)
        _check(lib().kicp_reg_set_allreduce(self._h, self._cb, None))


class CloudLayout(C.Structure):
    """kicp_cloud_layout (include/kicp.h): where x, y, z and the per-point stamp sit inside a PointCloud2 record."""
    _fields_ = [("point_step", C.c_uint), ("offset_x", C.c_uint), ("offset_y", C.c_uint), ("offset_z", C.c_uint),
                ("stamp_datatype", C.c_int), ("offset_stamp", C.c_uint)]


FIELD_UINT32, FIELD_FLOAT32, FIELD_FLOAT64 = 6, 7, 8  # sensor_msgs::msg::PointField datatype codes


class PreSteps:
    """The pipeline's pre-steps on the GPU: kiss_icp::Preprocessor::Preprocess + transform_points, kiss_icp::VoxelDownsample
    (pipeline/KinematicICP.cpp:54-62).  Results live in numbered device buffers; frame(b) wraps one for ComputeRobotMotion."""

    def __init__(self, device=0):
        h = C.c_void_p()
        _check(lib().kicp_pre_create(device, C.byref(h)))
        self._h, self.device = h, device

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.kicp_pre_destroy(self._h)
            self._h = None

    def set_probe_limit(self, limit):
        """probe length at which the reference's tsl::robin_map grows its table (128: robin-map 0.6.x, default; 8192: 1.x)"""
        _check(lib().kicp_pre_set_probe_limit(self._h, int(limit)))

    def last_max_probe(self):
        """Largest robin-hood displacement the last VoxelDownsample's replay saw (kicp_pre_last_max_probe)."""
        return int(lib().kicp_pre_last_max_probe(self._h))

    def Preprocess(self, frame, timestamps, relative_motion, lidar_to_base, max_range, min_range, deskew, dst=0):
        a, p = _d(frame)
        t, tp = _d(timestamps if timestamps is not None else np.zeros(0))
        _, r = _d(relative_motion)
        _, e = _d(lidar_to_base)
        n = C.c_size_t()
        _check(lib().kicp_pre_preprocess(self._h, p, a.size // 3, tp, t.size, r, e, max_range, min_range, int(deskew), dst, C.byref(n)))
        return n.value

    def Ingest(self, raw, n_points, point_step, offset_x, offset_y, offset_z, stamp_datatype=0, offset_stamp=0, sensor_pose=None):
        """PointCloud2 wire-format ingest (RosUtils.cpp:30-39 PointCloud2ToEigen + TimeStampHandler.cpp:57-128): ships the raw
        message bytes and decodes them on the GPU.  Returns (min_stamp, max_stamp) in seconds (0, 0 without a stamp field)."""
        buf = np.ascontiguousarray(np.frombuffer(raw, dtype=np.uint8))
        layout = CloudLayout(point_step, offset_x, offset_y, offset_z, stamp_datatype, offset_stamp)
        q = None if sensor_pose is None else _d(sensor_pose)[1]
        lo, hi = C.c_double(), C.c_double()
        _check(lib().kicp_pre_ingest(self._h, buf.ctypes.data if buf.size else None, n_points, C.byref(layout), q, C.byref(lo), C.byref(hi)))
        return lo.value, hi.value

    def IngestAhead(self, raw, n_points, point_step, offset_x, offset_y, offset_z, stamp_datatype=0, offset_stamp=0, sensor_pose=None):
        """kicp_pre_ingest_ahead: announce the NEXT message (`raw` must be a buffer whose memory the later Ingest call passes again - a
        numpy uint8 array); it is uploaded and decoded by the next Frame() call on the current one."""
        buf = np.ascontiguousarray(np.frombuffer(raw, dtype=np.uint8))
        layout = CloudLayout(point_step, offset_x, offset_y, offset_z, stamp_datatype, offset_stamp)
        q = None if sensor_pose is None else _d(sensor_pose)[1]
        self._ahead_keep = (buf, raw)  # (borrowed by the backend until the Ingest call for the same bytes)
        _check(lib().kicp_pre_ingest_ahead(self._h, buf.ctypes.data if buf.size else None, n_points, C.byref(layout), q))

    def set_option(self, name, value):
        _check(lib().kicp_pre_set_option(self._h, name.encode(), float(value)))

    def get_option(self, name):
        return float(lib().kicp_pre_get_option(self._h, name.encode()))

    def ahead_hits(self):
        return int(lib().kicp_pre_ahead_hits(self._h))

    def PreprocessIngested(self, relative_motion, lidar_to_base, max_range, min_range, deskew, dst=0):
        _, r = _d(relative_motion)
        _, e = _d(lidar_to_base)
        n = C.c_size_t()
        _check(lib().kicp_pre_preprocess_ingested(self._h, r, e, max_range, min_range, int(deskew), dst, C.byref(n)))
        return n.value

    def ingested(self):
        """(xyz (n,3) fp64, normalised stamps (n,) or None): what PointCloud2ToEigen / ProcessTimestamps hand to RegisterFrame."""
        n, has = C.c_size_t(), C.c_int()
        _check(lib().kicp_pre_ingested(self._h, None, None, 0, C.byref(n), C.byref(has)))
        xyz, st = np.empty((n.value, 3)), np.empty(n.value)
        _check(lib().kicp_pre_ingested(self._h, xyz.ctypes.data_as(_dp), st.ctypes.data_as(_dp), n.value, C.byref(n), C.byref(has)))
        return xyz, (st if has.value else None)

    def Frame(self, frame, timestamps, relative_motion, lidar_to_base, max_range, min_range, deskew, voxel_a, voxel_b, want_frame=True):
        """kicp_pre_frame / kicp_pre_frame_ingested (frame=None: the ingested cloud): Preprocess + the two downsamples behind one
        synchronisation.  Returns (counts, preprocessed frame or None)."""
        _, r = _d(relative_motion)
        _, e = _d(lidar_to_base)
        counts = (C.c_size_t * 3)()
        if frame is None:
            n_in = lib().kicp_pre_ingested_count(self._h)
            out = np.empty((n_in, 3), dtype=np.float64) if want_frame else None
            rc = lib().kicp_pre_frame_ingested(self._h, r, e, max_range, min_range, int(deskew), voxel_a, voxel_b,
                                               out.ctypes.data_as(_dp) if want_frame and n_in else None, n_in, counts)
        else:
            a, p = _d(frame)
            t, tp = _d(timestamps if timestamps is not None else np.zeros(0))
            n_in = a.size // 3
            out = np.empty((n_in, 3), dtype=np.float64) if want_frame else None
            rc = lib().kicp_pre_frame(self._h, p, n_in, tp, t.size, r, e, max_range, min_range, int(deskew), voxel_a, voxel_b,
                                      out.ctypes.data_as(_dp) if want_frame and n_in else None, n_in, counts)
        if rc < 0:
            message = lib().kicp_last_error().decode(errors="replace")
            if want_frame and n_in:  # the backend's helper thread may hold a pointer into `out`: collect (and drop) the download before the array can go
                lib().kicp_pre_download_finish(self._h, 0, None, 0, None)
            raise KicpError(rc, message)
        self.last_status = rc
        if want_frame and n_in:
            n = C.c_size_t()
            _check(lib().kicp_pre_download_finish(self._h, 0, out.ctypes.data_as(_dp), n_in, C.byref(n)))
            out = out[:counts[0]]
        return [int(c) for c in counts], out

    def VoxelDownsample(self, src, voxel_size, dst):
        n = C.c_size_t()
        self.last_status = _check(lib().kicp_pre_voxel_downsample(self._h, src, voxel_size, dst, C.byref(n)))  # KICP_WARN_TABLE_ORDER: see kicp.h
        return n.value

    def upload(self, buffer, points):
        a, p = _d(points)
        _check(lib().kicp_pre_upload(self._h, buffer, p, a.size // 3))

    def download(self, buffer):
        n = C.c_size_t()
        _check(lib().kicp_pre_download(self._h, buffer, None, 0, C.byref(n)))
        out = np.empty((n.value, 3), dtype=np.float64)
        _check(lib().kicp_pre_download(self._h, buffer, out.ctypes.data_as(_dp), n.value, C.byref(n)))
        return out

    def download_begin(self, buffer):
        """Start copying a buffer to the host in the background; collect it with download_finish."""
        _check(lib().kicp_pre_download_begin(self._h, buffer))

    def download_finish(self, buffer):
        n = C.c_size_t()
        _check(lib().kicp_pre_download(self._h, buffer, None, 0, C.byref(n)))  # (size only)
        out = np.empty((n.value, 3), dtype=np.float64)
        _check(lib().kicp_pre_download_finish(self._h, buffer, out.ctypes.data_as(_dp), n.value, C.byref(n)))
        return out[:n.value]

    def frame(self, buffer):
        """A DeviceFrame view of a buffer (no copy; valid until the buffer is overwritten)."""
        n = C.c_size_t()
        ptr = lib().kicp_pre_device_ptr(self._h, buffer, C.byref(n))
        f = DeviceFrame.__new__(DeviceFrame)
        f.n, f.device, f.ptr, f._borrowed = n.value, self.device, C.c_void_p(ptr), True
        return f


def comm_unique_id():
    buf = C.create_string_buffer(COMM_ID_BYTES)
    _check(lib().kicp_comm_unique_id(buf))
    return buf.raw
