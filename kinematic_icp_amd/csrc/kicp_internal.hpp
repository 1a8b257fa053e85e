// kicp_internal.hpp -- what the translation units behind include/kicp.h share: error handling, tracing, the staging
// policy for caller memory, the HBM mirror of the voxel map and the handle behind kicp_map.  Host code only; the kernels
// live in kicp_kernels.hpp (registration passes), kicp_mapdev.hpp (map maintenance) and kicp_pre.hpp (pre-steps), all with
// internal linkage so that every translation unit may include what it launches.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cfloat>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/kicp.h"
#include "kicp_common.hpp"
#include "kicp_host_map.hpp"
#include "kicp_se3.hpp"

namespace kicp {
namespace host {

// KICP_TRACE=1 in the environment: every traced C-ABI call reports its wall time on stderr (debugging aid)
bool env_flag(const char *name);
extern const bool g_trace;
// KICP_ROCTX=1: the same calls open a roctx range (libroctx64, bound at run time), so that a rocprofv3 --marker-trace run shows
// pre-steps, registration passes and map updates as named spans around their kernels
extern const bool g_roctx;
void roctx_push(const char *name);
void roctx_pop();
extern thread_local std::chrono::steady_clock::time_point g_trace_t0;
struct TraceScope {
    const char *name;
    std::chrono::steady_clock::time_point t0;
    explicit TraceScope(const char *n) : name(n) {
        if (g_trace) t0 = g_trace_t0 = std::chrono::steady_clock::now();
        if (g_roctx) roctx_push(n);
    }
    ~TraceScope() {
        if (g_roctx) roctx_pop();
        if (g_trace) std::fprintf(stderr, "[kicp] %-32s %9.3f ms\n", name, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    }
};
// a point inside a traced call: microseconds since the innermost traced call of this thread began
extern thread_local std::chrono::steady_clock::time_point g_trace_t0;
inline void trace_lap(const char *what) {
    if (g_trace) std::fprintf(stderr, "[kicp]     +%7.1f us  %s\n", std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - g_trace_t0).count(), what);
}
#define KICP_TRACE_CALL() ::kicp::host::TraceScope trace_scope_(__func__)
struct RoctxScope {  // a named span inside a call (one ICP pass: dispatch -> rows -> solve)
    explicit RoctxScope(const char *n) {
        if (g_roctx) roctx_push(n);
    }
    ~RoctxScope() {
        if (g_roctx) roctx_pop();
    }
};

// last error message of the calling thread (kicp_last_error) and the one way to report a failure
std::string &last_error();
int fail(int code, const std::string &msg);
// The host side of a frame is a few dozen polls of, and copies through, pinned host memory the GPU writes over PCIe: on a two-socket
// host all of it is ~20 % slower (and bimodal from run to run) when the memory or the library's helper threads sit on the socket the
// GPU is NOT attached to.  pinned_alloc is hipHostMalloc with the calling thread's memory policy set to the GPU's NUMA node for the
// duration of the call (sysfs: /sys/bus/pci/devices/<bdf>/numa_node; nothing happens where that is unknown or there is one node);
// bind_thread_near_gpu moves the CALLING thread - only ever a thread the library created - onto that node's CPUs (local_cpulist,
// intersected with what the process may use).  KICP_NUMA=0 turns both off.  The caller's own threads are the caller's to place
// (numactl --cpunodebind).
hipError_t pinned_alloc(void **ptr, size_t bytes, unsigned int flags);
void bind_thread_near_gpu(int device);
int device_locality(int device, int *node, char *cpulist, size_t cap);
// One job at a time on a thread of its own (the pre-steps' look-ahead upload, the map update's launches): e.g. the look-ahead upload's host work (a 2 MB copy into the staging buffer and a dozen API
// calls, ~85 us) runs beside the calling thread's own queueing of the frame's kernels (~100 us of API calls) instead of after it.
struct JobThread {
    std::thread th;
    std::mutex m;
    std::condition_variable cv;
    std::function<int()> job;
    int state = 0;  // 0 idle | 1 posted | 2 done | -1 leaving
    int result = 0;
    std::string message;  // the job's error text (kicp_last_error is per thread: the waiting thread takes it over)
    int device = 0;
    void run() {
        hipSetDevice(device);
        bind_thread_near_gpu(device);
        std::unique_lock<std::mutex> lock(m);
        for (;;) {
            cv.wait(lock, [this] { return state == 1 || state == -1; });
            if (state == -1) return;
            lock.unlock();
            const int rc = job();
            lock.lock();
            result = rc, state = 2;
            if (rc < 0) message = last_error();
            cv.notify_all();
        }
    }
    void post(int dev, std::function<int()> fn) {
        device = dev;
        if (!th.joinable()) th = std::thread([this] { run(); });
        {
            std::lock_guard<std::mutex> lock(m);
            job = std::move(fn), state = 1;
        }
        cv.notify_all();
    }
    int wait() {  // (only after post)
        std::unique_lock<std::mutex> lock(m);
        cv.wait(lock, [this] { return state == 2; });
        state = 0;
        if (result < 0) last_error() = message;
        return result;
    }
    void stop() {
        if (!th.joinable()) return;
        {
            std::unique_lock<std::mutex> lock(m);
            cv.wait(lock, [this] { return state != 1; });
            state = -1;
        }
        cv.notify_all();
        th.join();
    }
};
#define HIP_TRY(expr)                                                                                                        \
    do {                                                                                                                     \
        hipError_t e_ = (expr);                                                                                              \
        if (e_ != hipSuccess)                                                                                                \
            return ::kicp::host::fail(KICP_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_) + " (" __FILE__ ":" + \
                                                        std::to_string(__LINE__) + ")");                                    \
    } while (0)

inline Pose pose_from(const double p[7]) { return Pose{p[0], p[1], p[2], p[3], p[4], p[5], p[6]}; }
inline void pose_to(const Pose &T, double p[7]) { p[0] = T.qx, p[1] = T.qy, p[2] = T.qz, p[3] = T.qw, p[4] = T.tx, p[5] = T.ty, p[6] = T.tz; }
inline int set_device(int device) {
    HIP_TRY(hipSetDevice(device));
    return KICP_OK;
}

// pinned staging buffer for uploads of pageable caller memory (see staged_upload)
struct HostStage {
    unsigned char *p = nullptr;
    unsigned char *dev = nullptr;  // the same memory as a kernel sees it (nullptr: not mapped on this platform)
    size_t cap = 0;
    hipEvent_t done = nullptr;  // recorded behind the last asynchronous copy that reads the buffer
    bool pending = false;       // such a copy may still be in flight: wait on `done` before writing the buffer again
    void release() {
        if (done) hipEventSynchronize(done), hipEventDestroy(done);
        if (p) hipHostFree(p);
        p = nullptr, dev = nullptr, cap = 0, done = nullptr, pending = false;
    }
};
// (policy and implementation: kicp_core.hip)
int stage_reserve(HostStage &hs, size_t bytes, hipStream_t stream);
int stage_begin(HostStage &hs, size_t bytes, hipStream_t stream);
int stage_end(HostStage &hs, hipStream_t stream);
int staged_upload(HostStage &hs, size_t offset, void *dst, const void *src, size_t bytes, hipStream_t stream);
int staged_download(HostStage &hs, void *dst, const void *src, size_t bytes, hipStream_t stream);

struct DeviceMirror {
    int device = -1;
    Slot *d_table = nullptr;
    double *d_pool = nullptr;
    MirrorPoint *d_pool16 = nullptr;
    size_t table_slots = 0, pool_doubles = 0;  // allocated sizes
    size_t live_slots = 0;                     // table size the mirror currently represents
    uint64_t synced_epoch = ~0ull, synced_generation = ~0ull;
    // staging for delta uploads (device)
    uint2 *d_stage = nullptr;
    uint32_t *d_index = nullptr;
    size_t stage_words = 0, index_cap = 0;
    size_t last_upload_bytes = 0;
    int last_upload_full = 1;
    MapView view{};
    // device-side maintenance (kicp_mapdev.hpp): per-slot and per-update scratch
    unsigned long long *d_keys64 = nullptr;
    uint32_t *d_cnt = nullptr, *d_seg_start = nullptr, *d_free_list = nullptr;
    DevMapCounters *d_ctr = nullptr;
    size_t aux_slots = 0, free_cap = 0;
    unsigned long long *h_ctr = nullptr;  // pinned, host-coherent: [0..4] the counters of an update whose end the caller collects later, [7] its sequence number (k_up_publish)
    unsigned long long *h_ctr_dev = nullptr, ctr_seq = 0;
    bool ctr_clean = false;  // the per-update counters in d_ctr are zero (left so by k_up_publish): the next frame-sized update skips its memset
    double *d_world = nullptr;
    uint32_t *d_slot_of = nullptr, *d_order = nullptr, *d_touched = nullptr;
    size_t upd_cap = 0;
    HostStage stage;  // pinned staging for transfers from / to caller memory (queries, Pointcloud)
    // Pointcloud() from the device copy
    double *d_pc = nullptr;
    uint32_t *d_pc_blocks = nullptr;  // per-256-slot-block counts / offsets, then the total
    size_t pc_points = 0, pc_blocks = 0;
};
}  // namespace host
}  // namespace kicp

struct kicp_map {
    kicp::HostMap host;
    kicp::host::DeviceMirror mirror;
    // set while the HBM copy is newer than the host copy (after a device-side Update); the host copy is refreshed on
    // demand by ensure_host_current().  Counters of the device state for the cheap queries:
    bool device_ahead = false;
    kicp::DevMapCounters dev{};
    int last_update_on_device = 0;
    unsigned long long device_updates = 0;  // updates that ran (and were collected) on the GPU so far (kicp_map_device_updates)
    // a voxel coordinate beyond +-2^20 was seen: the packed keys of the device-side maintenance cannot hold it, so this map's
    // updates stay on the host from now on (until Clear); registration and queries on the device are unaffected
    bool host_updates_only = false;
    // kicp_map_update_pose_device_begin: the update's kernels and the copy of its counters are queued, nothing has been waited for;
    // kicp_map_update_finish (or any other call on the map) collects it.  The points stay borrowed until then (host fallback).
    bool pending_update = false;
    const double *pending_points = nullptr;
    size_t pending_n = 0;
    kicp::Pose pending_pose{};
    double pending_origin[3] = {0.0, 0.0, 0.0};
    bool pending_has_origin = false;
    // preferred device for bulk host-side insertions (kicp_map_set_device; -1 = none: host insertion) and their staging
    int bulk_device = -1;
    double *d_bulk = nullptr;
    size_t bulk_cap = 0;
    kicp_map(double vs, double md, uint32_t cap) : host(vs, md, cap) {}
};

namespace kicp {
namespace host {
// make the HBM mirror on `device` current / bring the host copy up to date after device-side updates (kicp_map.hip)
int map_sync(kicp_map *map, int device, hipStream_t stream);
int ensure_host_current(kicp_map *map);
int map_finish_pending(kicp_map *map);  // collect an update begun with kicp_map_update_pose_device_begin (no-op without one)
}  // namespace host
}  // namespace kicp
