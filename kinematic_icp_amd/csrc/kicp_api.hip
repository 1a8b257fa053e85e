// kicp_api.hip -- implementation of include/kicp.h: handles, HBM mirror of the voxel map, the registration loop
// (kernel enqueue / early exit / read-back), and the optional RCCL all-reduce.  gfx950 only; no CPU fallback.
#include <dlfcn.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>  // declarations only; the library is bound at run time (see CommApi)

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/kicp.h"
#include "kicp_host_map.hpp"
#include "kicp_kernels.hpp"
#include "kicp_mapdev.hpp"
#include "kicp_pre.hpp"

namespace {
using namespace kicp;

// KICP_TRACE=1 in the environment: every traced C-ABI call reports its wall time on stderr (debugging aid)
// (plain function on purpose: hipcc gave two namespace-scope initialiser lambdas of this shape the same closure symbol and
// ran the first one's body for both)
bool env_flag(const char *name) {
    const char *e = std::getenv(name);
    return e && *e && *e != '0';
}
const bool g_trace = env_flag("KICP_TRACE");
struct TraceScope {
    const char *name;
    std::chrono::steady_clock::time_point t0;
    explicit TraceScope(const char *n) : name(n) {
        if (g_trace) t0 = std::chrono::steady_clock::now();
    }
    ~TraceScope() {
        if (g_trace) std::fprintf(stderr, "[kicp] %-32s %9.3f ms\n", name, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    }
};
#define KICP_TRACE_CALL() TraceScope trace_scope_(__func__)

thread_local std::string g_error;
int fail(int code, const std::string &msg) {
    g_error = msg;
    return code;
}
#define HIP_TRY(expr)                                                                                          \
    do {                                                                                                       \
        hipError_t e_ = (expr);                                                                                \
        if (e_ != hipSuccess)                                                                                  \
            return fail(KICP_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_) + " (" __FILE__ ":" + \
                                          std::to_string(__LINE__) + ")");                                    \
    } while (0)

// host twin of kicp::limbs_to_double (same operations, so host- and device-side solves see the same doubles)
double host_limbs_to_double(const long long l[3]) {
    unsigned __int128 t = static_cast<unsigned __int128>(static_cast<__int128>(l[0]));
    t += static_cast<unsigned __int128>(static_cast<__int128>(l[1])) << 40;
    t += static_cast<unsigned __int128>(static_cast<__int128>(l[2])) << 80;
    const bool neg = static_cast<__int128>(t) < 0;
    if (neg) t = ~t + 1;
    const double mag = static_cast<double>(static_cast<unsigned long long>(t >> 64)) * 18446744073709551616.0 +
                       static_cast<double>(static_cast<unsigned long long>(t));
    return (neg ? -mag : mag) / kFixScale;
}

Pose pose_from(const double p[7]) { return Pose{p[0], p[1], p[2], p[3], p[4], p[5], p[6]}; }
void pose_to(const Pose &T, double p[7]) { p[0] = T.qx, p[1] = T.qy, p[2] = T.qz, p[3] = T.qw, p[4] = T.tx, p[5] = T.ty, p[6] = T.tz; }

// ---- RCCL, bound lazily so that single-GPU users never load it ------------------------------------------------
struct CommApi {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    bool load(std::string &err) {
        if (handle) return true;
        // prefer an RCCL that is already in the process (e.g. the one torch.distributed loaded)
        const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char *nm : names)
            if ((handle = dlopen(nm, RTLD_NOW | RTLD_NOLOAD))) break;
        if (!handle)
            for (const char *nm : names)
                if ((handle = dlopen(nm, RTLD_NOW | RTLD_GLOBAL))) break;
        if (!handle) {
            err = std::string("cannot load librccl: ") + dlerror();
            return false;
        }
        GetUniqueId = reinterpret_cast<decltype(GetUniqueId)>(dlsym(handle, "ncclGetUniqueId"));
        CommInitRank = reinterpret_cast<decltype(CommInitRank)>(dlsym(handle, "ncclCommInitRank"));
        CommDestroy = reinterpret_cast<decltype(CommDestroy)>(dlsym(handle, "ncclCommDestroy"));
        AllReduce = reinterpret_cast<decltype(AllReduce)>(dlsym(handle, "ncclAllReduce"));
        GetErrorString = reinterpret_cast<decltype(GetErrorString)>(dlsym(handle, "ncclGetErrorString"));
        if (!GetUniqueId || !CommInitRank || !CommDestroy || !AllReduce || !GetErrorString) {
            err = "librccl lacks an expected symbol";
            return false;
        }
        return true;
    }
};
CommApi g_comm;

// pinned staging buffer for uploads of pageable caller memory (see staged_upload)
struct HostStage {
    unsigned char *p = nullptr;
    size_t cap = 0;
    void release() {
        if (p) hipHostFree(p);
        p = nullptr, cap = 0;
    }
};
struct DeviceMirror {
    int device = -1;
    Slot *d_table = nullptr;
    double *d_pool = nullptr;
    float4 *d_pool32 = nullptr;
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
    double *d_world = nullptr;
    uint32_t *d_slot_of = nullptr, *d_order = nullptr, *d_touched = nullptr;
    size_t upd_cap = 0;
    HostStage stage;  // pinned staging for transfers from / to caller memory (queries, Pointcloud)
    // Pointcloud() from the device copy
    double *d_pc = nullptr;
    uint32_t *d_pc_blocks = nullptr;  // per-256-slot-block counts / offsets, then the total
    size_t pc_points = 0, pc_blocks = 0;
};
}  // namespace

struct kicp_map {
    HostMap host;
    DeviceMirror mirror;
    // set while the HBM copy is newer than the host copy (after a device-side Update); the host copy is refreshed on
    // demand by ensure_host_current().  Counters of the device state for the cheap queries:
    bool device_ahead = false;
    DevMapCounters dev{};
    int last_update_on_device = 0;
    kicp_map(double vs, double md, uint32_t cap) : host(vs, md, cap) {}
};

struct BinBuffers {
    unsigned long long *cell_keys = nullptr;
    uint32_t *cell_count = nullptr, *cell_start = nullptr, *cell_list = nullptr, *counters = nullptr;
    uint2 *qinfo = nullptr, *items = nullptr;
    double *sorted_src = nullptr;
    uint32_t mask = 0;
    size_t cap_n = 0;
};

struct kicp_reg {
    kicp_reg_config cfg{};
    int device = 0;
    int num_cus = 256;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    hipEvent_t evp[2 * KICP_MAX_LOG_PASSES] = {};  // per-pass events ("timing" == 2), created on first use
    IcpState *d_state = nullptr;
    HostRecord *rec = nullptr;    // host-mapped pinned result record (host view)
    HostRecord *d_rec = nullptr;  // same memory, device view
    unsigned long long call_id = 0;
    unsigned long long *d_partials = nullptr;  // limb rows of the reduction tree
    unsigned int *d_tickets = nullptr;
    size_t partial_blocks = 0;
    // mode 4 hand-off: tagged rows of the first-level groups in host-mapped pinned memory, added up by the host
    unsigned long long *rows = nullptr, *d_rows = nullptr;  // host / device view
    size_t rows_groups = 0;
    uint32_t tag = 0;       // tag of the last pass (1..65535)
    int group_rows = 1;     // option "group_rows": 1 = mode 4 (default), 0 = the device folds everything (mode 2)
    double *d_frame = nullptr;  // device copy of host frames
    size_t frame_cap = 0;
    HostStage stage;            // pinned staging for transfers from / to caller memory
    BinBuffers bin;
    // options
    int pass_kernel = 3;  // 3 fp32-mirror gather (default), 0 fp64 gather, 1 lds (given order), 2 binned by cell
    int block = 128;      // workgroup size of variants 0/3
    int loop_mode = 1;    // 0 enqueue every iteration up front; 1 stepped: keep one iteration queued ahead, poll the stop flag
    int wait_mode = 0;    // 0 poll the host-mapped record; 1 hipStreamSynchronize
    int waves_per_cu = 12; // persistent grid of variants 1/2
    int timing = 0;       // record HIP events around the call -> stats.gpu_ms
    int dbg = 0;
    int query_every = 64;  // polls between hipStreamQuery calls while waiting
    int speculate = 0;     // stepped loop: queue iteration it+1 before the stop flag of it is known (adapts to the last scan)
    int lanes_per_query = 0;  // variant 3: sub-lanes sharing one query (1, 2 or 4); 0 = by scan size
    int host_solve = 1;    // 1: the pass kernel publishes the limb totals and the host solves (default); 0: device-side solve
    // multi-GPU
    ncclComm_t comm = nullptr;
    int nranks = 1, rank = 0;
    kicp_allreduce_fn allreduce_fn = nullptr;
    void *allreduce_user = nullptr;
    // node-wide shared segment (multi-process, no device collective)
    struct ShmSlot {
        unsigned long long seq;
        long long words[kReduceWords];
        unsigned long long pad[7];  // 256 bytes
    };
    ShmSlot *shm = nullptr;    // host view: [2 buffers][nranks]
    ShmSlot *d_shm = nullptr;  // device view of the same memory
    size_t shm_bytes = 0;
    unsigned long long shm_step = 0;  // hand-offs issued so far (same on every rank)
    std::string shm_name;
};

namespace {

int set_device(int device) {
    HIP_TRY(hipSetDevice(device));
    return KICP_OK;
}

// Transfers from / to caller memory never hand the caller's pointer to the HIP runtime.  The runtime pins a pageable
// buffer for the DMA and remembers pinned ranges by address; with buffers that live at new or recycled addresses every
// frame (a new message, a new std::vector) a 4 MB scan took 17-27 ms to upload instead of 0.16 ms in most processes we
// measured (always in multiples of ~9 ms, and whether a process was hit depended on its allocation pattern only).  So
// both directions go through a pinned staging buffer owned by the handle: CPU copy in 1 MB pieces (~0.03 ms each), each
// followed by its asynchronous DMA - about 0.2 ms for that scan, every time.  KICP_DIRECT_UPLOAD=1 restores the direct
// DMA for callers that pass pinned (hipHostMalloc / hipHostRegister) memory.  The caller drains `stream` before the staging
// buffer is used again (every entry point here ends in a sync).
const bool g_direct_upload = env_flag("KICP_DIRECT_UPLOAD");
int stage_reserve(HostStage &hs, size_t bytes, hipStream_t stream) {
    if (bytes <= hs.cap) return KICP_OK;
    HIP_TRY(hipStreamSynchronize(stream));
    hs.release();
    const size_t want = bytes + bytes / 2 + (1u << 20);
    HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&hs.p), want, hipHostMallocDefault));
    hs.cap = want;
    return KICP_OK;
}
constexpr size_t kStagePiece = 1u << 20;
// `offset`: where in the staging buffer this transfer may start (several may be in flight within one call; reserve first)
int staged_upload(HostStage &hs, size_t offset, void *dst, const void *src, size_t bytes, hipStream_t stream) {
    if (bytes == 0) return KICP_OK;
    if (g_direct_upload) {
        HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, stream));
        return KICP_OK;
    }
    if (offset == 0)
        if (int rc = stage_reserve(hs, bytes, stream)) return rc;
    if (offset + bytes > hs.cap) return fail(KICP_ERR_ARG, "staging buffer too small for a follow-up transfer");
    for (size_t off = 0; off < bytes; off += kStagePiece) {
        const size_t len = std::min(kStagePiece, bytes - off);
        std::memcpy(hs.p + offset + off, static_cast<const unsigned char *>(src) + off, len);
        HIP_TRY(hipMemcpyAsync(static_cast<unsigned char *>(dst) + off, hs.p + offset + off, len, hipMemcpyHostToDevice, stream));
    }
    return KICP_OK;
}
// device -> caller memory; returns with the data in place
int staged_download(HostStage &hs, void *dst, const void *src, size_t bytes, hipStream_t stream) {
    if (bytes == 0) return KICP_OK;
    if (g_direct_upload) {
        HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        return KICP_OK;
    }
    if (int rc = stage_reserve(hs, bytes, stream)) return rc;
    HIP_TRY(hipMemcpyAsync(hs.p, src, bytes, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    std::memcpy(dst, hs.p, bytes);
    return KICP_OK;
}

int ensure_host_current(kicp_map *map);
// release every device buffer of a mirror (on its own device) and reset it
void free_mirror(DeviceMirror &mr) {
    if (mr.device >= 0) {
        hipSetDevice(mr.device);
        hipFree(mr.d_table), hipFree(mr.d_pool), hipFree(mr.d_pool32), hipFree(mr.d_stage), hipFree(mr.d_index);
        hipFree(mr.d_keys64), hipFree(mr.d_cnt), hipFree(mr.d_seg_start), hipFree(mr.d_free_list), hipFree(mr.d_ctr);
        hipFree(mr.d_world), hipFree(mr.d_slot_of), hipFree(mr.d_order), hipFree(mr.d_touched);
        hipFree(mr.d_pc), hipFree(mr.d_pc_blocks);
        mr.stage.release();
    }
    mr = DeviceMirror{};
}

// per-slot helper arrays, free list and counters of the device-side maintenance, rebuilt after every upload
int sync_aux(kicp_map *map, hipStream_t stream) {
    DeviceMirror &mr = map->mirror;
    const HostMap &h = map->host;
    const size_t slots = h.table().size();
    if (slots != mr.aux_slots) {
        hipFree(mr.d_keys64), hipFree(mr.d_cnt), hipFree(mr.d_seg_start);
        mr.d_keys64 = nullptr, mr.d_cnt = nullptr, mr.d_seg_start = nullptr;
        HIP_TRY(hipMalloc(&mr.d_keys64, slots * 8));
        HIP_TRY(hipMalloc(&mr.d_cnt, slots * 4));
        HIP_TRY(hipMalloc(&mr.d_seg_start, slots * 4));
        HIP_TRY(hipMemsetAsync(mr.d_cnt, 0, slots * 4, stream));
        mr.aux_slots = slots;
    }
    const size_t bucket_cap = mr.pool_doubles / (static_cast<size_t>(h.cap()) * 3);
    if (bucket_cap > mr.free_cap) {
        hipFree(mr.d_free_list);
        mr.d_free_list = nullptr;
        HIP_TRY(hipMalloc(&mr.d_free_list, (bucket_cap + 1) * 4));
        mr.free_cap = bucket_cap;
    }
    if (!mr.d_ctr) HIP_TRY(hipMalloc(&mr.d_ctr, sizeof(DevMapCounters)));
    hipLaunchKernelGGL(k_build_keys64, dim3(static_cast<uint32_t>(std::min<size_t>((slots + 255) / 256, 4096))), dim3(256), 0, stream, mr.d_table,
                       static_cast<uint32_t>(slots), mr.d_keys64);
    DevMapCounters c{};
    c.n_points = h.num_points(), c.n_voxels = static_cast<uint32_t>(h.num_voxels()), c.n_entries = static_cast<uint32_t>(h.num_entries());
    c.n_buckets_hi = static_cast<uint32_t>(h.buckets_in_use_hi()), c.free_count = static_cast<uint32_t>(h.free_list().size());
    if (c.free_count) HIP_TRY(hipMemcpyAsync(mr.d_free_list, h.free_list().data(), c.free_count * 4, hipMemcpyHostToDevice, stream));
    HIP_TRY(hipMemcpyAsync(mr.d_ctr, &c, sizeof c, hipMemcpyHostToDevice, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    map->dev = c;
    return KICP_OK;
}

// scatter `rows` staged rows of `row_words` 8-byte words each into dst at the given row indices
int upload_rows(DeviceMirror &mr, const std::vector<uint2> &staged, const std::vector<uint32_t> &index, uint32_t row_words, void *dst,
                hipStream_t stream) {
    if (index.empty()) return KICP_OK;
    if (staged.size() > mr.stage_words) {
        if (mr.d_stage) HIP_TRY(hipFree(mr.d_stage));
        mr.d_stage = nullptr;
        mr.stage_words = staged.size() + staged.size() / 2;
        HIP_TRY(hipMalloc(&mr.d_stage, mr.stage_words * sizeof(uint2)));
    }
    if (index.size() > mr.index_cap) {
        if (mr.d_index) HIP_TRY(hipFree(mr.d_index));
        mr.d_index = nullptr;
        mr.index_cap = index.size() + index.size() / 2;
        HIP_TRY(hipMalloc(&mr.d_index, mr.index_cap * sizeof(uint32_t)));
    }
    HIP_TRY(hipMemcpyAsync(mr.d_stage, staged.data(), staged.size() * sizeof(uint2), hipMemcpyHostToDevice, stream));
    HIP_TRY(hipMemcpyAsync(mr.d_index, index.data(), index.size() * sizeof(uint32_t), hipMemcpyHostToDevice, stream));
    const size_t total = staged.size();
    const uint32_t grid = static_cast<uint32_t>(std::min<size_t>((total + 255) / 256, 4096));
    hipLaunchKernelGGL(k_scatter_rows, dim3(grid), dim3(256), 0, stream, mr.d_stage, mr.d_index, static_cast<uint32_t>(index.size()), row_words,
                       static_cast<uint2 *>(dst));
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(stream));  // the staging vectors are reused by the caller
    mr.last_upload_bytes += staged.size() * sizeof(uint2) + index.size() * sizeof(uint32_t);
    return KICP_OK;
}

int map_sync(kicp_map *map, int device, hipStream_t stream) {
    DeviceMirror &mr = map->mirror;
    HostMap &h = map->host;
    if (map->device_ahead) {
        if (mr.device == device) return KICP_OK;  // the HBM copy is the current one
        if (int rc = ensure_host_current(map)) return rc;
    }
    if (mr.device == device && mr.synced_epoch == h.epoch()) return KICP_OK;
    if (int rc = set_device(device)) return rc;
    if (mr.device != device && mr.device >= 0) {  // mirror lives on another GPU: drop it
        free_mirror(mr);
        hipSetDevice(device);
    }
    mr.device = device;
    const size_t slots = h.table().size();
    const uint32_t cap = h.cap();
    const size_t pool_doubles = h.buckets_in_use_hi() * static_cast<size_t>(cap) * 3;
    bool full = mr.synced_generation != h.generation() || mr.live_slots != slots;
    if (slots > mr.table_slots) {
        if (mr.d_table) HIP_TRY(hipFree(mr.d_table));
        mr.d_table = nullptr;
        HIP_TRY(hipMalloc(&mr.d_table, slots * sizeof(Slot)));
        mr.table_slots = slots;
        full = true;
    }
    if (pool_doubles > mr.pool_doubles) {  // grow the pools, keeping what is already there (device-side copy)
        const size_t want = pool_doubles + pool_doubles / 2 + 3 * 1024;
        double *np = nullptr;
        float4 *np32 = nullptr;
        HIP_TRY(hipMalloc(&np, want * sizeof(double)));
        HIP_TRY(hipMalloc(&np32, want / 3 * sizeof(float4)));
        if (mr.d_pool && !full) {
            HIP_TRY(hipMemcpyAsync(np, mr.d_pool, mr.pool_doubles * sizeof(double), hipMemcpyDeviceToDevice, stream));
            HIP_TRY(hipMemcpyAsync(np32, mr.d_pool32, mr.pool_doubles / 3 * sizeof(float4), hipMemcpyDeviceToDevice, stream));
            HIP_TRY(hipStreamSynchronize(stream));
        }
        if (mr.d_pool) HIP_TRY(hipFree(mr.d_pool));
        if (mr.d_pool32) HIP_TRY(hipFree(mr.d_pool32));
        mr.d_pool = np, mr.d_pool32 = np32, mr.pool_doubles = want;
    }
    mr.last_upload_bytes = 0;
    // delta only pays off while the changed part is small
    if (!full && (h.dirty_slots().size() * 4 > slots || h.dirty_buckets().size() * 2 > h.buckets_in_use_hi())) full = true;
    if (full) {
        HIP_TRY(hipMemcpyAsync(mr.d_table, h.table().data(), slots * sizeof(Slot), hipMemcpyHostToDevice, stream));
        if (pool_doubles) HIP_TRY(hipMemcpyAsync(mr.d_pool, h.pool().data(), pool_doubles * sizeof(double), hipMemcpyHostToDevice, stream));
        if (pool_doubles) HIP_TRY(hipMemcpyAsync(mr.d_pool32, h.pool32().data(), pool_doubles / 3 * sizeof(float4), hipMemcpyHostToDevice, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        mr.last_upload_bytes = slots * sizeof(Slot) + pool_doubles * sizeof(double) + pool_doubles / 3 * sizeof(float4);
    } else {
        // gather the changed rows on the host, ship them with their indices, scatter on the device
        std::vector<uint2> staged;
        const std::vector<uint32_t> &ds = h.dirty_slots(), &db = h.dirty_buckets();
        staged.resize(ds.size() * (sizeof(Slot) / 8));
        for (size_t i = 0; i < ds.size(); ++i) std::memcpy(&staged[i * (sizeof(Slot) / 8)], &h.table()[ds[i]], sizeof(Slot));
        if (int rc = upload_rows(mr, staged, ds, sizeof(Slot) / 8, mr.d_table, stream)) return rc;
        staged.resize(db.size() * static_cast<size_t>(cap) * 3);
        for (size_t i = 0; i < db.size(); ++i)
            std::memcpy(&staged[i * static_cast<size_t>(cap) * 3], &h.pool()[static_cast<size_t>(db[i]) * cap * 3], static_cast<size_t>(cap) * 24);
        if (int rc = upload_rows(mr, staged, db, cap * 3, mr.d_pool, stream)) return rc;
        staged.resize(db.size() * static_cast<size_t>(cap) * 2);
        for (size_t i = 0; i < db.size(); ++i)
            std::memcpy(&staged[i * static_cast<size_t>(cap) * 2], &h.pool32()[static_cast<size_t>(db[i]) * cap * 4], static_cast<size_t>(cap) * 16);
        if (int rc = upload_rows(mr, staged, db, cap * 2, mr.d_pool32, stream)) return rc;
    }
    mr.last_upload_full = full ? 1 : 0;
    if (int rc = sync_aux(map, stream)) return rc;
    h.mark_synced(full);
    mr.view = MapView{mr.d_table, static_cast<uint32_t>(slots - 1), mr.d_pool, mr.d_pool32, cap, h.voxel_size()};
    mr.synced_epoch = h.epoch(), mr.synced_generation = h.generation(), mr.live_slots = slots;
    return KICP_OK;
}

// bring the host copy up to date after device-side updates: same layouts, so this is a plain download
int ensure_host_current(kicp_map *map) {
    if (!map->device_ahead) return KICP_OK;
    TraceScope trace_scope_("  (host copy refreshed from HBM)");
    DeviceMirror &mr = map->mirror;
    if (int rc = set_device(mr.device)) return rc;
    HIP_TRY(hipDeviceSynchronize());
    DevMapCounters c{};
    HIP_TRY(hipMemcpy(&c, mr.d_ctr, sizeof c, hipMemcpyDeviceToHost));
    const uint32_t cap = map->host.cap();
    std::vector<Slot> table(mr.live_slots);
    std::vector<double> pool(static_cast<size_t>(c.n_buckets_hi) * cap * 3);
    std::vector<float> pool32(static_cast<size_t>(c.n_buckets_hi) * cap * 4);
    std::vector<uint32_t> free_list(c.free_count);
    HIP_TRY(hipMemcpy(table.data(), mr.d_table, table.size() * sizeof(Slot), hipMemcpyDeviceToHost));
    if (!pool.empty()) HIP_TRY(hipMemcpy(pool.data(), mr.d_pool, pool.size() * 8, hipMemcpyDeviceToHost));
    if (!pool32.empty()) HIP_TRY(hipMemcpy(pool32.data(), mr.d_pool32, pool32.size() * 4, hipMemcpyDeviceToHost));
    if (!free_list.empty()) HIP_TRY(hipMemcpy(free_list.data(), mr.d_free_list, free_list.size() * 4, hipMemcpyDeviceToHost));
    map->host.Adopt(std::move(table), std::move(pool), std::move(pool32), c.n_buckets_hi, std::move(free_list));
    map->host.mark_synced(true);
    mr.synced_epoch = map->host.epoch(), mr.synced_generation = map->host.generation();  // the mirror already holds this state
    map->device_ahead = false;
    return KICP_OK;
}

// make the device pools (and the free-list stack) hold at least `want_buckets` buckets, keeping their contents
int grow_pools(kicp_map *map, size_t want_buckets) {
    DeviceMirror &mr = map->mirror;
    const uint32_t cap = map->host.cap();
    const size_t have = mr.pool_doubles / (static_cast<size_t>(cap) * 3);
    if (want_buckets <= have && have <= mr.free_cap) return KICP_OK;
    const size_t buckets = std::max(want_buckets + want_buckets / 2 + 1024, have);
    const size_t doubles = buckets * cap * 3;
    double *np = nullptr;
    float4 *np32 = nullptr;
    uint32_t *nf = nullptr;
    HIP_TRY(hipMalloc(&np, doubles * sizeof(double)));
    HIP_TRY(hipMalloc(&np32, doubles / 3 * sizeof(float4)));
    HIP_TRY(hipMalloc(&nf, (buckets + 1) * 4));
    if (mr.d_pool) HIP_TRY(hipMemcpy(np, mr.d_pool, mr.pool_doubles * sizeof(double), hipMemcpyDeviceToDevice));
    if (mr.d_pool32) HIP_TRY(hipMemcpy(np32, mr.d_pool32, mr.pool_doubles / 3 * sizeof(float4), hipMemcpyDeviceToDevice));
    if (mr.d_free_list && mr.free_cap) HIP_TRY(hipMemcpy(nf, mr.d_free_list, std::min(mr.free_cap, buckets) * 4, hipMemcpyDeviceToDevice));
    hipFree(mr.d_pool), hipFree(mr.d_pool32), hipFree(mr.d_free_list);
    mr.d_pool = np, mr.d_pool32 = np32, mr.d_free_list = nf, mr.pool_doubles = doubles, mr.free_cap = buckets;
    mr.view.pool = mr.d_pool, mr.view.pool32 = mr.d_pool32;
    return KICP_OK;
}

int ensure_update_scratch(DeviceMirror &mr, size_t n) {
    if (n <= mr.upd_cap) return KICP_OK;
    hipFree(mr.d_world), hipFree(mr.d_slot_of), hipFree(mr.d_order), hipFree(mr.d_touched);
    mr.d_world = nullptr, mr.d_slot_of = nullptr, mr.d_order = nullptr, mr.d_touched = nullptr;
    const size_t cap = n + n / 4 + 1024;
    HIP_TRY(hipMalloc(&mr.d_world, cap * 24));
    HIP_TRY(hipMalloc(&mr.d_slot_of, cap * 4));
    HIP_TRY(hipMalloc(&mr.d_order, cap * 4));
    HIP_TRY(hipMalloc(&mr.d_touched, cap * 4));
    mr.upd_cap = cap;
    return KICP_OK;
}

// Move the live entries of the device table into a fresh table with room for `extra_entries` more at a load factor of
// at most 0.25 (the host map's ReserveEntries rule).  The HBM copy becomes the authoritative one.
int device_rehash(kicp_map *map, size_t extra_entries) {
    DeviceMirror &mr = map->mirror;
    hipStream_t st = nullptr;
    const size_t old_slots = mr.live_slots;
    HIP_TRY(hipMemsetAsync(&mr.d_ctr->touched, 0, 8, st));  // touched (borrowed as the live counter) + error
    const uint32_t grid_old = static_cast<uint32_t>(std::min<size_t>((old_slots + 255) / 256, 8192));
    hipLaunchKernelGGL(k_rehash_count, dim3(grid_old), dim3(256), 0, st, mr.d_table, static_cast<uint32_t>(old_slots), &mr.d_ctr->touched);
    uint32_t live = 0;
    HIP_TRY(hipMemcpy(&live, &mr.d_ctr->touched, 4, hipMemcpyDeviceToHost));
    size_t want = 1024;
    while ((live + extra_entries) * 4 > want) want *= 2;
    if (want > (1ull << 31)) return fail(KICP_ERR_CAPACITY, "voxel table would exceed 2^31 slots");
    Slot *nt = nullptr;
    unsigned long long *nk = nullptr;
    uint32_t *nc = nullptr, *ns = nullptr;
    HIP_TRY(hipMalloc(&nt, want * sizeof(Slot)));
    HIP_TRY(hipMalloc(&nk, want * 8));
    HIP_TRY(hipMalloc(&nc, want * 4));
    HIP_TRY(hipMalloc(&ns, want * 4));
    const uint32_t grid_new = static_cast<uint32_t>(std::min<size_t>((want + 255) / 256, 8192));
    hipLaunchKernelGGL(k_table_clear, dim3(grid_new), dim3(256), 0, st, nt, static_cast<uint32_t>(want));
    HIP_TRY(hipMemsetAsync(nk, 0xFF, want * 8, st));
    HIP_TRY(hipMemsetAsync(nc, 0, want * 4, st));
    hipLaunchKernelGGL(k_rehash_move, dim3(grid_old), dim3(256), 0, st, mr.d_table, static_cast<uint32_t>(old_slots), nt, nk,
                       static_cast<uint32_t>(want - 1), &mr.d_ctr->error);
    HIP_TRY(hipGetLastError());
    map->dev.n_entries = live, map->dev.touched = 0;
    HIP_TRY(hipMemcpyAsync(&mr.d_ctr->n_entries, &map->dev.n_entries, 4, hipMemcpyHostToDevice, st));
    uint32_t err = 0;
    HIP_TRY(hipMemcpy(&err, &mr.d_ctr->error, 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipStreamSynchronize(st));
    hipFree(mr.d_table), hipFree(mr.d_keys64), hipFree(mr.d_cnt), hipFree(mr.d_seg_start);
    mr.d_table = nt, mr.d_keys64 = nk, mr.d_cnt = nc, mr.d_seg_start = ns;
    mr.table_slots = mr.live_slots = mr.aux_slots = want;
    mr.view.table = nt, mr.view.mask = static_cast<uint32_t>(want - 1);
    map->device_ahead = true;
    if (err) return fail(KICP_ERR_CAPACITY, "a voxel coordinate left the +-2^20 range of the device-side map update");
    return KICP_OK;
}

// VoxelHashMap::Update(points, pose) with the points already in HBM.  Runs on the device, table growth and pool growth
// included; only degenerate calls (no points) take the host path.
int map_update_device(kicp_map *map, int device, const double *d_points, size_t n, const Pose &pose) {
    DeviceMirror &mr = map->mirror;
    map->last_update_on_device = 0;
    auto host_fallback = [&]() -> int {
        std::vector<double> pts(3 * n);
        if (n) HIP_TRY(hipMemcpy(pts.data(), d_points, n * 24, hipMemcpyDeviceToHost));
        if (int rc = ensure_host_current(map)) return rc;
        map->host.ReserveEntries(32 * n + 1024);
        return map->host.Update(pts.data(), n, pose) ? KICP_OK : fail(KICP_ERR_CAPACITY, "more than 2^24-2 voxels");
    };
    if (n == 0 || n > 0x7FFFFFF0ull / 3) return host_fallback();
    if (int rc = set_device(device)) return rc;
    if (int rc = map_sync(map, device, nullptr)) return rc;
    const uint32_t cap = map->host.cap();
    // room for the points' own voxels: <= n new buckets (the pools grow on the device) ...
    if (map->dev.n_buckets_hi + n > 0xFFFFFEull) return fail(KICP_ERR_CAPACITY, "more than 2^24-2 voxels");
    if (int rc = grow_pools(map, map->dev.n_buckets_hi + n)) return rc;
    if (int rc = ensure_update_scratch(mr, n)) return rc;
    const uint32_t grid = static_cast<uint32_t>((n + 255) / 256);
    hipStream_t st = nullptr;
    UpdateParams up{};
    DevMapCounters c{};
    for (int attempt = 0;; ++attempt) {
        // ... and <= n new table entries without leaving the probing regime: re-hash (on the device) when the table is short
        if ((map->dev.n_entries + n) * 2 > mr.live_slots)
            if (int rc = device_rehash(map, 32 * n + 1024)) return rc;
        const size_t slots = mr.live_slots, bucket_cap = mr.pool_doubles / (static_cast<size_t>(cap) * 3);
        up.m = DevMap{mr.d_table, mr.d_keys64, static_cast<uint32_t>(slots - 1), mr.d_pool, mr.d_pool32, cap, static_cast<uint32_t>(bucket_cap),
                      map->host.voxel_size(), map->host.max_distance(), mr.d_free_list, mr.d_cnt, mr.d_seg_start, mr.d_ctr};
        up.in = d_points, up.n = static_cast<uint32_t>(n), up.pose = pose, up.world = mr.d_world, up.slot_of = mr.d_slot_of, up.order = mr.d_order;
        up.touched = mr.d_touched;
        HIP_TRY(hipMemsetAsync(&mr.d_ctr->touched, 0, 8, st));  // touched + error
        hipLaunchKernelGGL(k_up_claim, dim3(grid), dim3(256), 0, st, up);
        HIP_TRY(hipMemcpyAsync(&c, mr.d_ctr, sizeof c, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        map->device_ahead = true;  // the table now carries the new voxels' (still empty) entries
        if (c.error) return fail(KICP_ERR_CAPACITY, "a voxel coordinate left the +-2^20 range of the device-side map update");
        // every newly occupied voxel may add up to 26 halo entries: only continue with head-room
        const size_t new_entries = c.n_entries - map->dev.n_entries;
        map->dev = c;
        if ((c.n_entries + 26 * new_entries) * 4 <= slots * 3) break;
        if (attempt) return fail(KICP_ERR_CAPACITY, "device-side map update found no room after a re-hash");
        // Too tight.  The entries just claimed are still plain halo entries without an occupied neighbour, so a re-hash drops
        // them together with the per-slot counters of this attempt; then claim again in the larger table.
        if (int rc = device_rehash(map, 32 * n + 1024)) return rc;
    }
    const size_t slots = mr.live_slots;
    hipLaunchKernelGGL(k_up_scan, dim3(1), dim3(1024), 0, st, up);
    hipLaunchKernelGGL(k_up_scatter, dim3(grid), dim3(256), 0, st, up);
    hipLaunchKernelGGL(k_up_apply, dim3((c.touched + 63) / 64), dim3(64), 0, st, up);
    hipLaunchKernelGGL(k_up_remove, dim3(static_cast<uint32_t>(std::min<size_t>((slots + 255) / 256, 8192))), dim3(256), 0, st, up.m, pose.tx,
                       pose.ty, pose.tz);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(&c, mr.d_ctr, sizeof c, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (c.error) return fail(KICP_ERR_CAPACITY, "device-side map update ran out of room");
    map->dev = c;
    map->last_update_on_device = 1;
    return KICP_OK;
}

template <int BLOCK>
void launch_gather(const PassParams &p, uint32_t grid, hipStream_t s) {
    hipLaunchKernelGGL(k_pass_gather<BLOCK>, dim3(grid), dim3(BLOCK), 0, s, p);
}
int normalized_block(int b) { return (b == 64 || b == 256) ? b : 128; }
// Sub-lanes per query of variant 3.  Small scans are latency bound (few waves, each lane's chain of dependent bucket
// visits decides the kernel time): spreading a query's neighbour voxels over 2-4 lanes shortens that chain.  Large
// scans already fill the machine and only pay for the extra waves.
int lanes_for(const kicp_reg *r, size_t n) {
    if (r->lanes_per_query > 0) return r->lanes_per_query;
    return n <= 4096 ? 4 : (n <= 32768 ? 2 : 1);
}
uint32_t pass_grid(const kicp_reg *r, size_t n) {
    if (r->pass_kernel == 1 || r->pass_kernel == 2) {  // persistent one-wave workgroups
        const size_t max_groups = (n + 63) / 64 + (r->pass_kernel == 2 ? n / 8 : 0);  // more workgroups than groups would only idle
        const size_t want = static_cast<size_t>(r->num_cus) * r->waves_per_cu;
        return static_cast<uint32_t>(std::max<size_t>(1, std::min(want, max_groups)));
    }
    const int block = normalized_block(r->block);
    const size_t threads = r->pass_kernel == 3 ? n * static_cast<size_t>(lanes_for(r, n)) : n;
    return static_cast<uint32_t>(std::max<size_t>(1, (threads + block - 1) / block));
}
void launch_pass(const kicp_reg *r, const PassParams &p) {
    const uint32_t grid = pass_grid(r, p.n);
    if (r->pass_kernel == 2) {
        hipLaunchKernelGGL(k_pass_binned, dim3(grid), dim3(64), 0, r->stream, p);
        return;
    }
    if (r->pass_kernel == 1) {
        hipLaunchKernelGGL(k_pass_lds, dim3(grid), dim3(64), 0, r->stream, p);
        return;
    }
    if (r->pass_kernel == 3) {
        const int b = normalized_block(r->block), g = lanes_for(r, p.n);
#define KICP_G32(B, G) hipLaunchKernelGGL((k_pass_gather32<B, G>), dim3(grid), dim3(B), 0, r->stream, p)
        if (g == 1) { if (b == 64) KICP_G32(64, 1); else if (b == 256) KICP_G32(256, 1); else KICP_G32(128, 1); }
        else if (g == 2) { if (b == 64) KICP_G32(64, 2); else if (b == 256) KICP_G32(256, 2); else KICP_G32(128, 2); }
        else { if (b == 64) KICP_G32(64, 4); else if (b == 256) KICP_G32(256, 4); else KICP_G32(128, 4); }
#undef KICP_G32
        return;
    }
    switch (normalized_block(r->block)) {
        case 64: launch_gather<64>(p, grid, r->stream); break;
        case 256: launch_gather<256>(p, grid, r->stream); break;
        default: launch_gather<128>(p, grid, r->stream); break;
    }
}

int ensure_partials(kicp_reg *r, size_t blocks) {
    if (blocks <= r->partial_blocks) return KICP_OK;
    if (r->d_partials) HIP_TRY(hipFree(r->d_partials));
    if (r->d_tickets) HIP_TRY(hipFree(r->d_tickets));
    r->d_partials = nullptr, r->d_tickets = nullptr;
    const size_t want = blocks + blocks / 2 + 64, groups = want / kGroup + 2;
    HIP_TRY(hipMalloc(&r->d_partials, (want + groups) * kReduceWords * sizeof(unsigned long long)));
    HIP_TRY(hipMalloc(&r->d_tickets, groups * kTicketStride * sizeof(unsigned int)));
    HIP_TRY(hipMemsetAsync(r->d_tickets, 0, groups * kTicketStride * sizeof(unsigned int), r->stream));
    HIP_TRY(hipMemsetAsync(r->d_partials, 0, (want + groups) * kReduceWords * sizeof(unsigned long long), r->stream));  // tag 0 = never valid
    r->partial_blocks = want;
    return KICP_OK;
}
// host-mapped rows of the first-level groups (mode 4)
int ensure_rows(kicp_reg *r, size_t groups) {
    if (groups <= r->rows_groups) return KICP_OK;
    HIP_TRY(hipStreamSynchronize(r->stream));
    if (r->rows) HIP_TRY(hipHostFree(r->rows));
    r->rows = nullptr, r->d_rows = nullptr, r->rows_groups = 0;
    const size_t want = groups + groups / 2 + 64;
    HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&r->rows), want * kReduceWords * sizeof(unsigned long long), hipHostMallocMapped | hipHostMallocCoherent));
    std::memset(r->rows, 0, want * kReduceWords * sizeof(unsigned long long));
    HIP_TRY(hipHostGetDevicePointer(reinterpret_cast<void **>(&r->d_rows), r->rows, 0));
    r->rows_groups = want;
    return KICP_OK;
}
// next pass tag; when the 16-bit tag wraps, every buffer that holds tagged words is cleared so that a word left over
// from 65535 passes ago can never be mistaken for a fresh one
int next_tag(kicp_reg *r, uint32_t *tag) {
    if (r->tag >= 0xFFFFu) {
        HIP_TRY(hipStreamSynchronize(r->stream));
        if (r->rows) std::memset(r->rows, 0, r->rows_groups * kReduceWords * sizeof(unsigned long long));
        if (r->d_partials) {
            const size_t groups = r->partial_blocks / kGroup + 2;
            HIP_TRY(hipMemsetAsync(r->d_partials, 0, (r->partial_blocks + groups) * kReduceWords * sizeof(unsigned long long), r->stream));
        }
        r->tag = 0;
    }
    *tag = ++r->tag;
    return KICP_OK;
}
int ensure_frame(kicp_reg *r, size_t n) {
    if (n <= r->frame_cap) return KICP_OK;
    if (r->d_frame) HIP_TRY(hipFree(r->d_frame));
    r->d_frame = nullptr;
    const size_t want = n + n / 4 + 1024;
    HIP_TRY(hipMalloc(&r->d_frame, want * 3 * sizeof(double)));
    r->frame_cap = want;
    return KICP_OK;
}
void free_bin(BinBuffers &b) {
    hipFree(b.cell_keys), hipFree(b.cell_count), hipFree(b.cell_start), hipFree(b.cell_list), hipFree(b.counters);
    hipFree(b.qinfo), hipFree(b.items), hipFree(b.sorted_src);
    b = BinBuffers{};
}
int ensure_bin(kicp_reg *r, size_t n) {
    BinBuffers &b = r->bin;
    if (n <= b.cap_n) return KICP_OK;
    free_bin(b);
    const size_t cap = n + n / 4 + 1024;
    size_t slots = 1024;
    while (slots < 2 * cap) slots <<= 1;  // load factor <= 0.5 even if every query had its own cell
    HIP_TRY(hipMalloc(&b.cell_keys, slots * sizeof(unsigned long long)));
    HIP_TRY(hipMalloc(&b.cell_count, slots * sizeof(uint32_t)));
    HIP_TRY(hipMalloc(&b.cell_start, slots * sizeof(uint32_t)));
    HIP_TRY(hipMalloc(&b.cell_list, cap * sizeof(uint32_t)));
    HIP_TRY(hipMalloc(&b.counters, 16 * sizeof(uint32_t)));
    HIP_TRY(hipMalloc(&b.qinfo, cap * sizeof(uint2)));
    HIP_TRY(hipMalloc(&b.items, (cap / kRunLen + cap + 2) * sizeof(uint2)));
    HIP_TRY(hipMalloc(&b.sorted_src, cap * 3 * sizeof(double)));
    HIP_TRY(hipMemsetAsync(b.cell_keys, 0xFF, slots * sizeof(unsigned long long), r->stream));
    HIP_TRY(hipMemsetAsync(b.cell_count, 0, slots * sizeof(uint32_t), r->stream));
    HIP_TRY(hipMemsetAsync(b.counters, 0, 16 * sizeof(uint32_t), r->stream));
    b.mask = static_cast<uint32_t>(slots - 1), b.cap_n = cap;
    return KICP_OK;
}
// counting sort of the scan by cell at the predicted pose: three launches, once per scan
void launch_binning(kicp_reg *r, const double *d_frame, size_t n, const Pose &T0, double voxel_size) {
    BinBuffers &b = r->bin;
    BinParams bp{};
    bp.src = d_frame, bp.n = static_cast<uint32_t>(n), bp.pose0 = T0, bp.voxel_size = voxel_size;
    bp.cell_keys = b.cell_keys, bp.cell_count = b.cell_count, bp.cell_start = b.cell_start, bp.cell_list = b.cell_list;
    bp.mask = b.mask, bp.counters = b.counters, bp.qinfo = b.qinfo, bp.sorted_src = b.sorted_src, bp.items = b.items;
    const uint32_t grid = static_cast<uint32_t>((n + 255) / 256);
    hipLaunchKernelGGL(k_bin_count, dim3(grid), dim3(256), 0, r->stream, bp);
    hipLaunchKernelGGL(k_bin_scan, dim3(1), dim3(1024), 0, r->stream, bp);
    hipLaunchKernelGGL(k_bin_scatter, dim3(grid), dim3(256), 0, r->stream, bp);
}

// enqueue the collective between the limb reduction and the solve (multi-GPU only)
int enqueue_allreduce(kicp_reg *r) {
    long long *buf = r->d_state->reduce;
    if (r->allreduce_fn) {
        if (r->allreduce_fn(r->allreduce_user, buf, kReduceWords, static_cast<void *>(r->stream)) != 0)
            return fail(KICP_ERR_COMM, "user all-reduce callback failed");
        return KICP_OK;
    }
    const ncclResult_t rc = g_comm.AllReduce(buf, buf, kReduceWords, ncclInt64, ncclSum, r->comm, r->stream);
    if (rc != ncclSuccess) return fail(KICP_ERR_COMM, std::string("ncclAllReduce: ") + g_comm.GetErrorString(rc));
    return KICP_OK;
}

// wait until the record carries `call_id` with at least `min_iter` completed iterations (or its done bit);
// returns the observed seq.  Polls host-mapped memory; falls back to a stream sync when asked to or on a fault.
int wait_record(kicp_reg *r, unsigned long long call_id, unsigned min_iter, bool need_done, unsigned long long *seq_out) {
    volatile unsigned long long *seq = &r->rec->seq;
    auto ready = [&](unsigned long long s) {
        return (s >> 16) == call_id && ((s & 0x8000ull) || (!need_done && (s & 0x7FFFull) >= min_iter));
    };
    if (r->wait_mode == 1) {
        HIP_TRY(hipStreamSynchronize(r->stream));
        const unsigned long long s = __atomic_load_n(seq, __ATOMIC_ACQUIRE);
        if (!ready(s)) return fail(KICP_ERR_HIP, "result record not written after stream synchronisation");
        *seq_out = s;
        return KICP_OK;
    }
    // Poll the host-mapped record.  hipStreamQuery every `query_every` polls: it makes the runtime flush any command
    // it still holds back (some HIP runtimes batch the tail of the queue) and reports device faults.
    const unsigned query_every = r->query_every > 0 ? static_cast<unsigned>(r->query_every) : 64u;
    unsigned drained = 0;
    for (unsigned long long spins = 1;; ++spins) {
        const unsigned long long s = __atomic_load_n(seq, __ATOMIC_ACQUIRE);
        if (ready(s)) {
            *seq_out = s;
            return KICP_OK;
        }
        if (spins % query_every == 0) {
            const hipError_t q = hipStreamQuery(r->stream);
            if (q != hipSuccess && q != hipErrorNotReady) return fail(KICP_ERR_HIP, std::string("stream fault: ") + hipGetErrorString(q));
            if (q == hipSuccess && ++drained > 4 && !ready(__atomic_load_n(seq, __ATOMIC_ACQUIRE)))
                return fail(KICP_ERR_HIP, "kernels finished without publishing a result");
        }
    }
}

// mode 4: add the tagged rows of the `groups` first-level groups as they arrive (word = value << 16 | tag)
int wait_rows(kicp_reg *r, size_t groups, uint32_t tag, long long out_words[kReduceWords]) {
    for (int i = 0; i < kReduceWords; ++i) out_words[i] = 0;
    const unsigned query_every = r->query_every > 0 ? static_cast<unsigned>(r->query_every) : 64u;
    unsigned drained = 0;
    unsigned long long spins = 0;
    for (size_t g = 0; g < groups; ++g) {
        const unsigned long long *row = r->rows + g * kReduceWords;
        long long v[kReduceWords];
        for (;;) {
            bool ok = true;
            for (int i = 0; i < kReduceWords; ++i) {
                const unsigned long long w = __atomic_load_n(row + i, __ATOMIC_RELAXED);
                ok = ok && (static_cast<uint32_t>(w) & 0xFFFFu) == tag;
                v[i] = static_cast<long long>(w) >> 16;
            }
            if (ok) break;
            if (r->wait_mode == 1 || ++spins % query_every == 0) {
                // the query makes the runtime flush commands it may still hold back, and reports device faults
                const hipError_t q = r->wait_mode == 1 ? hipStreamSynchronize(r->stream) : hipStreamQuery(r->stream);
                if (q != hipSuccess && q != hipErrorNotReady) return fail(KICP_ERR_HIP, std::string("stream fault: ") + hipGetErrorString(q));
                if (q == hipSuccess && ++drained > 4) return fail(KICP_ERR_HIP, "kernels finished without publishing a result");
            }
        }
        for (int i = 0; i < kReduceWords; ++i) out_words[i] += v[i];
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    return KICP_OK;
}

// wait until every rank's slot of the current buffer carries `value`, then add the limb words (exact, order independent)
int wait_shm(kicp_reg *r, unsigned long long value, long long out_words[kReduceWords]) {
    const kicp_reg::ShmSlot *buf = r->shm + ((value - 1) & 1) * r->nranks;
    for (int i = 0; i < kReduceWords; ++i) out_words[i] = 0;
    for (int k = 0; k < r->nranks; ++k) {
        const volatile unsigned long long *seq = &buf[k].seq;
        for (unsigned long long spins = 1; __atomic_load_n(seq, __ATOMIC_ACQUIRE) != value; ++spins) {
            if (spins % 4096 == 0) {
                const hipError_t q = hipStreamQuery(r->stream);
                if (q != hipSuccess && q != hipErrorNotReady) return fail(KICP_ERR_HIP, std::string("stream fault: ") + hipGetErrorString(q));
                if (spins > (1ull << 34)) return fail(KICP_ERR_COMM, "timed out waiting for a peer rank's hand-off");
            }
        }
        for (int i = 0; i < kReduceWords; ++i) out_words[i] += buf[k].words[i];
    }
    return KICP_OK;
}

int run_registration(kicp_reg *r, kicp_map *map, const double *d_frame, size_t n, const double last_pose_qt[7],
                     const double rel_odom_qt[7], double tau, double out_pose_qt[7], kicp_stats *stats) {
    if (!r || !map || !last_pose_qt || !rel_odom_qt || !out_pose_qt) return fail(KICP_ERR_ARG, "null argument");
    if (stats) std::memset(stats, 0, sizeof(*stats));
    // current_estimate = last_robot_pose * relative_wheel_odometry   (Registration.cpp:156)
    const Pose T0 = pose_mul(pose_from(last_pose_qt), pose_from(rel_odom_qt));
    if (kicp_map_empty(map)) {  // Registration.cpp:157
        pose_to(T0, out_pose_qt);
        if (stats) stats->empty_map = 1;
        return KICP_OK;
    }
    const int max_it = r->cfg.max_num_iterations;
    if (max_it <= 0) {  // the reference's loop body never runs: the prediction is returned (Registration.cpp:179,189)
        pose_to(T0, out_pose_qt);
        return KICP_OK;
    }
    if (max_it > 0x7FFF) return fail(KICP_ERR_ARG, "max_num_iterations > 32767");
    if (n > 0x7FFFFFF0ull / 3) return fail(KICP_ERR_CAPACITY, "frame too large");
    if (int rc = set_device(r->device)) return rc;
    if (int rc = map_sync(map, r->device, r->stream)) return rc;
    const bool binned = r->pass_kernel == 2;
    if (binned)
        if (int rc = ensure_bin(r, n ? n : 1)) return rc;
    if (int rc = ensure_partials(r, pass_grid(r, n))) return rc;
    const bool shm = r->shm != nullptr;
    const bool multi = r->comm != nullptr || r->allreduce_fn != nullptr;
    if (shm && !r->host_solve) return fail(KICP_ERR_ARG, "the shared-segment mode needs host_solve = 1");
    const unsigned long long call_id = ++r->call_id;

    PassParams pp{};
    pp.src = d_frame, pp.n = static_cast<uint32_t>(n), pp.map = map->mirror.view, pp.tau = tau, pp.st = r->d_state;
    pp.bin = BinView{r->bin.sorted_src, r->bin.items, r->bin.counters};
    pp.partials = r->d_partials, pp.tickets = r->d_tickets;
    pp.dbg = r->dbg;
    SolveParams &sp = pp.sol;
    sp.pose0 = T0, sp.max_iterations = max_it, sp.convergence_criterion = r->cfg.convergence_criterion;
    sp.adaptive = r->cfg.use_adaptive_odometry_regularization, sp.fixed_regularization = r->cfg.fixed_regularization;
    sp.mode = multi ? 1 : 0, sp.call_id = call_id, sp.rec = r->d_rec;

    if (r->timing) HIP_TRY(hipEventRecord(r->ev0, r->stream));
    if (binned) launch_binning(r, d_frame, n, T0, map->host.voxel_size());
    const bool pass_events = r->timing == 2;
    if (pass_events && !r->evp[0])
        for (auto &e : r->evp) HIP_TRY(hipEventCreate(&e));
    auto enqueue_iteration = [&](int it) -> int {
        sp.pass = it;
        const bool ev = pass_events && it < KICP_MAX_LOG_PASSES;
        if (ev) HIP_TRY(hipEventRecord(r->evp[2 * it], r->stream));
        launch_pass(r, pp);
        if (ev) HIP_TRY(hipEventRecord(r->evp[2 * it + 1], r->stream));
        if (multi) {
            if (int rc = enqueue_allreduce(r)) return rc;
            hipLaunchKernelGGL(k_solve, dim3(1), dim3(64), 0, r->stream, r->d_state, sp);
        }
        return KICP_OK;
    };
    unsigned long long seq = 0;
    if (r->host_solve) {
        // ---- host-side solve: one launch per iteration, the pose travels as a kernel argument ----------------------
        HostRecord *rec = r->rec;
        Pose T = T0;
        double beta = 0.0;
        int iter = 0, converged = 0, nan_flag = 0;
        for (int it = 0; it < max_it; ++it) {
            const bool rows_mode = !multi && r->group_rows != 0;
            sp.pass = it, sp.pose0 = T, sp.mode = multi ? 3 : (rows_mode ? 4 : 2);
            long long words[kReduceWords];
            unsigned long long shm_value = 0;
            kicp_reg::ShmSlot *mine_host = nullptr;
            if (shm) {  // this rank's slot of the shared segment, double-buffered by hand-off parity
                const unsigned long long step = r->shm_step++;
                kicp_reg::ShmSlot *mine = r->d_shm + (step & 1) * r->nranks + r->rank;
                mine_host = r->shm + (step & 1) * r->nranks + r->rank;
                sp.pub_words = mine->words, sp.pub_seq = &mine->seq, sp.pub_value = shm_value = step + 1;
            } else {
                sp.pub_words = r->d_rec->words, sp.pub_seq = &r->d_rec->seq;
                sp.pub_value = (call_id << 16) | static_cast<unsigned long long>(it + 1);
            }
            const size_t groups = (pass_grid(r, n) + kGroup - 1) / kGroup;
            if (rows_mode) {
                if (int rc = ensure_rows(r, groups)) return rc;
                if (int rc = next_tag(r, &sp.tag)) return rc;
                sp.pub_rows = r->d_rows;
            }
            const bool ev = pass_events && it < KICP_MAX_LOG_PASSES;
            if (ev) HIP_TRY(hipEventRecord(r->evp[2 * it], r->stream));
            launch_pass(r, pp);
            if (ev) HIP_TRY(hipEventRecord(r->evp[2 * it + 1], r->stream));
            if (multi) {
                if (int rc = enqueue_allreduce(r)) return rc;
                hipLaunchKernelGGL(k_publish_words, dim3(1), dim3(64), 0, r->stream, r->d_state, r->d_rec, call_id, it);
            }
            if (rows_mode) {
                if (int rc = wait_rows(r, groups, sp.tag, words)) return rc;
                if (shm) {  // this rank's totals go into its slot from the host side; then every rank adds all slots
                    for (int i = 0; i < kReduceWords; ++i) mine_host->words[i] = words[i];
                    __atomic_store_n(&mine_host->seq, shm_value, __ATOMIC_RELEASE);
                    if (int rc = wait_shm(r, shm_value, words)) return rc;
                }
            } else if (shm) {
                if (int rc = wait_shm(r, sp.pub_value, words)) return rc;
            } else {
                if (int rc = wait_record(r, call_id, static_cast<unsigned>(it + 1), false, &seq)) return rc;
                for (int i = 0; i < kReduceWords; ++i) words[i] = rec->words[i];
            }
            double sums[kNumSums];
            for (int i = 0; i < kNumSums; ++i) sums[i] = host_limbs_to_double(words + 3 * i);
            const bool range_error = words[kNumLimbs] != 0;
            const double n = sums[6];
            if (it == 0)  // ComputeOdometryRegularization at the predicted pose (Registration.cpp:48-60,171-177)
                beta = r->cfg.use_adaptive_odometry_regularization ? 1.0 / (sums[5] / n + DBL_MIN) : r->cfg.fixed_regularization;
            double dx0, dx1;
            solve_perturbation(sums, n, beta, dx0, dx1);    // Registration.cpp:119-125
            T = pose_mul(T, motion_model(dx0, dx1));        // Registration.cpp:159-167,181-182
            iter = it + 1;
            if (stats && it < KICP_MAX_LOG_PASSES) {
                stats->n_corr[it] = n;
                for (int j = 0; j < 6; ++j) stats->sums[it][j] = sums[j];
                stats->dx[it][0] = dx0, stats->dx[it][1] = dx1;
            }
            if (std::sqrt(dx0 * dx0 + dx1 * dx1) < r->cfg.convergence_criterion) {  // Registration.cpp:184
                converged = 1;
                break;
            }
            if (!(n > 0.0) || range_error) {  // 0/0: NaN pose from here on, exactly as in the reference
                nan_flag = range_error ? 2 : 1;
                break;
            }
        }
        HIP_TRY(hipGetLastError());
        if (r->timing) HIP_TRY(hipEventRecord(r->ev1, r->stream));
        pose_to(T, out_pose_qt);
        if (stats) {
            stats->iterations = iter, stats->converged = converged, stats->beta = beta;
            if (r->timing) {
                float ms = 0.f;
                HIP_TRY(hipEventSynchronize(r->ev1));
                HIP_TRY(hipEventElapsedTime(&ms, r->ev0, r->ev1));
                stats->gpu_ms = ms;
                for (int i = 0; pass_events && i < iter && i < KICP_MAX_LOG_PASSES; ++i) {
                    HIP_TRY(hipEventElapsedTime(&ms, r->evp[2 * i], r->evp[2 * i + 1]));
                    stats->pass_ms[i] = ms;
                }
            }
        }
        if (nan_flag == 2) return fail(KICP_ERR_CAPACITY, "a per-point term exceeded the exact-accumulation range (|x| >= 2^23)");
        return nan_flag ? KICP_WARN_NO_CORRESPONDENCES : KICP_OK;
    }
    if (r->loop_mode == 0) {
        for (int it = 0; it < max_it; ++it)
            if (int rc = enqueue_iteration(it)) return rc;
        HIP_TRY(hipGetLastError());
        if (r->timing) HIP_TRY(hipEventRecord(r->ev1, r->stream));
        if (int rc = wait_record(r, call_id, 0, true, &seq)) return rc;
    } else {
        // stepped: the host polls the stop flag after every iteration.  When the previous scan needed more than one
        // iteration, iteration it+1 is queued before the flag of iteration it is known, so the GPU never idles on the
        // host (at most one queued iteration turns out to be unnecessary and exits at once); when scans converge in
        // one iteration - the usual case with good wheel odometry - nothing is queued speculatively.
        int queued = 0;
        if (int rc = enqueue_iteration(queued++)) return rc;
        for (int it = 0;; ++it) {
            if (r->speculate && queued < max_it && queued == it + 1)
                if (int rc = enqueue_iteration(queued++)) return rc;
            if (int rc = wait_record(r, call_id, static_cast<unsigned>(it + 1), false, &seq)) return rc;
            if (seq & 0x8000ull) break;
            if (queued == it + 1)
                if (int rc = enqueue_iteration(queued++)) return rc;
        }
        r->speculate = (seq & 0x7FFFull) > 1 ? 1 : 0;
        HIP_TRY(hipGetLastError());
        if (r->timing) HIP_TRY(hipEventRecord(r->ev1, r->stream));
    }
    const HostRecord *rec = r->rec;  // stable: after `done` no kernel writes the record again
    pose_to(rec->T, out_pose_qt);
    if (stats) {
        stats->iterations = rec->iter, stats->converged = rec->converged, stats->beta = rec->beta;
        const int k = rec->iter < KICP_MAX_LOG_PASSES ? rec->iter : KICP_MAX_LOG_PASSES;
        for (int i = 0; i < k; ++i) {
            stats->n_corr[i] = rec->log_ncorr[i];
            for (int j = 0; j < 6; ++j) stats->sums[i][j] = rec->log_sums[i][j];
            stats->dx[i][0] = rec->log_dx[i][0], stats->dx[i][1] = rec->log_dx[i][1];
        }
        if (r->timing) {
            float ms = 0.f;
            HIP_TRY(hipEventSynchronize(r->ev1));
            HIP_TRY(hipEventElapsedTime(&ms, r->ev0, r->ev1));
            stats->gpu_ms = ms;
            for (int i = 0; pass_events && i < k; ++i) {
                HIP_TRY(hipEventElapsedTime(&ms, r->evp[2 * i], r->evp[2 * i + 1]));
                stats->pass_ms[i] = ms;
            }
        }
    }
    if (rec->nan_flag == 2) return fail(KICP_ERR_CAPACITY, "a per-point term exceeded the exact-accumulation range (|x| >= 2^23)");
    return rec->nan_flag ? KICP_WARN_NO_CORRESPONDENCES : KICP_OK;
}

}  // namespace

// =====================================================================================================================
extern "C" {

const char *kicp_last_error(void) { return g_error.c_str(); }
int kicp_version(void) { return KICP_VERSION; }
int kicp_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return -1;
    return n;
}

// ---- map ------------------------------------------------------------------------------------------------------------
int kicp_map_create(double voxel_size, double max_distance, unsigned int max_points_per_voxel, kicp_map **out) {
    if (!out || !(voxel_size > 0.0) || max_points_per_voxel == 0) return fail(KICP_ERR_ARG, "bad map parameters");
    if (max_points_per_voxel > kMaxPointsPerVoxel) return fail(KICP_ERR_CAPACITY, "max_points_per_voxel > 255");
    *out = new kicp_map(voxel_size, max_distance, max_points_per_voxel);
    return KICP_OK;
}
void kicp_map_destroy(kicp_map *map) {
    if (!map) return;
    free_mirror(map->mirror);
    delete map;
}
int kicp_map_clear(kicp_map *map) {
    KICP_TRACE_CALL();
    if (!map) return fail(KICP_ERR_ARG, "null map");
    map->device_ahead = false;  // whatever the device holds is obsolete now
    map->host.Clear();
    return KICP_OK;
}
int kicp_map_empty(const kicp_map *map) {
    if (!map) return 1;
    return (map->device_ahead ? map->dev.n_voxels == 0 : map->host.Empty()) ? 1 : 0;
}
int kicp_map_add_points(kicp_map *map, const double *xyz, size_t n) {
    KICP_TRACE_CALL();
    if (!map || (!xyz && n)) return fail(KICP_ERR_ARG, "null argument");
    if (int rc = ensure_host_current(map)) return rc;
    return map->host.AddPoints(xyz, n) ? KICP_OK : fail(KICP_ERR_CAPACITY, "more than 2^24-2 voxels");
}
int kicp_map_remove_far(kicp_map *map, const double origin[3]) {
    KICP_TRACE_CALL();
    if (!map || !origin) return fail(KICP_ERR_ARG, "null argument");
    if (int rc = ensure_host_current(map)) return rc;
    map->host.RemovePointsFarFromLocation(origin);
    return KICP_OK;
}
int kicp_map_update_origin(kicp_map *map, const double *xyz, size_t n, const double origin[3]) {
    KICP_TRACE_CALL();
    if (!map || (!xyz && n) || !origin) return fail(KICP_ERR_ARG, "null argument");
    if (int rc = ensure_host_current(map)) return rc;
    return map->host.Update(xyz, n, origin) ? KICP_OK : fail(KICP_ERR_CAPACITY, "more than 2^24-2 voxels");
}
int kicp_map_update_pose(kicp_map *map, const double *xyz, size_t n, const double pose_qt[7]) {
    KICP_TRACE_CALL();
    if (!map || (!xyz && n) || !pose_qt) return fail(KICP_ERR_ARG, "null argument");
    if (int rc = ensure_host_current(map)) return rc;
    return map->host.Update(xyz, n, pose_from(pose_qt)) ? KICP_OK : fail(KICP_ERR_CAPACITY, "more than 2^24-2 voxels");
}
int kicp_map_update_pose_device(kicp_map *map, int device, const double *d_points_xyz, size_t n, const double pose_qt[7]) {
    KICP_TRACE_CALL();
    if (!map || (!d_points_xyz && n) || !pose_qt) return fail(KICP_ERR_ARG, "null argument");
    return map_update_device(map, device, d_points_xyz, n, pose_from(pose_qt));
}
int kicp_map_last_update_on_device(const kicp_map *map) { return map ? map->last_update_on_device : 0; }
size_t kicp_map_num_points(const kicp_map *map) {
    if (!map) return 0;
    return map->device_ahead ? static_cast<size_t>(map->dev.n_points) : map->host.num_points();
}
size_t kicp_map_num_voxels(const kicp_map *map) {
    if (!map) return 0;
    return map->device_ahead ? map->dev.n_voxels : map->host.num_voxels();
}
size_t kicp_map_pointcloud(const kicp_map *cmap, double *out_xyz, size_t cap_points) {
    KICP_TRACE_CALL();
    if (!cmap) return 0;
    kicp_map *map = const_cast<kicp_map *>(cmap);  // logically const: only scratch buffers / the host copy are touched
    if (!map->device_ahead) return map->host.Pointcloud(out_xyz, out_xyz ? cap_points : 0);
    // the HBM copy is the current one: gather the points there (same table order) and download just them
    const size_t total = static_cast<size_t>(map->dev.n_points);
    const size_t want = out_xyz ? std::min(total, cap_points) : 0;
    if (want == 0) return total;
    DeviceMirror &mr = map->mirror;
    auto gather = [&]() -> int {
        if (int rc = set_device(mr.device)) return rc;
        const size_t slots = mr.live_slots, blocks = (slots + 255) / 256;
        if (blocks + 1 > mr.pc_blocks) {
            hipFree(mr.d_pc_blocks);
            mr.d_pc_blocks = nullptr, mr.pc_blocks = 0;
            HIP_TRY(hipMalloc(&mr.d_pc_blocks, (blocks + 1) * 4));
            mr.pc_blocks = blocks + 1;
        }
        if (total > mr.pc_points) {
            hipFree(mr.d_pc);
            mr.d_pc = nullptr, mr.pc_points = 0;
            HIP_TRY(hipMalloc(&mr.d_pc, (total + total / 4 + 1024) * 24));
            mr.pc_points = total + total / 4 + 1024;
        }
        hipStream_t st = nullptr;
        hipLaunchKernelGGL(k_pc_count, dim3(static_cast<uint32_t>(blocks)), dim3(256), 0, st, mr.d_table, static_cast<uint32_t>(slots), mr.d_pc_blocks);
        hipLaunchKernelGGL(k_scan_blocks, dim3(1), dim3(1024), 0, st, mr.d_pc_blocks, static_cast<uint32_t>(blocks), mr.d_pc_blocks + blocks);
        hipLaunchKernelGGL(k_pc_gather, dim3(static_cast<uint32_t>(blocks)), dim3(256), 0, st, mr.d_table, static_cast<uint32_t>(slots), mr.d_pool,
                           map->host.cap(), mr.d_pc_blocks, mr.d_pc);
        HIP_TRY(hipGetLastError());
        uint32_t counted = 0;
        HIP_TRY(hipMemcpy(&counted, mr.d_pc_blocks + blocks, 4, hipMemcpyDeviceToHost));
        if (counted != total) return fail(KICP_ERR_HIP, "device map counters disagree with the table");
        if (int rc = staged_download(mr.stage, out_xyz, mr.d_pc, want * 24, st)) return rc;
        return KICP_OK;
    };
    if (gather() != KICP_OK) {  // fall back to refreshing the host copy
        if (ensure_host_current(map) != KICP_OK) return 0;
        return map->host.Pointcloud(out_xyz, cap_points);
    }
    return total;
}
size_t kicp_map_check(const kicp_map *map) {
    if (!map || ensure_host_current(const_cast<kicp_map *>(map)) != KICP_OK) return 1;
    return map->host.CheckInvariants();
}
int kicp_map_sync(kicp_map *map, int device) {
    KICP_TRACE_CALL();
    if (!map) return fail(KICP_ERR_ARG, "null map");
    if (int rc = set_device(device)) return rc;
    return map_sync(map, device, nullptr);
}
int kicp_map_last_upload(const kicp_map *map, size_t *bytes, int *was_full) {
    if (!map || !bytes || !was_full) return fail(KICP_ERR_ARG, "null argument");
    *bytes = map->mirror.last_upload_bytes, *was_full = map->mirror.last_upload_full;
    return KICP_OK;
}
int kicp_map_closest(kicp_map *map, int device, const double *queries_xyz, size_t n, double *out_nn_xyz, double *out_dist) {
    KICP_TRACE_CALL();
    if (!map || (!queries_xyz && n) || !out_nn_xyz || !out_dist) return fail(KICP_ERR_ARG, "null argument");
    if (n == 0) return KICP_OK;
    if (kicp_map_empty(map)) {
        for (size_t i = 0; i < n; ++i) out_nn_xyz[3 * i] = out_nn_xyz[3 * i + 1] = out_nn_xyz[3 * i + 2] = 0.0, out_dist[i] = DBL_MAX;
        return KICP_OK;
    }
    if (int rc = kicp_map_sync(map, device)) return rc;
    double *d_q = nullptr, *d_nn = nullptr, *d_d = nullptr;
    HIP_TRY(hipMalloc(&d_q, n * 24));
    HIP_TRY(hipMalloc(&d_nn, n * 24));
    HIP_TRY(hipMalloc(&d_d, n * 8));
    if (int rc = staged_upload(map->mirror.stage, 0, d_q, queries_xyz, n * 24, nullptr)) return rc;
    hipLaunchKernelGGL(k_closest, dim3(static_cast<uint32_t>((n + 255) / 256)), dim3(256), 0, nullptr, d_q, static_cast<uint32_t>(n),
                       map->mirror.view, d_nn, d_d);
    HIP_TRY(hipGetLastError());
    if (int rc = staged_download(map->mirror.stage, out_nn_xyz, d_nn, n * 24, nullptr)) return rc;
    if (int rc = staged_download(map->mirror.stage, out_dist, d_d, n * 8, nullptr)) return rc;
    hipFree(d_q), hipFree(d_nn), hipFree(d_d);
    return KICP_OK;
}

// ---- registration ---------------------------------------------------------------------------------------------------
int kicp_reg_create(const kicp_reg_config *config, int device, kicp_reg **out) {
    if (!config || !out) return fail(KICP_ERR_ARG, "null argument");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(KICP_ERR_HIP, "no HIP device visible: this library has no CPU fallback");
    if (device < 0 || device >= ndev) return fail(KICP_ERR_ARG, "device index out of range");
    if (int rc = set_device(device)) return rc;
    kicp_reg *r = new kicp_reg;
    r->cfg = *config, r->device = device;
    hipError_t e = hipStreamCreateWithFlags(&r->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreate(&r->ev0);
    if (e == hipSuccess) e = hipEventCreate(&r->ev1);
    if (e == hipSuccess) e = hipMalloc(&r->d_state, sizeof(IcpState));
    if (e == hipSuccess) e = hipMemset(r->d_state, 0, sizeof(IcpState));
    if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void **>(&r->rec), sizeof(HostRecord), hipHostMallocMapped | hipHostMallocCoherent);
    if (e == hipSuccess) std::memset(r->rec, 0, sizeof(HostRecord));
    if (e == hipSuccess) e = hipHostGetDevicePointer(reinterpret_cast<void **>(&r->d_rec), r->rec, 0);
    if (e == hipSuccess) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) r->num_cus = prop.multiProcessorCount;
    }
    if (e != hipSuccess) {
        kicp_reg_destroy(r);
        return fail(KICP_ERR_HIP, std::string("kicp_reg_create: ") + hipGetErrorString(e));
    }
    if (const char *env = std::getenv("KICP_PASS_KERNEL")) r->pass_kernel = std::atoi(env);
    if (const char *env = std::getenv("KICP_BLOCK")) r->block = normalized_block(std::atoi(env));
    if (const char *env = std::getenv("KICP_LOOP")) r->loop_mode = std::atoi(env);
    if (const char *env = std::getenv("KICP_WAIT")) r->wait_mode = std::atoi(env);
    if (const char *env = std::getenv("KICP_QUERY_EVERY")) r->query_every = std::atoi(env);
    *out = r;
    return KICP_OK;
}
void kicp_reg_destroy(kicp_reg *reg) {
    if (!reg) return;
    hipSetDevice(reg->device);
    if (reg->comm) g_comm.CommDestroy(reg->comm);
    if (reg->stream) hipStreamSynchronize(reg->stream);
    if (reg->shm) kicp_reg_shm_destroy(reg);
    if (reg->d_state) hipFree(reg->d_state);
    if (reg->rec) hipHostFree(reg->rec);
    if (reg->rows) hipHostFree(reg->rows);
    reg->stage.release();
    if (reg->d_partials) hipFree(reg->d_partials);
    if (reg->d_tickets) hipFree(reg->d_tickets);
    free_bin(reg->bin);
    if (reg->d_frame) hipFree(reg->d_frame);
    if (reg->ev0) hipEventDestroy(reg->ev0);
    if (reg->ev1) hipEventDestroy(reg->ev1);
    for (auto &e : reg->evp)
        if (e) hipEventDestroy(e);
    if (reg->stream) hipStreamDestroy(reg->stream);
    delete reg;
}
int kicp_reg_get_config(const kicp_reg *reg, kicp_reg_config *out) {
    if (!reg || !out) return fail(KICP_ERR_ARG, "null argument");
    *out = reg->cfg;
    return KICP_OK;
}
int kicp_reg_set_config(kicp_reg *reg, const kicp_reg_config *config) {
    if (!reg || !config) return fail(KICP_ERR_ARG, "null argument");
    reg->cfg = *config;
    return KICP_OK;
}
int kicp_reg_set_option(kicp_reg *reg, const char *name, double value) {
    if (!reg || !name) return fail(KICP_ERR_ARG, "null argument");
    const std::string k(name);
    if (k == "pass_kernel") reg->pass_kernel = static_cast<int>(value);
    else if (k == "block") reg->block = normalized_block(static_cast<int>(value));
    else if (k == "loop") reg->loop_mode = static_cast<int>(value);
    else if (k == "wait") reg->wait_mode = static_cast<int>(value);
    else if (k == "host_solve") reg->host_solve = static_cast<int>(value);
    else if (k == "group_rows") reg->group_rows = static_cast<int>(value);
    else if (k == "debug_tag") reg->tag = static_cast<uint32_t>(value) & 0xFFFFu;  // tests: jump next to the 16-bit tag's wrap-around
    else if (k == "lanes_per_query") reg->lanes_per_query = (value >= 4) ? 4 : (value >= 2 ? 2 : (value >= 1 ? 1 : 0));
    else if (k == "waves_per_cu") reg->waves_per_cu = std::max(1, static_cast<int>(value));
    else if (k == "timing") reg->timing = static_cast<int>(value);
    else if (k == "dbg") reg->dbg = static_cast<int>(value);
    else if (k == "query_every") reg->query_every = static_cast<int>(value);
    else return fail(KICP_ERR_ARG, "unknown option " + k);
    return KICP_OK;
}
double kicp_reg_get_option(const kicp_reg *reg, const char *name) {
    if (!reg || !name) return -1.0;
    const std::string k(name);
    if (k == "pass_kernel") return reg->pass_kernel;
    if (k == "block") return reg->block;
    if (k == "loop") return reg->loop_mode;
    if (k == "wait") return reg->wait_mode;
    if (k == "host_solve") return reg->host_solve;
    if (k == "group_rows") return reg->group_rows;
    if (k == "debug_tag") return reg->tag;
    if (k == "lanes_per_query") return reg->lanes_per_query;
    if (k == "waves_per_cu") return reg->waves_per_cu;
    if (k == "timing") return reg->timing;
    if (k == "last_not_staged") return reg->rec ? reg->rec->not_staged : -1.0;
    return -1.0;
}

int kicp_register_device(kicp_reg *reg, kicp_map *map, const double *d_frame_xyz, size_t n, const double last_pose_qt[7],
                         const double rel_odom_qt[7], double max_correspondence_distance, double out_pose_qt[7],
                         kicp_stats *stats) {
    KICP_TRACE_CALL();
    if (!d_frame_xyz && n) return fail(KICP_ERR_ARG, "null frame");
    return run_registration(reg, map, d_frame_xyz, n, last_pose_qt, rel_odom_qt, max_correspondence_distance, out_pose_qt, stats);
}
int kicp_register(kicp_reg *reg, kicp_map *map, const double *frame_xyz, size_t n, const double last_pose_qt[7],
                  const double rel_odom_qt[7], double max_correspondence_distance, double out_pose_qt[7], kicp_stats *stats) {
    KICP_TRACE_CALL();
    if (!reg || !map || (!frame_xyz && n)) return fail(KICP_ERR_ARG, "null argument");
    if (!kicp_map_empty(map) && n) {
        if (int rc = set_device(reg->device)) return rc;
        if (int rc = ensure_frame(reg, n)) return rc;
        if (int rc = staged_upload(reg->stage, 0, reg->d_frame, frame_xyz, n * 24, reg->stream)) return rc;
    }
    return run_registration(reg, map, reg->d_frame, n, last_pose_qt, rel_odom_qt, max_correspondence_distance, out_pose_qt, stats);
}
static int pass_once(kicp_reg *reg, kicp_map *map, const double *frame_xyz, size_t n, const double pose_qt[7],
                     double max_correspondence_distance, double out_sums[7], long long out_words[24]);
int kicp_pass_sums(kicp_reg *reg, kicp_map *map, const double *frame_xyz, size_t n, const double pose_qt[7],
                   double max_correspondence_distance, double out_sums[7]) {
    if (!out_sums) return fail(KICP_ERR_ARG, "null argument");
    return pass_once(reg, map, frame_xyz, n, pose_qt, max_correspondence_distance, out_sums, nullptr);
}
int kicp_pass_words(kicp_reg *reg, kicp_map *map, const double *frame_xyz, size_t n, const double pose_qt[7],
                    double max_correspondence_distance, long long out_words[24]) {
    if (!out_words) return fail(KICP_ERR_ARG, "null argument");
    double sums[7];
    return pass_once(reg, map, frame_xyz, n, pose_qt, max_correspondence_distance, sums, out_words);
}
static int pass_once(kicp_reg *reg, kicp_map *map, const double *frame_xyz, size_t n, const double pose_qt[7],
                     double max_correspondence_distance, double out_sums[7], long long out_words[24]) {
    if (!reg || !map || (!frame_xyz && n) || !pose_qt) return fail(KICP_ERR_ARG, "null argument");
    for (int i = 0; i < 7; ++i) out_sums[i] = 0.0;
    if (out_words)
        for (int i = 0; i < kReduceWords; ++i) out_words[i] = 0;
    if (kicp_map_empty(map) || n == 0) return KICP_OK;
    if (int rc = set_device(reg->device)) return rc;
    if (int rc = map_sync(map, reg->device, reg->stream)) return rc;
    if (int rc = ensure_frame(reg, n)) return rc;
    const bool binned = reg->pass_kernel == 2;
    if (binned)
        if (int rc = ensure_bin(reg, n)) return rc;
    if (int rc = ensure_partials(reg, pass_grid(reg, n))) return rc;
    if (int rc = staged_upload(reg->stage, 0, reg->d_frame, frame_xyz, n * 24, reg->stream)) return rc;
    const unsigned long long call_id = ++reg->call_id;
    PassParams pp{};
    pp.partials = reg->d_partials, pp.tickets = reg->d_tickets;
    pp.src = reg->d_frame, pp.n = static_cast<uint32_t>(n), pp.map = map->mirror.view, pp.tau = max_correspondence_distance;
    pp.st = reg->d_state, pp.bin = BinView{reg->bin.sorted_src, reg->bin.items, reg->bin.counters};
    pp.sol.pose0 = pose_from(pose_qt), pp.sol.pass = 0, pp.sol.mode = 1, pp.sol.call_id = call_id, pp.sol.rec = reg->d_rec;
    if (binned) launch_binning(reg, reg->d_frame, n, pp.sol.pose0, map->host.voxel_size());
    launch_pass(reg, pp);
    hipLaunchKernelGGL(k_publish_sums, dim3(1), dim3(64), 0, reg->stream, reg->d_state, reg->d_rec, call_id);
    HIP_TRY(hipGetLastError());
    unsigned long long seq = 0;
    if (int rc = wait_record(reg, call_id, 1, true, &seq)) return rc;
    for (int i = 0; i < 7; ++i) out_sums[i] = reg->rec->sums[i];
    if (out_words) HIP_TRY(hipMemcpy(out_words, reg->d_state->reduce, sizeof(long long) * kReduceWords, hipMemcpyDeviceToHost));
    return KICP_OK;
}

// ---- pre-steps ------------------------------------------------------------------------------------------------------
}  // extern "C"
struct kicp_pre {
    int device = 0;
    hipStream_t stream = nullptr;
    double *buf[KICP_PRE_BUFFERS] = {};
    size_t buf_cap[KICP_PRE_BUFFERS] = {}, buf_n[KICP_PRE_BUFFERS] = {};
    double *d_in = nullptr, *d_ts = nullptr, *d_staged = nullptr;
    uint32_t *d_flags = nullptr, *d_block_counts = nullptr, *d_slot_of = nullptr, *d_misc = nullptr;  // misc: [0] total, [1] error
    unsigned long long *d_keys = nullptr;
    uint32_t *d_min_index = nullptr;
    size_t cap_n = 0, table_slots = 0;
    // wire-format ingest: the raw message bytes, the stamps' extrema, what d_in / d_ts currently hold
    unsigned char *d_raw = nullptr;
    size_t raw_cap = 0;
    mutable HostStage stage;  // pinned staging for transfers from / to caller memory
    unsigned long long *d_minmax = nullptr;
    size_t ingested_n = 0;
    bool ingested = false, ingested_stamps = false;
};
namespace {
int pre_ensure(kicp_pre *p, size_t n) {
    if (n <= p->cap_n) return KICP_OK;
    const size_t cap = n + n / 4 + 1024;
    hipFree(p->d_in), hipFree(p->d_ts), hipFree(p->d_staged), hipFree(p->d_flags), hipFree(p->d_block_counts), hipFree(p->d_slot_of);
    hipFree(p->d_keys), hipFree(p->d_min_index);
    HIP_TRY(hipMalloc(&p->d_in, cap * 24));
    HIP_TRY(hipMalloc(&p->d_ts, cap * 8));
    HIP_TRY(hipMalloc(&p->d_staged, cap * 24));
    HIP_TRY(hipMalloc(&p->d_flags, cap * 4));
    HIP_TRY(hipMalloc(&p->d_block_counts, (cap / 256 + 2) * 4));
    HIP_TRY(hipMalloc(&p->d_slot_of, cap * 4));
    size_t slots = 1024;
    while (slots < 2 * cap) slots <<= 1;
    HIP_TRY(hipMalloc(&p->d_keys, slots * 8));
    HIP_TRY(hipMalloc(&p->d_min_index, slots * 4));
    p->cap_n = cap, p->table_slots = slots;
    return KICP_OK;
}
int pre_ensure_buf(kicp_pre *p, int b, size_t n) {
    if (n <= p->buf_cap[b]) return KICP_OK;
    if (p->buf[b]) HIP_TRY(hipFree(p->buf[b]));
    p->buf[b] = nullptr;
    const size_t cap = n + n / 4 + 1024;
    HIP_TRY(hipMalloc(&p->buf[b], cap * 24));
    p->buf_cap[b] = cap;
    return KICP_OK;
}
// flags + block counts are in place: scan, compact staged -> buffer dst, return the survivor count
int pre_compact(kicp_pre *p, const double *staged, size_t n, int dst, size_t *out_n) {
    const uint32_t grid = static_cast<uint32_t>((n + 255) / 256);
    if (int rc = pre_ensure_buf(p, dst, n)) return rc;
    hipLaunchKernelGGL(k_scan_blocks, dim3(1), dim3(1024), 0, p->stream, p->d_block_counts, grid, p->d_misc);
    hipLaunchKernelGGL(k_compact, dim3(grid), dim3(256), 0, p->stream, staged, p->d_flags, p->d_block_counts, static_cast<uint32_t>(n), p->buf[dst]);
    HIP_TRY(hipGetLastError());
    uint32_t misc[2] = {0, 0};
    HIP_TRY(hipMemcpyAsync(misc, p->d_misc, sizeof misc, hipMemcpyDeviceToHost, p->stream));
    HIP_TRY(hipStreamSynchronize(p->stream));
    if (misc[1]) return fail(KICP_ERR_CAPACITY, "a voxel coordinate left the +-2^20 range of the downsampling table");
    p->buf_n[dst] = misc[0];
    if (out_n) *out_n = misc[0];
    return KICP_OK;
}
// k_preprocess over what d_in / d_ts hold, then compaction into buffer dst
int pre_run_preprocess(kicp_pre *p, size_t n, bool do_deskew, const double relative_motion_qt[7], const double lidar_to_base_qt[7],
                       double max_range, double min_range, int dst_buffer, size_t *out_n) {
    if (n == 0) {
        p->buf_n[dst_buffer] = 0;
        if (out_n) *out_n = 0;
        return KICP_OK;
    }
    PreprocessParams pp{};
    pp.in = p->d_in, pp.timestamps = p->d_ts, pp.n = static_cast<uint32_t>(n), pp.deskew = do_deskew ? 1 : 0;
    const Pose rel = pose_from(relative_motion_qt);
    pose_log(rel, pp.omega);
    pp.motion_inverse = pose_inverse(rel), pp.lidar_to_base = pose_from(lidar_to_base_qt);
    pp.max_range = max_range, pp.min_range = min_range;
    pp.flags = p->d_flags, pp.staged = p->d_staged, pp.block_counts = p->d_block_counts;
    hipLaunchKernelGGL(k_preprocess, dim3(static_cast<uint32_t>((n + 255) / 256)), dim3(256), 0, p->stream, pp);
    return pre_compact(p, p->d_staged, n, dst_buffer, out_n);
}
}  // namespace
extern "C" {
int kicp_pre_create(int device, kicp_pre **out) {
    if (!out) return fail(KICP_ERR_ARG, "null argument");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(KICP_ERR_HIP, "no HIP device visible: this library has no CPU fallback");
    if (device < 0 || device >= ndev) return fail(KICP_ERR_ARG, "device index out of range");
    if (int rc = set_device(device)) return rc;
    kicp_pre *p = new kicp_pre;
    p->device = device;
    hipError_t e = hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipMalloc(&p->d_misc, 16);
    if (e == hipSuccess) e = hipMemset(p->d_misc, 0, 16);
    if (e != hipSuccess) {
        kicp_pre_destroy(p);
        return fail(KICP_ERR_HIP, std::string("kicp_pre_create: ") + hipGetErrorString(e));
    }
    *out = p;
    return KICP_OK;
}
void kicp_pre_destroy(kicp_pre *p) {
    if (!p) return;
    hipSetDevice(p->device);
    if (p->stream) hipStreamSynchronize(p->stream);
    for (double *b : p->buf) hipFree(b);
    hipFree(p->d_in), hipFree(p->d_ts), hipFree(p->d_staged), hipFree(p->d_flags), hipFree(p->d_block_counts), hipFree(p->d_slot_of);
    hipFree(p->d_keys), hipFree(p->d_min_index), hipFree(p->d_misc), hipFree(p->d_raw), hipFree(p->d_minmax);
    p->stage.release();
    if (p->stream) hipStreamDestroy(p->stream);
    delete p;
}
int kicp_pre_preprocess(kicp_pre *p, const double *frame_xyz, size_t n, const double *timestamps, size_t n_timestamps,
                        const double relative_motion_qt[7], const double lidar_to_base_qt[7], double max_range, double min_range,
                        int deskew, int dst_buffer, size_t *out_n) {
    KICP_TRACE_CALL();
    if (!p || (!frame_xyz && n) || !relative_motion_qt || !lidar_to_base_qt || dst_buffer < 0 || dst_buffer >= KICP_PRE_BUFFERS)
        return fail(KICP_ERR_ARG, "bad argument");
    const bool do_deskew = deskew && n_timestamps != 0;  // Preprocessing.cpp: `if (deskew_ && !timestamps.empty())`
    if (do_deskew && (!timestamps || n_timestamps < n)) return fail(KICP_ERR_ARG, "one timestamp per point is required for deskewing");
    if (int rc = set_device(p->device)) return rc;
    if (n > 0x7FFFFFF0ull / 3) return fail(KICP_ERR_CAPACITY, "frame too large");
    if (n) {
        if (int rc = pre_ensure(p, n)) return rc;
        p->ingested = false;  // d_in / d_ts are overwritten
        if (int rc = stage_reserve(p->stage, n * 32, p->stream)) return rc;  // one buffer for both arrays
        if (int rc = staged_upload(p->stage, 0, p->d_in, frame_xyz, n * 24, p->stream)) return rc;
        if (do_deskew)
            if (int rc = staged_upload(p->stage, n * 24, p->d_ts, timestamps, n * 8, p->stream)) return rc;
    }
    return pre_run_preprocess(p, n, do_deskew, relative_motion_qt, lidar_to_base_qt, max_range, min_range, dst_buffer, out_n);
}
int kicp_pre_ingest(kicp_pre *p, const void *data, size_t n_points, const kicp_cloud_layout *layout, const double sensor_pose_qt[7],
                    double *out_min_stamp, double *out_max_stamp) {
    KICP_TRACE_CALL();
    if (!p || !layout || (!data && n_points)) return fail(KICP_ERR_ARG, "bad argument");
    const kicp_cloud_layout &L = *layout;
    const int st = L.stamp_datatype;
    if (st != 0 && st != KICP_FIELD_UINT32 && st != KICP_FIELD_FLOAT32 && st != KICP_FIELD_FLOAT64)
        return fail(KICP_ERR_ARG, "timestamp field type not supported");  // TimeStampHandler.cpp:103
    const uint32_t stamp_bytes = st == KICP_FIELD_FLOAT64 ? 8u : 4u;
    if (L.point_step == 0 || L.offset_x + 4ull > L.point_step || L.offset_y + 4ull > L.point_step || L.offset_z + 4ull > L.point_step ||
        (st != 0 && L.offset_stamp + static_cast<unsigned long long>(stamp_bytes) > L.point_step))
        return fail(KICP_ERR_ARG, "field offsets do not fit inside point_step");
    if (n_points > 0x7FFFFFF0ull / 3) return fail(KICP_ERR_CAPACITY, "cloud too large");
    if (int rc = set_device(p->device)) return rc;
    p->ingested = true, p->ingested_n = n_points, p->ingested_stamps = st != 0 && n_points != 0;
    if (out_min_stamp) *out_min_stamp = 0.0;
    if (out_max_stamp) *out_max_stamp = 0.0;
    if (n_points == 0) return KICP_OK;
    if (int rc = pre_ensure(p, n_points)) return rc;
    const size_t bytes = n_points * static_cast<size_t>(L.point_step);
    if (bytes > p->raw_cap) {
        hipFree(p->d_raw);
        p->d_raw = nullptr, p->raw_cap = 0;
        HIP_TRY(hipMalloc(&p->d_raw, bytes + bytes / 4 + 4096));
        p->raw_cap = bytes + bytes / 4 + 4096;
    }
    if (!p->d_minmax) HIP_TRY(hipMalloc(&p->d_minmax, 16));
    const unsigned long long init[2] = {~0ull, 0ull};
    HIP_TRY(hipMemcpyAsync(p->d_minmax, init, 16, hipMemcpyHostToDevice, p->stream));
    if (int rc = staged_upload(p->stage, 0, p->d_raw, data, bytes, p->stream)) return rc;
    IngestParams ip{};
    ip.raw = p->d_raw, ip.n = static_cast<uint32_t>(n_points), ip.point_step = L.point_step;
    ip.off_x = L.offset_x, ip.off_y = L.offset_y, ip.off_z = L.offset_z, ip.off_t = L.offset_stamp, ip.stamp_type = st;
    ip.transform = sensor_pose_qt ? 1 : 0;
    if (sensor_pose_qt) ip.T = pose_from(sensor_pose_qt);
    ip.out_xyz = p->d_in, ip.out_stamps = p->d_ts, ip.minmax = p->d_minmax;
    const uint32_t grid = static_cast<uint32_t>((n_points + 255) / 256);
    hipLaunchKernelGGL(k_ingest, dim3(grid), dim3(256), 0, p->stream, ip);
    if (st != 0) {
        hipLaunchKernelGGL(k_normalize_stamps, dim3(grid), dim3(256), 0, p->stream, p->d_ts, ip.n, p->d_minmax);
        unsigned long long mm[2];
        HIP_TRY(hipMemcpyAsync(mm, p->d_minmax, 16, hipMemcpyDeviceToHost, p->stream));
        HIP_TRY(hipStreamSynchronize(p->stream));
        if (out_min_stamp) *out_min_stamp = ordered_value(mm[0]);
        if (out_max_stamp) *out_max_stamp = ordered_value(mm[1]);
    } else {
        HIP_TRY(hipStreamSynchronize(p->stream));  // `data` is borrowed for the call only
    }
    HIP_TRY(hipGetLastError());
    return KICP_OK;
}
int kicp_pre_preprocess_ingested(kicp_pre *p, const double relative_motion_qt[7], const double lidar_to_base_qt[7], double max_range,
                                 double min_range, int deskew, int dst_buffer, size_t *out_n) {
    KICP_TRACE_CALL();
    if (!p || !relative_motion_qt || !lidar_to_base_qt || dst_buffer < 0 || dst_buffer >= KICP_PRE_BUFFERS)
        return fail(KICP_ERR_ARG, "bad argument");
    if (!p->ingested) return fail(KICP_ERR_ARG, "no ingested cloud: call kicp_pre_ingest first");
    if (int rc = set_device(p->device)) return rc;
    return pre_run_preprocess(p, p->ingested_n, deskew && p->ingested_stamps, relative_motion_qt, lidar_to_base_qt, max_range, min_range,
                              dst_buffer, out_n);
}
int kicp_pre_ingested(const kicp_pre *p, double *out_xyz, double *out_stamps, size_t cap_points, size_t *out_n, int *out_has_stamps) {
    if (!p) return fail(KICP_ERR_ARG, "bad argument");
    if (!p->ingested) return fail(KICP_ERR_ARG, "no ingested cloud: call kicp_pre_ingest first");
    if (int rc = set_device(p->device)) return rc;
    const size_t k = std::min(p->ingested_n, cap_points);
    if (k && out_xyz)
        if (int rc = staged_download(p->stage, out_xyz, p->d_in, k * 24, p->stream)) return rc;
    if (k && out_stamps && p->ingested_stamps)
        if (int rc = staged_download(p->stage, out_stamps, p->d_ts, k * 8, p->stream)) return rc;
    if (out_n) *out_n = p->ingested_n;
    if (out_has_stamps) *out_has_stamps = p->ingested_stamps ? 1 : 0;
    return KICP_OK;
}
int kicp_pre_voxel_downsample(kicp_pre *p, int src, double voxel_size, int dst, size_t *out_n) {
    KICP_TRACE_CALL();
    if (!p || src < 0 || src >= KICP_PRE_BUFFERS || dst < 0 || dst >= KICP_PRE_BUFFERS || src == dst || !(voxel_size > 0.0))
        return fail(KICP_ERR_ARG, "bad argument");
    if (int rc = set_device(p->device)) return rc;
    const size_t n = p->buf_n[src];
    if (n == 0) {
        p->buf_n[dst] = 0;
        if (out_n) *out_n = 0;
        return KICP_OK;
    }
    if (int rc = pre_ensure(p, n)) return rc;
    size_t slots = 1024;  // this call's table: the smallest power of two >= 2n keeps the memset small
    while (slots < 2 * n) slots <<= 1;
    HIP_TRY(hipMemsetAsync(p->d_keys, 0xFF, slots * 8, p->stream));
    HIP_TRY(hipMemsetAsync(p->d_min_index, 0xFF, slots * 4, p->stream));
    DownsampleParams dp{};
    dp.in = p->buf[src], dp.n = static_cast<uint32_t>(n), dp.voxel_size = voxel_size, dp.keys = p->d_keys, dp.min_index = p->d_min_index;
    dp.mask = static_cast<uint32_t>(slots - 1), dp.slot_of = p->d_slot_of, dp.flags = p->d_flags, dp.block_counts = p->d_block_counts;
    dp.error = p->d_misc + 1;
    const uint32_t grid = static_cast<uint32_t>((n + 255) / 256);
    hipLaunchKernelGGL(k_downsample_claim, dim3(grid), dim3(256), 0, p->stream, dp);
    hipLaunchKernelGGL(k_downsample_flag, dim3(grid), dim3(256), 0, p->stream, dp);
    return pre_compact(p, p->buf[src], n, dst, out_n);
}
int kicp_pre_upload(kicp_pre *p, int buffer, const double *xyz, size_t n) {
    KICP_TRACE_CALL();
    if (!p || buffer < 0 || buffer >= KICP_PRE_BUFFERS || (!xyz && n)) return fail(KICP_ERR_ARG, "bad argument");
    if (int rc = set_device(p->device)) return rc;
    if (int rc = pre_ensure_buf(p, buffer, n ? n : 1)) return rc;
    if (n)
        if (int rc = staged_upload(p->stage, 0, p->buf[buffer], xyz, n * 24, p->stream)) return rc;
    HIP_TRY(hipStreamSynchronize(p->stream));
    p->buf_n[buffer] = n;
    return KICP_OK;
}
int kicp_pre_download(const kicp_pre *p, int buffer, double *out_xyz, size_t cap_points, size_t *out_n) {
    KICP_TRACE_CALL();
    if (!p || buffer < 0 || buffer >= KICP_PRE_BUFFERS) return fail(KICP_ERR_ARG, "bad argument");
    if (int rc = set_device(p->device)) return rc;
    const size_t n = p->buf_n[buffer], k = std::min(n, cap_points);
    if (k && out_xyz)
        if (int rc = staged_download(p->stage, out_xyz, p->buf[buffer], k * 24, p->stream)) return rc;
    if (out_n) *out_n = n;
    return KICP_OK;
}
const double *kicp_pre_device_ptr(const kicp_pre *p, int buffer, size_t *out_n) {
    if (!p || buffer < 0 || buffer >= KICP_PRE_BUFFERS) return nullptr;
    if (out_n) *out_n = p->buf_n[buffer];
    return p->buf[buffer];
}

// ---- device helpers -------------------------------------------------------------------------------------------------
int kicp_device_malloc(int device, size_t bytes, void **out_dptr) {
    if (!out_dptr) return fail(KICP_ERR_ARG, "null argument");
    if (int rc = set_device(device)) return rc;
    HIP_TRY(hipMalloc(out_dptr, bytes ? bytes : 1));
    return KICP_OK;
}
int kicp_device_free(int device, void *dptr) {
    if (int rc = set_device(device)) return rc;
    HIP_TRY(hipFree(dptr));
    return KICP_OK;
}
int kicp_device_upload(int device, void *dst_dptr, const void *src_host, size_t bytes) {
    if (int rc = set_device(device)) return rc;
    HIP_TRY(hipMemcpy(dst_dptr, src_host, bytes, hipMemcpyHostToDevice));
    return KICP_OK;
}
int kicp_device_synchronize(int device) {
    if (int rc = set_device(device)) return rc;
    HIP_TRY(hipDeviceSynchronize());
    return KICP_OK;
}

// ---- multi-GPU ------------------------------------------------------------------------------------------------------
int kicp_comm_unique_id(char id[KICP_COMM_ID_BYTES]) {
    static_assert(KICP_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "id size");
    static_assert(KICP_REDUCE_WORDS == kReduceWords, "payload size");
    if (!id) return fail(KICP_ERR_ARG, "null argument");
    std::string err;
    if (!g_comm.load(err)) return fail(KICP_ERR_COMM, err);
    ncclUniqueId uid;
    const ncclResult_t rc = g_comm.GetUniqueId(&uid);
    if (rc != ncclSuccess) return fail(KICP_ERR_COMM, std::string("ncclGetUniqueId: ") + g_comm.GetErrorString(rc));
    std::memcpy(id, uid.internal, KICP_COMM_ID_BYTES);
    return KICP_OK;
}
int kicp_reg_comm_init(kicp_reg *reg, int nranks, int rank, const char id[KICP_COMM_ID_BYTES]) {
    if (!reg || !id || nranks < 1 || rank < 0 || rank >= nranks) return fail(KICP_ERR_ARG, "bad communicator arguments");
    std::string err;
    if (!g_comm.load(err)) return fail(KICP_ERR_COMM, err);
    if (int rc = set_device(reg->device)) return rc;
    if (reg->comm) g_comm.CommDestroy(reg->comm), reg->comm = nullptr;
    ncclUniqueId uid;
    std::memcpy(uid.internal, id, KICP_COMM_ID_BYTES);
    const ncclResult_t rc = g_comm.CommInitRank(&reg->comm, nranks, uid, rank);
    if (rc != ncclSuccess) {
        reg->comm = nullptr;
        return fail(KICP_ERR_COMM, std::string("ncclCommInitRank: ") + g_comm.GetErrorString(rc));
    }
    reg->nranks = nranks, reg->rank = rank;
    return KICP_OK;
}
int kicp_reg_comm_destroy(kicp_reg *reg) {
    if (!reg) return fail(KICP_ERR_ARG, "null argument");
    if (reg->comm) {
        hipSetDevice(reg->device);
        hipStreamSynchronize(reg->stream);
        g_comm.CommDestroy(reg->comm);
        reg->comm = nullptr;
    }
    reg->nranks = 1, reg->rank = 0;
    return KICP_OK;
}
int kicp_reg_shm_destroy(kicp_reg *reg) {
    if (!reg) return fail(KICP_ERR_ARG, "null argument");
    if (reg->shm) {
        hipSetDevice(reg->device);
        hipStreamSynchronize(reg->stream);
        hipHostUnregister(reg->shm);
        munmap(reg->shm, reg->shm_bytes);
        if (reg->rank == 0) shm_unlink(reg->shm_name.c_str());
        reg->shm = nullptr, reg->d_shm = nullptr, reg->shm_bytes = 0;
    }
    reg->nranks = 1, reg->rank = 0;
    return KICP_OK;
}
int kicp_reg_shm_init(kicp_reg *reg, int nranks, int rank, const char *name) {
    if (!reg || !name || nranks < 1 || rank < 0 || rank >= nranks) return fail(KICP_ERR_ARG, "bad shared-segment arguments");
    if (reg->comm) return fail(KICP_ERR_ARG, "an RCCL communicator is already attached");
    kicp_reg_shm_destroy(reg);
    if (int rc = set_device(reg->device)) return rc;
    const size_t bytes = 2 * static_cast<size_t>(nranks) * sizeof(kicp_reg::ShmSlot);
    const std::string nm = std::string(name[0] == '/' ? "" : "/") + name;
    const int fd = shm_open(nm.c_str(), rank == 0 ? (O_CREAT | O_RDWR) : O_RDWR, 0600);
    if (fd < 0) return fail(KICP_ERR_COMM, "shm_open(" + nm + ") failed (rank 0 must create it first)");
    if (rank == 0 && ftruncate(fd, static_cast<off_t>(bytes)) != 0) {
        close(fd);
        return fail(KICP_ERR_COMM, "ftruncate on the shared segment failed");
    }
    void *ptr = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (ptr == MAP_FAILED) return fail(KICP_ERR_COMM, "mmap of the shared segment failed");
    if (rank == 0) std::memset(ptr, 0, bytes);
    hipError_t e = hipHostRegister(ptr, bytes, hipHostRegisterMapped | hipHostRegisterPortable);
    void *dptr = nullptr;
    if (e == hipSuccess) e = hipHostGetDevicePointer(&dptr, ptr, 0);
    if (e != hipSuccess) {
        munmap(ptr, bytes);
        return fail(KICP_ERR_HIP, std::string("registering the shared segment: ") + hipGetErrorString(e));
    }
    reg->shm = static_cast<kicp_reg::ShmSlot *>(ptr), reg->d_shm = static_cast<kicp_reg::ShmSlot *>(dptr);
    reg->shm_bytes = bytes, reg->shm_step = 0, reg->shm_name = nm, reg->nranks = nranks, reg->rank = rank;
    return KICP_OK;
}
int kicp_reg_set_allreduce(kicp_reg *reg, kicp_allreduce_fn fn, void *user) {
    if (!reg) return fail(KICP_ERR_ARG, "null argument");
    reg->allreduce_fn = fn, reg->allreduce_user = user;
    return KICP_OK;
}

}  // extern "C"
