// kicp_reg_api.hip -- the C-ABI entry points of the registration (see kicp_reg_internal.hpp)
#include "kicp_reg_internal.hpp"

using namespace kicp;
using namespace kicp::host;

namespace {
constexpr size_t kBarFramePoints = 8192;  // host frames up to this size travel through the BAR (kicp_register)
int ensure_frame(kicp_reg *r, size_t n) {
    if (n <= r->frame_cap) return KICP_OK;
    if (int rc = aql_quiesce(r)) return rc;
    if (r->d_frame) HIP_TRY(hipFree(r->d_frame));
    r->d_frame = nullptr;
    const size_t want = n + n / 4 + 1024;
    HIP_TRY(hipMalloc(&r->d_frame, want * 3 * sizeof(double)));
    r->frame_cap = want;
    return KICP_OK;
}
// A piece of a host frame, read by the GPU straight out of the handle's pinned staging buffer (host-mapped memory, over PCIe) and
// written as fp64 into the device frame: float32 sources are widened on the way - static_cast<double>(float) is exact, i.e. what
// the reference's host-side conversion produces (ros/src/kinematic_icp_ros/utils/RosUtils.cpp:30-39).  16 bytes per lane and load.
template <typename T>
__global__ __launch_bounds__(256) void k_fetch_frame(const T *__restrict__ staged, double *__restrict__ dst, uint32_t count) {
    constexpr uint32_t kPer = 16 / sizeof(T);  // scalars per 16-byte load
    const uint32_t i = (blockIdx.x * 256u + threadIdx.x) * kPer;
    if (i + kPer <= count) {
        typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
        const u32x4 w = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(staged + i));
        T v[kPer];
        __builtin_memcpy(v, &w, 16);
#pragma unroll
        for (uint32_t k = 0; k < kPer; ++k) dst[i + k] = static_cast<double>(v[k]);
    } else {
        for (uint32_t k = i; k < count; ++k) dst[k] = static_cast<double>(staged[k]);
    }
}
// Upload of a whole host frame (`n` points of T = double | float) into r->d_frame: the calling thread copies the caller's memory
// into the pinned staging buffer piece by piece and launches k_fetch_frame behind each piece, so the GPU pulls piece k over PCIe
// while the CPU copies piece k + 1.  One kernel launch per piece costs the host ~3 us where a hipMemcpyAsync costs ~10
// (profiles/r03_time_presteps.txt), and nothing but the copy itself is left on the calling thread.
constexpr size_t kFetchPiece = 384u << 10;  // bytes of caller memory per piece (a multiple of 16)
template <typename T>
int fetch_upload(kicp_reg *r, const T *src, size_t n) {
    const size_t bytes = n * 3 * sizeof(T);
    if (int rc = stage_begin(r->stage, bytes, r->stream)) return rc;
    if (!r->stage.dev) {  // the platform does not map pinned host memory into the device's address space: the DMA engine moves the frame
        if (int rc = stage_end(r->stage, r->stream)) return rc;
        if (sizeof(T) == sizeof(double)) return staged_upload(r->stage, 0, r->d_frame, src, bytes, r->stream);
        std::vector<double> wide(n * 3);  // (float32: widened on the host first - static_cast<double>(float) is exact)
        for (size_t i = 0; i < wide.size(); ++i) wide[i] = static_cast<double>(src[i]);
        if (int rc = staged_upload(r->stage, 0, r->d_frame, wide.data(), wide.size() * sizeof(double), r->stream)) return rc;
        HIP_TRY(hipStreamSynchronize(r->stream));  // (`wide` goes out of scope; staged_upload has copied it into the pinned buffer, the DMAs may lag)
        return KICP_OK;
    }
    const unsigned char *from = reinterpret_cast<const unsigned char *>(src);
    for (size_t off = 0; off < bytes; off += kFetchPiece) {
        const size_t len = std::min(kFetchPiece, bytes - off);
        std::memcpy(r->stage.p + off, from + off, len);
        const uint32_t count = static_cast<uint32_t>(len / sizeof(T));
        const uint32_t grid = static_cast<uint32_t>((len + 4095) / 4096);  // 256 lanes x 16 bytes
        hipLaunchKernelGGL(k_fetch_frame<T>, dim3(grid), dim3(256), 0, r->stream, reinterpret_cast<const T *>(r->stage.dev + off),
                           r->d_frame + off / sizeof(T), count);
    }
    HIP_TRY(hipGetLastError());
    return stage_end(r->stage, r->stream);
}
}  // namespace

extern "C" {

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
    if (e == hipSuccess) e = pinned_alloc(reinterpret_cast<void **>(&r->rec), sizeof(HostRecord), hipHostMallocMapped | hipHostMallocCoherent);
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
    if (const char *env = std::getenv("KICP_WAIT")) r->wait_mode = std::atoi(env);
    if (const char *env = std::getenv("KICP_QUERY_EVERY")) r->query_every = std::atoi(env);
    if (const char *env = std::getenv("KICP_SMALL")) r->use_small = std::atoi(env) != 0;
    if (const char *env = std::getenv("KICP_SMALL_RESIDENT")) r->small_resident = std::atoi(env) != 0;
    if (const char *env = std::getenv("KICP_SMALL_CMD")) r->small_cmd = std::atoi(env) != 0;
    if (const char *env = std::getenv("KICP_P2P_ROWS")) r->p2p_rows = std::atoi(env) == 2 ? 2 : 1;  // (test hook: 2 = the one-row format launches of more than 32 groups use)
    *out = r;
    return KICP_OK;
}
void kicp_reg_destroy(kicp_reg *reg) {
    if (!reg) return;
    for (kicp_reg *lane : reg->batch_lanes) kicp_reg_destroy(lane);
    reg->batch_lanes.clear();
    hipSetDevice(reg->device);
    if (reg->comm) {
        for (ncclComm_t &c : reg->lane_comms)
            if (c) g_comm.CommDestroy(c), c = nullptr;
        g_comm.CommDestroy(reg->comm);
    }
    (void)reg->aql.drain(5.0);
    if (reg->stream) hipStreamSynchronize(reg->stream);
    if (reg->shm) kicp_reg_shm_destroy(reg);
    if (reg->p2p_box) kicp_reg_p2p_destroy(reg);
    if (reg->d_state) hipFree(reg->d_state);
    if (reg->rec) hipHostFree(reg->rec);
    if (reg->rows) hipHostFree(reg->rows);
    if (reg->cmd) hipHostFree(reg->cmd);
    if (reg->bar_frame) reg->aql.free_bar(reg->bar_frame);
    if (reg->d_trace) hipFree(reg->d_trace);
    if (reg->scans_bar) reg->aql.free_bar(reg->scans_bar);
    else if (reg->d_scans) hipFree(reg->d_scans);
    if (reg->cmd_bar) reg->aql.free_bar(reg->cmd_bar);
    else if (reg->d_cmd_copies) hipFree(reg->d_cmd_copies);
    reg->stage.release();
    if (reg->d_partials) hipFree(reg->d_partials);
    if (reg->d_tickets) hipFree(reg->d_tickets);
    if (reg->d_group_acc) hipFree(reg->d_group_acc);
    if (reg->d_frame) hipFree(reg->d_frame);
    if (reg->ev0) hipEventDestroy(reg->ev0);
    if (reg->ev1) hipEventDestroy(reg->ev1);
    for (auto &e : reg->evp)
        if (e) hipEventDestroy(e);
    reg->aql.release();
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
    if (k == "timing") {
        reg->timing = static_cast<int>(value);
    }
    else if (k == "wait") reg->wait_mode = static_cast<int>(value);
    else if (k == "debug_p2p_one_row") reg->p2p_rows = value != 0.0 ? 2 : 1;  // tests: this rank sends its total as ONE row, as launches of more than 32 groups do
    else if (k == "debug_tag") reg->tag = static_cast<uint32_t>(value) & 0xFFFFu;  // tests: jump next to the 16-bit tag's wrap-around
    else if (k == "lanes_per_query") reg->lanes_per_query = (value >= 4) ? 4 : (value >= 2 ? 2 : (value >= 1 ? 1 : 0));
    else if (k == "resident_generic") reg->resident_generic = value != 0.0;
    else if (k == "batch_resident") reg->batch_resident = value != 0.0;
    else if (k == "batch_queues") reg->batch_queues = std::min<int>(std::max(static_cast<int>(value), 0), kMaxBatchQueues);
    else if (k == "batch_rotate") reg->batch_rotate = value != 0.0;
    else if (k == "batch_threads") reg->batch_threads = std::min<int>(std::max(static_cast<int>(value), 0), kMaxBatchQueues + 1);
    else if (k == "batch_depth") reg->batch_depth = std::min<int>(std::max(static_cast<int>(value), 1), kPipeSlots);
    else if (k == "latency_kernel") reg->latency_kernel = value == 2.0 ? 2 : (value == 1.0 ? 1 : 0);
    else if (k == "aql") reg->use_aql = value != 0.0 ? 1 : 0;
    else if (k == "bar_frame") reg->use_bar_frame = value != 0.0 ? 1 : 0;
    else if (k == "fetch_upload") reg->fetch_frames = value != 0.0 ? 1 : 0;
    else if (k == "small") reg->use_small = value != 0.0 ? 1 : 0;
    else if (k == "small_resident") reg->small_resident = value == 2.0 ? 2 : (value != 0.0 ? 1 : 0);  // 1 adaptive (default), 2 always, 0 never
    else if (k == "small_wave") reg->small_wave = value != 0.0 ? 1 : 0;
    else if (k == "small_trace") {  // debugging aid: per-pass wall-clock stamps of workgroup 0 + host-side phase times
        if (value != 0.0 && !reg->d_trace) {
            HIP_TRY(hipMalloc(reinterpret_cast<void **>(&reg->d_trace), 1024 * 4 * sizeof(long long)));
            HIP_TRY(hipMemset(reg->d_trace, 0, 1024 * 4 * sizeof(long long)));
        }
        reg->trace_pass = value >= 1.0 ? static_cast<uint32_t>(value) : 1u;  // (the value: which pass of a launch is stamped)
        reg->trace_host_us = reg->trace_dev_us = reg->trace_first_us = 0.0, reg->trace_n = reg->trace_first_n = 0;
    }
    else if (k == "small_cmd") {
        if (reg->cmd_bar || (value != 0.0) == (reg->small_cmd != 0)) return KICP_OK;  // (once the copies live in the BAR they stay there)
        reg->small_cmd = value != 0.0 ? 1 : 0;
    }
    else if (k == "small_timeout_us") reg->small_timeout_us = value;
    else if (k == "debug_stall_us") reg->debug_stall_us = value;
    else if (k == "dbg") {
#ifdef KICP_DBG_BUILD
        reg->dbg = static_cast<int>(value);
#else
        if (value != 0.0) return fail(KICP_ERR_ARG, "this library is built without the pass kernels' ablation switches: load libkicp_amd_dbg.so (make dbg) for option dbg");
#endif
    }
    else return fail(KICP_ERR_ARG, "unknown option " + k);
    return KICP_OK;
}
double kicp_reg_get_option(const kicp_reg *reg, const char *name) {
    if (!reg || !name) return -1.0;
    const std::string k(name);
    if (k == "wait") return reg->wait_mode;
    if (k == "debug_tag") return reg->tag;
    if (k == "lanes_per_query") return reg->lanes_per_query;
    if (k == "resident_generic") return reg->resident_generic;
    if (k == "resident_passes") return reg->last_resident_passes;
    if (k == "batch_resident") return reg->batch_resident;
    if (k == "batch_threads") return reg->batch_threads;
    if (k == "batch_threads_active") return reg->last_batch_threads;
    if (k == "batch_depth") return reg->batch_depth;
    if (k == "batch_rotate") return reg->batch_rotate;
    if (k == "batch_queues") return reg->batch_queues;
    if (k == "batch_queue_passes") return static_cast<double>(reg->batch_queue_passes);
    if (k == "batch_resident_passes") return static_cast<double>(reg->batch_resident_passes);
    if (k == "latency_kernel") return reg->latency_kernel;
    if (k == "timing") return reg->timing;
    if (k == "aql") return reg->use_aql;
    if (k == "aql_kernarg") return !reg->aql.ready ? -1.0 : (std::strcmp(reg->aql.kernarg_place(), "host memory") == 0 ? 0.0 : (std::strcmp(reg->aql.kernarg_place(), "device memory") == 0 ? 1.0 : 2.0));
    if (k == "bar_frame") return reg->bar_frame ? 1.0 : (reg->use_bar_frame ? 0.5 : 0.0);  // 1: in use; 0.5: enabled, not (yet) set up
    if (k == "comm_ranks") {  // ranks the attached RCCL communicator itself reports (ncclCommCount); 0: none attached
        int count = 0;
        if (reg->comm && g_comm.CommCount && g_comm.CommCount(reg->comm, &count) == ncclSuccess) return count;
        return reg->comm ? reg->nranks : 0;
    }
    if (k == "fetch_upload") return reg->fetch_frames;
    if (k == "small") return reg->use_small;
    if (k == "small_resident") return reg->small_resident;
    if (k == "small_wave") return reg->small_wave;
    if (k == "trace_host_us") return reg->trace_n ? reg->trace_host_us / static_cast<double>(reg->trace_n) : 0.0;
    if (k == "trace_device_us") return reg->trace_n ? reg->trace_dev_us / static_cast<double>(reg->trace_n) : 0.0;
    if (k == "trace_first_us") return reg->trace_first_n ? reg->trace_first_us / static_cast<double>(reg->trace_first_n) : 0.0;
    if (k.rfind("trace_stamp_", 0) == 0) {  // trace_stamp_<i>: word i of the device stamps of the LAST call (100 MHz ticks), [workgroup][4]
        if (!reg->d_trace) return -1.0;
        static long long v[4096];
        const int i = std::atoi(k.c_str() + 12);
        if (i == 0 && hipMemcpy(v, reg->d_trace, sizeof v, hipMemcpyDeviceToHost) != hipSuccess) return -1.0;  // (word 0 refreshes the copy)
        return (i >= 0 && i < 4096) ? static_cast<double>(v[i]) : -1.0;
    }
    if (k == "small_cmd") return (reg->small_cmd == 1 && reg->cmd_bar) ? 1.0 : (reg->small_cmd ? 0.5 : 0.0);  // 1: BAR copies in use; 0.5: requested, not yet set up
    if (k == "small_timeout_us") return reg->small_timeout_us;
    if (k == "small_active") return reg->last_small;  // path of the last registration: 0 generic, 1 small (sub-lanes per query), 2 small (wave per query)
    if (k == "small_relaunches") return static_cast<double>(reg->small_relaunches);
    if (k == "aql_active") return (reg->aql.ready && reg->last_via_aql) ? 1.0 : 0.0;  // was the last pass dispatched through the AQL queue
    return -1.0;
}

int kicp_register_device(kicp_reg *reg, kicp_map *map, const double *d_frame_xyz, size_t n, const double last_pose_qt[7],
                         const double rel_odom_qt[7], double max_correspondence_distance, double out_pose_qt[7],
                         kicp_stats *stats) {
    KICP_TRACE_CALL();
    if (!d_frame_xyz && n) return fail(KICP_ERR_ARG, "null frame");
    return run_registration(reg, map, d_frame_xyz, n, last_pose_qt, rel_odom_qt, max_correspondence_distance, out_pose_qt, stats);
}
int kicp_register_device_batch(kicp_reg *reg, kicp_map *map, size_t count, const double *const *d_frames_xyz, const size_t *n,
                               const double *last_poses_qt, const double *rel_odoms_qt, double max_correspondence_distance,
                               double *out_poses_qt, int *out_iterations) {
    KICP_TRACE_CALL();
    if (count && (!d_frames_xyz || !n || !last_poses_qt || !rel_odoms_qt || !out_poses_qt)) return fail(KICP_ERR_ARG, "null argument");
    int worst = KICP_OK;
    kicp_stats st;
    for (size_t k = 0; k < count; ++k)
        if (!d_frames_xyz[k] && n[k]) return fail(KICP_ERR_ARG, "null frame");
    size_t first = 0;
    if (reg) reg->last_batch_threads = 0;
    if (reg && map) {  // scans that leave most of the device empty: several resident kernels, each with a part of the batch and a host thread
        const int rc = run_batch_resident_threads(reg, map, count, d_frames_xyz, n, last_poses_qt, rel_odoms_qt, max_correspondence_distance, out_poses_qt,
                                                  out_iterations, &worst);
        if (rc < 0) return rc;
        if (rc != 1) return worst;
    }
    if (reg && map) {  // large scans: several in flight, a queue each
        const int rc = run_batch_queues(reg, map, count, d_frames_xyz, n, last_poses_qt, rel_odoms_qt, max_correspondence_distance, out_poses_qt, out_iterations,
                                        &first, &worst);
        if (rc < 0) return rc;
        if (rc != 1 && first == count) return worst;
    }
    if (reg && map && first == 0) {  // a pass kernel resident across the batch's scans, where the batch is one for it
        const int rc = run_batch_resident(reg, map, count, d_frames_xyz, n, last_poses_qt, rel_odoms_qt, max_correspondence_distance, out_poses_qt,
                                          out_iterations, &first, &worst);
        if (rc < 0) return rc;
    }
    for (size_t k = first; k < count; ++k) {
        const int rc = run_registration(reg, map, d_frames_xyz[k], n[k], last_poses_qt + 7 * k, rel_odoms_qt + 7 * k, max_correspondence_distance,
                                        out_poses_qt + 7 * k, out_iterations ? &st : nullptr);
        if (rc < 0) return rc;
        worst = std::max(worst, rc);
        if (out_iterations) out_iterations[k] = st.iterations;
    }
    return worst;
}
int kicp_register_device_concurrent(kicp_reg *const *regs, size_t lanes, kicp_map *map, size_t count, const double *const *d_frames_xyz,
                                    const size_t *n, const double *last_poses_qt, const double *rel_odoms_qt, double max_correspondence_distance,
                                    double *out_poses_qt, int *out_iterations) {
    KICP_TRACE_CALL();
    if (!regs || lanes == 0 || !map) return fail(KICP_ERR_ARG, "null argument");
    if (count && (!d_frames_xyz || !n || !last_poses_qt || !rel_odoms_qt || !out_poses_qt)) return fail(KICP_ERR_ARG, "null argument");
    for (size_t t = 0; t < lanes; ++t) {
        if (!regs[t] || regs[t]->device != regs[0]->device) return fail(KICP_ERR_ARG, "the lanes' handles must exist and live on one device");
        for (size_t u = 0; u < t; ++u)
            if (regs[u] == regs[t]) return fail(KICP_ERR_ARG, "every lane needs a handle of its own");
        if (regs[t]->comm || regs[t]->allreduce_fn || regs[t]->shm || regs[t]->d_p2p_table)
            return fail(KICP_ERR_ARG, "independent scans are not sharded: detach the multi-GPU exchange from the lanes' handles");
    }
    for (size_t k = 0; k < count; ++k)
        if (!d_frames_xyz[k] && n[k]) return fail(KICP_ERR_ARG, "null frame");
    // the map's HBM copy is brought up to date HERE, once: the lanes then only read it
    if (int rc = set_device(regs[0]->device)) return rc;
    if (!kicp_map_empty(map)) {
        if (int rc = map_sync(map, regs[0]->device, regs[0]->stream)) return rc;
        HIP_TRY(hipStreamSynchronize(regs[0]->stream));
    }
    lanes = std::min(lanes, std::max<size_t>(count, 1));
    // Small scans: one launch per pass while several lanes are in flight.  A resident kernel waits for its host, which waits for
    // the rows of ALL its workgroups - with several such kernels on the device, workgroups of one may have to wait for CUs held by
    // the others, and only the give-up time-out would untangle that.
    // Large scans: the four-waves-per-SIMD build.  The latency-oriented build trades occupancy for a shorter chain per wave - two
    // workgroups per CU, which ONE scan of <= 131 072 points cannot exceed anyway, but which leaves no room for a second scan's
    // workgroups next to the first's (measured: 94k instead of 137k scans/s with four lanes on cfg2).
    std::vector<int> resident(lanes), latency(lanes);
    for (size_t t = 0; t < lanes; ++t) {
        resident[t] = regs[t]->small_resident, latency[t] = regs[t]->latency_kernel;
        if (lanes > 1) regs[t]->small_resident = 0, regs[t]->latency_kernel = 0;
    }
    std::atomic<size_t> next{0};
    std::atomic<int> worst{KICP_OK}, failed{KICP_OK};
    std::string failure;
    std::mutex failure_lock;
    auto lane = [&](size_t t) {
        kicp_stats st;
        for (;;) {
            const size_t k = next.fetch_add(1, std::memory_order_relaxed);
            if (k >= count || failed.load(std::memory_order_relaxed) < 0) return;
            const int rc = run_registration(regs[t], map, d_frames_xyz[k], n[k], last_poses_qt + 7 * k, rel_odoms_qt + 7 * k, max_correspondence_distance,
                                            out_poses_qt + 7 * k, out_iterations ? &st : nullptr);
            if (rc < 0) {
                std::lock_guard<std::mutex> hold(failure_lock);
                if (failed.load() == KICP_OK) failed = rc, failure = kicp_last_error();  // (the message is per thread: carry it over)
                return;
            }
            int seen = worst.load();
            while (rc > seen && !worst.compare_exchange_weak(seen, rc)) {}
            if (out_iterations) out_iterations[k] = st.iterations;
        }
    };
    lane_pool().run(lanes, lane);
    for (size_t t = 0; t < lanes; ++t) regs[t]->small_resident = resident[t], regs[t]->latency_kernel = latency[t];
    if (failed.load() < 0) return fail(failed.load(), failure);
    return worst.load();
}
int kicp_register(kicp_reg *reg, kicp_map *map, const double *frame_xyz, size_t n, const double last_pose_qt[7],
                  const double rel_odom_qt[7], double max_correspondence_distance, double out_pose_qt[7], kicp_stats *stats) {
    KICP_TRACE_CALL();
    if (!reg || !map || (!frame_xyz && n)) return fail(KICP_ERR_ARG, "null argument");
    if (!kicp_map_empty(map) && n) {
        if (int rc = set_device(reg->device)) return rc;
        // A scan of the size the pipeline registers (<= 8192 points = 192 KB) is written straight into HBM through the PCIe BAR
        // (write-combined stores, one fence): a few microseconds, no pinned staging copy, no DMA packet, nothing on the HIP stream
        // - so the pass can go out through the AQL queue at once.  The host had the results of every earlier call before it
        // got here, so no kernel is still reading the buffer.
        if (reg->use_bar_frame && n <= kBarFramePoints) {
            if (!reg->bar_frame && !reg->bar_frame_tried) {
                reg->bar_frame_tried = true;
                if (aql_up(reg)) reg->bar_frame = static_cast<double *>(reg->aql.alloc_bar(kBarFramePoints * 24));
            }
            if (reg->bar_frame) {
                std::memcpy(reg->bar_frame, frame_xyz, n * 24);
                _mm_sfence();
                return run_registration(reg, map, reg->bar_frame, n, last_pose_qt, rel_odom_qt, max_correspondence_distance, out_pose_qt, stats);
            }
        }
        if (int rc = ensure_frame(reg, n)) return rc;
        if (int rc = aql_quiesce(reg)) return rc;
        reg->stream_dirty = true;  // (the kernels of earlier calls have long read d_frame: the host had their results)
        if (reg->fetch_frames) {
            if (int rc = fetch_upload<double>(reg, frame_xyz, n)) return rc;
        } else if (int rc = staged_upload(reg->stage, 0, reg->d_frame, frame_xyz, n * 24, reg->stream)) {
            return rc;
        }
    }
    return run_registration(reg, map, reg->d_frame, n, last_pose_qt, rel_odom_qt, max_correspondence_distance, out_pose_qt, stats);
}
// ComputeRobotMotion on a frame that is still float32 - what a PointCloud2 carries on the wire (RosUtils.cpp:30-39 widens every
// coordinate with static_cast<double> on the host before the reference ever sees it): half the bytes cross PCIe, the widening
// happens on the device (exact, so the registration sees the very doubles the reference sees).
int kicp_register_f32(kicp_reg *reg, kicp_map *map, const float *frame_xyz_f32, size_t n, const double last_pose_qt[7],
                      const double rel_odom_qt[7], double max_correspondence_distance, double out_pose_qt[7], kicp_stats *stats) {
    KICP_TRACE_CALL();
    if (!reg || !map || (!frame_xyz_f32 && n)) return fail(KICP_ERR_ARG, "null argument");
    if (!kicp_map_empty(map) && n) {
        if (int rc = set_device(reg->device)) return rc;
        if (reg->use_bar_frame && n <= kBarFramePoints) {  // small frames: widened by the CPU on their way through the BAR
            if (!reg->bar_frame && !reg->bar_frame_tried) {
                reg->bar_frame_tried = true;
                if (aql_up(reg)) reg->bar_frame = static_cast<double *>(reg->aql.alloc_bar(kBarFramePoints * 24));
            }
            if (reg->bar_frame) {
                for (size_t i = 0; i < 3 * n; ++i) reg->bar_frame[i] = static_cast<double>(frame_xyz_f32[i]);
                _mm_sfence();
                return run_registration(reg, map, reg->bar_frame, n, last_pose_qt, rel_odom_qt, max_correspondence_distance, out_pose_qt, stats);
            }
        }
        if (n > 0x7FFFFFF0ull / 3) return fail(KICP_ERR_CAPACITY, "frame too large");
        if (int rc = ensure_frame(reg, n)) return rc;
        if (int rc = aql_quiesce(reg)) return rc;
        reg->stream_dirty = true;
        if (int rc = fetch_upload<float>(reg, frame_xyz_f32, n)) return rc;
    }
    return run_registration(reg, map, reg->d_frame, n, last_pose_qt, rel_odom_qt, max_correspondence_distance, out_pose_qt, stats);
}
// KinematicRegistration(const KinematicRegistration &): the reference's struct is a plain copyable aggregate
// (Registration.hpp:32-50).  A new handle on the same device with the same parameters and tuning options, and workspaces of its
// own; multi-GPU exchanges (communicator, shared segment, mailboxes, callback) are per handle and are NOT carried over.
int kicp_reg_clone(const kicp_reg *reg, kicp_reg **out) {
    if (!reg || !out) return fail(KICP_ERR_ARG, "null argument");
    kicp_reg *c = nullptr;
    if (int rc = kicp_reg_create(&reg->cfg, reg->device, &c)) return rc;
    c->use_bar_frame = reg->use_bar_frame, c->fetch_frames = reg->fetch_frames;
    c->wait_mode = reg->wait_mode, c->timing = reg->timing;
    c->query_every = reg->query_every, c->lanes_per_query = reg->lanes_per_query, c->latency_kernel = reg->latency_kernel;
    c->p2p_rows = reg->p2p_rows, c->use_aql = reg->use_aql;
    c->small_cmd = reg->cmd_bar ? 1 : reg->small_cmd, c->use_small = reg->use_small, c->small_block = reg->small_block, c->small_wave = reg->small_wave;
    c->wave_block = reg->wave_block, c->small_resident = reg->small_resident, c->small_timeout_us = reg->small_timeout_us, c->small_group_rows = reg->small_group_rows;
    c->resident_generic = reg->resident_generic, c->batch_resident = reg->batch_resident, c->batch_depth = reg->batch_depth, c->batch_rotate = reg->batch_rotate, c->batch_queues = reg->batch_queues, c->batch_threads = reg->batch_threads ;
    *out = c;
    return KICP_OK;
}
// DataAssociation's output for one pose (Registration.cpp:62-81), from the very kernel the handle would register this scan with: a
// registration of ONE iteration at `pose` (last pose = pose, odometry = identity) whose pass kernel is the EXPORT instantiation of the
// build that scan size and the handle's options select (launch_pass / launch_small).
int kicp_pass_correspondences(kicp_reg *reg, kicp_map *map, const double *frame_xyz, size_t n, const double pose_qt[7], double max_correspondence_distance,
                              int32_t *out_index, double *out_d2, double *out_nn_xyz) {
    KICP_TRACE_CALL();
    if (!reg || !map || (!frame_xyz && n) || !pose_qt || (n && (!out_index || !out_d2 || !out_nn_xyz))) return fail(KICP_ERR_ARG, "null argument");
    if (n == 0) return KICP_OK;
    if (n > 0x7FFFFFF0ull / 3) return fail(KICP_ERR_CAPACITY, "frame too large");
    if (reg->comm || reg->allreduce_fn || reg->shm || reg->d_p2p_table) return fail(KICP_ERR_ARG, "detach the multi-GPU exchange first: correspondences are exported per device");
    if (int rc = set_device(reg->device)) return rc;
    if (int rc = ensure_frame(reg, n)) return rc;
    if (int rc = staged_upload(reg->stage, 0, reg->d_frame, frame_xyz, n * 24, reg->stream)) return rc;
    reg->stream_dirty = true;
    unsigned char *buf = nullptr;
    HIP_TRY(hipMalloc(&buf, n * 36));
    reg->corr_nn = reinterpret_cast<double *>(buf), reg->corr_d2 = reg->corr_nn + 3 * n, reg->corr_index = reinterpret_cast<int32_t *>(reg->corr_d2 + n);
    hipError_t e = hipMemsetAsync(buf, 0, n * 32, reg->stream);
    if (e == hipSuccess) e = hipMemsetAsync(reg->corr_index, 0xFF, n * 4, reg->stream);  // (-1: an empty map returns before any kernel runs)
    const int max_it = reg->cfg.max_num_iterations;
    reg->cfg.max_num_iterations = 1;
    const double identity[7] = {0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0};
    double pose_out[7];
    int rc = e == hipSuccess ? run_registration(reg, map, reg->d_frame, n, pose_qt, identity, max_correspondence_distance, pose_out, nullptr) : KICP_ERR_HIP;
    reg->cfg.max_num_iterations = max_it;
    reg->corr_index = nullptr, reg->corr_d2 = reg->corr_nn = nullptr;
    if (e == hipSuccess && rc >= 0) e = hipStreamSynchronize(reg->stream);
    if (e == hipSuccess && rc >= 0) e = hipMemcpy(out_nn_xyz, buf, n * 24, hipMemcpyDeviceToHost);
    if (e == hipSuccess && rc >= 0) e = hipMemcpy(out_d2, buf + n * 24, n * 8, hipMemcpyDeviceToHost);
    if (e == hipSuccess && rc >= 0) e = hipMemcpy(out_index, buf + n * 32, n * 4, hipMemcpyDeviceToHost);
    (void)hipFree(buf);
    if (e != hipSuccess) return fail(KICP_ERR_HIP, std::string("kicp_pass_correspondences: ") + hipGetErrorString(e));
    if (rc < 0) return rc;
    for (size_t i = 0; i < n; ++i)
        if (out_index[i] < 0) out_d2[i] = DBL_MAX;
    return KICP_OK;  // (a pass without correspondences is a result here, not a warning)
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
    if (int rc = ensure_partials(reg, pass_grid(reg, n))) return rc;
    if (int rc = staged_upload(reg->stage, 0, reg->d_frame, frame_xyz, n * 24, reg->stream)) return rc;
    const unsigned long long call_id = ++reg->call_id;
    PassParams pp{};
    pp.partials = reg->d_partials, pp.tickets = reg->d_tickets;
    pp.src = reg->d_frame, pp.n = static_cast<uint32_t>(n), pp.map = map->mirror.view, pp.tau = max_correspondence_distance;
    pp.st = reg->d_state, pp.search = search_params(max_correspondence_distance, map->mirror.view.voxel_size);
    set_pose(pp.sol, pose_from(pose_qt));
    pp.sol.pass = 0, pp.sol.mode = 1, pp.sol.call_id = call_id, pp.sol.rec = reg->d_rec;
    if (int rc = launch_pass(reg, pp)) return rc;
    hipLaunchKernelGGL(k_publish_sums, dim3(1), dim3(64), 0, reg->stream, reg->d_state, reg->d_rec, call_id);
    HIP_TRY(hipGetLastError());
    unsigned long long seq = 0;
    if (int rc = wait_record(reg, call_id, 1, true, &seq)) return rc;
    for (int i = 0; i < 7; ++i) out_sums[i] = reg->rec->sums[i];
    if (out_words) HIP_TRY(hipMemcpy(out_words, reg->d_state->reduce, sizeof(long long) * kReduceWords, hipMemcpyDeviceToHost));
    return KICP_OK;
}

size_t kicp_aql_kernel_names(char *out, size_t cap) {
    // every (template instantiation of a) kernel launch_pass / launch_small may dispatch through the AQL queue, in the form
    // aql_kernel_for / aql_small_kernel_for look it up
    std::string all;
    char name[128];
    all += "void kicp::k_pass_gather32<256, 1, 2, false, true, false>(\n";
    all += "void kicp::k_pass_gather32<256, 1, 4, false, false, false>(\n";
    all += "void kicp::k_pass_gather32<256, 2, 4, true, false, false>(\n";
    all += "void kicp::k_pass_gather32<256, 4, 4, false, false, false>(\n";
    for (int g : {1, 2, 4}) {
        std::snprintf(name, sizeof name, "void kicp::k_pass_small<256, %d, false>(\n", g);
        all += name;
    }
    for (int b : {256, 512, 1024}) {
        std::snprintf(name, sizeof name, "void kicp::k_pass_wave<%d, false>(\n", b);
        all += name;
    }
    all += "void kicp::k_pass_resident<256, 2, true>(\n";
    if (out && cap) {
        const size_t n = std::min(cap - 1, all.size());
        std::memcpy(out, all.data(), n);
        out[n] = '\0';
    }
    return all.size() + 1;
}

}  // extern "C"
