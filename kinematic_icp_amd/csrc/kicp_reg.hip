// kicp_reg.hip -- KinematicRegistration::ComputeRobotMotion behind include/kicp.h: the registration handle, the per-iteration
// launch / hand-off / host-side solve loop (kicp_kernels.hpp), and the multi-GPU exchanges (host shared segment, RCCL bound
// at run time, caller-supplied all-reduce).
#include <dlfcn.h>
#include <fcntl.h>
#include <sched.h>
#include <rccl/rccl.h>  // declarations only; the library is bound at run time (see CommApi)
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>

#include "kicp_aql.hpp"
#include "kicp_internal.hpp"
#include "kicp_kernels.hpp"
#include "kicp_small.hpp"

using namespace kicp;
using namespace kicp::host;

namespace {
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

// ---- RCCL, bound lazily so that single-GPU users never load it ------------------------------------------------
struct CommApi {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;  // (optional: diagnostics only)
    // (optional: one sub-communicator per lane of a sharded batch call - run_batch_queues; without it such a batch registers scan after scan)
    ncclResult_t (*CommSplit)(ncclComm_t, int, int, ncclComm_t *, void *) = nullptr;
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
        CommCount = reinterpret_cast<decltype(CommCount)>(dlsym(handle, "ncclCommCount"));
        CommSplit = reinterpret_cast<decltype(CommSplit)>(dlsym(handle, "ncclCommSplit"));
        if (!GetUniqueId || !CommInitRank || !CommDestroy || !AllReduce || !GetErrorString) {
            err = "librccl lacks an expected symbol";
            return false;
        }
        return true;
    }
};
CommApi g_comm;

}  // namespace

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
    unsigned long long *d_group_acc = nullptr;  // the resident kernels' group accumulators (finish_pass, ROWS_ONLY)
    bool acc_dirty = false;       // a resident launch was left before all its passes were collected: accumulators / tickets may hold partial counts
    size_t partial_blocks = 0;
    // mode 4 hand-off: tagged rows of the first-level groups in host-mapped pinned memory, added up by the host
    unsigned long long *rows = nullptr, *d_rows = nullptr;  // host / device view
    size_t rows_groups = 0;
    uint32_t tag = 0;       // tag of the last pass (1..65535)
    double *d_frame = nullptr;  // device copy of host frames
    size_t frame_cap = 0;
    HostStage stage;            // pinned staging for transfers from / to caller memory
    // small host frames skip the DMA engine altogether: the CPU writes them through the PCIe BAR into host-visible HBM
    double *bar_frame = nullptr;  // (the same address on both sides)
    int use_bar_frame = 1;        // option "bar_frame"
    int fetch_frames = 1;         // option "fetch_upload": larger host frames are pulled by the GPU out of the staging buffer piece by piece (1) | DMA engine (0)
    bool bar_frame_tried = false;
    // options
    int wait_mode = 0;    // 0 poll the host-mapped record; 1 hipStreamSynchronize
    int timing = 0;       // record HIP events around the call -> stats.gpu_ms
    int dbg = 0;
    // kicp_pass_correspondences: device buffers the EXPORT instantiations of the pass kernels write the per-query decisions to (set for
    // the duration of that call only)
    int32_t *corr_index = nullptr;
    double *corr_d2 = nullptr, *corr_nn = nullptr;
    int query_every = 512; // polls between hipStreamQuery calls while waiting (a call costs ~1 us of host time)
    int lanes_per_query = 0;  // variant 3: sub-lanes sharing one query (1, 2 or 4); 0 = by scan size
    int latency_kernel = 1;   // variant 3, one lane per query: the two-voxels-per-round build (0 never | 1 scans <= kLatencyMaxPoints | 2 always)
    // multi-GPU
    // RCCL: the communicator of single calls; a sharded batch call (run_batch_queues) gives every lane a communicator of its own, split off
    // this one on first use, so that each lane issues ITS collectives in its own fixed order on its own stream (lane_comms: owned here,
    // lent to the lanes' handles for the duration of a call)
    ncclComm_t lane_comms[8] = {};
    bool lane_comms_failed = false;
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
    void *shm_base = nullptr;  // start of the mapping (header slot first)
    ShmSlot *shm = nullptr;    // host view: [2 buffers][nranks]
    ShmSlot *d_shm = nullptr;  // device view of the same memory
    size_t shm_bytes = 0;
    unsigned long long shm_step = 0;  // hand-offs issued so far (same on every rank)
    // Sharded batches with several scans in flight (run_batch_queues): lane j of the batch call owns the slots [2 buffers][nranks]
    // behind the single-call area, area 1 + j, and counts its own hand-offs - lane j registers scans j, j + lanes, j + 2 lanes ... on
    // EVERY rank, so its sequence of exchanges is the same everywhere whatever order the lanes' passes complete in.
    static constexpr int kShmLanes = 8;  // (= kMaxBatchQueues)
    unsigned long long shm_lane_step[kShmLanes] = {};
    bool shm_poisoned = false;  // a sharded batch failed half-way: the ranks' lane counters may disagree until the segment is set up again
    std::string shm_name;
    // one-shot exchange over peer mappings (kicp_reg_p2p_*): this rank's mailbox in its own HBM (fine-grained), the peers'
    // mailboxes as IPC mappings, and the table of all of them the pass kernel reads
    unsigned long long *p2p_box = nullptr;
    int p2p_rows = 1;  // peer mailboxes, wire format (the same on every rank): 1 the first-level group rows themselves - a launch of more than
                       // kP2pMaxGroups groups sends its total as one row -, 2 always that single row, 0 the totals as tagged halves (round 2's)
    void *p2p_mapped[kP2pMaxRanks] = {};
    unsigned long long **d_p2p_table = nullptr;
    unsigned long long p2p_step = 0;  // exchanges issued so far (same on every rank)
    bool p2p_poisoned = false;        // a registration failed while the mailboxes were attached: the ranks may be out of step
    // direct AQL dispatch of the pass kernel (kicp_aql.hpp): the handle's own user-mode queue next to its HIP stream
    AqlDispatcher aql;
    int use_aql = 1;            // option "aql": 1 (default) dispatch the pass kernel with hand-written AQL packets where possible, 0 always through HIP
    bool aql_tried = false;     // set-up attempted (it is lazy: the first registration pays for it)
    bool stream_dirty = true;   // HIP work may be pending on `stream`: synchronise before the next AQL dispatch
    bool last_via_aql = false;  // how the pass the host is waiting for was launched
    std::map<int, const AqlKernel *> aql_kernels;
    // small-scan path (kicp_small.hpp): the command line the resident kernel polls (host-mapped, 64-byte aligned), the
    // sequence number of the last command issued, and the knobs
    unsigned long long *cmd = nullptr, *d_cmd = nullptr;
    unsigned long long *d_cmd_copies = nullptr;  // kCmdReplicas copies of the command line in device memory
    unsigned long long *cmd_bar = nullptr;       // host view of the same copies when they live in BAR-writable HBM (option "small_cmd" 1)
    int small_cmd = 1;            // option "small_cmd": 1 (default) the host writes the command copies through the BAR; 0 workgroup 0 relays the host line
    unsigned long long cmd_seq = 0;
    int use_small = 1;            // option "small": scans of up to kSmallMaxLanes lanes take k_pass_small
    int small_block = 256;        // option "small_block": its workgroup size (256 | 512 | 1024)
    int small_wave = 1;           // option "small_wave": scans of up to kWaveMaxPoints points take k_pass_wave (one wave per query)
    int wave_block = 0;           // option "wave_block": its workgroup size (256 | 512 | 1024; 0 = by scan size)
    int small_resident = 1;       // option "small_resident": the kernel stays for the call's later iterations
    int small_group_rows = 1;     // option "small_group_rows": the small-scan kernels' workgroups hand their sums over through their groups' counting
                                  // accumulators - one row per 32 workgroups crosses PCIe - 2 always | 0 never (round 3: every workgroup sends a row of
                                  // its own) | 1 (default) where it measured faster: the wave-per-query kernel with ONE pass out at a time (-1 us per
                                  // pass on cfg4: the host adds 5 rows instead of 135); with several passes in flight the rows' crossing is hidden
                                  // anyway and the accumulators' extra round trip to the L2 is not (+0.2 us), and k_pass_small's few rows gain nothing
    double small_timeout_us = 20000.0;  // option "small_timeout_us": how long a resident workgroup waits for a command
    double debug_stall_us = 0.0;  // tests: stall the host once before its next CONTINUE command (exercises the give-up path)
    unsigned long long small_relaunches = 0;  // launches repeated because a resident kernel gave up waiting
    int last_small = 0;           // 1 when the last registration ran on the small path
    int resident_generic = 1;     // option "resident_generic": scans beyond the small-scan kernels keep the generic kernel resident for a call's later iterations
    int batch_queues = 4;         // option "batch_queues": large scans of a batch in flight at a time, each on a queue of its own (run_batch_queues); < 2: off
    std::vector<kicp_reg *> batch_lanes;  // the handles those queues belong to (clones of this one, made on first use)
    unsigned long long batch_queue_passes = 0;  // passes served that way so far (get-only "batch_queue_passes")
    int batch_rotate = 1;         // option "batch_rotate": the workgroups of that kernel take turns at the parts of a scan (k_pass_resident)
    int batch_depth = 3;          // option "batch_depth": scans of a batch in flight at a time in that mode (run_batch_resident)
    int last_batch_threads = 0;   // resident kernels (= host threads) the last batch call ran side by side (get-only "batch_threads_active"; 0: not that path)
    int batch_threads = 8;        // option "batch_threads": batches of scans that leave most of the device empty: up to this many resident kernels at a
                                  // time - as many as fit the device side by side -, each serving a contiguous part of the batch from a host thread
                                  // of its own (run_batch_resident_threads); < 2: one kernel, the caller's thread
    int batch_resident = 1;       // option "batch_resident": kicp_register_device_batch keeps that kernel resident ACROSS the scans of the batch
    ScanRef *d_scans = nullptr;   // the batch's scan table (device memory)
    ScanRef *scans_bar = nullptr; // the same memory as the CPU writes it through the PCIe BAR (nullptr: d_scans is plain device memory)
    size_t scans_cap = 0;
    unsigned long long batch_resident_passes = 0;  // passes served that way so far (get-only "batch_resident_passes")
    int last_resident_passes = 0; // passes of the last call that a resident launch of the GENERIC kernel served (get-only "resident_passes")
    int small_prev_iters = 2;     // iterations of the previous small-path call: a scan that converged at once makes the next launch leave after its first pass
    uint32_t trace_pass = 1;      // the pass of a launch the stamps are taken on (the option's value)
    long long *d_trace = nullptr; // option "small_trace": device buffer of the kernel's per-pass wall-clock stamps
    double trace_host_us = 0.0, trace_dev_us = 0.0, trace_first_us = 0.0;  // host: rows seen -> command sent; device: command sent -> rows seen; launch -> first rows
    unsigned long long trace_n = 0, trace_first_n = 0;
};

namespace {

// Every spin-wait below is bounded by wall-clock time (default 20 s, KICP_WAIT_TIMEOUT_S): a wedged kernel or a dead peer
// rank turns into KICP_ERR_HIP / KICP_ERR_COMM instead of a hung caller.
double wait_timeout_s() {
    static const double t = [] {
        const char *e = std::getenv("KICP_WAIT_TIMEOUT_S");
        const double v = e ? std::atof(e) : 0.0;
        return v > 0.0 ? v : 20.0;
    }();
    return t;
}
struct Deadline {
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    bool passed() const { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > wait_timeout_s(); }
};

constexpr int kPassBlock = 256;  // workgroup size of every build of the generic pass kernel
// Sub-lanes per query of variant 3.  Small scans are latency bound (few waves, each lane's chain of dependent bucket
// visits decides the kernel time): spreading a query's neighbour voxels over 2-4 lanes shortens that chain.  Large
// scans already fill the machine and only pay for the extra waves.
constexpr size_t kLatencyMaxPoints = 131072;  // two waves per SIMD on 256 CUs
int lanes_for(const kicp_reg *r, size_t n) {
    if (r->lanes_per_query > 0) return r->lanes_per_query;
    return n <= 4096 ? 4 : (n <= 32768 ? 2 : 1);
}
uint32_t pass_grid(const kicp_reg *r, size_t n) {
    const size_t threads = n * static_cast<size_t>(lanes_for(r, n));
    return static_cast<uint32_t>(std::max<size_t>(1, (threads + kPassBlock - 1) / kPassBlock));
}
// AQL kernel objects, looked up once per template instantiation by DEMANGLED name (kicp_aql.hpp), or nullptr
bool aql_up(kicp_reg *r) {
    if (!r->aql_tried) {
        r->aql_tried = true;
        if (r->aql.init(r->device) != 0 && env_flag("KICP_TRACE")) std::fprintf(stderr, "[kicp] AQL dispatch unavailable: %s\n", r->aql.why.c_str());
        else if (env_flag("KICP_TRACE")) std::fprintf(stderr, "[kicp] AQL dispatch ready, kernel arguments in %s%s%s\n", r->aql.kernarg_place(), r->aql.why.empty() ? "" : "; ", r->aql.why.c_str());
    }
    if (r->aql.ready && r->aql.queue_error) {  // a dead queue: forget it, the handle goes on through its HIP stream
        if (env_flag("KICP_TRACE")) std::fprintf(stderr, "[kicp] AQL queue error %d: falling back to the HIP stream\n", r->aql.queue_error);
        r->aql.disable();
    }
    return r->aql.ready;
}
const AqlKernel *aql_lookup(kicp_reg *r, int key, const char *demangled_prefix) {
    if (!aql_up(r)) return nullptr;
    auto it = r->aql_kernels.find(key);
    if (it != r->aql_kernels.end()) return it->second;
    const AqlKernel &k = r->aql.kernel(demangled_prefix);
    return r->aql_kernels[key] = k.usable ? &k : nullptr;
}
// the names below must agree with tools/aql_kernel_names.py (tests/test_host.py checks them against build/kicp_reg.hsaco)
const AqlKernel *aql_kernel_for(kicp_reg *r, int b, int g, int occ, bool split, bool lat) {
    char name[128];
    std::snprintf(name, sizeof name, "void kicp::k_pass_gather32<%d, %d, %d, %s, %s, false>(", b, g, occ, split ? "true" : "false", lat ? "true" : "false");
    return aql_lookup(r, b * 1000 + g * 100 + occ * 10 + (split ? 1 : 0) + (lat ? 2 : 0), name);
}
const AqlKernel *aql_resident_kernel_for(kicp_reg *r, bool lat) {
    return lat ? aql_lookup(r, -7, "void kicp::k_pass_resident<256, 2, true>(") : nullptr;  // (the resident generic kernel exists as the latency-oriented build only)
}
const AqlKernel *aql_small_kernel_for(kicp_reg *r, int block, int g, bool wave) {
    char name[128];
    if (wave) std::snprintf(name, sizeof name, "void kicp::k_pass_wave<%d, false>(", block);
    else std::snprintf(name, sizeof name, "void kicp::k_pass_small<%d, %d, false>(", block, g);
    return aql_lookup(r, -(block * 10 + (wave ? 9 : g)), name);
}
// Before HIP work follows kernels that went through the handle's AQL queue: wait for them.  A time-out is an error (a kernel
// of ours may still be writing the buffers the next launch reuses); a queue error retires the dispatcher instead - its kernels
// are gone with the queue - and the handle goes on through HIP.
int aql_quiesce(kicp_reg *r) {
    if (!r->aql.busy()) return KICP_OK;
    if (r->aql.drain(wait_timeout_s())) return KICP_OK;
    if (r->aql.queue_error) {
        r->aql.disable();
        return KICP_OK;
    }
    return fail(KICP_ERR_HIP, "the AQL queue did not drain (KICP_WAIT_TIMEOUT_S)");
}
// allow_aql: nothing on the handle's HIP stream has to be ordered behind this kernel and the host will poll for the result
// The generic pass kernel comes in four builds, all of 256-thread workgroups (round 6: the 64 / 128 / 512-thread workgroups, the
// three-waves register budget, the voxel-dealing pair of sub-lanes and the plain fp64 gather never won an A/B and are gone):
//   one lane per query, two waves per SIMD, two neighbour voxels per round (LAT)   scans of up to 131 072 points, one call at a time
//   one lane per query, four waves per SIMD                                        larger scans; several scans in flight
//   two lanes per query sharing every bucket / four lanes per query               scans of up to 32 768 / 4 096 points
int launch_pass(kicp_reg *r, const PassParams &p, bool allow_aql = false) {
    const uint32_t grid = pass_grid(r, p.n);
    const int g = lanes_for(r, p.n);
    // the latency-oriented build (two neighbour voxels per round, two waves per SIMD): scans of one lane per query that
    // leave the machine at most two waves per SIMD anyway
    const bool lat = g == 1 && (r->latency_kernel == 2 || (r->latency_kernel == 1 && p.n <= kLatencyMaxPoints));
    const int occ = lat ? 2 : 4;
    const bool split = g == 2;
    // While HIP work may be pending on the handle's stream (a frame upload, a mirror refresh, a clear) the kernel goes
    // through the stream, ordered behind it; once the host has that pass's result the stream is known to be idle.
    if (p.corr_index) {  // kicp_pass_correspondences: the same build with the per-query decisions written out, through the HIP stream
        if (int rc = aql_quiesce(r)) return rc;
        r->last_via_aql = false;
        if (lat) hipLaunchKernelGGL((k_pass_gather32<kPassBlock, 1, 2, false, true, true>), dim3(grid), dim3(kPassBlock), 0, r->stream, p);
        else if (g == 1) hipLaunchKernelGGL((k_pass_gather32<kPassBlock, 1, 4, false, false, true>), dim3(grid), dim3(kPassBlock), 0, r->stream, p);
        else if (g == 2) hipLaunchKernelGGL((k_pass_gather32<kPassBlock, 2, 4, true, false, true>), dim3(grid), dim3(kPassBlock), 0, r->stream, p);
        else hipLaunchKernelGGL((k_pass_gather32<kPassBlock, 4, 4, false, false, true>), dim3(grid), dim3(kPassBlock), 0, r->stream, p);
        return KICP_OK;
    }
    if (allow_aql && r->use_aql && !r->stream_dirty) {
        if (const AqlKernel *k = aql_kernel_for(r, kPassBlock, g, occ, split, lat)) {
            // Fences of the packet.  Acquire: agent scope - the kernel start invalidates the vector / scalar L1s and the
            // XCDs' L2 lines of device memory, so everything earlier kernels released and every DMA the host has waited
            // for is seen; it is what makes a kernarg slot re-read from host memory, too (no acquire: stale arguments).
            // System scope costs 3.4 us more per dispatch on this part (measured: 25.9 vs 22.6 us per cfg2 scan).
            // (Dropping the acquire for the later passes of a call - same frame, same map - was measured too: no gain.)
            // Release: agent scope; the results leave through system-scope stores into host-mapped memory, and
            // AqlDispatcher::drain() puts a system-scope release behind the kernels before HIP work follows them.
            if (r->aql.dispatch(*k, grid, static_cast<uint32_t>(kPassBlock), &p, sizeof p, HSA_FENCE_SCOPE_AGENT, HSA_FENCE_SCOPE_AGENT)) {
                r->last_via_aql = true;
                return KICP_OK;
            }
        }
    }
    if (allow_aql) r->stream_dirty = false;  // the host waits for this pass: by then everything queued before it is done
    if (int rc = aql_quiesce(r)) return rc;  // kernels dispatched through the AQL queue come first
    r->last_via_aql = false;
    if (lat) hipLaunchKernelGGL((k_pass_gather32<kPassBlock, 1, 2, false, true>), dim3(grid), dim3(kPassBlock), 0, r->stream, p);
    else if (g == 1) hipLaunchKernelGGL((k_pass_gather32<kPassBlock, 1, 4, false>), dim3(grid), dim3(kPassBlock), 0, r->stream, p);
    else if (g == 2) hipLaunchKernelGGL((k_pass_gather32<kPassBlock, 2, 4, true>), dim3(grid), dim3(kPassBlock), 0, r->stream, p);
    else hipLaunchKernelGGL((k_pass_gather32<kPassBlock, 4, 4, false>), dim3(grid), dim3(kPassBlock), 0, r->stream, p);
    return KICP_OK;
}
int ensure_partials(kicp_reg *r, size_t blocks) {
    if (blocks <= r->partial_blocks) return KICP_OK;
    if (int rc = aql_quiesce(r)) return rc;
    if (r->d_partials) HIP_TRY(hipFree(r->d_partials));
    if (r->d_tickets) HIP_TRY(hipFree(r->d_tickets));
    if (r->d_group_acc) HIP_TRY(hipFree(r->d_group_acc));
    r->d_partials = nullptr, r->d_tickets = nullptr, r->d_group_acc = nullptr;
    const size_t want = blocks + blocks / 2 + 64, groups = want / kGroup + 2;
    HIP_TRY(hipMalloc(&r->d_partials, (want + groups) * kReduceWords * sizeof(unsigned long long)));
    HIP_TRY(hipMalloc(&r->d_tickets, groups * kTicketStride * sizeof(unsigned int)));
    HIP_TRY(hipMalloc(&r->d_group_acc, 2 * groups * kAccStride * sizeof(unsigned long long)));  // (two sets: an ordinary launch takes the set of its tag's parity)
    r->stream_dirty = true;
    HIP_TRY(hipMemsetAsync(r->d_tickets, 0, groups * kTicketStride * sizeof(unsigned int), r->stream));
    HIP_TRY(hipMemsetAsync(r->d_group_acc, 0, 2 * groups * kAccStride * sizeof(unsigned long long), r->stream));
    HIP_TRY(hipMemsetAsync(r->d_partials, 0, (want + groups) * kReduceWords * sizeof(unsigned long long), r->stream));  // tag 0 = never valid
    r->partial_blocks = want;
    return KICP_OK;
}
// host-mapped rows of the first-level groups (mode 4)
int ensure_rows(kicp_reg *r, size_t groups) {
    if (groups <= r->rows_groups) return KICP_OK;
    if (int rc = aql_quiesce(r)) return rc;
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
        if (int rc = aql_quiesce(r)) return rc;
        r->stream_dirty = true;
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
    const Deadline deadline;
    for (unsigned long long spins = 1;; ++spins) {
        const unsigned long long s = __atomic_load_n(seq, __ATOMIC_ACQUIRE);
        if (ready(s)) {
            *seq_out = s;
            return KICP_OK;
        }
        if (spins % query_every == 0) {
            if (r->last_via_aql) {  // the kernel went through the handle's own AQL queue: its error callback is the fault check
                if (r->aql.queue_error) return fail(KICP_ERR_HIP, "the AQL queue reported error " + std::to_string(r->aql.queue_error));
                if (deadline.passed()) return fail(KICP_ERR_HIP, "timed out waiting for the registration kernels (KICP_WAIT_TIMEOUT_S)");
                continue;
            }
            const hipError_t q = hipStreamQuery(r->stream);
            if (q != hipSuccess && q != hipErrorNotReady) return fail(KICP_ERR_HIP, std::string("stream fault: ") + hipGetErrorString(q));
            if (q == hipSuccess && ++drained > 4 && !ready(__atomic_load_n(seq, __ATOMIC_ACQUIRE)))
                return fail(KICP_ERR_HIP, "kernels finished without publishing a result");
            if (deadline.passed()) return fail(KICP_ERR_HIP, "timed out waiting for the registration kernels (KICP_WAIT_TIMEOUT_S)");
        }
    }
}

// The flag word of a GROUP's row (finish_pass): the sum over its <= kGroup workgroups of range_error (0 / 1 each) + kLostRowUnit once if
// the group's reader gave a row up + kGaveUpUnit per workgroup that left without a command.  The fields cannot run into each other
// inside one group's row (<= 32 in each), but their SUMS over the groups of a launch can (a cfg5 launch has 62 groups: 256 range
// errors would read as a lost row - ADVICE r4), so the host never adds flag words: every row's word is reduced to its three facts
// first and those are OR-ed.
long long row_flags(long long w) {
    const unsigned long long u = static_cast<unsigned long long>(w);
    return static_cast<long long>(((u & 0xFFull) ? 1ull : 0ull) | (((u >> 8) & 0xFFull) ? kLostRowUnit : 0ull) | ((u >> 16) ? kGaveUpUnit : 0ull));
}
// mode 4: add the tagged rows of the `groups` first-level groups as they arrive (word = value << 16 | tag)
int wait_rows(kicp_reg *r, size_t groups, uint32_t tag, long long out_words[kReduceWords], size_t first_row = 0) {
    for (int i = 0; i < kReduceWords; ++i) out_words[i] = 0;
    const unsigned query_every = r->query_every > 0 ? static_cast<unsigned>(r->query_every) : 64u;
    unsigned drained = 0;
    unsigned long long spins = 0;
    const Deadline deadline;
    for (size_t g = 0; g < groups; ++g) {
        const unsigned long long *row = r->rows + (first_row + g) * kReduceWords;
        long long v[kReduceWords];
        for (;;) {
            bool ok = true;
            for (int i = 0; i < kReduceWords; ++i) {
                const unsigned long long w = __atomic_load_n(row + i, __ATOMIC_RELAXED);
                ok = ok && (static_cast<uint32_t>(w) & 0xFFFFu) == tag;
                v[i] = static_cast<long long>(w) >> 16;
            }
            if (ok) break;
            if (r->last_via_aql) {
                if (++spins % query_every == 0) {
                    if (r->aql.queue_error) return fail(KICP_ERR_HIP, "the AQL queue reported error " + std::to_string(r->aql.queue_error));
                    if (deadline.passed()) return fail(KICP_ERR_HIP, "timed out waiting for the pass kernel's rows (KICP_WAIT_TIMEOUT_S)");
                }
                continue;
            }
            if (r->wait_mode == 1 || ++spins % query_every == 0) {
                // the query makes the runtime flush commands it may still hold back, and reports device faults
                const hipError_t q = r->wait_mode == 1 ? hipStreamSynchronize(r->stream) : hipStreamQuery(r->stream);
                if (q != hipSuccess && q != hipErrorNotReady) return fail(KICP_ERR_HIP, std::string("stream fault: ") + hipGetErrorString(q));
                if (q == hipSuccess && ++drained > 4) return fail(KICP_ERR_HIP, "kernels finished without publishing a result");
                if (deadline.passed()) return fail(KICP_ERR_HIP, "timed out waiting for the pass kernel's rows (KICP_WAIT_TIMEOUT_S)");
            }
        }
        for (int i = 0; i < kReduceWords; ++i)
            if (i != kNumLimbs) out_words[i] += v[i];
        out_words[kNumLimbs] |= row_flags(v[kNumLimbs]);
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    return KICP_OK;
}

// wait until every rank's slot of the current buffer carries `value`, then add the limb words (exact, order independent)
int wait_shm(kicp_reg *r, unsigned long long value, long long out_words[kReduceWords]) {
    const kicp_reg::ShmSlot *buf = r->shm + ((value - 1) & 1) * r->nranks;
    for (int i = 0; i < kReduceWords; ++i) out_words[i] = 0;
    const Deadline deadline;
    for (int k = 0; k < r->nranks; ++k) {
        const volatile unsigned long long *seq = &buf[k].seq;
        for (unsigned long long spins = 1; __atomic_load_n(seq, __ATOMIC_ACQUIRE) != value; ++spins) {
            if (spins % 4096 == 0) {
                const hipError_t q = hipStreamQuery(r->stream);
                if (q != hipSuccess && q != hipErrorNotReady) return fail(KICP_ERR_HIP, std::string("stream fault: ") + hipGetErrorString(q));
                if (deadline.passed()) return fail(KICP_ERR_COMM, "timed out waiting for a peer rank's hand-off (KICP_WAIT_TIMEOUT_S)");
            }
        }
        for (int i = 0; i < kReduceWords; ++i) out_words[i] += buf[k].words[i];
    }
    return KICP_OK;
}

// ---- the small-scan path (kicp_small.hpp) ------------------------------------------------------------------------------
// How the small-scan path runs a scan of n points: one wave per query (k_pass_wave, up to kWaveMaxPoints points) or G
// sub-lanes per query (k_pass_small, up to kSmallMaxLanes lanes); grid == 0: the scan does not fit, the generic path takes it.
struct SmallPlan {
    uint32_t grid = 0;
    int block = 256, g = 1;
    bool wave = false;
    bool generic = false, lat = false;  // generic: the generic pass kernel, resident (k_pass_resident); rows = group rows
};
// do the rows of a launch of plan `pl` reach the host per GROUP of workgroups (`pipelined`: several passes of the launch are out at a time)
bool grouped_rows(const kicp_reg *r, const SmallPlan &pl, bool pipelined) {
    if (pl.generic || r->small_group_rows == 2) return true;
    return r->small_group_rows == 1 && pl.wave && !pipelined;
}
SmallPlan small_plan(const kicp_reg *r, size_t n) {
    SmallPlan pl;
    if (n == 0) return pl;
    if (r->small_wave && n <= kWaveMaxPoints) {
        pl.wave = true;
        pl.block = r->wave_block ? r->wave_block : (n <= 512 ? 256 : (n <= 2176 ? 512 : 1024));  // <= kWaveMaxRows rows; 512 measured best at 1 080 points
        const size_t per_group = static_cast<size_t>(pl.block) / 64;
        pl.grid = static_cast<uint32_t>((n + per_group - 1) / per_group);
        if (pl.grid <= static_cast<uint32_t>(kWaveMaxRows)) return pl;
        pl.block = 1024, pl.grid = static_cast<uint32_t>((n + 15) / 16);
        return pl;
    }
    pl.g = lanes_for(r, n), pl.block = r->small_block;
    const size_t lanes = n * static_cast<size_t>(pl.g);
    if (lanes <= static_cast<size_t>(kSmallMaxLanes)) {
        pl.grid = static_cast<uint32_t>((lanes + pl.block - 1) / pl.block);
        return pl;
    }
    // larger scans: the generic kernel, resident while every workgroup fits on the device at once (one lane per query)
    // (the latency-oriented build only: with 235 VGPRs it keeps everything in registers across the pass loop, the four-waves-per-
    // SIMD build does not; two workgroups per CU)
    pl.lat = r->latency_kernel != 0;
    if (!pl.lat || !r->resident_generic || r->lanes_per_query > 1 || 
        n > kLatencyMaxPoints * static_cast<size_t>(std::max(1, r->num_cus)) / 256)
        return pl;  // (explicit kernel-shape options keep the plain kernel they name)
    pl.generic = true, pl.g = 1, pl.block = 256;
    pl.grid = static_cast<uint32_t>((n + 255) / 256);
    return pl;
}
int ensure_cmd(kicp_reg *r) {
    if (!r->cmd) {
        HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&r->cmd), kPipeSlots * kCmdWords * sizeof(unsigned long long), hipHostMallocMapped | hipHostMallocCoherent));
        std::memset(r->cmd, 0, kPipeSlots * kCmdWords * sizeof(unsigned long long));
        HIP_TRY(hipHostGetDevicePointer(reinterpret_cast<void **>(&r->d_cmd), r->cmd, 0));
    }
    const size_t bytes = static_cast<size_t>(kCmdReplicas) * kCmdStrideWords * sizeof(unsigned long long);
    if (r->small_cmd == 1 && !r->cmd_bar) {  // the copies in host-writable HBM: needs the HSA side of the AQL dispatcher
        if (r->d_cmd_copies) {
            if (int rc = aql_quiesce(r)) return rc;
            HIP_TRY(hipStreamSynchronize(r->stream));
            HIP_TRY(hipFree(r->d_cmd_copies));
            r->d_cmd_copies = nullptr;
        }
        if (aql_up(r)) r->cmd_bar = static_cast<unsigned long long *>(r->aql.alloc_bar(bytes));
        if (r->cmd_bar) {
            for (size_t i = 0; i < bytes / 8; ++i) r->cmd_bar[i] = 0ull;
            _mm_sfence();
            r->d_cmd_copies = r->cmd_bar;
        } else {
            if (env_flag("KICP_TRACE")) std::fprintf(stderr, "[kicp] command line in BAR-writable HBM unavailable (%s): relaying through workgroup 0\n", r->aql.why.c_str());
            r->small_cmd = 0;
        }
    }
    if (!r->d_cmd_copies) {
        HIP_TRY(hipMalloc(reinterpret_cast<void **>(&r->d_cmd_copies), bytes));
        HIP_TRY(hipMemset(r->d_cmd_copies, 0, bytes));
    }
    return KICP_OK;
}
// `count` consecutive pass tags (the same wrap rule as next_tag)
int next_tag_range(kicp_reg *r, uint32_t count, uint32_t *first) {
    if (r->tag + count > 0xFFFFu) r->tag = 0xFFFFu;  // not enough room before the wrap: wrap now
    if (int rc = next_tag(r, first)) return rc;
    r->tag += count - 1;
    return KICP_OK;
}
// the command that starts pass `seq - seq_base` of the resident kernel: seven pose words, then the control word (release); it
// travels in line seq % kPipeSlots (kicp_small.hpp).  `scan`: the scan of the launch's table the pass belongs to (batches)
void send_command(kicp_reg *r, unsigned long long seq, uint32_t op, const Pose &T, uint32_t scan = 0u) {
    unsigned long long w[kCmdWords];
    const double v[7] = {T.qx, T.qy, T.qz, T.qw, T.tx, T.ty, T.tz};
    std::memcpy(w, v, 7 * sizeof(double));
    w[7] = (((seq & 0xFFFFFFFFull) << 32) | (static_cast<unsigned long long>(scan) << 8) | op) ^ cmd_fold(w);
    const size_t slot = static_cast<size_t>(seq % kPipeSlots) * kCmdWords;
    if (r->small_cmd == 1 && r->cmd_bar) {  // straight into the copies the workgroups poll (write-combined BAR stores)
        for (int c = 0; c < kCmdReplicas; ++c)
            for (int i = 0; i < kCmdWords; ++i) r->cmd_bar[static_cast<size_t>(c) * kCmdStrideWords + slot + i] = w[i];
        _mm_sfence();
        return;
    }
    for (int i = 0; i < 7; ++i) __atomic_store_n(r->cmd + slot + i, w[i], __ATOMIC_RELAXED);
    __atomic_store_n(r->cmd + slot + 7, w[7], __ATOMIC_RELEASE);
}
int launch_small(kicp_reg *r, const SmallParams &sp, const SmallPlan &pl) {
    const int b = pl.block, g = pl.g;
    const uint32_t grid = pl.grid;
    if (sp.p.corr_index) {  // kicp_pass_correspondences: the same kernels with the per-query decisions written out, through the HIP stream
        if (pl.generic) return fail(KICP_ERR_ARG, "correspondences are exported by one-pass launches");
        if (int rc = aql_quiesce(r)) return rc;
        r->last_via_aql = false, r->stream_dirty = false;
        if (pl.wave && b == 1024) hipLaunchKernelGGL((k_pass_wave<1024, true>), dim3(grid), dim3(1024), 0, r->stream, sp);
        else if (pl.wave && b == 512) hipLaunchKernelGGL((k_pass_wave<512, true>), dim3(grid), dim3(512), 0, r->stream, sp);
        else if (pl.wave) hipLaunchKernelGGL((k_pass_wave<256, true>), dim3(grid), dim3(256), 0, r->stream, sp);
        else if (g == 1) hipLaunchKernelGGL((k_pass_small<256, 1, true>), dim3(grid), dim3(256), 0, r->stream, sp);
        else if (g == 2) hipLaunchKernelGGL((k_pass_small<256, 2, true>), dim3(grid), dim3(256), 0, r->stream, sp);
        else hipLaunchKernelGGL((k_pass_small<256, 4, true>), dim3(grid), dim3(256), 0, r->stream, sp);
        HIP_TRY(hipGetLastError());
        return KICP_OK;
    }
    if (r->use_aql && !r->stream_dirty) {
        if (const AqlKernel *k = pl.generic ? aql_resident_kernel_for(r, pl.lat) : aql_small_kernel_for(r, b, g, pl.wave)) {
            if (r->aql.dispatch(*k, grid, static_cast<uint32_t>(b), &sp, sizeof sp, HSA_FENCE_SCOPE_AGENT, HSA_FENCE_SCOPE_AGENT)) {
                r->last_via_aql = true;
                return KICP_OK;
            }
        }
    }
    r->stream_dirty = false;  // the host waits for this kernel's rows: by then everything queued before it is done
    if (int rc = aql_quiesce(r)) return rc;
    r->last_via_aql = false;
    if (pl.generic && pl.lat) hipLaunchKernelGGL((k_pass_resident<256, 2, true>), dim3(grid), dim3(256), 0, r->stream, sp);
    else if (pl.generic) return fail(KICP_ERR_ARG, "the resident generic kernel exists as the latency-oriented build only");
    else if (pl.wave && b == 1024) hipLaunchKernelGGL((k_pass_wave<1024>), dim3(grid), dim3(1024), 0, r->stream, sp);
    else if (pl.wave && b == 512) hipLaunchKernelGGL((k_pass_wave<512>), dim3(grid), dim3(512), 0, r->stream, sp);
    else if (pl.wave) hipLaunchKernelGGL((k_pass_wave<256>), dim3(grid), dim3(256), 0, r->stream, sp);
    else if (g == 1) hipLaunchKernelGGL((k_pass_small<256, 1>), dim3(grid), dim3(256), 0, r->stream, sp);
    else if (g == 2) hipLaunchKernelGGL((k_pass_small<256, 2>), dim3(grid), dim3(256), 0, r->stream, sp);
    else hipLaunchKernelGGL((k_pass_small<256, 4>), dim3(grid), dim3(256), 0, r->stream, sp);
    HIP_TRY(hipGetLastError());
    return KICP_OK;
}
// add the rows of the `grid` workgroups as they arrive; out_words in the layout of the all-reduce payload (three 40-bit limbs
// per sum, then the range flag).  *gave_up: a resident workgroup left without having seen the command of this pass.
int wait_rows_small(kicp_reg *r, uint32_t grid, uint32_t tag, uint32_t parity, long long out_words[kReduceWords], bool *gave_up) {
    __int128 total[kNumSums] = {};
    unsigned long long flags = 0;
    const unsigned query_every = r->query_every > 0 ? static_cast<unsigned>(r->query_every) : 64u;
    unsigned drained = 0;
    unsigned long long spins = 0;
    const Deadline deadline;
    for (uint32_t g = 0; g < grid; ++g) {
        const unsigned long long *row = r->rows + (static_cast<size_t>(parity) * grid + g) * kSmallRowWords;
        unsigned long long w[kSmallRowWords];
        // the rows land within a few microseconds of each other, and every line the device has just written misses the CPU's
        // caches: ask for the lines a few rows ahead while this row is being checked
        __builtin_prefetch(row + 6 * kSmallRowWords), __builtin_prefetch(row + 6 * kSmallRowWords + 8);
        for (;;) {
            bool ok = true;
            for (int i = 0; i < kSmallRowWords; ++i) {
                w[i] = __atomic_load_n(row + i, __ATOMIC_RELAXED);
                ok = ok && (static_cast<uint32_t>(w[i]) & 0xFFFFu) == tag;
            }
            if (ok) break;
            if (++spins % query_every != 0) continue;
            if (r->last_via_aql) {
                if (r->aql.queue_error) return fail(KICP_ERR_HIP, "the AQL queue reported error " + std::to_string(r->aql.queue_error));
            } else {
                const hipError_t q = hipStreamQuery(r->stream);  // makes the runtime flush what it may hold back; reports faults
                if (q != hipSuccess && q != hipErrorNotReady) return fail(KICP_ERR_HIP, std::string("stream fault: ") + hipGetErrorString(q));
                if (q == hipSuccess && ++drained > 4) return fail(KICP_ERR_HIP, "kernels finished without publishing a result");
            }
            if (deadline.passed()) return fail(KICP_ERR_HIP, "timed out waiting for the small-scan kernel's rows (KICP_WAIT_TIMEOUT_S)");
        }
        for (int i = 0; i < kNumSums; ++i)
            total[i] += static_cast<__int128>(w[2 * i] >> 16) + (static_cast<__int128>(static_cast<long long>(w[2 * i + 1]) >> 16) << 48);
        flags |= w[2 * kNumSums] >> 16;
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    for (int i = 0; i < kReduceWords; ++i) out_words[i] = 0;
    const unsigned __int128 m40 = (static_cast<unsigned __int128>(1) << 40) - 1;
    for (int i = 0; i < kNumSums; ++i) {
        const unsigned __int128 u = static_cast<unsigned __int128>(total[i]);
        out_words[3 * i] = static_cast<long long>(u & m40), out_words[3 * i + 1] = static_cast<long long>((u >> 40) & m40);
        out_words[3 * i + 2] = static_cast<long long>(total[i] >> 80);
    }
    out_words[kNumLimbs] = (flags & 1ull) ? 1 : 0;
    *gave_up = (flags & kSmallGaveUp) != 0;
    return KICP_OK;
}

// What the host does with the exact sums of one pass: Registration.cpp:119-125 (solve), :159-167,181-182 (update), :184 (stop
// test), and on pass 0 the regularisation of :48-60,171-177.  Shared by the generic and the small-scan loops.
struct HostLoop {
    Pose T;
    double beta = 0.0;
    int iter = 0, converged = 0, nan_flag = 0;
    // returns true when the loop ends with this pass
    bool step(const kicp_reg *r, const long long words[kReduceWords], kicp_stats *stats) {
        const int it = iter;
        double sums[kNumSums];
        for (int i = 0; i < kNumSums; ++i) sums[i] = host_limbs_to_double(words + 3 * i);
        const bool range_error = words[kNumLimbs] != 0;
        const double n = sums[6];
        if (it == 0) beta = r->cfg.use_adaptive_odometry_regularization ? 1.0 / (sums[5] / n + DBL_MIN) : r->cfg.fixed_regularization;
        double dx0, dx1;
        solve_perturbation(sums, n, beta, dx0, dx1);
        T = pose_mul(T, motion_model(dx0, dx1));
        iter = it + 1;
        if (stats && it < KICP_MAX_LOG_PASSES) {
            stats->n_corr[it] = n;
            for (int j = 0; j < 6; ++j) stats->sums[it][j] = sums[j];
            stats->dx[it][0] = dx0, stats->dx[it][1] = dx1;
        }
        if (std::sqrt(dx0 * dx0 + dx1 * dx1) < r->cfg.convergence_criterion) {  // Registration.cpp:184
            converged = 1;
            return true;
        }
        if (!(n > 0.0) || range_error) {
            // 0/0: the pose is NaN from here on.  The reference keeps iterating to max_num_iterations (no NaN ever passes the
            // stop test, every later association is empty, Registration.cpp:179-187); those passes cannot change anything, so
            // they are accounted for without being run.
            nan_flag = range_error ? 2 : 1;
            const int max_it = r->cfg.max_num_iterations;
            for (int j = iter; stats && j < max_it && j < KICP_MAX_LOG_PASSES; ++j) {
                stats->n_corr[j] = 0.0;
                for (int q = 0; q < 6; ++q) stats->sums[j][q] = 0.0;
                stats->dx[j][0] = stats->dx[j][1] = std::nan("");
            }
            iter = max_it;
            return true;
        }
        return iter >= r->cfg.max_num_iterations;
    }
};

// A workgroup of an earlier resident launch of the generic kernel gave up waiting for its command (k_pass_resident sets the word),
// or the host left a launch with passes still out: counts of a round that never completed - the call ended first - may be left
// behind in the groups' accumulators (and tickets).  Clear them before they are counted into this call's passes.
int clear_stale_tickets(kicp_reg *r) {
    if ((__atomic_load_n(&r->rec->reserved[0], __ATOMIC_RELAXED) == 0u && !r->acc_dirty) || !r->d_tickets) return KICP_OK;
    if (int rc = aql_quiesce(r)) return rc;
    r->stream_dirty = true;
    HIP_TRY(hipMemsetAsync(r->d_tickets, 0, (r->partial_blocks / kGroup + 2) * kTicketStride * sizeof(unsigned int), r->stream));
    HIP_TRY(hipMemsetAsync(r->d_group_acc, 0, 2 * (r->partial_blocks / kGroup + 2) * kAccStride * sizeof(unsigned long long), r->stream));
    __atomic_store_n(&r->rec->reserved[0], 0u, __ATOMIC_RELAXED);
    r->acc_dirty = false;
    return KICP_OK;
}

constexpr int kMaxGiveUps = 16;  // launches in a row that may end without a completed pass before the call fails
int run_small(kicp_reg *r, kicp_map *map, const double *d_frame, size_t n, const SmallPlan &pl, const Pose &T0, double tau, double out_pose_qt[7],
              kicp_stats *stats) {
    const int max_it = r->cfg.max_num_iterations;
    const uint32_t grid = pl.grid;
    // generic plan: a launch that will not stay goes out as the plain pass kernel (launch_pass, its own grid)
    const size_t groups_resident = (grid + kGroup - 1) / kGroup, groups_plain = (pass_grid(r, n) + kGroup - 1) / kGroup;
    // (rows, tickets and host rows of a resident launch are double-buffered by pass parity: finish_pass, small_publish)
    // grouped: the launch's rows are GROUP rows (the generic kernel's; the small-scan kernels' with "small_group_rows")
    const bool grouped = grouped_rows(r, pl, false);
    if (grouped) {
        if (int rc = ensure_partials(r, std::max<uint32_t>(kPipeSlots * grid, pl.generic ? pass_grid(r, n) : 0u))) return rc;
        if (int rc = ensure_rows(r, std::max(kPipeSlots * groups_resident, pl.generic ? groups_plain : size_t(0)))) return rc;
    } else if (int rc = ensure_rows(r, kPipeSlots * static_cast<size_t>(grid))) {
        return rc;
    }
    if (int rc = ensure_cmd(r)) return rc;
    SmallParams sp{};
    PassParams &pp = sp.p;
    pp.src = d_frame, pp.n = static_cast<uint32_t>(n), pp.map = map->mirror.view, pp.tau = tau, pp.st = r->d_state;
    pp.search = search_params(tau, map->mirror.view.voxel_size);
    pp.sol.max_iterations = max_it, pp.sol.convergence_criterion = r->cfg.convergence_criterion, pp.sol.mode = 4;
    pp.dbg = r->dbg;  // (0, or 14: the in-process A/B switch of the plain launch's hand-over)
    pp.corr_index = r->corr_index, pp.corr_d2 = r->corr_d2, pp.corr_nn = r->corr_nn;  // (kicp_pass_correspondences; nullptr otherwise)
    if (grouped) pp.partials = r->d_partials, pp.tickets = r->d_tickets, pp.group_acc = r->d_group_acc, pp.sol.pub_rows = r->d_rows, pp.sol.call_id = ++r->call_id, pp.sol.rec = r->d_rec;
    if (grouped)
        if (int rc = clear_stale_tickets(r)) return rc;
    sp.cmd = r->d_cmd, sp.rows = r->d_rows, sp.cmd_dev = r->d_cmd_copies, sp.relay = (r->small_cmd == 1 && r->cmd_bar) ? 0 : 1;
    sp.group_rows = grouped ? 1 : 0;
    sp.timeout_ticks = static_cast<long long>(std::max(50.0, r->small_timeout_us) * 100.0);  // 100 MHz wall clock
    HostLoop loop;
    loop.T = T0;
    bool finished = false;
    int give_ups = 0;  // consecutive launches that ended in a give-up without a pass completed
    while (!finished) {
        if (give_ups > kMaxGiveUps)
            return fail(KICP_ERR_HIP, "the resident pass kernel gave up waiting for its command in " + std::to_string(give_ups) + " launches in a row (small_timeout_us too short for this host?)");
        if (give_ups > 0 && grouped) {
            // workgroups of the launch that gave up may have added (partial, marked) contributions to the accumulators / tickets of the
            // slot the fresh launch's pass will use: wait for that kernel to be gone and clear them
            r->acc_dirty = true;
            if (int rc = clear_stale_tickets(r)) return rc;
        }
        const uint32_t left = static_cast<uint32_t>(max_it - loop.iter);
        // Residency pays from the second pass on and costs ~1 us when there is none (the kernel lingers until it sees STOP, and
        // the next dispatch waits for it).  Consecutive scans of a drive need about the same number of iterations, so the first
        // launch of a call stays resident only if the previous call needed more than one; a call that turns out to need more
        // gets a resident launch for the rest.
        const bool stay = r->small_resident == 1 ? (loop.iter > 0 || r->small_prev_iters > 1) : r->small_resident != 0;
        const uint32_t cnt = stay ? std::min(left, kSmallMaxPasses) : 1u;
        if (int rc = next_tag_range(r, cnt, &sp.tag0)) return rc;
        set_pose(pp.sol, loop.T), pp.sol.pass = loop.iter;
        sp.max_passes = cnt, sp.seq_base = r->cmd_seq;
        r->cmd_seq += cnt;  // every sequence number this launch may wait for is now spent
        sp.trace = r->d_trace, sp.trace_pass = r->trace_pass;
        auto t_sent = std::chrono::steady_clock::now();
        const bool plain = pl.generic && cnt == 1;
        const int iter_at_launch = loop.iter;
        if (plain) {
            pp.sol.tag = sp.tag0;
            if (int rc = launch_pass(r, pp, true)) return rc;
        } else if (int rc = launch_small(r, sp, pl)) {
            return rc;
        }
        for (uint32_t k = 0; k < cnt; ++k) {
            const RoctxScope pass_span("icp pass: rows -> solve -> command");
            long long words[kReduceWords];
            bool gave_up = false;
            int rc_rows;
            if (grouped) {
                rc_rows = wait_rows(r, plain ? groups_plain : groups_resident, sp.tag0 + k, words, plain ? 0 : (k % kPipeSlots) * groups_resident);
                // workgroups that left without a command (kGaveUpUnit each), or a row that never reached its group's reader
                // (kLostRowUnit): either way this pass is run again, in a fresh launch
                gave_up = (static_cast<unsigned long long>(words[kNumLimbs]) >> 8) != 0ull;
                words[kNumLimbs] &= 0xFFll;
            } else {
                rc_rows = wait_rows_small(r, grid, sp.tag0 + k, k % kPipeSlots, words, &gave_up);
            }
            const auto t_rows = std::chrono::steady_clock::now();
            if (r->d_trace) {
                const double us = std::chrono::duration<double, std::micro>(t_rows - t_sent).count();
                if (k == 0) r->trace_first_us += us, ++r->trace_first_n;
                else r->trace_dev_us += us, ++r->trace_n;
            }
            if (int rc = rc_rows) {
                if (k + 1 < cnt) send_command(r, sp.seq_base + k + 1, kCmdStop, loop.T);
                r->acc_dirty = true;
                return rc;
            }
            if (gave_up) {  // (part of) the kernel left while this thread was away: run this pass and the rest in a fresh launch
                ++r->small_relaunches, ++give_ups;
                if (k + 1 < cnt) send_command(r, sp.seq_base + k + 1, kCmdStop, loop.T);  // workgroups that did see the command
                break;
            }
            give_ups = 0;
            finished = loop.step(r, words, stats);
            if (k + 1 == cnt) break;
            if (r->debug_stall_us > 0.0) {  // tests: be late once
                const auto t0 = std::chrono::steady_clock::now();
                while (std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() < r->debug_stall_us) {
                }
                r->debug_stall_us = 0.0;
            }
            send_command(r, sp.seq_base + k + 1, finished ? kCmdStop : kCmdContinue, loop.T);
            if (r->d_trace) {
                t_sent = std::chrono::steady_clock::now();
                r->trace_host_us += std::chrono::duration<double, std::micro>(t_sent - t_rows).count();
            }
            if (finished) break;
        }
        if (pl.generic && !plain) r->last_resident_passes += loop.iter - iter_at_launch;
    }
    pose_to(loop.T, out_pose_qt);
    if (stats) stats->iterations = loop.iter, stats->converged = loop.converged, stats->beta = loop.beta;
    r->last_small = pl.generic ? 0 : (pl.wave ? 2 : 1);
    r->small_prev_iters = loop.iter;
    if (loop.nan_flag == 2) return fail(KICP_ERR_CAPACITY, "a per-point term exceeded the exact-accumulation range (|x| >= 2^43)");
    return loop.nan_flag ? KICP_WARN_NO_CORRESPONDENCES : KICP_OK;
}

int run_registration_impl(kicp_reg *r, kicp_map *map, const double *d_frame, size_t n, const double last_pose_qt[7],
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

    if (n > 0x7FFFFFF0ull / 3) return fail(KICP_ERR_CAPACITY, "frame too large");
    if (int rc = set_device(r->device)) return rc;
    const uint64_t epoch_before = map->mirror.synced_epoch;
    if (int rc = map_sync(map, r->device, r->stream)) return rc;
    if (map->mirror.synced_epoch != epoch_before) r->stream_dirty = true;  // the mirror was (re)uploaded through the HIP stream
    const bool shm = r->shm != nullptr;
    const bool multi = r->comm != nullptr || r->allreduce_fn != nullptr;
    const bool p2p = r->d_p2p_table != nullptr;
    // (the single-record hand-offs - the device collectives and the peer mailboxes - count iterations in 15 bits of their sequence
    // word; the default tagged-row hand-offs and the small-scan path have no such limit)
    if (max_it > 0x7FFF && (multi || p2p))
        return fail(KICP_ERR_ARG, "max_num_iterations > 32767 with a single-record hand-off (RCCL / callback / peer-mailbox exchange)");
    r->last_small = 0, r->last_resident_passes = 0;
    if (r->use_small && !shm && !multi && !p2p && r->timing == 0 && r->wait_mode == 0 && (r->dbg == 0 || r->dbg == 14)) {
        const SmallPlan pl = small_plan(r, n);
        if (pl.grid) return run_small(r, map, d_frame, n, pl, T0, tau, out_pose_qt, stats);
    }
    if (int rc = ensure_partials(r, pass_grid(r, n))) return rc;
    if (int rc = clear_stale_tickets(r)) return rc;
    if (p2p && (multi || shm)) return fail(KICP_ERR_ARG, "the peer-mailbox mode needs no other exchange attached");
    if (shm && multi) return fail(KICP_ERR_ARG, "the shared-segment mode needs no other exchange attached");
    if (p2p && r->p2p_poisoned)
        return fail(KICP_ERR_COMM, "the peer-mailbox exchange is out of step after an earlier failure: kicp_reg_p2p_destroy, _export and _connect again on every rank");
    const unsigned long long call_id = ++r->call_id;

    PassParams pp{};
    pp.src = d_frame, pp.n = static_cast<uint32_t>(n), pp.map = map->mirror.view, pp.tau = tau, pp.st = r->d_state;
    pp.search = search_params(tau, map->mirror.view.voxel_size);
    pp.partials = r->d_partials, pp.tickets = r->d_tickets, pp.group_acc = r->d_group_acc;
    pp.dbg = r->dbg;
    pp.corr_index = r->corr_index, pp.corr_d2 = r->corr_d2, pp.corr_nn = r->corr_nn;  // (kicp_pass_correspondences; nullptr otherwise)
    SolveParams &sp = pp.sol;
    set_pose(sp, T0), sp.max_iterations = max_it, sp.convergence_criterion = r->cfg.convergence_criterion;
    sp.adaptive = r->cfg.use_adaptive_odometry_regularization, sp.fixed_regularization = r->cfg.fixed_regularization;
    sp.mode = multi ? 1 : 0, sp.call_id = call_id, sp.rec = r->d_rec;

    if (r->timing) HIP_TRY(hipEventRecord(r->ev0, r->stream));
    const bool pass_events = r->timing == 2;
    if (pass_events && !r->evp[0])
        for (auto &e : r->evp) HIP_TRY(hipEventCreate(&e));
    unsigned long long seq = 0;
    {
        // ---- one launch per iteration, the pose travels as a kernel argument, the host solves (Registration.cpp:119-125,159-167,
        //      181-184).  (Round 6: the device-side solve - last workgroup of the launch, stepped or queued up front - is gone: it
        //      lost every A/B since round 2 and no exchange needs it.)
        HostRecord *rec = r->rec;
        HostLoop loop;
        loop.T = T0;
        int passes_run = 0;
        for (int it = 0; it < max_it; ++it) {
            const RoctxScope pass_span("icp pass: launch -> rows -> solve");
            ++passes_run;
            const bool rows_mode = !multi && !p2p;
            const size_t groups = (pass_grid(r, n) + kGroup - 1) / kGroup;
            // peer mailboxes: the groups' rows travel themselves when the launch has few enough of them (one reduction level less)
            // (every rank must use the same wire format - option "p2p_rows" - but may be on either side of the group limit)
            const bool p2p_rows = p2p && r->p2p_rows == 1 && groups <= static_cast<size_t>(kP2pMaxGroups);  // (2: always the single row - tests)
            set_pose(sp, loop.T);
            sp.pass = it, sp.mode = multi ? 3 : (p2p ? (p2p_rows ? 6 : 7) : 4);
            if (p2p) {  // every rank issues the same sequence of exchanges: the step number doubles as tag and buffer parity
                const unsigned long long step = r->p2p_step++;
                sp.p2p_peers = r->d_p2p_table, sp.p2p_nranks = r->nranks, sp.p2p_rank = r->rank;
                sp.p2p_tag = static_cast<uint32_t>(step % 65535ull) + 1u, sp.p2p_parity = static_cast<uint32_t>(step & 1ull);
                sp.p2p_timeout_ticks = static_cast<long long>(wait_timeout_s() * 0.8 * 1.0e8);  // the kernel gives up before the host does
            }
            long long words[kReduceWords];
            unsigned long long shm_value = 0;
            kicp_reg::ShmSlot *mine_host = nullptr;
            if (shm) {  // this rank's slot of the shared segment, double-buffered by hand-off parity
                const unsigned long long step = r->shm_step++;
                mine_host = r->shm + (step & 1) * r->nranks + r->rank;
                sp.pub_value = shm_value = step + 1;
            } else {
                sp.pub_words = r->d_rec->words, sp.pub_seq = &r->d_rec->seq;
                sp.pub_value = (call_id << 16) | static_cast<unsigned long long>(it + 1);
            }
            if (rows_mode) {
                if (int rc = ensure_rows(r, groups)) return rc;
                if (int rc = next_tag(r, &sp.tag)) return rc;
                sp.pub_rows = r->d_rows;
            } else if (p2p_rows) {
                if (int rc = next_tag(r, &sp.tag)) return rc;  // (the workgroups' rows inside a group are tagged like mode 4's)
            }
            const bool ev = pass_events && it < KICP_MAX_LOG_PASSES;
            if (ev) HIP_TRY(hipEventRecord(r->evp[2 * it], r->stream));
            // direct AQL dispatch when the host polls for the result and nothing follows the kernel on the HIP stream
            if (int rc = launch_pass(r, pp, !multi && r->timing == 0 && r->wait_mode == 0)) return rc;
            if (ev) HIP_TRY(hipEventRecord(r->evp[2 * it + 1], r->stream));
            if (multi) {
                if (int rc = enqueue_allreduce(r)) return rc;
                hipLaunchKernelGGL(k_publish_words, dim3(1), dim3(64), 0, r->stream, r->d_state, r->d_rec, call_id, it);
            }
            if (rows_mode) {
                if (int rc = wait_rows(r, groups, sp.tag, words)) {
                    r->acc_dirty = true;  // (the groups' counting accumulators may hold part of this pass: cleared before the next call's)
                    return rc;
                }
                if ((static_cast<unsigned long long>(words[kNumLimbs]) >> 8) != 0ull)
                    return fail(KICP_ERR_HIP, "a workgroup's row did not reach its group's reader in time (kRowWaitTicks)");
                words[kNumLimbs] &= 0xFFll;
                if (shm) {  // this rank's totals go into its slot from the host side; then every rank adds all slots
                    for (int i = 0; i < kReduceWords; ++i) mine_host->words[i] = words[i];
                    __atomic_store_n(&mine_host->seq, shm_value, __ATOMIC_RELEASE);
                    if (int rc = wait_shm(r, shm_value, words)) return rc;
                }
            } else {
                if (int rc = wait_record(r, call_id, static_cast<unsigned>(it + 1), false, &seq)) return rc;
                for (int i = 0; i < kReduceWords; ++i) words[i] = rec->words[i];
                if (p2p && words[kNumLimbs + 1] != 0) return fail(KICP_ERR_COMM, "a peer rank's totals did not arrive in this rank's mailbox in time");
            }
            if (loop.step(r, words, stats)) break;
        }
        HIP_TRY(hipGetLastError());
        if (r->timing) HIP_TRY(hipEventRecord(r->ev1, r->stream));
        pose_to(loop.T, out_pose_qt);
        if (stats) {
            stats->iterations = loop.iter, stats->converged = loop.converged, stats->beta = loop.beta;
            if (r->timing) {
                float ms = 0.f;
                HIP_TRY(hipEventSynchronize(r->ev1));
                HIP_TRY(hipEventElapsedTime(&ms, r->ev0, r->ev1));
                stats->gpu_ms = ms;
                for (int i = 0; pass_events && i < passes_run && i < KICP_MAX_LOG_PASSES; ++i) {
                    HIP_TRY(hipEventElapsedTime(&ms, r->evp[2 * i], r->evp[2 * i + 1]));
                    stats->pass_ms[i] = ms;
                }
            }
        }
        if (loop.nan_flag == 2) return fail(KICP_ERR_CAPACITY, "a per-point term exceeded the exact-accumulation range (|x| >= 2^43)");
        return loop.nan_flag ? KICP_WARN_NO_CORRESPONDENCES : KICP_OK;
    }
}

// Peer-mailbox mode: the ranks stay in step only while every exchange completes on every rank (tags and buffer parity are
// the step number).  A registration that fails after it has started an exchange - a peer's slot that did not arrive in time,
// a device fault - leaves this rank's later steps paired with other scans' steps on the peers, silently.  So the state is
// poisoned: every later call fails with KICP_ERR_COMM until the caller has torn the mailboxes down and connected them again
// on every rank (kicp_reg_p2p_destroy / _export / _connect), which resets the step counters.
int run_registration(kicp_reg *r, kicp_map *map, const double *d_frame, size_t n, const double last_pose_qt[7], const double rel_odom_qt[7],
                     double tau, double out_pose_qt[7], kicp_stats *stats) {
    const unsigned long long step_before = r ? r->p2p_step : 0ull;
    const int rc = run_registration_impl(r, map, d_frame, n, last_pose_qt, rel_odom_qt, tau, out_pose_qt, stats);
    if (rc < 0 && r && r->d_p2p_table && r->p2p_step != step_before) r->p2p_poisoned = true;
    return rc;
}

// kicp_register_device_batch with ONE pass kernel RESIDENT ACROSS THE SCANS of the batch (batches that run_batch_queues does not
// take: small scans only, "batch_queues" < 2).  What starts a pass is a command polled by a kernel that is already on the device
// (~1.5 us) instead of a dispatch of 2 048 waves (~4.5 us), for pass 0 of a scan as for its later passes (run_small).  The launch
// carries the table of the batch's scans (pointer, size); every command names the scan its pass belongs to, and "batch_depth"
// scans are in flight at a time (below).  With depth 1 scan k + 1's first pass is only started when scan k's last solve is done, as
// a loop of ComputeRobotMotion calls would.
// Returns 1 when the batch is not one for this path - the caller then runs the plain loop -, else a kicp status; *done = scans
// completed from the front.  On a give-up of the kernel (a workgroup that saw no command in time) the scans in hand and the rest
// are left to the plain loop, too.
constexpr uint32_t kBatchMaxPasses = 1024;  // passes (= tags) one launch may serve
// one hand-off on lane `lane` of o's segment: this rank's words go out, every rank's come back - summed into `sum`, and (per_rank != nullptr)
// one by one; blocking, bounded by KICP_WAIT_TIMEOUT_S
int shm_lane_exchange(kicp_reg *o, int lane, const long long *mine, long long *sum, long long *per_rank = nullptr) {  // per_rank: [nranks][kReduceWords]
    const unsigned long long step = o->shm_lane_step[lane]++;
    kicp_reg::ShmSlot *slots = o->shm + 2 * static_cast<size_t>(o->nranks) * (1 + lane) + (step & 1) * o->nranks;
    for (int i = 0; i < kReduceWords; ++i) slots[o->rank].words[i] = mine[i];
    __atomic_store_n(&slots[o->rank].seq, step + 1, __ATOMIC_RELEASE);
    const Deadline deadline;
    unsigned polls = 0;
    for (int k = 0; k < o->nranks; ++k)
        while (__atomic_load_n(&slots[k].seq, __ATOMIC_ACQUIRE) != step + 1)
            if (++polls % 4096u == 0u && deadline.passed()) return fail(KICP_ERR_COMM, "timed out waiting for a peer rank's hand-off (KICP_WAIT_TIMEOUT_S)");
    for (int i = 0; i < kReduceWords; ++i) sum[i] = 0;
    for (int k = 0; k < o->nranks; ++k)
        for (int i = 0; i < kReduceWords; ++i) {
            sum[i] += slots[k].words[i];  // (exact integers: the order does not matter)
            if (per_rank) per_rank[static_cast<size_t>(k) * kReduceWords + i] = slots[k].words[i];
        }
    return KICP_OK;
}
int depth_of(const kicp_reg *r) { return std::min<int>(std::max(r->batch_depth, 1), static_cast<int>(kPipeSlots)); }
int run_batch_resident(kicp_reg *r, kicp_map *map, size_t count, const double *const *d_frames, const size_t *n, const double *last_poses_qt,
                       const double *rel_odoms_qt, double tau, double *out_poses_qt, int *out_iterations, size_t *done, int *worst) {
    *done = 0;
    const int max_it = r->cfg.max_num_iterations;
    // (a batch call in this mode costs ~8 us of its own - the table, the kernel's leaving, the queue drained before the next call -
    //  against ~2.2 us saved per scan: from eight scans on it pays; measured in-process, cfg2 and cfg4, batches of 2 / 4 / 16 / 256)
    constexpr size_t kBatchResidentMinScans = 8;
    if (!r->batch_resident || !r->resident_generic || count < kBatchResidentMinScans || count > kCmdMaxScans || max_it <= 0 || kicp_map_empty(map)) return 1;
    if (!(r->use_small && !r->shm && !r->comm && !r->allreduce_fn && !r->d_p2p_table &&
          r->timing == 0 && r->wait_mode == 0 && (r->dbg == 0 || (r->dbg >= 2 && r->dbg <= 5) || r->dbg == 9 || r->dbg == 14) && r->small_resident != 0))
        return 1;
    // one kind of kernel serves the whole batch: the generic one (scans beyond the small-scan kernels, up to what the device holds at
    // once) or one wave per query (scans of up to kWaveMaxPoints points); anything else - or a mix - takes the plain loop
    size_t n_max = 0, n_min = ~size_t(0);
    for (size_t k = 0; k < count; ++k) n_max = std::max(n_max, n[k]), n_min = std::min(n_min, n[k]);
    if (n_min == 0) return 1;
    SmallPlan pl = small_plan(r, n_max);
    const SmallPlan pl_min = small_plan(r, n_min);
    if (!(pl.generic && pl_min.generic) && !(pl.wave && pl_min.wave && pl.grid)) return 1;
    if (int rc = set_device(r->device)) return rc;
    const uint64_t epoch_before = map->mirror.synced_epoch;
    if (int rc = map_sync(map, r->device, r->stream)) return rc;
    if (map->mirror.synced_epoch != epoch_before) r->stream_dirty = true;
    const bool wave = pl.wave;
    const uint32_t grid = wave ? pl.grid : static_cast<uint32_t>((n_max + 255) / 256);
    const size_t groups = (grid + kGroup - 1) / kGroup;
    const bool grouped = grouped_rows(r, pl, depth_of(r) > 1);  // the rows the host adds are group rows (the wave kernel's: "small_group_rows")
    if (!grouped) {
        if (int rc = ensure_rows(r, kPipeSlots * static_cast<size_t>(grid))) return rc;
    } else {
        if (int rc = ensure_partials(r, kPipeSlots * grid)) return rc;
        if (int rc = ensure_rows(r, kPipeSlots * groups)) return rc;
    }
    if (int rc = ensure_cmd(r)) return rc;
    if (grouped)
        if (int rc = clear_stale_tickets(r)) return rc;
    // The batch's scan table.  Where the CPU can write HBM through the PCIe BAR (the kernarg ring and the command copies live there
    // already) the table is written in place - a microsecond, no copy, no synchronisation; the launch's acquire makes it visible like
    // the kernel arguments.  Otherwise it is copied through the stream (and waited for: ~15 us per batch call).
    if (count > r->scans_cap) {
        if (int rc = aql_quiesce(r)) return rc;
        if (r->scans_bar) r->aql.free_bar(r->scans_bar);
        else if (r->d_scans) HIP_TRY(hipFree(r->d_scans));
        r->d_scans = nullptr, r->scans_bar = nullptr, r->scans_cap = 0;
        const size_t cap = count + count / 2 + 64;
        if (aql_up(r)) r->scans_bar = static_cast<ScanRef *>(r->aql.alloc_bar(cap * sizeof(ScanRef)));
        if (r->scans_bar) r->d_scans = r->scans_bar;
        else HIP_TRY(hipMalloc(reinterpret_cast<void **>(&r->d_scans), cap * sizeof(ScanRef)));
        r->scans_cap = cap;
    }
    // (the previous batch's kernel has left: the host had its last rows and sent STOP before it returned)
    if (int rc = aql_quiesce(r)) return rc;
    if (r->scans_bar) {
        for (size_t k = 0; k < count; ++k) r->scans_bar[k] = ScanRef{n[k] ? d_frames[k] : reinterpret_cast<const double *>(r->d_state), n[k]};  // (an idle lane still reads point 0)
        _mm_sfence();
    } else {
        std::vector<ScanRef> table(count);
        for (size_t k = 0; k < count; ++k) table[k] = ScanRef{n[k] ? d_frames[k] : reinterpret_cast<const double *>(r->d_state), n[k]};
        r->stream_dirty = true;
        HIP_TRY(hipMemcpyAsync(r->d_scans, table.data(), count * sizeof(ScanRef), hipMemcpyHostToDevice, r->stream));
        HIP_TRY(hipStreamSynchronize(r->stream));  // (`table` is pageable and about to go out of scope)
        r->stream_dirty = false;
    }
    if (!wave) pl.generic = true, pl.lat = true, pl.g = 1, pl.block = 256, pl.grid = grid;
    SmallParams sp{};
    PassParams &pp = sp.p;
    pp.src = n[0] ? d_frames[0] : reinterpret_cast<const double *>(r->d_state), pp.n = static_cast<uint32_t>(n[0]), pp.map = map->mirror.view, pp.tau = tau, pp.st = r->d_state;
    pp.search = search_params(tau, map->mirror.view.voxel_size);
    pp.dbg = r->dbg;
    pp.sol.max_iterations = max_it, pp.sol.convergence_criterion = r->cfg.convergence_criterion, pp.sol.mode = 4;
    pp.partials = r->d_partials, pp.tickets = r->d_tickets, pp.group_acc = r->d_group_acc, pp.sol.pub_rows = r->d_rows, pp.sol.call_id = ++r->call_id, pp.sol.rec = r->d_rec;
    sp.cmd = r->d_cmd, sp.rows = r->d_rows, sp.cmd_dev = r->d_cmd_copies, sp.relay = (r->small_cmd == 1 && r->cmd_bar) ? 0 : 1;
    sp.timeout_ticks = static_cast<long long>(std::max(50.0, r->small_timeout_us) * 100.0);
    sp.scans = r->d_scans;
    sp.group_rows = grouped ? 1 : 0;
    // (the workgroups' shares of a scan move on by about 0.38 of the grid per pass - far from where they were, and back only after many passes)
    if (!wave && r->batch_rotate && depth_of(r) > 1) sp.rotate = (static_cast<uint32_t>(grid * 0.381966) | 1u) % grid;
    // Several scans of the batch are in flight at a time (option "batch_depth", 1 .. kPipeSlots; 1: one).  The scans of a batch do not
    // depend on each other - every one starts from its own pose, the map does not change -, so while the host adds, solves and
    // answers the rows of pass k (a round trip of ~3 us over PCIe), the workgroups are already searching pass k + 1, which belongs
    // to ANOTHER scan; and a workgroup that is done with its part of a pass finds the next command waiting instead of waiting for
    // the slowest workgroup and the host.  Passes are numbered in the order their commands go out; with depth d the command of
    // pass k + d goes out when every row of pass k is in (kPipeSlots command lines, row buffers and sets of tickets, all taken in
    // turn: kicp_small.hpp, finish_pass).
    struct InFlight {
        HostLoop loop;
        kicp_stats st;
        size_t k = 0;          // the scan
        bool active = false;   // holds a scan that is not finished
        bool waiting = false;  // a pass of it is out
    };
    const int depth = depth_of(r);
    InFlight slots[kPipeSlots];
    int order[kPipeSlots] = {}, out = 0;   // slots whose passes are out, oldest first
    uint32_t order_pass[kPipeSlots] = {};
    uint32_t pass = 0, budget = 0;  // next pass index inside the current launch; passes that launch may serve (0: no kernel on the device)
    size_t next_scan = 0, front = 0;  // next scan to start; scans [0, front) are complete
    std::vector<unsigned char> complete(count, 0);
    bool stop_sent = false;
    auto stop_kernel = [&]() {
        const Pose ident{0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0};
        if (budget && pass < budget && !stop_sent) send_command(r, sp.seq_base + pass, kCmdStop, ident);
        budget = 0, stop_sent = true;
    };
    auto leave = [&](int rc) {  // hand back what is complete from the front; the caller's plain loop takes the rest
        if (out > 0 || rc != KICP_OK) r->acc_dirty = true;  // (passes still out will not be collected)
        stop_kernel();
        if (rc < 0) {  // nothing of this call may still be reading the caller's frames when it returns with an error (as run_batch_queues)
            (void)aql_quiesce(r);
            (void)hipStreamSynchronize(r->stream);
        }
        while (front < count && complete[front]) ++front;
        *done = front;
        return rc;
    };
    r->last_small = wave ? 2 : 0, r->last_resident_passes = 0;
    for (;;) {
        // ---- send out what can go out ------------------------------------------------------------------------------------
        while (out < depth) {
            int s = -1;
            bool starts = false;
            for (int j = 0; j < depth && s < 0; ++j)
                if (slots[j].active && !slots[j].waiting) s = j;
            if (s < 0 && next_scan < count) {
                for (int j = 0; j < depth && s < 0; ++j)
                    if (!slots[j].active) s = j;
                if (s >= 0) {
                    InFlight &f = slots[s];
                    f = InFlight{};
                    f.k = next_scan++, f.active = true, starts = true;
                    f.loop.T = pose_mul(pose_from(last_poses_qt + 7 * f.k), pose_from(rel_odoms_qt + 7 * f.k));  // Registration.cpp:156
                    std::memset(&f.st, 0, sizeof f.st);
                }
            }
            if (s < 0) break;  // nothing to send
            InFlight &f = slots[s];
            if (budget == 0 || pass >= budget) {  // no kernel on the device (any more: a launch that has served all its passes has left)
                if (out > 0) {  // (its last passes are still being collected)
                    if (starts) f.active = false, --next_scan;
                    break;
                }
                // tags are reserved per launch: no more than the rest of the batch can use
                unsigned long long rest = static_cast<unsigned long long>(count - next_scan) * static_cast<unsigned long long>(max_it);
                for (int j = 0; j < depth; ++j)
                    if (slots[j].active) rest += static_cast<unsigned long long>(std::max(1, max_it - slots[j].loop.iter));
                const uint32_t cnt = static_cast<uint32_t>(std::min<unsigned long long>(kBatchMaxPasses, rest));
                if (int rc = next_tag_range(r, cnt, &sp.tag0)) return leave(rc);
                set_pose(pp.sol, f.loop.T), pp.sol.pass = f.loop.iter;
                sp.max_passes = cnt, sp.seq_base = r->cmd_seq, sp.scan0 = static_cast<uint32_t>(f.k);
                r->cmd_seq += cnt;
                sp.trace = r->d_trace, sp.trace_pass = r->trace_pass;
                if (int rc = launch_small(r, sp, pl)) return leave(rc);
                pass = 0, budget = cnt, stop_sent = false;
            } else {
                if (r->debug_stall_us > 0.0) {  // tests: be late once (the kernel gives up, the plain loop takes over)
                    const auto t0 = std::chrono::steady_clock::now();
                    while (std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() < r->debug_stall_us) {
                    }
                    r->debug_stall_us = 0.0;
                }
                send_command(r, sp.seq_base + pass, starts ? kCmdNewScan : kCmdContinue, f.loop.T, static_cast<uint32_t>(f.k));
            }
            f.waiting = true;
            order[out] = s, order_pass[out] = pass, ++out, ++pass;
        }
        if (out == 0) break;  // every scan is complete
        // ---- the rows of the oldest pass that is out ---------------------------------------------------------------------
        InFlight &f = slots[order[0]];
        const uint32_t at = order_pass[0];
        for (int j = 1; j < out; ++j) order[j - 1] = order[j], order_pass[j - 1] = order_pass[j];
        --out;
        f.waiting = false;
        long long words[kReduceWords];
        bool gave_up = false;
        if (int rc = !grouped ? wait_rows_small(r, grid, sp.tag0 + at, at % kPipeSlots, words, &gave_up) : wait_rows(r, groups, sp.tag0 + at, words, (at % kPipeSlots) * groups))
            return leave(rc);
        if (grouped) gave_up = (static_cast<unsigned long long>(words[kNumLimbs]) >> 8) != 0ull, words[kNumLimbs] &= 0xFFll;
        if (gave_up) {  // (part of) the kernel has left: the scans in hand and the rest go through the plain loop
            ++r->small_relaunches;
            return leave(KICP_OK);
        }
        ++r->batch_resident_passes;
        if (!f.loop.step(r, words, &f.st)) continue;
        pose_to(f.loop.T, out_poses_qt + 7 * f.k);
        if (out_iterations) out_iterations[f.k] = f.loop.iter;
        r->small_prev_iters = f.loop.iter;
        complete[f.k] = 1, f.active = false;
        if (f.loop.nan_flag == 2) return leave(fail(KICP_ERR_CAPACITY, "a per-point term exceeded the exact-accumulation range (|x| >= 2^43)"));
        if (f.loop.nan_flag) *worst = std::max(*worst, static_cast<int>(KICP_WARN_NO_CORRESPONDENCES));
    }
    stop_kernel();
    *done = count;
    return KICP_OK;
}


// kicp_register_device_batch, large scans: SEVERAL SCANS IN FLIGHT ON SEVERAL QUEUES, one host thread.  The scans of a batch do
// not depend on each other - every one starts from its own pose, the map does not change - so the call keeps option
// "batch_queues" (default 4) of them going at a time, each on a handle of its own (clones of the caller's, made on first use and
// kept): own HSA queue, own reduction scratch and rows.  Every pass is an ordinary launch of the pass kernel in its
// four-waves-per-SIMD build (the latency-oriented build fills the register file with ONE scan's waves and leaves no room for a
// second scan's next to them); the device takes workgroups from all queues as wave slots fall free, so a pass's slow workgroups
// no longer hold anything up - the next scan's workgroups fill the slots the fast ones have left - and the host's answer to one
// scan's rows (add, solve, next launch: ~2 us) is hidden behind the other scans' searches.  This thread goes round the scans in
// flight: rows complete -> Registration.cpp:119-125, 159-167, 184 on the host -> next pass or next scan.
// What kicp_register_device_concurrent does with a host thread per lane, done by one thread that never sleeps on a lane.
// Returns 1 when the batch is not one for this path (the caller goes on to run_batch_resident / the plain loop), else a kicp
// status; *done = scans completed from the front.
constexpr int kMaxBatchQueues = 8;
static_assert(kMaxBatchQueues == kicp_reg::kShmLanes, "a lane of a sharded batch call owns one area of the shared segment");
struct BatchFlight {
    kicp_reg *h = nullptr;
    HostLoop loop;
    PassParams pp{};     // large scans: the pass kernel's arguments
    SmallParams sp{};    // small scans (kicp_small.hpp): a launch that serves ONE pass and leaves
    SmallPlan pl;
    bool small = false;    // a small-scan kernel's launch ...
    bool own_rows = false; // ... whose workgroups send rows of their own ("small_group_rows" 0)
    size_t k = 0, rows = 0, row_next = 0;  // rows of the pass in flight: the groups' (large) / the workgroups' (small); how many are in
    uint32_t tag = 0;
    bool active = false;
    unsigned polls = 0;
    long long words[kReduceWords] = {};            // sums of the rows that are in (large scans: the reduce payload's layout)
    __int128 total[kNumSums] = {};                 // (small scans: two 48-bit halves per sum and row)
    unsigned long long flags = 0;
    Deadline since;
    // sharded batches: this lane's next scan (static deal), and - once this rank's rows are in - the hand-off it waits for
    size_t next = 0;
    bool at_peers = false;
    unsigned long long shm_value = 0;
    // sharded over RCCL: the pass ends in the device-side tree + ncclAllReduce + k_publish_words on the lane's stream; the record's
    // sequence word the host polls for
    bool via_comm = false;
    unsigned long long comm_seq = 0;
};
// the rows of a flight's pass that have arrived since the last look: 1 all in (sums in out_words), 0 not yet, < 0 error
int flight_rows(BatchFlight &f, long long out_words[kReduceWords]) {
    kicp_reg *h = f.h;
    if (f.via_comm) {  // the all-reduced totals arrive as ONE record behind the collective (k_publish_words)
        if (__atomic_load_n(&h->rec->seq, __ATOMIC_ACQUIRE) != f.comm_seq) {
            if (++f.polls % 256u == 0u) {
                const hipError_t q = hipStreamQuery(h->stream);
                if (q != hipSuccess && q != hipErrorNotReady) return fail(KICP_ERR_HIP, std::string("stream fault: ") + hipGetErrorString(q));
                if (f.since.passed()) return fail(KICP_ERR_COMM, "timed out waiting for a lane's all-reduce (KICP_WAIT_TIMEOUT_S)");
            }
            return 0;
        }
        for (int i = 0; i < kReduceWords; ++i) out_words[i] = h->rec->words[i];
        out_words[kNumLimbs] = out_words[kNumLimbs] != 0 ? 1 : 0;
        return 1;
    }
    const uint32_t tag = f.tag;
    const int row_words = f.own_rows ? kSmallRowWords : kReduceWords;
    for (; f.row_next < f.rows; ++f.row_next) {
        const unsigned long long *row = h->rows + f.row_next * row_words;
        unsigned long long w[kReduceWords];
        bool ok = true;
        for (int i = 0; i < row_words; ++i) {
            w[i] = __atomic_load_n(row + i, __ATOMIC_RELAXED);
            ok = ok && (static_cast<uint32_t>(w[i]) & 0xFFFFu) == tag;
        }
        if (!ok) {
            if (++f.polls % 256u == 0u) {
                if (h->last_via_aql) {
                    if (h->aql.queue_error) return fail(KICP_ERR_HIP, "the AQL queue reported error " + std::to_string(h->aql.queue_error));
                } else {  // (the query makes the runtime flush commands it may still hold back, and reports device faults)
                    const hipError_t q = hipStreamQuery(h->stream);
                    if (q != hipSuccess && q != hipErrorNotReady) return fail(KICP_ERR_HIP, std::string("stream fault: ") + hipGetErrorString(q));
                }
                if (f.since.passed()) return fail(KICP_ERR_HIP, "timed out waiting for the pass kernel's rows (KICP_WAIT_TIMEOUT_S)");
            }
            return 0;
        }
        if (f.own_rows) {
            for (int i = 0; i < kNumSums; ++i)
                f.total[i] += static_cast<__int128>(w[2 * i] >> 16) + (static_cast<__int128>(static_cast<long long>(w[2 * i + 1]) >> 16) << 48);
            f.flags |= w[2 * kNumSums] >> 16;
        } else {
            for (int i = 0; i < kReduceWords; ++i)
                if (i != kNumLimbs) f.words[i] += static_cast<long long>(w[i]) >> 16;
            f.words[kNumLimbs] |= row_flags(static_cast<long long>(w[kNumLimbs]) >> 16);
        }
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    if (f.own_rows) {  // (the layout of the all-reduce payload: three 40-bit limbs per sum, then the range flag - wait_rows_small)
        for (int i = 0; i < kReduceWords; ++i) out_words[i] = 0;
        const unsigned __int128 m40 = (static_cast<unsigned __int128>(1) << 40) - 1;
        for (int i = 0; i < kNumSums; ++i) {
            const unsigned __int128 u = static_cast<unsigned __int128>(f.total[i]);
            out_words[3 * i] = static_cast<long long>(u & m40), out_words[3 * i + 1] = static_cast<long long>((u >> 40) & m40);
            out_words[3 * i + 2] = static_cast<long long>(f.total[i] >> 80);
        }
        out_words[kNumLimbs] = (f.flags & 1ull) ? 1 : 0;
    } else {
        for (int i = 0; i < kReduceWords; ++i) out_words[i] = f.words[i];
    }
    return 1;
}
int flight_launch(BatchFlight &f, const kicp_map *map, const double *d_frame, size_t n, double tau) {
    kicp_reg *h = f.h;
    f.pl = h->use_small ? small_plan(h, n) : SmallPlan();
    f.small = f.pl.grid != 0 && !f.pl.generic;
    f.row_next = 0, f.flags = 0, f.polls = 0;
    for (auto &w : f.words) w = 0;
    for (auto &t : f.total) t = 0;
    PassParams &pp = f.small ? f.sp.p : f.pp;
    if (n == 0) d_frame = reinterpret_cast<const double *>(h->d_state);  // (an empty shard: the one workgroup's lanes are all idle, but an idle lane still reads point 0)
    pp.src = d_frame, pp.n = static_cast<uint32_t>(n), pp.map = map->mirror.view, pp.tau = tau, pp.st = h->d_state;
    pp.search = search_params(tau, map->mirror.view.voxel_size);
    pp.dbg = h->dbg;
    SolveParams &sol = pp.sol;
    set_pose(sol, f.loop.T);
    sol.pass = f.loop.iter, sol.mode = 4, sol.max_iterations = h->cfg.max_num_iterations;
    sol.convergence_criterion = h->cfg.convergence_criterion;
    f.own_rows = f.small && !grouped_rows(h, f.pl, false);
    if (f.small) {  // one wave per query / sub-lanes per query; the launch serves this pass only
        if (f.own_rows) {  // every workgroup's row goes straight to the host
            f.rows = f.pl.grid;
            if (int rc = ensure_rows(h, (f.rows * kSmallRowWords + kReduceWords - 1) / kReduceWords)) return rc;
        } else {  // one row per group of 32 workgroups (counting accumulators)
            f.rows = (f.pl.grid + kGroup - 1) / kGroup;
            if (int rc = ensure_partials(h, kPipeSlots * f.pl.grid)) return rc;
            if (int rc = ensure_rows(h, kPipeSlots * f.rows)) return rc;
            if (int rc = clear_stale_tickets(h)) return rc;
            pp.group_acc = h->d_group_acc, sol.pub_rows = h->d_rows, sol.rec = h->d_rec;
        }
        if (int rc = ensure_cmd(h)) return rc;
        if (int rc = next_tag(h, &f.tag)) return rc;
        SmallParams &sp = f.sp;
        sp.group_rows = f.own_rows ? 0 : 1;
        sp.cmd = h->d_cmd, sp.rows = h->d_rows, sp.cmd_dev = h->d_cmd_copies, sp.relay = (h->small_cmd == 1 && h->cmd_bar) ? 0 : 1;
        sp.timeout_ticks = 5000, sp.trace = nullptr, sp.scans = nullptr, sp.rotate = 0;
        sp.tag0 = f.tag, sp.max_passes = 1, sp.seq_base = h->cmd_seq;
        h->cmd_seq += 1;
        f.since = Deadline();
        return launch_small(h, sp, f.pl);
    }
    const uint32_t grid = pass_grid(h, n);
    f.rows = (grid + kGroup - 1) / kGroup;
    if (int rc = ensure_partials(h, grid)) return rc;
    if (int rc = ensure_rows(h, f.rows)) return rc;
    if (int rc = clear_stale_tickets(h)) return rc;  // (nothing to do unless an earlier call left a pass uncollected)
    pp.partials = h->d_partials, pp.tickets = h->d_tickets, pp.group_acc = h->d_group_acc;
    sol.call_id = ++h->call_id, sol.rec = h->d_rec, sol.pub_rows = h->d_rows;
    if (int rc = next_tag(h, &sol.tag)) return rc;
    f.tag = sol.tag;
    f.since = Deadline();
    if (f.via_comm) {  // device-side tree -> all-reduce on the lane's communicator -> the totals to the host, all on the lane's stream
        sol.mode = 3;
        f.comm_seq = (sol.call_id << 16) | static_cast<unsigned long long>(f.loop.iter + 1);
        if (int rc = launch_pass(h, pp, false)) return rc;
        if (int rc = enqueue_allreduce(h)) return rc;
        hipLaunchKernelGGL(k_publish_words, dim3(1), dim3(64), 0, h->stream, h->d_state, h->d_rec, sol.call_id, f.loop.iter);
        HIP_TRY(hipGetLastError());
        return KICP_OK;
    }
    return launch_pass(h, pp, true);
}
int run_batch_queues(kicp_reg *r, kicp_map *map, size_t count, const double *const *d_frames, const size_t *n, const double *last_poses_qt,
                     const double *rel_odoms_qt, double tau, double *out_poses_qt, int *out_iterations, size_t *done, int *worst) {
    *done = 0;
    const int queues = std::min(r->batch_queues, kMaxBatchQueues);
    const int max_it = r->cfg.max_num_iterations;
    // SHARDED batches (the shared segment attached, kicp_reg_shm_init): every rank calls with ITS shard of every scan, the lanes'
    // exchanges go through the segment (below).  Every decision up to here and in the loop must then be the same on every rank - so
    // none of them looks at the shard sizes, which differ.
    // ... or through RCCL: lane j owns a sub-communicator of the handle's (ncclCommSplit on first use: every rank comes here with the
    // same arguments, so the splits line up), its collectives go out in the lane's own fixed order - scans j, j + lanes, ..., pass by
    // pass - on the lane's own stream, and the lanes' collectives interleave freely (round 6: until then an RCCL batch registered scan
    // after scan while the shared segment kept four in flight)
    const bool over_rccl = r->comm != nullptr && !r->shm && g_comm.CommSplit != nullptr && !r->lane_comms_failed;
    const bool sharded = r->shm != nullptr || over_rccl;
    if (queues < 2 || count < 2u * static_cast<size_t>(queues) || max_it <= 0 || kicp_map_empty(map)) return 1;
    if (!(r->use_aql && (!r->comm || over_rccl) && !r->allreduce_fn && !r->d_p2p_table && r->timing == 0 &&
          r->wait_mode == 0 && (r->dbg == 0 || r->dbg == 11 || r->dbg == 12 || r->dbg == 14)))
        return 1;
    if (r->shm && r->shm_poisoned)
        return fail(KICP_ERR_COMM, "the shared-segment exchange is out of step after a sharded batch that failed: kicp_reg_shm_destroy and _init again on every rank");
    // a batch of small scans only (kicp_small.hpp) is better off with ONE resident kernel and several scans in flight inside it
    // (run_batch_resident): a launch and a sweep over every workgroup's row per pass is more than one host thread can turn round in
    // the 4.5 us such a pass takes (measured, cfg4: 5.0 us per scan on four queues, 4.5 resident).  Mixed batches come here.
    bool any_large = sharded;  // (a shard goes through the generic kernel whatever its size: its group rows feed the exchange)
    for (size_t k = 0; k < count && !sharded; ++k) {
        if (n[k] == 0) return 1;
        if (!any_large) {
            const SmallPlan pl = r->use_small ? small_plan(r, n[k]) : SmallPlan();
            any_large = pl.grid == 0 || pl.generic;
        }
    }
    if (!any_large) return 1;
    if (int rc = set_device(r->device)) return rc;
    const uint64_t epoch_before = map->mirror.synced_epoch;
    if (int rc = map_sync(map, r->device, r->stream)) return rc;
    if (map->mirror.synced_epoch != epoch_before) HIP_TRY(hipStreamSynchronize(r->stream));  // (the lanes only read the copy)
    while (static_cast<int>(r->batch_lanes.size()) < queues) {
        kicp_reg *c = nullptr;
        if (int rc = kicp_reg_clone(r, &c)) return rc;
        r->batch_lanes.push_back(c);
    }
    BatchFlight flights[kMaxBatchQueues];
    for (int j = 0; j < queues; ++j) {
        kicp_reg *h = r->batch_lanes[j];
        h->cfg = r->cfg, h->lanes_per_query = r->lanes_per_query;
        h->query_every = r->query_every, h->dbg = r->dbg, h->latency_kernel = 0, h->small_resident = 0, h->batch_queues = 0;
        h->small_group_rows = r->small_group_rows;
        h->use_small = sharded ? 0 : r->use_small, h->small_block = r->small_block, h->small_wave = r->small_wave, h->wave_block = r->wave_block;
        flights[j].h = h;
        flights[j].next = static_cast<size_t>(j);  // (sharded: lane j's first scan)
        flights[j].via_comm = over_rccl;
        if (over_rccl) {
            if (!r->lane_comms[j]) {
                const ncclResult_t rc = g_comm.CommSplit(r->comm, 0, r->rank, &r->lane_comms[j], nullptr);
                if (rc != ncclSuccess || !r->lane_comms[j]) {  // (every rank fails alike: the same library, the same arguments)
                    r->lane_comms[j] = nullptr, r->lane_comms_failed = true;
                    for (int i = 0; i < j; ++i) flights[i].h->comm = nullptr;
                    return 1;  // this and later batches go scan after scan over the handle's own communicator
                }
            }
            h->comm = r->lane_comms[j], h->nranks = r->nranks, h->rank = r->rank;
        }
    }
    r->last_small = 0, r->last_resident_passes = 0;
    {
        const SmallPlan first = r->use_small ? small_plan(flights[0].h, n[0]) : SmallPlan();
        if (first.grid && !first.generic) r->last_small = first.wave ? 2 : 1;  // ("small_active": the path of the batch's first scan)
    }
    std::vector<unsigned char> complete(count, 0);
    size_t next_scan = 0, front = 0, finished_scans = 0;
    auto leave = [&](int rc) {  // nothing of this call may still be running when it returns: the caller owns the frames
        for (int j = 0; j < queues; ++j) {
            (void)aql_quiesce(flights[j].h);
            (void)hipStreamSynchronize(flights[j].h->stream);
            if (rc < 0 && flights[j].active) flights[j].h->acc_dirty = true;  // (a pass that was not collected: its accumulators may be part full)
        }
        if (rc < 0 && r->shm) r->shm_poisoned = true;  // (the ranks' lane counters can no longer be assumed equal)
        for (int j = 0; j < queues && over_rccl; ++j) flights[j].h->comm = nullptr;  // (the communicators stay this handle's)
        while (front < count && complete[front]) ++front;
        *done = front;
        return rc;
    };
    // the slots of lane j's hand-off `step` in the shared segment: [nranks], double-buffered by the hand-off's parity
    auto lane_slots = [&](int j, unsigned long long step) { return r->shm + 2 * static_cast<size_t>(r->nranks) * (1 + j) + (step & 1) * r->nranks; };
    while (finished_scans < count) {
        for (int j = 0; j < queues; ++j) {
            BatchFlight &f = flights[j];
            if (!f.active) {
                if (sharded) {  // the deal is static - lane j registers scans j, j + queues, ... - so that every rank's lane j issues the same exchanges
                    if (f.next >= count) continue;
                    f.k = f.next, f.next += static_cast<size_t>(queues), f.active = true;
                } else {
                    if (next_scan >= count) continue;
                    f.k = next_scan++, f.active = true;
                }
                f.at_peers = false;
                f.loop = HostLoop();
                f.loop.T = pose_mul(pose_from(last_poses_qt + 7 * f.k), pose_from(rel_odoms_qt + 7 * f.k));  // Registration.cpp:156
                if (int rc = flight_launch(f, map, d_frames[f.k], n[f.k], tau)) return leave(rc);
                continue;
            }
            long long words[kReduceWords];
            if (!f.at_peers) {
                const int ready = flight_rows(f, words);
                if (ready < 0) return leave(ready);
                if (ready == 0) continue;
                if ((static_cast<unsigned long long>(words[kNumLimbs]) >> 8) != 0ull)
                    return leave(fail(KICP_ERR_HIP, "a workgroup's row did not reach its group's reader in time (kRowWaitTicks)"));
                if (r->shm) {  // this rank's totals of the pass go into its slot of the lane's area; then the lane waits for every rank's
                    const unsigned long long step = r->shm_lane_step[j]++;
                    kicp_reg::ShmSlot *mine = lane_slots(j, step) + r->rank;
                    for (int i = 0; i < kReduceWords; ++i) mine->words[i] = words[i];
                    __atomic_store_n(&mine->seq, step + 1, __ATOMIC_RELEASE);
                    f.at_peers = true, f.shm_value = step + 1, f.since = Deadline(), f.polls = 0;
                }
            }
            if (r->shm) {  // (a non-blocking look: the other lanes' rows and hand-offs are served meanwhile)
                const kicp_reg::ShmSlot *slots = lane_slots(j, f.shm_value - 1);
                bool all_in = true;
                for (int k = 0; k < r->nranks && all_in; ++k) all_in = __atomic_load_n(&slots[k].seq, __ATOMIC_ACQUIRE) == f.shm_value;
                if (!all_in) {
                    if (++f.polls % 4096u == 0u && f.since.passed()) return leave(fail(KICP_ERR_COMM, "timed out waiting for a peer rank's hand-off (KICP_WAIT_TIMEOUT_S)"));
                    continue;
                }
                for (int i = 0; i < kReduceWords; ++i) words[i] = 0;
                for (int k = 0; k < r->nranks; ++k)
                    for (int i = 0; i < kReduceWords; ++i) words[i] += slots[k].words[i];  // (exact integers: the order does not matter)
                words[kNumLimbs] = words[kNumLimbs] != 0 ? 1 : 0;
                f.at_peers = false;
            }
            ++r->batch_queue_passes;
            if (!f.loop.step(f.h, words, nullptr)) {
                if (int rc = flight_launch(f, map, d_frames[f.k], n[f.k], tau)) return leave(rc);
                continue;
            }
            pose_to(f.loop.T, out_poses_qt + 7 * f.k);
            if (out_iterations) out_iterations[f.k] = f.loop.iter;
            complete[f.k] = 1, f.active = false, ++finished_scans;
            if (f.loop.nan_flag == 2) return leave(fail(KICP_ERR_CAPACITY, "a per-point term exceeded the exact-accumulation range (|x| >= 2^43)"));
            if (f.loop.nan_flag) *worst = std::max(*worst, static_cast<int>(KICP_WARN_NO_CORRESPONDENCES));
        }
    }
    for (int j = 0; j < queues && over_rccl; ++j) flights[j].h->comm = nullptr;
    *done = count;
    return KICP_OK;
}

}  // namespace

namespace {
// The host threads of kicp_register_device_concurrent's lanes: started on first use, kept for the life of the process (asleep on
// a condition variable between calls).  They must not inherit a caller's pinning - a caller bound to one core (OMP_PROC_BIND
// binds the initial thread of many a process) would have every lane spin on that core, and threads created for each call would
// spend most of a short call there before the scheduler spreads them (measured: 47k instead of 135k scans/s under `taskset -c 0`)
// - so each one asks for every CPU once, when it starts, and has long found a core of its own by the time work arrives.
struct LanePool {
    std::mutex mutex;
    std::condition_variable work, done;
    std::vector<std::thread> threads;
    const std::function<void(size_t)> *job = nullptr;
    size_t lanes = 0, finished = 0;
    unsigned long long epoch = 0;
    bool busy = false;
    void worker(size_t index) {
        cpu_set_t all;
        CPU_ZERO(&all);
        for (int c = 0; c < CPU_SETSIZE; ++c) CPU_SET(c, &all);
        (void)sched_setaffinity(0, sizeof all, &all);  // (what the cpuset allows is what remains)
        unsigned long long seen = 0;
        std::unique_lock<std::mutex> lock(mutex);
        for (;;) {
            work.wait(lock, [&] { return epoch != seen; });
            seen = epoch;
            if (index >= lanes) continue;
            const std::function<void(size_t)> *f = job;
            lock.unlock();
            (*f)(index);
            lock.lock();
            if (++finished == lanes) done.notify_all();
        }
    }
    void run(size_t n, const std::function<void(size_t)> &f) {
        std::unique_lock<std::mutex> lock(mutex);
        done.wait(lock, [&] { return !busy; });  // one call at a time drives the pool
        busy = true;
        while (threads.size() < n) {
            const size_t index = threads.size();
            threads.emplace_back([this, index] { worker(index); });
            threads.back().detach();
        }
        job = &f, lanes = n, finished = 0, ++epoch;
        work.notify_all();
        done.wait(lock, [&] { return finished == lanes; });
        busy = false, job = nullptr;
        done.notify_all();
    }
};
LanePool &lane_pool() {
    static LanePool *pool = new LanePool;  // (never destroyed: its threads outlive main's statics)
    return *pool;
}
// CPUs this process may keep busy with spinning lane threads: what a thread of the lane pool is ALLOWED to run on once it has asked for
// every CPU (sched_getaffinity after the reset: the cpuset of a container - docker --cpuset-cpus, Slurm, a k8s static CPU policy -
// is what remains; measured on a short-lived thread of its own, not on the caller: many a runtime pins the thread that initialised
// it - under torch the bench's main thread is allowed ONE CPU), cut down to the cgroup's CPU quota where there is one
// (v2: cpu.max "quota period"; v1: cpu.cfs_quota_us / cpu.cfs_period_us).  The batch paths that spend a host thread per resident
// kernel take min(option, budget - 1) of them and fall back to ONE kernel driven by the caller's thread below two.
int host_cpu_budget() {
    static const int budget = [] {
        long long cpus = sysconf(_SC_NPROCESSORS_ONLN);
        if (cpus < 1) cpus = 1;
        long long allowed = 0;
        std::thread([&allowed] {
            cpu_set_t all, got;
            CPU_ZERO(&all);
            for (int c = 0; c < CPU_SETSIZE; ++c) CPU_SET(c, &all);
            (void)sched_setaffinity(0, sizeof all, &all);
            CPU_ZERO(&got);
            if (sched_getaffinity(0, sizeof got, &got) == 0) allowed = CPU_COUNT(&got);
        }).join();
        if (allowed > 0) cpus = std::min(cpus, allowed);
        auto cap = [&cpus](long long quota, long long period) {
            if (quota > 0 && period > 0) cpus = std::min<long long>(cpus, std::max<long long>(1, (quota + period - 1) / period));
        };
        if (FILE *f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {  // cgroup v2 ("max 100000": no quota - fscanf stops at "max")
            long long quota = 0, period = 0;
            if (std::fscanf(f, "%lld %lld", &quota, &period) == 2) cap(quota, period);
            std::fclose(f);
        }
        long long q1 = 0, p1 = 0;  // cgroup v1
        for (const char *dir : {"/sys/fs/cgroup/cpu", "/sys/fs/cgroup/cpu,cpuacct"}) {
            if (FILE *f = std::fopen((std::string(dir) + "/cpu.cfs_quota_us").c_str(), "r")) {
                if (std::fscanf(f, "%lld", &q1) != 1) q1 = 0;
                std::fclose(f);
            }
            if (FILE *f = std::fopen((std::string(dir) + "/cpu.cfs_period_us").c_str(), "r")) {
                if (std::fscanf(f, "%lld", &p1) != 1) p1 = 0;
                std::fclose(f);
            }
            if (q1 > 0 && p1 > 0) break;
        }
        cap(q1, p1);
        if (const char *e = std::getenv("KICP_CPU_BUDGET"))  // (tests; a deployment that knows better)
            if (std::atoi(e) > 0) cpus = std::atoi(e);
        return static_cast<int>(std::max<long long>(1, std::min<long long>(cpus, 1 << 20)));
    }();
    return budget;
}
}  // namespace

namespace {
// Batches of scans that leave most of the device empty, round 5: small scans only (one wave per query: <= 4 096 points each), or
// scans of the generic kernel of <= kThreadsMaxGenericPoints points each.  One resident kernel with three scans in flight
// serves a batch of 1 080-point scans at ~4.3 us per scan however many scans there are: such a scan occupies 135 of the device's 256
// CUs, and a workgroup needs its ~4 us per pass (search, hand-over, next command); a 16 384-point scan's resident kernel (64
// workgroups, 32 CUs) takes 8.1 us per scan where four queues of ordinary launches take 4.6 - the command processor starts a kernel
// every ~4.5 us whatever the number of queues (6 and 8 queues measured SLOWER than 4), a resident kernel needs no dispatch at all.
// The rest of the device takes MORE resident kernels:
// the batch is cut into `batch_threads` contiguous parts, part t is served by run_batch_resident on handle t (the caller's, then clones
// of it: the lanes of run_batch_queues) from a host thread of the lane pool - the scans are independent, every pose stays bit-equal to
// registering that scan alone.  All kernels must be co-resident (a resident kernel waits for its host, which waits for the rows of ALL
// its workgroups): T x workgroups x waves per workgroup must fit the device at 16 waves per CU - the generic kernel's latency build:
// 8 waves = two workgroups per CU -, else fewer threads; generic scans come here only if three kernels fit (two would not beat the queues).
// Returns 1 when the batch is not one for this path.
constexpr size_t kThreadsMaxGenericPoints = 24576;  // (five and more such kernels fit the device)
int run_batch_resident_threads(kicp_reg *r, kicp_map *map, size_t count, const double *const *d_frames, const size_t *n, const double *last_poses_qt,
                               const double *rel_odoms_qt, double tau, double *out_poses_qt, int *out_iterations, int *worst) {
    constexpr size_t kMinScansPerThread = 16;
    // (sharded batches - the shared segment attached - go through the queues: resident kernels that wait for their PEERS' rows as well as
    //  for their host stalled intermittently with several ranks on one device; the option that enabled them is gone, round 6)
    if (r->shm) return 1;
    // every part's host thread spins on its kernel's rows: no more parts than CPUs this process may keep busy, one left for the rest
    const int threads_cap = std::max(1, host_cpu_budget() - 1);
    int threads = std::min({r->batch_threads, kMaxBatchQueues + 1, threads_cap});
    if (threads < 2 || count < 2 * kMinScansPerThread || r->cfg.max_num_iterations <= 0 || kicp_map_empty(map)) return 1;
    if (!(r->batch_resident && r->resident_generic && r->use_small && r->small_wave && r->use_aql &&
          !r->comm && !r->allreduce_fn && !r->d_p2p_table && r->timing == 0 && r->wait_mode == 0 && r->dbg == 0 && r->small_resident != 0 && r->debug_stall_us == 0.0))
        return 1;
    size_t n_max = 0, n_min = ~size_t(0);
    for (size_t k = 0; k < count; ++k) n_max = std::max(n_max, n[k]), n_min = std::min(n_min, n[k]);
    if (n_min == 0) return 1;
    const SmallPlan pl = small_plan(r, n_max), pl_min = small_plan(r, n_min);
    const bool wave = pl.wave && pl_min.wave && pl.grid, generic = pl.generic && pl_min.generic;
    if (!wave && !generic) return 1;
    // kernels that fit the device side by side; generic scans: the latency build (two workgroups per CU) where three and more of its
    // kernels fit, else the four-waves build (four per CU) where two and more do - 131 072-point scans: two kernels of 512 workgroups
    const size_t grid_g = std::max<size_t>(1, (n_max + 255) / 256);
    const size_t fit_lat = n_max > kThreadsMaxGenericPoints ? 0 : static_cast<size_t>(r->num_cus) * 2 / grid_g;
    if (generic && fit_lat < 3) return 1;  // (two kernels of the latency build would not beat the queues)
    const size_t fit = wave ? static_cast<size_t>(r->num_cus) * 16 / std::max<size_t>(1, static_cast<size_t>(pl.grid) * static_cast<size_t>(pl.block / 64)) : fit_lat;
    threads = static_cast<int>(std::min<size_t>({static_cast<size_t>(threads), count / kMinScansPerThread, fit}));
    if (threads < (wave ? 2 : 3)) return 1;
    if (int rc = set_device(r->device)) return rc;
    if (int rc = map_sync(map, r->device, r->stream)) return rc;  // (once, here: the lanes then only read the copy)
    HIP_TRY(hipStreamSynchronize(r->stream));
    while (static_cast<int>(r->batch_lanes.size()) < threads - 1) {
        kicp_reg *c = nullptr;
        if (int rc = kicp_reg_clone(r, &c)) return rc;
        r->batch_lanes.push_back(c);
    }
    std::vector<kicp_reg *> handles{r};
    std::vector<unsigned long long> passes_before, relaunches_before;
    for (int t = 1; t < threads; ++t) {
        kicp_reg *h = r->batch_lanes[t - 1];
        h->cfg = r->cfg, h->query_every = r->query_every, h->dbg = 0, h->latency_kernel = r->latency_kernel, h->batch_queues = 0, h->batch_threads = 0;
        h->use_small = r->use_small, h->small_wave = r->small_wave, h->wave_block = r->wave_block, h->small_block = r->small_block, h->small_resident = r->small_resident;
        h->small_group_rows = r->small_group_rows, h->small_timeout_us = r->small_timeout_us, h->batch_depth = r->batch_depth, h->batch_rotate = r->batch_rotate;
        h->batch_resident = 1, h->resident_generic = 1, h->small_cmd = r->cmd_bar ? 1 : r->small_cmd;
        handles.push_back(h);
    }
    for (kicp_reg *h : handles) passes_before.push_back(h->batch_resident_passes), relaunches_before.push_back(h->small_relaunches);
    std::vector<int> rcs(static_cast<size_t>(threads), KICP_OK), worsts(static_cast<size_t>(threads), KICP_OK);
    std::vector<std::string> messages(static_cast<size_t>(threads));
    const std::function<void(size_t)> lane = [&](size_t t) {
        kicp_reg *h = handles[t];
        const size_t lo = count * t / static_cast<size_t>(threads), hi = count * (t + 1) / static_cast<size_t>(threads);
        size_t done = 0;
        int rc = run_batch_resident(h, map, hi - lo, d_frames + lo, n + lo, last_poses_qt + 7 * lo, rel_odoms_qt + 7 * lo, tau, out_poses_qt + 7 * lo,
                                    out_iterations ? out_iterations + lo : nullptr, &done, &worsts[t]);
        kicp_stats st;
        for (size_t k = lo + done; rc >= 0 && k < hi; ++k) {  // (not a batch for the resident kernel after all, or its kernel gave up: one call per scan)
            rc = run_registration(h, map, d_frames[k], n[k], last_poses_qt + 7 * k, rel_odoms_qt + 7 * k, tau, out_poses_qt + 7 * k, out_iterations ? &st : nullptr);
            if (rc >= 0) worsts[t] = std::max(worsts[t], rc);
            if (rc >= 0 && out_iterations) out_iterations[k] = st.iterations;
        }
        rcs[t] = rc < 0 ? rc : KICP_OK;
        if (rc < 0) messages[t] = kicp_last_error();  // (the message is per thread: carry it over)
    };
    lane_pool().run(static_cast<size_t>(threads), lane);
    r->last_batch_threads = threads;
    for (int t = 1; t < threads; ++t) {  // (the caller reads the counters on its own handle)
        r->batch_resident_passes += handles[t]->batch_resident_passes - passes_before[t];
        r->small_relaunches += handles[t]->small_relaunches - relaunches_before[t];
    }
    for (int t = 0; t < threads; ++t) {
        if (rcs[t] < 0) {
            return fail(rcs[t], messages[t]);
        }
        *worst = std::max(*worst, worsts[t]);
    }
    return KICP_OK;
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
static void destroy_lane_comms(kicp_reg *reg) {  // the lanes' sub-communicators go before the communicator they were split off
    for (kicp_reg *lane : reg->batch_lanes)
        if (lane->stream) hipStreamSynchronize(lane->stream), lane->comm = nullptr;
    for (ncclComm_t &c : reg->lane_comms) {
        if (c) g_comm.CommDestroy(c);
        c = nullptr;
    }
    reg->lane_comms_failed = false;
}
int kicp_reg_comm_init(kicp_reg *reg, int nranks, int rank, const char id[KICP_COMM_ID_BYTES]) {
    if (!reg || !id || nranks < 1 || rank < 0 || rank >= nranks) return fail(KICP_ERR_ARG, "bad communicator arguments");
    std::string err;
    if (!g_comm.load(err)) return fail(KICP_ERR_COMM, err);
    if (int rc = set_device(reg->device)) return rc;
    if (reg->comm) destroy_lane_comms(reg), g_comm.CommDestroy(reg->comm), reg->comm = nullptr;
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
        destroy_lane_comms(reg);
        g_comm.CommDestroy(reg->comm);
        reg->comm = nullptr;
    }
    reg->nranks = 1, reg->rank = 0;
    return KICP_OK;
}
// Shared segment layout: one header slot (magic word written LAST by rank 0, then the rank count) followed by the
// [2 buffers][nranks] hand-off slots of single calls and, behind them, one such area per lane of a sharded batch call.
constexpr unsigned long long kShmMagic = 0x4B49435053484D31ull;  // "KICPSHM1"
int kicp_reg_shm_destroy(kicp_reg *reg) {
    if (!reg) return fail(KICP_ERR_ARG, "null argument");
    if (reg->shm) {
        hipSetDevice(reg->device);
        hipStreamSynchronize(reg->stream);
        if (reg->d_shm) (void)hipHostUnregister(reg->shm_base);
        munmap(reg->shm_base, reg->shm_bytes);
        if (reg->rank == 0) shm_unlink(reg->shm_name.c_str());
        reg->shm = nullptr, reg->d_shm = nullptr, reg->shm_base = nullptr, reg->shm_bytes = 0;
    }
    reg->nranks = 1, reg->rank = 0;
    return KICP_OK;
}
int kicp_reg_shm_init(kicp_reg *reg, int nranks, int rank, const char *name) {
    if (!reg || !name || nranks < 1 || rank < 0 || rank >= nranks) return fail(KICP_ERR_ARG, "bad shared-segment arguments");
    if (reg->comm) return fail(KICP_ERR_ARG, "an RCCL communicator is already attached");
    kicp_reg_shm_destroy(reg);
    if (int rc = set_device(reg->device)) return rc;
    const size_t bytes = (1 + 2 * static_cast<size_t>(nranks) * (1 + kicp_reg::kShmLanes)) * sizeof(kicp_reg::ShmSlot);  // header, single-call area, the lanes' areas
    const std::string nm = std::string(name[0] == '/' ? "" : "/") + name;
    void *ptr = MAP_FAILED;
    if (rank == 0) {
        // a segment of this name left behind by a crashed run must not be adopted: remove it, then create exclusively
        shm_unlink(nm.c_str());
        const int fd = shm_open(nm.c_str(), O_CREAT | O_EXCL | O_RDWR, 0600);
        if (fd < 0) return fail(KICP_ERR_COMM, "shm_open(" + nm + ", O_CREAT | O_EXCL) failed");
        if (ftruncate(fd, static_cast<off_t>(bytes)) != 0) {
            close(fd);
            shm_unlink(nm.c_str());
            return fail(KICP_ERR_COMM, "ftruncate on the shared segment failed");
        }
        ptr = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        close(fd);
        if (ptr == MAP_FAILED) return fail(KICP_ERR_COMM, "mmap of the shared segment failed");
        std::memset(ptr, 0, bytes);
        auto *hdr = static_cast<kicp_reg::ShmSlot *>(ptr);
        hdr->words[0] = nranks;
        __atomic_store_n(&hdr->seq, kShmMagic, __ATOMIC_RELEASE);  // published last: the other ranks wait for it
    } else {
        // wait (bounded) until rank 0 has created, sized, zeroed and published the segment
        const Deadline deadline;
        for (;;) {
            const int fd = shm_open(nm.c_str(), O_RDWR, 0600);
            if (fd >= 0) {
                struct stat st {};
                if (fstat(fd, &st) == 0 && static_cast<size_t>(st.st_size) == bytes) {
                    ptr = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
                    close(fd);
                    if (ptr == MAP_FAILED) return fail(KICP_ERR_COMM, "mmap of the shared segment failed");
                    auto *hdr = static_cast<kicp_reg::ShmSlot *>(ptr);
                    while (__atomic_load_n(&hdr->seq, __ATOMIC_ACQUIRE) != kShmMagic) {
                        if (deadline.passed()) {
                            munmap(ptr, bytes);
                            return fail(KICP_ERR_COMM, "timed out waiting for rank 0 to publish the shared segment");
                        }
                        usleep(50);
                    }
                    if (hdr->words[0] != nranks) {
                        munmap(ptr, bytes);
                        return fail(KICP_ERR_COMM, "the shared segment was created for a different number of ranks");
                    }
                    break;
                }
                close(fd);
            }
            if (deadline.passed()) return fail(KICP_ERR_COMM, "timed out waiting for rank 0 to create shared segment " + nm);
            usleep(200);
        }
    }
    // The device view is only needed when the GPU itself writes the slot ("group_rows" = 0); by default the rank's host adds
    // its GPU's tagged rows and stores the totals, so a failed registration is not fatal.
    hipError_t e = hipHostRegister(ptr, bytes, hipHostRegisterMapped | hipHostRegisterPortable);
    void *dptr = nullptr;
    if (e == hipSuccess) e = hipHostGetDevicePointer(&dptr, ptr, 0);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        dptr = nullptr;
    }
    reg->shm_base = ptr;
    reg->shm = static_cast<kicp_reg::ShmSlot *>(ptr) + 1;
    reg->d_shm = dptr ? static_cast<kicp_reg::ShmSlot *>(dptr) + 1 : nullptr;
    reg->shm_bytes = bytes, reg->shm_step = 0, reg->shm_name = nm, reg->nranks = nranks, reg->rank = rank;
    for (auto &st : reg->shm_lane_step) st = 0;
    reg->shm_poisoned = false;
    return KICP_OK;
}
// ---- one-shot exchange over peer mappings -----------------------------------------------------
// Each rank owns a mailbox in its own HBM: [2 parities][nranks][kP2pWords] tagged words, fine-grained so that a peer's stores
// (over xGMI) become visible to a kernel that is polling it.  Export -> the caller gathers every rank's handle (any
// transport: torch.distributed, MPI, a file) -> connect opens the peers' mailboxes -> every pass kernel's last workgroup
// writes this rank's totals into all mailboxes and collects its own (mode 5, kicp_kernels.hpp::p2p_exchange).
int kicp_reg_p2p_destroy(kicp_reg *reg) {
    if (!reg) return fail(KICP_ERR_ARG, "null argument");
    if (reg->p2p_box || reg->d_p2p_table) {
        hipSetDevice(reg->device);
        hipStreamSynchronize(reg->stream);
        for (void *&m : reg->p2p_mapped)
            if (m) (void)hipIpcCloseMemHandle(m), m = nullptr;
        if (reg->d_p2p_table) hipFree(reg->d_p2p_table);
        if (reg->p2p_box) hipFree(reg->p2p_box);
        reg->d_p2p_table = nullptr, reg->p2p_box = nullptr;
        (void)hipGetLastError();
    }
    reg->nranks = 1, reg->rank = 0, reg->p2p_step = 0, reg->p2p_poisoned = false;
    return KICP_OK;
}
int kicp_reg_p2p_export(kicp_reg *reg, int nranks, int rank, char handle[KICP_P2P_HANDLE_BYTES]) {
    static_assert(KICP_P2P_HANDLE_BYTES == sizeof(hipIpcMemHandle_t), "handle size");
    static_assert(KICP_P2P_MAX_RANKS == kP2pMaxRanks, "rank limit");
    if (!reg || !handle || nranks < 1 || nranks > kP2pMaxRanks || rank < 0 || rank >= nranks) return fail(KICP_ERR_ARG, "bad peer-mailbox arguments");
    if (reg->comm || reg->shm || reg->allreduce_fn) return fail(KICP_ERR_ARG, "another exchange is already attached");
    kicp_reg_p2p_destroy(reg);
    if (int rc = set_device(reg->device)) return rc;
    const size_t bytes = p2p_box_words(nranks) * sizeof(unsigned long long);  // totals area (mode 5) + group-row area (mode 6)
    // fine-grained: stores arriving from a peer GPU must be visible to a wave that is polling (no stale L2 line)
    hipError_t e = hipExtMallocWithFlags(reinterpret_cast<void **>(&reg->p2p_box), bytes, hipDeviceMallocFinegrained);
    if (e != hipSuccess) {
        reg->p2p_box = nullptr;
        return fail(KICP_ERR_HIP, std::string("hipExtMallocWithFlags(fine-grained mailbox): ") + hipGetErrorString(e));
    }
    HIP_TRY(hipMemset(reg->p2p_box, 0, bytes));  // tag 0 never matches
    hipIpcMemHandle_t h;
    e = hipIpcGetMemHandle(&h, reg->p2p_box);
    if (e != hipSuccess) {
        hipFree(reg->p2p_box), reg->p2p_box = nullptr;
        return fail(KICP_ERR_COMM, std::string("hipIpcGetMemHandle: ") + hipGetErrorString(e));
    }
    std::memcpy(handle, &h, sizeof h);
    reg->nranks = nranks, reg->rank = rank;
    return KICP_OK;
}
int kicp_reg_p2p_connect(kicp_reg *reg, const char *handles) {
    if (!reg || !handles) return fail(KICP_ERR_ARG, "null argument");
    if (!reg->p2p_box) return fail(KICP_ERR_ARG, "kicp_reg_p2p_export first");
    if (reg->d_p2p_table) return fail(KICP_ERR_ARG, "already connected: kicp_reg_p2p_destroy / _export first");
    if (int rc = set_device(reg->device)) return rc;
    unsigned long long *table[kP2pMaxRanks] = {};
    for (int k = 0; k < reg->nranks; ++k) {
        if (k == reg->rank) {
            table[k] = reg->p2p_box;
            continue;
        }
        hipIpcMemHandle_t h;
        std::memcpy(&h, handles + static_cast<size_t>(k) * KICP_P2P_HANDLE_BYTES, sizeof h);
        void *ptr = nullptr;
        const hipError_t e = hipIpcOpenMemHandle(&ptr, h, hipIpcMemLazyEnablePeerAccess);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            return fail(KICP_ERR_COMM, "hipIpcOpenMemHandle(rank " + std::to_string(k) + "): " + hipGetErrorString(e));
        }
        reg->p2p_mapped[k] = ptr, table[k] = static_cast<unsigned long long *>(ptr);
    }
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&reg->d_p2p_table), sizeof table));
    HIP_TRY(hipMemcpy(reg->d_p2p_table, table, sizeof table, hipMemcpyHostToDevice));
    reg->p2p_step = 0;
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
int kicp_reg_set_allreduce(kicp_reg *reg, kicp_allreduce_fn fn, void *user) {
    if (!reg) return fail(KICP_ERR_ARG, "null argument");
    reg->allreduce_fn = fn, reg->allreduce_user = user;
    return KICP_OK;
}

}  // extern "C"
