// kicp_reg_internal.hpp -- what the translation units of the registration (kicp_reg_*.hip) share: the handle behind kicp_reg, the
// run-time binding of RCCL, and the functions that cross file boundaries.  Round 6 cut the former kicp_reg.hip (166 KB, one
// translation unit) into
//   kicp_reg_launch.hip   the pass kernels' launches (HIP stream / hand-written AQL packets), buffers, tags, the waits for rows and records
//                         - the only file that instantiates the pass kernels, and the source of the code object embedded for AQL dispatch
//   kicp_reg_run.hip      one registration: the host-side solve loop (Registration.cpp:151-190), the small-scan path with its resident kernel
//   kicp_reg_batch.hip    batches of independent scans on resident kernels (one kernel across the scans; several side by side, a thread each)
//   kicp_reg_queues.hip   batches with several scans in flight on queues of their own (also sharded: shared segment / RCCL lanes)
//   kicp_reg_comm.hip     the multi-GPU exchanges: RCCL communicator, host shared segment, peer mailboxes, caller-supplied all-reduce
//   kicp_reg_api.hip      the C-ABI entry points of include/kicp.h: create / destroy / options / kicp_register* / kicp_pass_*
#pragma once
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

// (an internal header of six translation units that all begin this way)
using namespace kicp;
using namespace kicp::host;

namespace kicp {
namespace host {
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
extern CommApi g_comm;
}  // namespace host
}  // namespace kicp

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

namespace kicp {
namespace host {
constexpr int kPassBlock = 256;  // workgroup size of every build of the generic pass kernel
constexpr size_t kLatencyMaxPoints = 131072;  // two waves per SIMD on 256 CUs
constexpr uint32_t kBatchMaxPasses = 1024;  // passes (= tags) one launch may serve
constexpr int kMaxBatchQueues = 8;
constexpr int kMaxGiveUps = 16;  // launches in a row that may end without a completed pass before the call fails
constexpr size_t kThreadsMaxGenericPoints = 24576;  // (five and more such kernels fit the device)
double wait_timeout_s();
double host_limbs_to_double(const long long l[3]);
struct Deadline {
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    bool passed() const { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > wait_timeout_s(); }
};
struct SmallPlan {
    uint32_t grid = 0;
    int block = 256, g = 1;
    bool wave = false;
    bool generic = false, lat = false;  // generic: the generic pass kernel, resident (k_pass_resident); rows = group rows
};
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
double wait_timeout_s();
int lanes_for(const kicp_reg *r, size_t n);
uint32_t pass_grid(const kicp_reg *r, size_t n);
bool aql_up(kicp_reg *r);
const AqlKernel *aql_lookup(kicp_reg *r, int key, const char *demangled_prefix);
const AqlKernel *aql_kernel_for(kicp_reg *r, int b, int g, int occ, bool split, bool lat);
const AqlKernel *aql_resident_kernel_for(kicp_reg *r, bool lat);
const AqlKernel *aql_small_kernel_for(kicp_reg *r, int block, int g, bool wave);
int aql_quiesce(kicp_reg *r);
int launch_pass(kicp_reg *r, const PassParams &p, bool allow_aql = false);
int ensure_partials(kicp_reg *r, size_t blocks);
int ensure_rows(kicp_reg *r, size_t groups);
int next_tag(kicp_reg *r, uint32_t *tag);
int enqueue_allreduce(kicp_reg *r);
int wait_record(kicp_reg *r, unsigned long long call_id, unsigned min_iter, bool need_done, unsigned long long *seq_out);
long long row_flags(long long w);
int wait_rows(kicp_reg *r, size_t groups, uint32_t tag, long long out_words[kReduceWords], size_t first_row = 0);
int wait_shm(kicp_reg *r, unsigned long long value, long long out_words[kReduceWords]);
bool grouped_rows(const kicp_reg *r, const SmallPlan &pl, bool pipelined);
SmallPlan small_plan(const kicp_reg *r, size_t n);
int ensure_cmd(kicp_reg *r);
int next_tag_range(kicp_reg *r, uint32_t count, uint32_t *first);
void send_command(kicp_reg *r, unsigned long long seq, uint32_t op, const Pose &T, uint32_t scan = 0u);
int launch_small(kicp_reg *r, const SmallParams &sp, const SmallPlan &pl);
int wait_rows_small(kicp_reg *r, uint32_t grid, uint32_t tag, uint32_t parity, long long out_words[kReduceWords], bool *gave_up);
int clear_stale_tickets(kicp_reg *r);
int run_small(kicp_reg *r, kicp_map *map, const double *d_frame, size_t n, const SmallPlan &pl, const Pose &T0, double tau, double out_pose_qt[7],
              kicp_stats *stats);
int run_registration_impl(kicp_reg *r, kicp_map *map, const double *d_frame, size_t n, const double last_pose_qt[7],
                          const double rel_odom_qt[7], double tau, double out_pose_qt[7], kicp_stats *stats);
int run_registration(kicp_reg *r, kicp_map *map, const double *d_frame, size_t n, const double last_pose_qt[7], const double rel_odom_qt[7],
                     double tau, double out_pose_qt[7], kicp_stats *stats);
int depth_of(const kicp_reg *r);
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
LanePool &lane_pool();
int run_batch_resident(kicp_reg *r, kicp_map *map, size_t count, const double *const *d_frames, const size_t *n, const double *last_poses_qt,
                       const double *rel_odoms_qt, double tau, double *out_poses_qt, int *out_iterations, size_t *done, int *worst);
int run_batch_queues(kicp_reg *r, kicp_map *map, size_t count, const double *const *d_frames, const size_t *n, const double *last_poses_qt,
                     const double *rel_odoms_qt, double tau, double *out_poses_qt, int *out_iterations, size_t *done, int *worst);
int host_cpu_budget();
int run_batch_resident_threads(kicp_reg *r, kicp_map *map, size_t count, const double *const *d_frames, const size_t *n, const double *last_poses_qt,
                               const double *rel_odoms_qt, double tau, double *out_poses_qt, int *out_iterations, int *worst);
}  // namespace host
}  // namespace kicp
