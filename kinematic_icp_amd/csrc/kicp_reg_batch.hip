// kicp_reg_batch.hip -- batches of independent scans on resident kernels (see kicp_reg_internal.hpp)
#include "kicp_reg_internal.hpp"

using namespace kicp;
using namespace kicp::host;

namespace kicp {
namespace host {
// kicp_register_device_batch with ONE pass kernel RESIDENT ACROSS THE SCANS of the batch (batches that run_batch_queues does not
// take: small scans only, "batch_queues" < 2).  What starts a pass is a command polled by a kernel that is already on the device
// (~1.5 us) instead of a dispatch of 2 048 waves (~4.5 us), for pass 0 of a scan as for its later passes (run_small).  The launch
// carries the table of the batch's scans (pointer, size); every command names the scan its pass belongs to, and "batch_depth"
// scans are in flight at a time (below).  With depth 1 scan k + 1's first pass is only started when scan k's last solve is done, as
// a loop of ComputeRobotMotion calls would.
// Returns 1 when the batch is not one for this path - the caller then runs the plain loop -, else a kicp status; *done = scans
// completed from the front.  On a give-up of the kernel (a workgroup that saw no command in time) the scans in hand and the rest
// are left to the plain loop, too.
// one hand-off on lane `lane` of o's segment: this rank's words go out, every rank's come back - summed into `sum`, and (per_rank != nullptr)
// one by one; blocking, bounded by KICP_WAIT_TIMEOUT_S
int shm_lane_exchange(kicp_reg *o, int lane, const long long *mine, long long *sum, long long *per_rank) {  // per_rank: [nranks][kReduceWords]
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
}  // namespace host
}  // namespace kicp
