// kicp_reg_queues.hip -- batches with several scans in flight on queues of their own, sharded or not (see kicp_reg_internal.hpp)
#include "kicp_reg_internal.hpp"

using namespace kicp;
using namespace kicp::host;

namespace kicp {
namespace host {
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
}  // namespace host
}  // namespace kicp
