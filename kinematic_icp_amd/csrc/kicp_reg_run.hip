// kicp_reg_run.hip -- one registration: the host-side solve loop and the small-scan path (see kicp_reg_internal.hpp)
#include "kicp_reg_internal.hpp"

using namespace kicp;
using namespace kicp::host;

namespace kicp {
namespace host {
// What the host does with the exact sums of one pass: Registration.cpp:119-125 (solve), :159-167,181-182 (update), :184 (stop
// test), and on pass 0 the regularisation of :48-60,171-177.  Shared by the generic and the small-scan loops.


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
        trace_lap("pass kernel dispatched");
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
            trace_lap("rows of a pass at the host");
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
    trace_lap("map in step with its pending update");
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
}  // namespace host
}  // namespace kicp
