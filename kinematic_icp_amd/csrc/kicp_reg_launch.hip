// kicp_reg_launch.hip -- the pass kernels' launches, the handle's buffers and tags, the waits for rows and records (see kicp_reg_internal.hpp)
#include "kicp_reg_internal.hpp"

using namespace kicp;
using namespace kicp::host;

namespace kicp {
namespace host {
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


// Sub-lanes per query of variant 3.  Small scans are latency bound (few waves, each lane's chain of dependent bucket
// visits decides the kernel time): spreading a query's neighbour voxels over 2-4 lanes shortens that chain.  Large
// scans already fill the machine and only pay for the extra waves.
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
int launch_pass(kicp_reg *r, const PassParams &p, bool allow_aql) {
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
    HIP_TRY(pinned_alloc(reinterpret_cast<void **>(&r->rows), want * kReduceWords * sizeof(unsigned long long), hipHostMallocMapped | hipHostMallocCoherent));
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
int wait_rows(kicp_reg *r, size_t groups, uint32_t tag, long long out_words[kReduceWords], size_t first_row) {
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
        HIP_TRY(pinned_alloc(reinterpret_cast<void **>(&r->cmd), kPipeSlots * kCmdWords * sizeof(unsigned long long), hipHostMallocMapped | hipHostMallocCoherent));
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
void send_command(kicp_reg *r, unsigned long long seq, uint32_t op, const Pose &T, uint32_t scan) {
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
}  // namespace host
}  // namespace kicp
