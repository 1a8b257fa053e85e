// kicp_small.hpp -- the small-scan registration path (BASELINE.json config 4).
//
// What the reference's pipeline actually registers is small: the double-downsampled source of
// pipeline/KinematicICP.cpp:38-44,68-72 (~1-2k points of a 128k-point frame) and the ~1k points of the 2-D LaserScan entry
// (ros/src/kinematic_icp_ros/nodes/online_node.cpp:53).  On such scans one ICP iteration (Registration.cpp:179-187) is a few
// microseconds of search sitting on a launch + reduction-tree + hand-off floor three times as long.  k_pass_small removes the
// floor's parts one by one:
//   * at most kSmallMaxGroups workgroups of 1024 lanes (G sub-lanes per query, the same search code as k_pass_gather32), so
//     there is NO inter-workgroup reduction tree: every workgroup folds its lanes in LDS and stores its row - two 64-byte lines of
//     self-validating words - straight into host-mapped memory, where the host adds the (<= 16) rows; or (round 5, group_rows:
//     the wave-per-query kernel's hundreds of workgroups) adds it into its group's counting accumulators, whose completing
//     additions send ONE row per 32 workgroups to the host (kicp_kernels.hpp::counting_hand_over);
//   * the kernel stays RESIDENT for the iterations of one ComputeRobotMotion call: after publishing the rows of pass k it
//     polls one 64-byte command line in host-mapped memory for the pose of pass k + 1 (or STOP).  A host -> GPU -> host round
//     trip through a polled line costs ~3.5 us against ~7.5 us through a launch (profiles/r02a_micro_handoff.txt); the kernel
//     occupies <= 16 of 256 CUs and never outlives the call: it leaves on STOP, after `max_passes` passes, or when no command
//     arrives within `timeout_ticks` (it then marks the rows of the pass it gave up on, and the host launches afresh).
// The command line is its own flag: seven pose words and one control word = (sequence << 32 | scan << 8 | opcode) XOR a 64-bit
// fold of the pose words, so a torn read (some words of the previous command) cannot pass for the command the kernel is waiting
// for.  There are kPipeSlots lines, taken in turn (command `seq` travels in line seq % kPipeSlots): the host may send the command
// that starts pass k + d (d <= kPipeSlots) as soon as it holds every row of pass k - while workgroups are still busy with the
// passes in between (run_batch_resident keeps several scans of a batch in flight that way).
// Frame and map cannot change while the call is running, so the acquire of the launch (AQL packet / HIP launch) covers all
// passes.
#pragma once
#include "kicp_kernels.hpp"

namespace kicp {

constexpr int kSmallMaxLanes = 16384;    // lanes (queries x sub-lanes) of the largest scan this path takes
constexpr int kSmallMaxGroups = 64;      // workgroups of one launch = rows the host adds (kSmallMaxLanes / 256)
constexpr int kSmallRowWords = 16;       // 14 sum words (two 48-bit halves per sum) + flag word + spare = two 64-byte lines
constexpr uint32_t kSmallMaxPasses = 48; // passes one launch may serve (bounds the tags reserved per launch)
constexpr int kCmdWords = 8;
// kCmdNewScan (a kernel resident across the scans of a batch): like kCmdContinue, the pass it starts being pass 0 of a scan.  Both
// carry the index (into the launch's scan table, SmallParams::scans) of the scan the pass belongs to: a batch's scans take turns.
enum : uint32_t { kCmdContinue = 1u, kCmdStop = 2u, kCmdNewScan = 3u };
constexpr unsigned long long kCmdMaxScans = 1ull << 24;  // scan indices a command can carry
constexpr uint32_t kPipeSlots = 4;  // command lines / row buffers / ticket sets, taken in turn by consecutive passes: passes that may be out at a time
// one scan of a batch as the resident kernel sees it
struct ScanRef {
    const double *src;   // device pointer, AoS xyz fp64
    unsigned long long n;
};
constexpr unsigned long long kSmallGaveUp = 2ull;  // flag-word bit: the workgroup saw no command in time and left

// 64-bit fold of the seven pose words of a command (host and device)
KICP_HD unsigned long long cmd_fold(const unsigned long long w[7]) {
    unsigned long long x = 0x9E3779B97F4A7C15ull;
    for (int i = 0; i < 7; ++i) {
        x = (x ^ w[i]) * 0xD6E8FEB86659FD93ull;
        x ^= x >> 32;
    }
    return x;
}

struct SmallParams {
    PassParams p;                   // (p.sol.mode is not used: the host always solves; p.partials / p.tickets unused)
    const unsigned long long *cmd;  // device view of the host-mapped command lines (kPipeSlots x kCmdWords words, 64-byte aligned)
    unsigned long long *cmd_dev;    // kCmdReplicas copies of the command lines in device memory (await_command)
    int32_t relay;                  // 1: workgroup 0 relays the host line into the copies; 0: the host writes the copies (BAR)
    int32_t group_rows;             // small-scan kernels: 1 the workgroups' sums go through their groups' counting accumulators and ONE row per group of
                                    // 32 reaches the host (p.group_acc, p.sol.pub_rows; round 5, default) | 0 every workgroup sends its own row (`rows`)
    long long *trace;               // debugging aid (option "small_trace"): workgroup 0 stamps its passes here, 4 wall-clock words each
    unsigned long long *rows;       // device view of the host-mapped rows [kPipeSlots][gridDim.x][kSmallRowWords]: pass k writes buffer k % kPipeSlots, so the
                                    // give-up marker of pass k + 1 never lands on a row of pass k the host has not read yet
    unsigned long long seq_base;    // the command that starts pass k (k >= 1) carries sequence seq_base + k
    uint32_t tag0;                  // pass k publishes with tag tag0 + k (the host reserves the range)
    uint32_t max_passes;            // passes this launch may serve; 1 = leave after the first (no residency)
    long long timeout_ticks;        // 100 MHz wall-clock ticks a workgroup waits for a command before it gives up
    const ScanRef *scans;           // k_pass_resident serving a batch: the batch's scans (device memory, written before the launch);
    uint32_t rotate, pad3_;         // k_pass_resident: the workgroups' shares of a scan move on by this many blocks from pass to pass (below)
    uint32_t scan0, trace_pass;     // (trace_pass: the pass of the launch that `trace` stamps)
                                    // the launch starts on scans[scan0]; later passes: the scan their command names.  nullptr: p.src / p.n
};

// Between the passes of a resident kernel NOTHING but the pose, the pass counter and the lane id is worth a register: the
// compiler would otherwise hoist every loop-invariant address and product out of the pass loop and keep it live through the
// search, which sits at the 128-VGPR limit of a 1024-lane workgroup.  So each pass re-reads its arguments through an opaque
// copy of the kernarg pointer (scalar loads from the scalar cache) and re-derives its indices from an opaque copy of the lane
// id.
typedef const SmallParams __attribute__((address_space(4))) *SmallKernarg;
__device__ __forceinline__ const SmallParams &fresh_args() {
    // the kernel's only explicit argument sits at offset 0 of its kernarg segment
    SmallKernarg q = (SmallKernarg)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("; per-pass view of the kernel arguments" : "+s"(q));
    return *(const SmallParams *)q;
}
__device__ __forceinline__ uint32_t fresh_tid() {
    uint32_t t = threadIdx.x;
    asm volatile("; per-pass copy of the lane id" : "+v"(t));
    return t;
}

// Wait for the command that starts pass `pass + 1` and hand its pose to the workgroup through LDS.  Hundreds of workgroups
// polling one line of HOST memory over PCIe starve each other (measured: 270 pollers, 80 us per command), so the command
// reaches the workgroups through DEVICE memory: `cmd_dev` holds kCmdReplicas copies of the 64-byte line (a workgroup polls copy
// blockIdx % kCmdReplicas with agent-scope loads, lanes 0..7 one word each).  Who fills them:
//   relay = 1: wave 0 of workgroup 0 polls the host line (ONE PCIe reader) and stores what it finds into every copy;
//   relay = 0: the host writes the copies itself through the PCIe BAR (the line lives in host-visible fine-grained HBM).
// Returns the command's opcode; 0: no command in time (the rows of the pass that will not run are then marked so that the host
// launches afresh).  Ends in a workgroup barrier.
constexpr int kCmdReplicas = 64;         // copies of the command line ...
constexpr int kCmdStrideWords = 544;     // ... 4352 bytes apart (4 KiB + 256 B), so that the pollers spread over memory channels
// POLL_WAVE: the wave that does the waiting - wave 1 where wave 0 is still busy handing over the rows of the pass (its
// ticket and, for the last workgroup of a group, the group's sum): the two round trips then run side by side.
template <bool MARK_ROWS = true, int POLL_WAVE = 0>
__device__ __forceinline__ uint32_t await_command(const SmallParams &sp, uint32_t tid, uint32_t pass, unsigned long long *s_cmd) {
    if ((tid >> 6) == POLL_WAVE) {
        const int lane = tid & 63;
        const unsigned long long want = sp.seq_base + pass + 1;
        const long long t0 = wall_clock64();
        const bool relay = sp.relay != 0 && blockIdx.x == 0;
        const size_t slot = static_cast<size_t>(want % kPipeSlots) * kCmdWords;  // (the lines of a copy are neighbours)
        const unsigned long long *line = (relay ? sp.cmd : sp.cmd_dev + static_cast<size_t>(blockIdx.x % kCmdReplicas) * kCmdStrideWords) + slot;
        unsigned long long w = 0, ctrl = 0;
        for (;;) {
            if (lane < kCmdWords)
                w = relay ? __hip_atomic_load(line + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)
                          : __hip_atomic_load(line + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            unsigned long long pose[7];
#pragma unroll
            for (int k = 0; k < 7; ++k) pose[k] = __shfl(w, k, 64);
            ctrl = __shfl(w, 7, 64) ^ cmd_fold(pose);
            if ((ctrl >> 32) == (want & 0xFFFFFFFFull) && (ctrl & 0xFFull) >= kCmdContinue && (ctrl & 0xFFull) <= kCmdNewScan) break;
            if (wall_clock64() - t0 > sp.timeout_ticks) {
                ctrl = 0ull;  // give up: mark the rows of the pass that will not run, then leave
                if (MARK_ROWS && !sp.group_rows && lane < kSmallRowWords)
                    __hip_atomic_store(sp.rows + (static_cast<size_t>((pass + 1u) % kPipeSlots) * gridDim.x + blockIdx.x) * kSmallRowWords + lane,
                                       ((lane == 2 * kNumSums ? kSmallGaveUp : 0ull) << 16) | (sp.tag0 + pass + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                break;
            }
        }
        if (relay && ctrl != 0ull && lane < kCmdWords) {  // pass the command on: every word is self-validating, no ordering needed
#pragma unroll
            for (int r = 0; r < kCmdReplicas; ++r)
                __hip_atomic_store(sp.cmd_dev + static_cast<size_t>(r) * kCmdStrideWords + slot + lane, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (lane < 7) s_cmd[lane] = w;
        if (lane == 7) s_cmd[7] = ctrl & 0xFFFFFFFFull;  // scan << 8 | opcode
    }
    __syncthreads();
    return static_cast<uint32_t>(s_cmd[7]) & 0xFFu;  // kCmdContinue / kCmdStop / kCmdNewScan; 0: no command in time
}
// the scan index the last command named (wave-uniform)
__device__ __forceinline__ uint32_t command_scan(const unsigned long long *s_cmd) { return static_cast<uint32_t>(uniform_i(static_cast<int>(s_cmd[7] >> 8))); }
// value of a double in lane l, wave-uniform
__device__ __forceinline__ double uniform_lane_d(double v, int l) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane(static_cast<int>(b), l), hi = __builtin_amdgcn_readlane(static_cast<int>(b >> 32), l);
    return __longlong_as_double((static_cast<long long>(hi) << 32) | static_cast<unsigned int>(lo));
}

// workgroup sum of the lanes' terms -> one row of self-validating words in host memory
template <int BLOCK>
__device__ __forceinline__ void small_publish(const Acc &a, const SmallParams &sp, uint32_t tid, uint32_t pass, int (*s_red)[kWaveLimbs], int *s_flag, int gave_up = 0) {
    const uint32_t tag = sp.tag0 + pass;
    const int lane = tid & 63, wave = tid >> 6;
    int limb[kWaveLimbs];
#pragma unroll
    for (int k = 0; k < kWaveLimbs; ++k) limb[k] = wave_sum_to_lane63(a.limb[k]);
    const int range_error = __any(a.range_error) ? 1 : 0;
    if (lane == 63) {
#pragma unroll
        for (int k = 0; k < kWaveLimbs; ++k) s_red[wave][k] = limb[k];
        if (range_error) atomicOr(s_flag, 2);
    }
    __syncthreads();
    if (wave != 0) return;
    I128 t{0ull, 0ll};
    if (lane < kNumSums)
        for (int w = 0; w < BLOCK / 64; ++w)
            i128_add_limb_sums(t, s_red[w][kTermLimbs * lane], s_red[w][kTermLimbs * lane + 1], s_red[w][kTermLimbs * lane + 2], s_red[w][kTermLimbs * lane + 3]);
    if (sp.group_rows) {  // one row per group of 32 workgroups reaches the host (kicp_kernels.hpp::counting_hand_over)
        counting_hand_over(t, (*s_flag & 2) ? 1 : 0, gave_up, sp.p, tag, pass % kPipeSlots, pass % kPipeSlots, lane);
        return;
    }
    // |workgroup sum| < 2^83 * BLOCK: bits 0..47 and 48..95 (the upper half carries the sign)
    const unsigned long long m48 = (1ull << 48) - 1;
    const unsigned long long h0 = t.lo & m48, h1 = ((t.lo >> 48) | (static_cast<unsigned long long>(t.hi) << 16)) & m48;
    const unsigned long long v0 = __shfl(h0, lane >> 1, 64), v1 = __shfl(h1, lane >> 1, 64);
    unsigned long long word = (lane & 1) ? v1 : v0;
    if (lane == 2 * kNumSums) word = (*s_flag & 2) ? 1ull : 0ull;
    if (lane > 2 * kNumSums) word = 0ull;
    if (lane < kSmallRowWords)
        __hip_atomic_store(sp.rows + (static_cast<size_t>(pass % kPipeSlots) * gridDim.x + blockIdx.x) * kSmallRowWords + lane, (word << 16) | tag, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_SYSTEM);
}

// G sub-lanes per query as in k_pass_gather32 (the neighbour voxels of a query are dealt round-robin to its sub-lanes).
// BLOCK = 256 (default): one wave per SIMD on as many CUs as the scan has workgroups - the search is a chain of dependent
// loads, and waves that share a SIMD and a CU's L1 path only lengthen it; 1024: a quarter of the rows for the host to add.
template <int BLOCK, int G, bool EXPORT = false>
__global__ __launch_bounds__(BLOCK) void k_pass_small(const SmallParams /* read through fresh_args() */) {
    __shared__ int s_red[BLOCK / 64][kWaveLimbs];
    __shared__ int s_flag;
    __shared__ unsigned long long s_cmd[kCmdWords];
    Pose T = fresh_args().p.sol.pose0;
    int gave_up = 0;  // (wave-uniform; group_rows) no command arrived in time: this round only hands over the marked empty sums
    for (uint32_t pass = 0;; ++pass) {
        const SmallParams &sp = fresh_args();
        const PassParams &p = sp.p;
        uint32_t tid = fresh_tid();
        if (tid == 0) s_flag = 0;
        Acc acc{};
        if (!gave_up) {
            const uint32_t gt = blockIdx.x * BLOCK + tid;
            const uint32_t i = gt / G;
            const int sub = static_cast<int>(gt % G);
            const float margin = p.search.margin_u;
            Lane L;
            start_lane(L, p, p.src, T, i, i < p.n);
            if (G > 1) {  // deal the set bits round-robin: the r-th occupied voxel goes to sub-lane r % G
                uint32_t rest = L.todo, mine = 0u;
                for (int r = 0; rest; ++r) {
                    const uint32_t low = rest & (0u - rest);
                    rest ^= low;
                    if (r % G == sub) mine |= low;
                }
                L.todo = mine;
            }
            float cull = L.t.b1;  // running minimum shared by the G sub-lanes (culling only)
            while (__any(L.todo != 0u)) {
                L.todo = cull_todo(L, cull, margin);
                if (L.todo) {
                    const int s = __ffs(L.todo) - 1;
                    L.todo &= L.todo - 1u;
                    visit_bucket<1>(L.q, L.t, p.map, s, margin);
                }
                cull = L.t.b1;
#pragma unroll
                for (int off = 1; off < G; off <<= 1) cull = fminf(cull, __shfl_xor(cull, off, 64));
            }
#pragma unroll
            for (int off = 1; off < G; off <<= 1) {
                Best3 o;
                o.b1 = __shfl_xor(L.t.b1, off, 64), o.b2 = __shfl_xor(L.t.b2, off, 64), o.b3 = __shfl_xor(L.t.b3, off, 64);
                o.i1 = __shfl_xor(L.t.i1, off, 64), o.i2 = __shfl_xor(L.t.i2, off, 64), o.o1 = __shfl_xor(L.t.o1, off, 64), o.o2 = __shfl_xor(L.t.o2, off, 64);
                best3_merge(L.t, o);
            }
            if (sub == 0) resolve_and_accumulate<EXPORT>(acc, p, false, p.src, T, L.i, L.t);
        }
        __syncthreads();  // s_flag is reset; (s_red of the previous pass has long been read)
        tid = fresh_tid();
        small_publish<BLOCK>(acc, sp, tid, pass, s_red, &s_flag, gave_up);
        if (gave_up || pass + 1 >= sp.max_passes) return;
        const uint32_t op = await_command(sp, tid, pass, s_cmd);
        if (op != kCmdContinue) {
            if (op != 0u || !sp.group_rows) return;  // STOP - or, with a row per workgroup, the marked row is out (await_command)
            gave_up = 1;  // (group rows: the mark travels through the group's accumulators, like k_pass_resident's)
            if (fresh_tid() == 0 && sp.p.sol.rec) __hip_atomic_store(&sp.p.sol.rec->reserved[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            continue;
        }
        T = Pose{uniform_d(__longlong_as_double(static_cast<long long>(s_cmd[0]))), uniform_d(__longlong_as_double(static_cast<long long>(s_cmd[1]))),
                 uniform_d(__longlong_as_double(static_cast<long long>(s_cmd[2]))), uniform_d(__longlong_as_double(static_cast<long long>(s_cmd[3]))),
                 uniform_d(__longlong_as_double(static_cast<long long>(s_cmd[4]))), uniform_d(__longlong_as_double(static_cast<long long>(s_cmd[5]))),
                 uniform_d(__longlong_as_double(static_cast<long long>(s_cmd[6])))};
        // (no barrier here: wave 0 reaches its next poll only through the two barriers of the next pass's epilogue)
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// k_pass_resident: the GENERIC pass kernel (one lane per query, kicp_kernels.hpp) resident for the iterations of a call - scans
// too large for the kernels above but small enough for every workgroup to be on the device at once (<= 512 workgroups of the
// latency-oriented build, <= 1024 of the four-waves-per-SIMD build on 256 CUs).  Same search, same exact sums, same reduction:
// tagged workgroup rows, a ticket per group of 32, the group's row to the host (finish_pass, mode 4) - with tag0 + pass as the
// tag of pass `pass`.  What a pass after the first saves is the launch (doorbell -> packet -> dispatch of 2 048 waves, ~5 us)
// against a polled command (~1.3 us).  A workgroup that sees no command in time hands over an EMPTY row marked kGaveUpUnit
// for the pass it will not run - through the same group protocol, so the host finds the mark in the group's row - and leaves.
template <int BLOCK, int OCC, bool LAT>
__global__ __launch_bounds__(BLOCK, OCC) void k_pass_resident(const SmallParams /* read through fresh_args() */) {
    __shared__ int s_red[BLOCK / 64][kWaveLimbs];
    __shared__ int s_flag;
    __shared__ unsigned long long s_cmd[kCmdWords];
    // the four-waves-per-SIMD build (LAT false; round 5: it fits its 128 registers now that the query is parked in LDS and the basis is
    // read where it is used): idle lanes take over voxels of loaded queries, as in k_pass_gather32's build of that shape
    constexpr bool kLends = !LAT;
    __shared__ int s_lend[kLends ? BLOCK / 64 : 1][kLendWords];
    __shared__ double s_park[kLends ? BLOCK * kParkWords : 1];
    Pose T = fresh_args().p.sol.pose0;
    int gave_up = 0;  // (wave-uniform) no command arrived in time: this round only hands over the marked empty row
    uint32_t scan = static_cast<uint32_t>(uniform_i(static_cast<int>(fresh_args().scan0)));  // (a batch: index into sp.scans of the scan this pass belongs to)
    // Which 256 points of the scan this workgroup searches moves on by `rotate` blocks with every pass.  Scans of one sensor are
    // heavy in the same places (the same index ranges hold the rays that meet the densest part of the map), and with several passes
    // in flight the pace is set by the workgroup that is behind: taking turns, a workgroup that had a heavy share catches up on the
    // lighter ones that follow (the sums are exact: who searches what does not change a bit of the result).
    uint32_t share = blockIdx.x;
    for (uint32_t pass = 0;; ++pass) {
        const SmallParams &sp = fresh_args();
        uint32_t tid = fresh_tid();
        if (tid == 0) s_flag = 0;
        const bool stamp = sp.trace != nullptr && tid == 0 && pass == sp.trace_pass;  // option "small_trace": every workgroup stamps its second pass, [workgroup][4]
        if (stamp) sp.trace[4 * blockIdx.x] = wall_clock64();
        Acc acc{};
        if (!gave_up) {
            const double *src = sp.p.src;
            uint32_t n = sp.p.n;
            if (sp.scans) src = sp.scans[scan].src, n = static_cast<uint32_t>(sp.scans[scan].n);  // (uniform: scalar loads)
            gather32_pass<BLOCK, 1, false, LAT, kLends>(sp.p, T, false, tid, acc, src, n, share, kLends ? &s_lend[kLends ? tid / 64 : 0][0] : nullptr, s_park);
            share += sp.rotate;
            if (share >= gridDim.x) share -= gridDim.x;
        }
        __syncthreads();  // s_flag is reset; (s_red of the previous pass has long been read)
        if (sp.trace != nullptr && fresh_tid() == 0 && pass == sp.trace_pass) sp.trace[4 * blockIdx.x + 1] = wall_clock64();  // every wave's search is done
        finish_pass<BLOCK, true>(acc, sp.p, s_red, &s_flag, sp.tag0 + pass, gave_up, pass % kPipeSlots);
        if (sp.trace != nullptr && fresh_tid() == 0 && pass == sp.trace_pass) sp.trace[4 * blockIdx.x + 2] = wall_clock64();  // row stored (a group's last workgroup: group row sent)
        if (gave_up || pass + 1 >= sp.max_passes) return;
        const uint32_t op = await_command<false, 1>(sp, fresh_tid(), pass, s_cmd);
        if (sp.trace != nullptr && fresh_tid() == 0 && pass == sp.trace_pass) sp.trace[4 * blockIdx.x + 3] = wall_clock64();  // next command seen
        if (op != kCmdContinue && op != kCmdNewScan) {
            if (op == kCmdStop) return;
            gave_up = 1;
            // tell the host that a workgroup of this launch gave up: should the call end before every workgroup has taken its ticket
            // of the give-up round (the host stops after the pass it was waiting for), the next launch must not inherit the
            // tickets taken so far - the host clears them when it finds this word set (run_small)
            if (fresh_tid() == 0 && sp.p.sol.rec) __hip_atomic_store(&sp.p.sol.rec->reserved[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            continue;
        }
        scan = command_scan(s_cmd);
        T = Pose{uniform_d(__longlong_as_double(static_cast<long long>(s_cmd[0]))), uniform_d(__longlong_as_double(static_cast<long long>(s_cmd[1]))),
                 uniform_d(__longlong_as_double(static_cast<long long>(s_cmd[2]))), uniform_d(__longlong_as_double(static_cast<long long>(s_cmd[3]))),
                 uniform_d(__longlong_as_double(static_cast<long long>(s_cmd[4]))), uniform_d(__longlong_as_double(static_cast<long long>(s_cmd[5]))),
                 uniform_d(__longlong_as_double(static_cast<long long>(s_cmd[6])))};
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// k_pass_wave: ONE WAVE PER QUERY, for scans of up to kWaveMaxPoints points (the 2-D LaserScan entry, the pipeline's
// double-downsampled source).  A thread-per-query wave walks ~600 VALU instructions per visited bucket plus the fp64 transform
// and accumulation of its 64 queries in lock step - ~10 us of dependent issue however few queries there are, on a machine of
// 1024 SIMDs of which a 1 080-point scan then uses 17.  Here the 64 lanes of a wave share ONE query: the transform, the probe
// and the culling are wave-uniform.  The probe is ONE access: lane l reads word l of the own voxel's 128-byte slot (key,
// neighbour mask and the 27 neighbours' bucket indices; lanes 32-63 read the next slot of the probe sequence).  A round of the
// search looks at up to SIX neighbour buckets at once, one point per lane and bucket (lanes 0-19 / 20-39 / 40-59, two buckets
// each): the fp64 point AND its 16-bit mirror word (which only says whether the slot holds a point) are loaded together, the
// distance is the reference's own fp64 expression - no pre-selection, no margin, no exact fall-back - and the minimum is a DPP
// reduction followed by the reference's rule among the lanes whose norm could tie (closer_by_norm, ties to the earlier one in
// visiting order): the generic kernel's decision, so the two paths give the same bits.  The terms of the normal equations
// come from the same function as the generic kernel's (correspondence_terms: the closed form over the pose's basis, wave-uniform
// here); lanes 0-5 take one each to convert and park (no wave reduction: a wave has ONE correspondence).  A query costs its wave a few hundred instructions and three dependent memory accesses (probe, buckets,
// nothing else: the source point stays in registers across the passes of a call); a 1 080-point scan occupies 1 080 SIMDs.
// Wave 0 adds the workgroup's (<= 16) correspondences and stores the row as small_publish does.  Resident loop and command
// protocol: as k_pass_small.
// ------------------------------------------------------------------------------------------------------------------------
constexpr uint32_t kWaveMaxPoints = 4096;
constexpr int kWaveMaxRows = 272;  // rows one launch may produce (ceil(kWaveMaxPoints / 16) = 256, and 1 080 / 4 = 270)

// exact 128-bit fixed-point term of x (to_fixed without the limb split)
__device__ __forceinline__ I128 to_fixed128(double x, int &range_error) {
    if (!(fabs(x) < kFixLimit)) {
        range_error = 1;
        return I128{0ull, 0ll};
    }
    const long long ip = __double2ll_rn(x);
    const long long fp = __double2ll_rn((x - static_cast<double>(ip)) * kFixScale);
    I128 t{static_cast<unsigned long long>(ip) << 40, ip >> 24};
    i128_add(t, I128{static_cast<unsigned long long>(fp), fp >> 63});
    return t;
}
// minimum of a non-negative float over the wave (as the unsigned order of its bits), in every lane
__device__ __forceinline__ float wave_min_nonneg(float v) {
    uint32_t b = __float_as_uint(v);
    b = min(b, static_cast<uint32_t>(__builtin_amdgcn_update_dpp(static_cast<int>(b), static_cast<int>(b), 0x111, 0xF, 0xF, false)));  // row_shr:1
    b = min(b, static_cast<uint32_t>(__builtin_amdgcn_update_dpp(static_cast<int>(b), static_cast<int>(b), 0x112, 0xF, 0xF, false)));  // row_shr:2
    b = min(b, static_cast<uint32_t>(__builtin_amdgcn_update_dpp(static_cast<int>(b), static_cast<int>(b), 0x114, 0xF, 0xF, false)));  // row_shr:4
    b = min(b, static_cast<uint32_t>(__builtin_amdgcn_update_dpp(static_cast<int>(b), static_cast<int>(b), 0x118, 0xF, 0xF, false)));  // row_shr:8
    // lane 15 of every row now holds the row's minimum: fold the four rows through readlane (wave-uniform result)
    const uint32_t r0 = __builtin_amdgcn_readlane(static_cast<int>(b), 15), r1 = __builtin_amdgcn_readlane(static_cast<int>(b), 31);
    const uint32_t r2 = __builtin_amdgcn_readlane(static_cast<int>(b), 47), r3 = __builtin_amdgcn_readlane(static_cast<int>(b), 63);
    return __uint_as_float(min(min(r0, r1), min(r2, r3)));
}

// minimum of a double >= 0 over the wave (as the unsigned order of its bits), wave-uniform
template <int CTRL>
__device__ __forceinline__ unsigned long long dpp_min_u64(unsigned long long b) {
    const uint32_t lo = static_cast<uint32_t>(__builtin_amdgcn_update_dpp(static_cast<int>(b), static_cast<int>(b), CTRL, 0xF, 0xF, false));
    const uint32_t hi = static_cast<uint32_t>(__builtin_amdgcn_update_dpp(static_cast<int>(b >> 32), static_cast<int>(b >> 32), CTRL, 0xF, 0xF, false));
    const unsigned long long o = (static_cast<unsigned long long>(hi) << 32) | lo;
    return o < b ? o : b;
}
__device__ __forceinline__ double wave_min_nonneg_d(double v) {
    unsigned long long b = static_cast<unsigned long long>(__double_as_longlong(v));
    // row_shr 1, 2, 4, 8: lane 15 of every row of 16 ends up with the row's minimum
    b = dpp_min_u64<0x111>(b), b = dpp_min_u64<0x112>(b), b = dpp_min_u64<0x114>(b), b = dpp_min_u64<0x118>(b);
    unsigned long long r = ~0ull;
#pragma unroll
    for (int l = 15; l < 64; l += 16) {
        const unsigned long long o = (static_cast<unsigned long long>(static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(b >> 32), l))) << 32) |
                                     static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(b), l));
        r = o < r ? o : r;
    }
    return __longlong_as_double(static_cast<long long>(r));
}
// the running best of a wave's search under the reference's rule: smallest norm, ties to the earlier one in visiting order
struct WaveBest {
    double d2;           // exact squared distance (the acceptance bound while idx == kNoIndex32)
    uint32_t idx, ord;   // pool index and visiting order (shift * kOrdStride + position in the bucket)
    double x, y, z;      // the point itself
};
__device__ __forceinline__ bool better_candidate(double cd, uint32_t co, const WaveBest &b) {
    if (b.idx == kNoIndex32) return true;
    return co < b.ord ? !closer_by_norm(b.d2, cd) : closer_by_norm(cd, b.d2);
}

template <int BLOCK, bool EXPORT = false>
__global__ __launch_bounds__(BLOCK) void k_pass_wave(const SmallParams /* read through fresh_args() */) {
    constexpr int kWaves = BLOCK / 64;
    constexpr int kTripPoints = static_cast<int>(kMirrorTrip);  // 20 lanes per bucket: three buckets side by side, two such sets per lane
    __shared__ unsigned long long s_term[kWaves][2 * kNumSums];  // lo / hi of the wave's seven 128-bit terms
    __shared__ int s_flag;
    __shared__ unsigned long long s_cmd[kCmdWords];
    Pose T = fresh_args().p.sol.pose0;
    // this wave's query (wave-uniform) and its source point: the same for every pass of a scan, so it is read once per scan
    // (a launch that serves a batch - sp.scans - takes up the scan each command names)
    double sx = 0.0, sy = 0.0, sz = 0.0;
    uint32_t scan = static_cast<uint32_t>(uniform_i(static_cast<int>(fresh_args().scan0))), n_scan = 0u;
    bool new_scan = true;
    int gave_up = 0;  // (wave-uniform; group_rows) no command arrived in time: this round only hands over the marked empty sums
    for (uint32_t pass = 0;; ++pass) {
        const SmallParams &sp = fresh_args();
        const PassParams &p = sp.p;
        const MapView &m = p.map;
        if (new_scan) {
            const double *src = sp.scans ? sp.scans[scan].src : p.src;
            n_scan = sp.scans ? static_cast<uint32_t>(sp.scans[scan].n) : p.n;
            const uint32_t q0 = blockIdx.x * kWaves + (fresh_tid() >> 6);
            sx = sy = sz = 0.0;
            if (q0 < n_scan) sx = src[3 * q0], sy = src[3 * q0 + 1], sz = src[3 * q0 + 2];
            new_scan = false;
        }
        uint32_t tid = fresh_tid();
        const int lane = tid & 63;
        const int wave = __builtin_amdgcn_readfirstlane(static_cast<int>(tid >> 6));
        if (tid == 0) s_flag = 0;
        const uint32_t qi = blockIdx.x * kWaves + static_cast<uint32_t>(wave);
        const bool stamp = sp.trace != nullptr && tid == 0 && pass == sp.trace_pass;  // every workgroup stamps its second pass: [workgroup][4]
        if (stamp) sp.trace[4 * blockIdx.x] = wall_clock64();
        bool accepted = false;
        double term_value = 0.0;  // lane k < 6: the k-th term of this wave's correspondence (lane 6: the count)
        if (qi < n_scan && !gave_up) {
            const SearchParams &sq = p.search;
            const float margin = sq.margin_u;
            const double vs = m.voxel_size;
            // T * source (Registration.cpp:74,88) and its voxel
            Query q;
            {
                double rx, ry, rz;
                quat_rotate(T, sx, sy, sz, rx, ry, rz);
                q.x = rx + T.tx, q.y = ry + T.ty, q.z = rz + T.tz;
                q.vx = voxel_coord(q.x, vs, sq.inv_vs), q.vy = voxel_coord(q.y, vs, sq.inv_vs), q.vz = voxel_coord(q.z, vs, sq.inv_vs);
            }
            // ---- probe: lane l reads word (l & 31) of the 128-byte slot h (lanes 0..31) or h + 1 (lanes 32..63): key, neighbour
            //      mask and the 27 neighbours' buckets arrive in ONE access
            uint32_t h = voxel_hash(q.vx, q.vy, q.vz) & m.mask;
            uint32_t todo = 0u, line = 0u;
            int base_lane = 0;  // lane that holds word 0 of the matching slot
            for (;;) {
                const uint32_t hh = (h + static_cast<uint32_t>(lane >> 5)) & m.mask;
                line = reinterpret_cast<const uint32_t *>(m.table + hh)[lane & 31];
                bool found = false, absent = false;
#pragma unroll
                for (int half = 0; half < 2 && !found && !absent; ++half) {
                    const int b0 = 32 * half;
                    const uint32_t val = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(line), b0 + 3));
                    if (val == kEmptyVal) {
                        absent = true;
                    } else if (__builtin_amdgcn_readlane(static_cast<int>(line), b0) == q.vx && __builtin_amdgcn_readlane(static_cast<int>(line), b0 + 1) == q.vy &&
                               __builtin_amdgcn_readlane(static_cast<int>(line), b0 + 2) == q.vz) {
                        found = true, base_lane = b0;
                        todo = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(line), b0 + 4));
                    }
                }
                if (found || absent) break;
                h = (h + 2u) & m.mask;
            }
            // the pose's part of the terms (J.col(0) = R UnitX, R UnitY; Registration.cpp:86-93): wave-uniform, issued here so that it runs
            // in the shadow of the bucket loads
            const PassBasis B = basis_of(T);
            Lane L;  // (the culling helpers' view: offsets inside the own voxel in mirror units)
            L.q.lx = static_cast<float>((q.x - q.vx * vs) * sq.upm), L.q.ly = static_cast<float>((q.y - q.vy * vs) * sq.upm),
            L.q.lz = static_cast<float>((q.z - q.vz * vs) * sq.upm);
            L.q.slot0 = 0u;
            const double bound = sq.bound;
            WaveBest best{bound, kNoIndex32, 0u, 0.0, 0.0, 0.0};
            float cur_u = sq.bound_u;  // the best squared distance so far in mirror units (culling only)
            const int group = lane / kTripPoints, kk = lane % kTripPoints;
            while (todo) {
                L.todo = todo;
                todo = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(cull_todo(L, cur_u, margin))));
                if (!todo) break;
                // up to six neighbour voxels this round (reference visiting order): lane group g takes the g-th and the (g + 3)-rd
                int sv[6], ns = 0;
#pragma unroll
                for (int g = 0; g < 6; ++g) {
                    sv[g] = 0;
                    if (todo) {
                        sv[g] = __ffs(todo) - 1;
                        todo &= todo - 1u;
                        ns = g + 1;
                    }
                }
                uint32_t bucket[2];
                int svl[2];
                bool active[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int g = group + 3 * u;
                    active[u] = group < 3 && g < ns;
                    svl[u] = u == 0 ? (group == 0 ? sv[0] : (group == 1 ? sv[1] : sv[2])) : (group == 0 ? sv[3] : (group == 1 ? sv[4] : sv[5]));
                    // the neighbour's bucket index sits in word 5 + s of the slot line (held by lane base_lane + 5 + s)
                    const uint32_t b0 = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(line), base_lane + 5 + sv[3 * u]));
                    const uint32_t b1 = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(line), base_lane + 5 + sv[3 * u + 1]));
                    const uint32_t b2 = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(line), base_lane + 5 + sv[3 * u + 2]));
                    bucket[u] = group == 0 ? b0 : (group == 1 ? b1 : b2);
                }
                bool more = true;
                for (uint32_t k0 = 0; more && k0 < m.cap16; k0 += kMirrorTrip) {
                    const uint32_t k = k0 + static_cast<uint32_t>(kk);
                    // both loads of both buckets are issued before anything is consumed: the 16-bit mirror says whether the slot holds
                    // a point, the fp64 pool is what the distance is computed from (exact: no pre-selection, no margin)
                    MirrorPoint mp[2];
                    double px[2], py[2], pz[2];
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        mp[u] = mirror_empty();
                        px[u] = py[u] = pz[u] = 0.0;
                        if (active[u]) {
                            mp[u] = m.pool16[static_cast<size_t>(bucket[u]) * m.cap16 + k];
                            if (k < m.cap) {
                                const double *t = m.pool + (static_cast<size_t>(bucket[u]) * m.cap + k) * 3;
                                px[u] = t[0], py[u] = t[1], pz[u] = t[2];
                            }
                        }
                    }
                    // this lane's better candidate of the two, then the wave's
                    double d2 = DBL_MAX;
                    uint32_t ord = 0u, gidx = kNoIndex32;
                    double cx = 0.0, cy = 0.0, cz = 0.0;
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const bool have = active[u] && k < m.cap && (mp[u].y >> 16) == 0u;
                        const double dx = px[u] - q.x, dy = py[u] - q.y, dz = pz[u] - q.z;
                        const double e = dx * dx + dy * dy + dz * dz;
                        const uint32_t o = static_cast<uint32_t>(svl[u]) * kOrdStride + k;
                        // (u = 1 is later in visiting order than u = 0: it only replaces on a strictly smaller norm)
                        if (have && e < bound && (gidx == kNoIndex32 || closer_by_norm(e, d2)))
                            d2 = e, ord = o, gidx = bucket[u] * m.cap + k, cx = px[u], cy = py[u], cz = pz[u];
                    }
                    const double dmin = wave_min_nonneg_d(d2);
                    if (dmin < DBL_MAX) {
                        // every lane whose squared distance could round to the minimum's norm (closer_by_norm's threshold)
                        unsigned long long near = __ballot(gidx != kNoIndex32 && d2 * (1.0 - 8.8817841970012523e-16) <= dmin);
                        while (near) {
                            const int l = __ffsll(static_cast<long long>(near)) - 1;
                            near &= near - 1ull;
                            const double cd = uniform_lane_d(d2, l);
                            const uint32_t co = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(ord), l));
                            if (better_candidate(cd, co, best)) {
                                best.d2 = cd, best.ord = co, best.idx = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(gidx), l));
                                best.x = uniform_lane_d(cx, l), best.y = uniform_lane_d(cy, l), best.z = uniform_lane_d(cz, l);
                            }
                        }
                        cur_u = fminf(cur_u, static_cast<float>(best.d2 * sq.upm * sq.upm) * 1.00001f + 1.0f);
                    }
                    // a bucket goes on where the last slot of this trip holds a point (any of the round's buckets)
                    more = __ballot((active[0] && kk == kTripPoints - 1 && (mp[0].y >> 16) == 0u) || (active[1] && kk == kTripPoints - 1 && (mp[1].y >> 16) == 0u)) != 0ull;
                }
            }
            const bool take = best.idx != kNoIndex32 && sqrt(best.d2) < p.tau;  // `distance < max_correspondance_distance`, Registration.cpp:75
            if (EXPORT && lane == 0) export_correspondence(p, qi, take ? best.idx : kNoIndex32, best.d2, take ? best.x : 0.0, take ? best.y : 0.0, take ? best.z : 0.0);
            if (take) {
                accepted = true;
                // the terms of kicp_kernels.hpp::correspondence_terms (one function for every pass kernel: the same doubles), lane k < 6
                // taking term k to convert and park; JTJ(0,0) is the expression basis_of converts
                double term[5];
                correspondence_terms(B, sx, sy, q.x, q.y, q.z, best.x, best.y, best.z, term);
                term_value = lane == 0 ? B.c0x * B.c0x + B.c0y * B.c0y + B.c0z * B.c0z
                                       : (lane == 1 ? term[0] : (lane == 2 ? term[1] : (lane == 3 ? term[2] : (lane == 4 ? term[3] : term[4]))));
                if (lane == 6) term_value = 1.0;  // the count
            }
        }
        if (stamp) sp.trace[4 * blockIdx.x + 1] = wall_clock64();
        int range_error = 0;
        if (lane < kNumSums) {
            const I128 t = accepted ? to_fixed128(term_value, range_error) : I128{0ull, 0ll};
            s_term[wave][2 * lane] = t.lo, s_term[wave][2 * lane + 1] = static_cast<unsigned long long>(t.hi);
        }
        const bool any_range_error = __ballot(range_error != 0) != 0ull;
        __syncthreads();  // (also: s_flag is reset)
        if (lane == 0 && any_range_error) atomicOr(&s_flag, 2);
        __syncthreads();
        tid = fresh_tid();
        if ((tid >> 6) == 0) {
            const int ln = tid & 63;
            I128 t{0ull, 0ll};
            if (ln < kNumSums)
                for (int w = 0; w < kWaves; ++w) i128_add(t, I128{s_term[w][2 * ln], static_cast<long long>(s_term[w][2 * ln + 1])});
            if (sp.group_rows) {  // one row per group of 32 workgroups reaches the host (kicp_kernels.hpp::counting_hand_over)
                counting_hand_over(t, (s_flag & 2) ? 1 : 0, gave_up, sp.p, sp.tag0 + pass, pass % kPipeSlots, pass % kPipeSlots, ln);
            } else {
                const unsigned long long m48 = (1ull << 48) - 1;
                const unsigned long long h0 = t.lo & m48, h1 = ((t.lo >> 48) | (static_cast<unsigned long long>(t.hi) << 16)) & m48;
                const unsigned long long v0 = __shfl(h0, ln >> 1, 64), v1 = __shfl(h1, ln >> 1, 64);
                unsigned long long word = (ln & 1) ? v1 : v0;
                if (ln == 2 * kNumSums) word = (s_flag & 2) ? 1ull : 0ull;
                if (ln > 2 * kNumSums) word = 0ull;
                if (ln < kSmallRowWords)
                    __hip_atomic_store(sp.rows + (static_cast<size_t>(pass % kPipeSlots) * gridDim.x + blockIdx.x) * kSmallRowWords + ln, (word << 16) | (sp.tag0 + pass),
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
        if (sp.trace != nullptr && tid == 0 && pass == sp.trace_pass) sp.trace[4 * blockIdx.x + 2] = wall_clock64();
        if (gave_up || pass + 1 >= sp.max_passes) return;
        const uint32_t op = await_command<true, 1>(sp, tid, pass, s_cmd);
        if (op != kCmdContinue && op != kCmdNewScan) {
            if (op != 0u || !sp.group_rows) return;  // STOP - or, with a row per workgroup, the marked row is out (await_command)
            gave_up = 1;  // (group rows: the mark travels through the group's accumulators, like k_pass_resident's)
            if (fresh_tid() == 0 && sp.p.sol.rec) __hip_atomic_store(&sp.p.sol.rec->reserved[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            continue;
        }
        if (sp.scans && command_scan(s_cmd) != scan) scan = command_scan(s_cmd), new_scan = true;
        if (sp.trace != nullptr && tid == 0 && pass == sp.trace_pass) sp.trace[4 * blockIdx.x + 3] = wall_clock64();
        T = Pose{uniform_d(__longlong_as_double(static_cast<long long>(s_cmd[0]))), uniform_d(__longlong_as_double(static_cast<long long>(s_cmd[1]))),
                 uniform_d(__longlong_as_double(static_cast<long long>(s_cmd[2]))), uniform_d(__longlong_as_double(static_cast<long long>(s_cmd[3]))),
                 uniform_d(__longlong_as_double(static_cast<long long>(s_cmd[4]))), uniform_d(__longlong_as_double(static_cast<long long>(s_cmd[5]))),
                 uniform_d(__longlong_as_double(static_cast<long long>(s_cmd[6])))};
    }
}

}  // namespace kicp
