// kicp_small.hpp -- the small-scan registration path (BASELINE.json config 4).
//
// What the reference's pipeline actually registers is small: the double-downsampled source of
// pipeline/KinematicICP.cpp:38-44,68-72 (~1-2k points of a 128k-point frame) and the ~1k points of the 2-D LaserScan entry
// (ros/src/kinematic_icp_ros/nodes/online_node.cpp:53).  On such scans one ICP iteration (Registration.cpp:179-187) is a few
// microseconds of search sitting on a launch + reduction-tree + hand-off floor three times as long.  k_pass_small removes the
// floor's parts one by one:
//   * at most kSmallMaxGroups workgroups of 1024 lanes (G sub-lanes per query, the same search code as k_pass_gather32), so
//     there is NO inter-workgroup reduction: every workgroup folds its lanes in LDS and stores its row - two 64-byte lines of
//     self-validating words - straight into host-mapped memory, where the host adds the (<= 16) rows;
//   * the kernel stays RESIDENT for the iterations of one ComputeRobotMotion call: after publishing the rows of pass k it
//     polls one 64-byte command line in host-mapped memory for the pose of pass k + 1 (or STOP).  A host -> GPU -> host round
//     trip through a polled line costs ~3.5 us against ~7.5 us through a launch (profiles/r02a_micro_handoff.txt); the kernel
//     occupies <= 16 of 256 CUs and never outlives the call: it leaves on STOP, after `max_passes` passes, or when no command
//     arrives within `timeout_ticks` (it then marks the rows of the pass it gave up on, and the host launches afresh).
// The command line is its own flag: seven pose words and one control word = (sequence << 8 | opcode) XOR a 64-bit fold of the
// pose words, so a torn read (some words of the previous command) cannot pass for the command the kernel is waiting for.
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
enum : uint32_t { kCmdContinue = 1u, kCmdStop = 2u };
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
    const unsigned long long *cmd;  // device view of the host-mapped command line (kCmdWords words, 64-byte aligned)
    unsigned long long *rows;       // device view of the host-mapped rows [gridDim.x][kSmallRowWords]
    unsigned long long seq_base;    // the command that starts pass k (k >= 1) carries sequence seq_base + k
    uint32_t tag0;                  // pass k publishes with tag tag0 + k (the host reserves the range)
    uint32_t max_passes;            // passes this launch may serve; 1 = leave after the first (no residency)
    long long timeout_ticks;        // 100 MHz wall-clock ticks a workgroup waits for a command before it gives up
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

// workgroup sum of the lanes' terms -> one row of self-validating words in host memory
template <int BLOCK>
__device__ __forceinline__ void small_publish(const Acc &a, const SmallParams &sp, uint32_t tid, uint32_t tag, int (*s_red)[kWaveLimbs], int *s_flag) {
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
    // |workgroup sum| < 2^83 * BLOCK: bits 0..47 and 48..95 (the upper half carries the sign)
    const unsigned long long m48 = (1ull << 48) - 1;
    const unsigned long long h0 = t.lo & m48, h1 = ((t.lo >> 48) | (static_cast<unsigned long long>(t.hi) << 16)) & m48;
    const unsigned long long v0 = __shfl(h0, lane >> 1, 64), v1 = __shfl(h1, lane >> 1, 64);
    unsigned long long word = (lane & 1) ? v1 : v0;
    if (lane == 2 * kNumSums) word = (*s_flag & 2) ? 1ull : 0ull;
    if (lane > 2 * kNumSums) word = 0ull;
    if (lane < kSmallRowWords)
        __hip_atomic_store(sp.rows + static_cast<size_t>(blockIdx.x) * kSmallRowWords + lane, (word << 16) | tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// G sub-lanes per query as in k_pass_gather32 (the neighbour voxels of a query are dealt round-robin to its sub-lanes).
// BLOCK = 256 (default): one wave per SIMD on as many CUs as the scan has workgroups - the search is a chain of dependent
// loads, and waves that share a SIMD and a CU's L1 path only lengthen it; 1024: a quarter of the rows for the host to add.
template <int BLOCK, int G>
__global__ __launch_bounds__(BLOCK) void k_pass_small(const SmallParams /* read through fresh_args() */) {
    __shared__ int s_red[BLOCK / 64][kWaveLimbs];
    __shared__ int s_flag;
    __shared__ unsigned long long s_cmd[kCmdWords];
    Pose T = fresh_args().p.sol.pose0;
    for (uint32_t pass = 0;; ++pass) {
        const SmallParams &sp = fresh_args();
        const PassParams &p = sp.p;
        uint32_t tid = fresh_tid();
        if (tid == 0) s_flag = 0;
        Acc acc{};
        {
            const uint32_t gt = blockIdx.x * BLOCK + tid;
            const uint32_t i = gt / G;
            const int sub = static_cast<int>(gt % G);
            const float margin = p.search.margin_u;
            Lane L;
            start_lane(L, p, T, i, i < p.n);
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
            if (sub == 0) resolve_and_accumulate(acc, p, T, L.i, L.t);
        }
        __syncthreads();  // s_flag is reset; (s_red of the previous pass has long been read)
        tid = fresh_tid();
        small_publish<BLOCK>(acc, sp, tid, sp.tag0 + pass, s_red, &s_flag);
        if (pass + 1 >= sp.max_passes) return;
        // ---- wait for the next command: wave 0 polls the host line, lanes 0..7 one word each (ONE 64-byte read) -------------
        if ((tid >> 6) == 0) {
            const int lane = tid & 63;
            const unsigned long long want = sp.seq_base + pass + 1;
            const long long t0 = wall_clock64();
            unsigned long long w = 0, ctrl = 0;
            for (;;) {
                if (lane < kCmdWords) w = __hip_atomic_load(sp.cmd + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                unsigned long long pose[7];
#pragma unroll
                for (int k = 0; k < 7; ++k) pose[k] = __shfl(w, k, 64);
                ctrl = __shfl(w, 7, 64) ^ cmd_fold(pose);
                if ((ctrl >> 8) == want && ((ctrl & 0xFFull) == kCmdContinue || (ctrl & 0xFFull) == kCmdStop)) break;
                if (wall_clock64() - t0 > sp.timeout_ticks) {
                    ctrl = 0ull;  // give up: mark the rows of the pass that will not run, then leave
                    if (lane < kSmallRowWords)
                        __hip_atomic_store(sp.rows + static_cast<size_t>(blockIdx.x) * kSmallRowWords + lane,
                                           ((lane == 2 * kNumSums ? kSmallGaveUp : 0ull) << 16) | (sp.tag0 + pass + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    break;
                }
            }
            if (lane < 7) s_cmd[lane] = w;
            if (lane == 7) s_cmd[7] = ctrl & 0xFFull;
        }
        __syncthreads();
        if (static_cast<uint32_t>(s_cmd[7]) != kCmdContinue) return;
        T = Pose{uniform_d(__longlong_as_double(static_cast<long long>(s_cmd[0]))), uniform_d(__longlong_as_double(static_cast<long long>(s_cmd[1]))),
                 uniform_d(__longlong_as_double(static_cast<long long>(s_cmd[2]))), uniform_d(__longlong_as_double(static_cast<long long>(s_cmd[3]))),
                 uniform_d(__longlong_as_double(static_cast<long long>(s_cmd[4]))), uniform_d(__longlong_as_double(static_cast<long long>(s_cmd[5]))),
                 uniform_d(__longlong_as_double(static_cast<long long>(s_cmd[6])))};
        // (no barrier here: wave 0 reaches its next poll only through the two barriers of the next pass's epilogue)
    }
}

}  // namespace kicp
