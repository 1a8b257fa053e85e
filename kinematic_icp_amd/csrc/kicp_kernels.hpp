// kicp_kernels.hpp -- gfx950 (CDNA4, wave64) kernels of the registration hot path.
//
//   k_pass_*    : fused DataAssociation + ComputePerturbation reduction (+ the pass-0 sum of
//                 ComputeOdometryRegularization): registration/Registration.cpp:62-81, 83-118, 48-55, with
//                 kiss_icp::VoxelHashMap::GetClosestNeighbor (kiss-icp v1.2.0) inlined as ONE table
//                 probe + the scan of the neighbour buckets that can matter.  No correspondence list is materialised.
//                 One launch = one ICP iteration; every variant ends in finish_pass (exact reduction + hand-off).
//       k_pass_gather32 variant 3 (default): thread (or 2 / 4 sub-lanes) per query; the 16-bit mirror pre-selects (packed
//                      fp32 arithmetic, integer-key tournament), the winner and anything within the error margin of it
//                      are resolved in fp64.
//                      (Removed after losing every A/B: round 1's variants with the neighbourhoods staged in LDS per wave, and - in
//                      round 6 - the plain fp64 gather that served as the ablation's baseline; k_closest still runs the plain
//                      fp64 search, search_global, which is also the exact fallback of the mirror search.)
//   finish_pass : wave / workgroup reduction of the exact sums, then either tagged rows for the host (default: the host
//                 adds the rows of the first-level groups and solves, Registration.cpp:119-125,159-167,181-184) or the
//                 full device-side tree whose last workgroup solves on one lane (host_solve = 0) / leaves the totals
//                 for an all-reduce (RCCL and callback modes).
//   k_closest   : GetClosestNeighbor for a batch of queries (API parity / tests).
//
// Roofline: gather + reduction, ~0.02 flop/B -> memory bound, no MFMA.
// Numerics: per-point arithmetic is fp64 in the reference's operation order (TU compiled with -ffp-contract=off),
// so NN decisions and per-correspondence terms equal the fp64 reference's.  The sums are accumulated EXACTLY:
// each term is rounded once to a multiple of 2^-40 and added as a 128-bit integer, so the result does not depend
// on summation order, kernel variant, binning order or the number of GPUs (bit-reproducible).
#pragma once
#include <cfloat>
#include <climits>

#include "kicp_common.hpp"
#include "kicp_se3.hpp"

namespace kicp {

constexpr int kMaxLog = 32;
constexpr int kNumSums = 7;           // JTJ00 JTJ01 JTJ11 JTr0 JTr1 ssq count
constexpr int kNumLimbs = 3 * kNumSums;
constexpr int kReduceWords = 24;      // all-reduce payload: 21 limb sums + padding (int64)
constexpr double kFixScale = 1099511627776.0;  // 2^40
constexpr double kFixLimit = 8796093022208.0;  // 2^43: |term| must stay below this (source points within ~2900 km of the base frame)

// Result record in host-mapped pinned memory: written by the finalising lane, polled by the host.
struct HostRecord {
    unsigned long long seq;  // (call_id << 16) | (done << 15) | completed iterations; written last, system-scope release
    int32_t done, iter, converged, nan_flag;
    Pose T;
    double beta;
    double log_ncorr[kMaxLog];
    double log_sums[kMaxLog][6];
    double log_dx[kMaxLog][2];
    double sums[kNumSums];  // last pass, for kicp_pass_sums
    uint32_t reserved[4];
    long long words[kReduceWords];  // limb totals of the last pass (host-side solve / kicp_pass_words)
};

// Device-resident loop state of one ComputeRobotMotion call.
struct IcpState {
    Pose T;  // current_estimate
    double beta;
    int32_t done, iter, converged, nan_flag;
    unsigned int reserved_;
    long long reduce[kReduceWords];      // limb sums of the running pass (multi-GPU: all-reduced in place)
    unsigned int pad_[32];
    unsigned int ticket;                 // second-level arrival counter (own cache line)
    unsigned int pad2_[31];
};

// What the per-correspondence terms need of the pose (Registration.cpp:86-93,108-113), the same for every
// correspondence of a pass: J.col(0) = R UnitX = c0 and J.col(1) = R (-s.y, s.x, 0) = -s.y c0 + s.x c1 with c1 = R UnitY, and the
// fixed-point term of JTJ(0,0) = |c0|^2, which every correspondence adds unchanged.  Computed ONCE per pass - by the host, next to
// the pose it sends (set_pose), or by the kernel where the pose comes from the device (device-side solve, resident kernels) - by
// this one function, so that every kernel and the host see the same doubles.
struct PassBasis {
    double c0x, c0y, c0z, c1x, c1y, c1z;
    int jtj00[4];  // to_fixed(|c0|^2): four signed 21-bit limbs (= 2^40 exactly for any unit quaternion)
};
struct SolveParams {
    Pose pose0;
    PassBasis basis;  // of pose0 (set_pose)
    int32_t pass;
    int32_t max_iterations;
    double convergence_criterion;
    int32_t adaptive;
    double fixed_regularization;
    int32_t mode;  // 0 = solve inside the pass kernel; 1 = leave the limb totals in st->reduce (multi-GPU: all-reduce follows);
                   // 3 = as 1, but the pose of every pass comes from the kernel argument (host-side solve, multi-GPU);
                   // 2 = publish the limb totals to the host record, the HOST solves (a CPU core does the ~2000 serial fp64
                   //     instructions of the solve in 0.2 us, one GPU lane needs ~5 us);
                   // 4 = as 2, but the reduction stops at the first-level groups: their tagged rows go to the host, which
                   //     adds them (default: two dependent device-scope round trips instead of six).  Sending every
                   //     workgroup's row was tried and is far slower: writes into host memory cost ~130 ns per transaction
    unsigned long long call_id;
    HostRecord *rec;  // device pointer to the host-mapped record
    // mode 2 hand-off target (host-mapped memory: the handle's own record, or this rank's slot of the node-wide
    // shared segment in multi-process mode): 24 limb words, then the sequence word set to pub_value
    long long *pub_words;
    unsigned long long *pub_seq;
    unsigned long long pub_value;
    // mode 4 hand-off (default single-GPU / shared-segment mode): every first-level group's row goes straight to the
    // host, each word carrying the 16-bit tag of this pass; the host adds the rows as they arrive
    unsigned long long *pub_rows;  // host-mapped [groups][kReduceWords]
    uint32_t tag;                  // 1..65535, unique per pass within an epoch (the buffers are cleared when it wraps)
    // mode 5: one-shot exchange over peer mappings (multi-GPU without a collective library).  The
    // last workgroup of the launch writes this rank's totals - tagged - into slot `p2p_rank` of EVERY rank's mailbox (its
    // own included; the peers' mailboxes are IPC mappings of their HBM, reached over xGMI), collects the p2p_nranks slots of
    // its own mailbox, adds them in rank order and hands the node-wide totals to its host as in mode 2.
    unsigned long long *const *p2p_peers;  // device array [p2p_nranks]: every rank's mailbox as seen from this GPU
    int32_t p2p_nranks, p2p_rank;
    uint32_t p2p_tag;     // 1..65535 (step % 65535 + 1); double-buffered by p2p_parity, so a stale slot can never match
    uint32_t p2p_parity;  // step & 1
    // mode 6: as 5 without the second reduction level - the last workgroup of every first-level group writes the group's row
    // into every rank's mailbox itself, and the last workgroup of group 0 adds ALL ranks' group rows (its own rank's included,
    // in rank and group order) as they arrive.  For launches of at most kP2pMaxGroups groups; a larger launch reduces on two
    // levels as mode 5 does and sends its total as a single row (mode 7), so ranks on either side of the limit still pair up.
    long long p2p_timeout_ticks;  // 100 MHz ticks a rank waits for its peers' slots (derived from the host's KICP_WAIT_TIMEOUT_S)
};
// a rank's mailbox: [2 parities][nranks][kP2pWords] tagged words; a 64-bit total travels as two tagged 32-bit halves
constexpr int kP2pWords = 2 * 24;
constexpr int kP2pMaxRanks = 16;
// ... followed by the area of mode 6: [2 parities][nranks][kP2pMaxGroups] rows of 24 tagged words (value << 16 | tag, like the
// rows of mode 4) - the first-level GROUP rows of every rank, written by the groups' last workgroups themselves
constexpr int kP2pMaxGroups = 32;  // <= 1024 workgroups per rank and launch; larger launches exchange totals (mode 5)
constexpr int kP2pCountWord = 23;  // word of a row that carries the sender's group count (padding in the single-GPU layout)
KICP_HD size_t p2p_rows_offset(int nranks) { return 2 * static_cast<size_t>(nranks) * kP2pWords; }
KICP_HD size_t p2p_box_words(int nranks) { return p2p_rows_offset(nranks) + 2 * static_cast<size_t>(nranks) * kP2pMaxGroups * 24; }

// Wave-uniform numbers of the pre-selection over the 16-bit mirror, computed once per call on the host (search_params()).
struct SearchParams {
    double bound;    // tau^2 (1 + 9.1e-13): acceptance bound on exact squared distances
    double upm;      // mirror units per metre = 65536 / voxel_size
    double inv_vs;   // 1 / voxel_size (voxel_coord's fast path)
    float bound_u;   // the bound in units^2, widened by the margin
    float margin_u;  // error margin of one decision between two mirror distances, units^2
    double tau2_lo, tau2_hi;  // tau^2 (1 -+ 2^-49): below / above, `sqrt(d2) < tau` is decided without taking the root (accepted_by_norm)
};
KICP_HD SearchParams search_params(double tau, double vs) {
    SearchParams sp;
    sp.bound = tau * tau * (1.0 + 9.1e-13);
    sp.upm = mirror_units_per_metre(vs);
    sp.inv_vs = 1.0 / vs;
    sp.tau2_lo = tau * tau * (1.0 - 1.7763568394002505e-15), sp.tau2_hi = tau * tau * (1.0 + 1.7763568394002505e-15);
    // error model in metres (k_pass_gather32): |delta| <= sqrt(3) * 1.05 units = 2.78e-5 vs; D off by <= 2 sqrt(D) |delta| +
    // |delta|^2 + 4.2e-6 D; both candidates of a decision, 10 % spare; D <= min(bound, 12 vs^2)
    const double bcap = fmin(sp.bound, 12.0 * vs * vs);
    const double margin = 2.2 * (5.55e-5 * sqrt(bcap) * vs + 4.2e-6 * bcap + 7.7e-10 * vs * vs);
    sp.margin_u = static_cast<float>(margin * sp.upm * sp.upm) * 1.00001f;
    sp.bound_u = static_cast<float>(sp.bound * sp.upm * sp.upm) * 1.00001f + sp.margin_u;
    return sp;
}

struct PassParams {
    const double *src;  // scan points, base frame, AoS xyz fp64 (device)
    uint32_t n;
    MapView map;
    double tau;
    SearchParams search;  // = search_params(tau, map.voxel_size)
    IcpState *st;
    unsigned long long *partials;  // [(grid + ceil(grid/32)) * 24] limb rows of the workgroups, then of the groups
    unsigned int *tickets;         // first-level arrival counters, one per group, 128 B apart, zero between launches
    unsigned long long *group_acc; // resident kernels: the groups' counting accumulators (finish_pass, ROWS_ONLY), zero between passes
    SolveParams sol;
    int32_t dbg;          // ablation switches for tools/dbg_census.py (0 = normal operation); read by the DBG build only (dbg_is)
    // kicp_pass_correspondences (the EXPORT instantiations of the pass kernels only; nullptr otherwise): what DataAssociation appends
    // for query i (Registration.cpp:73-77) - the chosen map point's index in the pool (-1: no correspondence), its squared distance
    // to T * source[i] and its coordinates
    int32_t *corr_index;
    double *corr_d2, *corr_nn;
};
// The ablation / attribution switches (`dbg`, tools/gpu_dbg.py, tools/ab_option.py, bench.py's floor and rounds census) exist in
// libkicp_amd_dbg.so only (make dbg: -DKICP_DBG_BUILD).  In the production library every test below is a compile-time constant:
// the pass kernels carry no branch, no scalar load and no register for them (round 6; they cost the issue-bound kernel SALU work).
#ifdef KICP_DBG_BUILD
constexpr bool kDbgBuild = true;
#else
constexpr bool kDbgBuild = false;
#endif
__device__ __forceinline__ bool dbg_is(const PassParams &p, int v) { return kDbgBuild && p.dbg == v; }
__device__ __forceinline__ bool dbg_not(const PassParams &p, int v) { return !kDbgBuild || p.dbg != v; }


// ------------------------------------------------------------------------------------------------------------
// exact accumulation
// ------------------------------------------------------------------------------------------------------------
struct I128 {
    unsigned long long lo;
    long long hi;
};
__device__ __forceinline__ void i128_add(I128 &a, const I128 &b) {
    const unsigned long long old = a.lo;
    a.lo += b.lo;
    a.hi += b.hi + (a.lo < old ? 1 : 0);
}
// T = l0 + l1*2^40 + l2*2^80 with 0 <= l0,l1 < 2^40, l2 signed
__device__ __forceinline__ void i128_to_limbs(const I128 &t, long long l[3]) {
    const unsigned long long m40 = (1ull << 40) - 1;
    l[0] = static_cast<long long>(t.lo & m40);
    l[1] = static_cast<long long>(((t.lo >> 40) | (static_cast<unsigned long long>(t.hi) << 24)) & m40);
    l[2] = t.hi >> 16;
}
__device__ __forceinline__ double limbs_to_double(const long long l[3]) {
    // recombine (limb sums may exceed 40 bits and be negative), then convert the 128-bit magnitude
    I128 t{static_cast<unsigned long long>(l[0]), l[0] >> 63};
    I128 a{static_cast<unsigned long long>(l[1]) << 40, l[1] >> 24};
    I128 b{0ull, static_cast<long long>(static_cast<unsigned long long>(l[2]) << 16)};
    i128_add(t, a), i128_add(t, b);
    const bool neg = t.hi < 0;
    if (neg) {
        t.lo = ~t.lo + 1ull;
        t.hi = ~t.hi + (t.lo == 0ull ? 1 : 0);
    }
    const double mag = static_cast<double>(static_cast<unsigned long long>(t.hi)) * 18446744073709551616.0 + static_cast<double>(t.lo);
    return (neg ? -mag : mag) / kFixScale;
}

// A lane's contribution to the seven sums of one pass: every lane serves at most ONE correspondence per pass, so each sum is
// a single fixed-point term x * 2^40 with |x| < 2^43, held as four 21-bit limbs (the top one signed) - a wave's 64 values of
// a limb then add up in int32 (finish_pass).  The term is formed without leaving the integers' range: x = ip + fr with
// ip = rint(x) (exact in int64) and fr = x - ip (exact, |fr| <= 1/2), so round(x * 2^40) = ip * 2^40 + round(fr * 2^40)
// - the same integer the single conversion gives wherever that one fits (ties included: ip * 2^40 is even).
constexpr int kTermLimbs = 4;
constexpr int kWaveLimbs = kTermLimbs * kNumSums;
constexpr int kLendJobs = 32;                     // voxels a wave's loaded queries can give away per round (gather32_pass)
constexpr int kLendWords = kLendJobs * (1 + 7 + 7);  // per wave: the jobs, the lent lanes' own records, the records they hand back
struct Acc {
    int limb[kWaveLimbs];
    int range_error;
};
// The term as an integer, T = rint(x 2^40) (|T| < 2^83; x 2^40 is exact, rint of a double beyond 2^52 is the double itself), split
// into four SIGNED limbs of 21 bits by truncating divisions carried out in fp64 - every step exact: q = trunc(t 2^-k) and
// t - q 2^k = fma(q, -2^k, t) are representable (the remainder keeps t's granularity and is shorter than t).
//   T = l0 + l1 2^21 + l2 2^42 + l3 2^63,  |lk| < 2^21, every limb with T's sign
// (round 4 built the same integer from two 64-bit conversions and split it with 128-bit integer arithmetic: ~30 instructions
// against ~18; the limbs were non-negative below the top one - any split of the same integer adds up to the same sums).
KICP_HD void to_fixed(double x, int *limb, int &range_error) {
    const bool ok = fabs(x) < kFixLimit;
    if (!ok) range_error = 1;
    const double t = rint((ok ? x : 0.0) * kFixScale);
    const double q3 = trunc(t * 0x1p-63);
    const double r3 = fma(q3, -0x1p63, t);
    const double q2 = trunc(r3 * 0x1p-42);
    const double r2 = fma(q2, -0x1p42, r3);
    const double q1 = trunc(r2 * 0x1p-21);
    const double r1 = fma(q1, -0x1p21, r2);
    limb[0] = static_cast<int>(r1), limb[1] = static_cast<int>(q1), limb[2] = static_cast<int>(q2), limb[3] = static_cast<int>(q3);
}
KICP_HD PassBasis basis_of(const Pose &T) {
    PassBasis b;
    quat_rotate(T, 1.0, 0.0, 0.0, b.c0x, b.c0y, b.c0z);  // J.col(0) = R * UnitX (Registration.cpp:90)
    quat_rotate(T, 0.0, 1.0, 0.0, b.c1x, b.c1y, b.c1z);
    int range_error = 0;  // (|c0|^2 is 1 to rounding for a unit quaternion; a NaN pose leaves zeros - its passes add nothing)
    to_fixed(b.c0x * b.c0x + b.c0y * b.c0y + b.c0z * b.c0z, b.jtj00, range_error);
    return b;
}
KICP_HD void set_pose(SolveParams &f, const Pose &T) { f.pose0 = T, f.basis = basis_of(T); }
// the 128-bit sum of one term's limb sums s[0..3] (each a sum of at most 2^10 limbs)
__device__ __forceinline__ void i128_add_limb_sums(I128 &t, long long s0, long long s1, long long s2, long long s3) {
    const long long low = s0 + s1 * 2097152ll;  // |.| < 2^53 (limb sums are signed)
    i128_add(t, I128{static_cast<unsigned long long>(low), low >> 63});
    i128_add(t, I128{static_cast<unsigned long long>(s2) << 42, s2 >> 22});
    i128_add(t, I128{static_cast<unsigned long long>(s3) << 63, s3 >> 1});
}

// ------------------------------------------------------------------------------------------------------------
// nearest-neighbour search helpers
// ------------------------------------------------------------------------------------------------------------
struct Query {
    double x, y, z;      // transformed point T*p
    int32_t vx, vy, vz;  // PointToVoxel(T*p)
};
// squared distances to the -/+ faces of the own voxel per axis, and the culling slack
struct Faces {
    double fm[3], fp[3];
    double slack;
};

__device__ __forceinline__ void make_query(Query &q, double x, double y, double z, double vs) {
    q.x = x, q.y = y, q.z = z;
    q.vx = static_cast<int32_t>(floor(x / vs)), q.vy = static_cast<int32_t>(floor(y / vs)), q.vz = static_cast<int32_t>(floor(z / vs));
}
__device__ __forceinline__ void make_faces(Faces &f, const Query &q, double vs) {
    const double lx = q.x - q.vx * vs, ly = q.y - q.vy * vs, lz = q.z - q.vz * vs;
    f.fm[0] = lx * lx, f.fm[1] = ly * ly, f.fm[2] = lz * lz;
    f.fp[0] = (vs - lx) * (vs - lx), f.fp[1] = (vs - ly) * (vs - ly), f.fp[2] = (vs - lz) * (vs - lz);
    // culling slack: far above fp64 rounding of the face distances, far below anything that matters
    f.slack = 4.0 * vs * 9.1e-13 * (fabs(q.x) + fabs(q.y) + fabs(q.z) + vs);
}

// lower bound of the squared distance from the query to any point of neighbour voxel (dx,dy,dz)
__device__ __forceinline__ double box_d2(const Faces &f, int dx, int dy, int dz) {
    double d = 0.0;
    d += dx > 0 ? f.fp[0] : (dx < 0 ? f.fm[0] : 0.0);
    d += dy > 0 ? f.fp[1] : (dy < 0 ? f.fm[1] : 0.0);
    d += dz > 0 ? f.fp[2] : (dz < 0 ? f.fm[2] : 0.0);
    return d;
}

// bucket value of voxel (x,y,z), or kEmptyVal when it holds no points (absent or halo entry)
__device__ __forceinline__ uint32_t table_lookup(const MapView &m, int32_t x, int32_t y, int32_t z) {
    uint32_t h = voxel_hash(x, y, z) & m.mask;
    for (;;) {
        const int4 e = *reinterpret_cast<const int4 *>(m.table + h);
        if (static_cast<uint32_t>(e.w) == kEmptyVal) return kEmptyVal;
        if (e.x == x && e.y == y && e.z == z) return val_count(static_cast<uint32_t>(e.w), m.cbits) ? static_cast<uint32_t>(e.w) : kEmptyVal;
        h = (h + 1) & m.mask;
    }
}
// the whole entry: bucket value and the 27-bit neighbour-occupancy mask (0 when the voxel has no entry at all,
// i.e. none of its 27 neighbours holds a point)
// slot index of the entry (for its nb[] record) and the 27-bit neighbour-occupancy mask; nbr == 0 when the voxel has
// no entry at all, i.e. none of its 27 neighbours holds a point
__device__ __forceinline__ void table_lookup_entry(const MapView &m, int32_t x, int32_t y, int32_t z, uint32_t &slot, uint32_t &nbr) {
    uint32_t h = voxel_hash(x, y, z) & m.mask;
    for (;;) {
        const int4 e = *reinterpret_cast<const int4 *>(m.table + h);
        const uint32_t n = m.table[h].nbr;  // same cache line: issued together with the key
        if (static_cast<uint32_t>(e.w) == kEmptyVal) {
            slot = 0u, nbr = 0u;
            return;
        }
        if (e.x == x && e.y == y && e.z == z) {
            slot = h, nbr = n;
            return;
        }
        h = (h + 1) & m.mask;
    }
}

// `a` beats the running minimum `best` (both exact squared distances) under the reference's rule: kiss-icp compares
// (p - query).norm(), i.e. the ROUNDED SQUARE ROOTS, with strict '<' (std::min_element inside a voxel, `distance <
// closest_distance` across voxels - kiss-icp v1.2.0 core/VoxelHashMap.cpp GetClosestNeighbor), so a later candidate whose
// squared distance is smaller by an ulp or two but whose square root rounds to the same double does NOT replace the earlier
// one.  The square roots are only taken when the squared distances are that close (sqrt halves a relative difference: beyond
// 2^-50 the roots differ by more than their rounding).
__device__ __forceinline__ bool closer_by_norm(double a, double best) {
    if (!(a < best)) return false;
    if (a < best * (1.0 - 8.8817841970012523e-16)) return true;  // 1 - 2^-50
    return sqrt(a) < sqrt(best);
}
// `distance < max_correspondance_distance` (Registration.cpp:75) as the reference evaluates it - on the ROUNDED square root - without
// paying for the root where the squared distance is not within 2^-49 of tau^2: d2 < tau^2 (1 - 2^-49) makes the rounded root smaller
// than tau whatever the roundings of tau * tau and of the root (each 2^-53), d2 >= tau^2 (1 + 2^-49) makes it larger.
__device__ __forceinline__ bool accepted_by_norm(double d2, double tau, const SearchParams &sp) {
    if (d2 < sp.tau2_lo) return true;
    if (!(d2 < sp.tau2_hi)) return false;
    return sqrt(d2) < tau;
}
// scan one bucket in insertion order; strict '<' on the norms keeps the first minimum (std::min_element + `distance < closest`)
__device__ __forceinline__ void scan_points(const double *__restrict__ p, uint32_t count, uint32_t base_index, const Query &q,
                                            double &best, uint32_t &best_idx) {
    uint32_t k = 0;
    // ten, then four points per trip: all of a trip's loads in flight together, then the comparisons in insertion order (a bucket of
    // the default 20 points is two round trips to memory instead of ten)
    for (; k + 10 <= count; k += 10) {
        double c[30];
#pragma unroll
        for (int j = 0; j < 30; ++j) c[j] = p[3 * k + j];
#pragma unroll
        for (int j = 0; j < 10; ++j) {
            const double dx = c[3 * j] - q.x, dy = c[3 * j + 1] - q.y, dz = c[3 * j + 2] - q.z;
            const double d2 = dx * dx + dy * dy + dz * dz;
            if (closer_by_norm(d2, best)) best = d2, best_idx = base_index + k + j;
        }
    }
    for (; k + 4 <= count; k += 4) {
        double c[12];
#pragma unroll
        for (int j = 0; j < 12; ++j) c[j] = p[3 * k + j];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const double dx = c[3 * j] - q.x, dy = c[3 * j + 1] - q.y, dz = c[3 * j + 2] - q.z;
            const double d2 = dx * dx + dy * dy + dz * dz;
            if (closer_by_norm(d2, best)) best = d2, best_idx = base_index + k + j;
        }
    }
    for (; k + 2 <= count; k += 2) {
        const double ax = p[3 * k], ay = p[3 * k + 1], az = p[3 * k + 2];
        const double bx = p[3 * k + 3], by = p[3 * k + 4], bz = p[3 * k + 5];
        const double adx = ax - q.x, ady = ay - q.y, adz = az - q.z;
        const double bdx = bx - q.x, bdy = by - q.y, bdz = bz - q.z;
        const double a2 = adx * adx + ady * ady + adz * adz;
        const double b2 = bdx * bdx + bdy * bdy + bdz * bdz;
        if (closer_by_norm(a2, best)) best = a2, best_idx = base_index + k;
        if (closer_by_norm(b2, best)) best = b2, best_idx = base_index + k + 1;
    }
    if (k < count) {
        const double dx = p[3 * k] - q.x, dy = p[3 * k + 1] - q.y, dz = p[3 * k + 2] - q.z;
        const double d2 = dx * dx + dy * dy + dz * dz;
        if (closer_by_norm(d2, best)) best = d2, best_idx = base_index + k;
    }
}

// 27-voxel 1-NN straight from HBM/L2.  `best` enters as the acceptance bound (or DBL_MAX).  `cull`: a squared distance some
// point of the map is KNOWN to have (the pre-selected winner's, exact) - voxels that cannot hold anything that close are skipped
// from the start; the winner is still chosen among everything that is visited, in visiting order, by the reference's rule.
__device__ __forceinline__ void search_global(const MapView &m, const Query &q, double &best, uint32_t &best_idx, double cull = 1.7976931348623157e308) {
    Faces f;
    make_faces(f, q, m.voxel_size);
#pragma unroll 1
    for (int s = 0; s < 27; ++s) {
        const int dx = shift_component(kShiftX, s), dy = shift_component(kShiftY, s), dz = shift_component(kShiftZ, s);
        if (box_d2(f, dx, dy, dz) > fmin(best, cull) + f.slack) continue;  // no point in there can beat `best` (or match `cull`)
        const uint32_t val = table_lookup(m, q.vx + dx, q.vy + dy, q.vz + dz);
        if (val == kEmptyVal) continue;
        const uint32_t bucket = val_bucket(val, m.cbits);
        scan_points(m.pool + static_cast<size_t>(bucket) * m.cap * 3, val_count(val, m.cbits), bucket * m.cap, q, best, best_idx);
    }
}

constexpr uint32_t kNoIndex32 = 0xFFFFFFFFu;
// exact fp64 evaluation of one candidate from the HBM pool
__device__ __forceinline__ double exact_d2_of(double tx, double ty, double tz, const Query &q) {
    const double dx = tx - q.x, dy = ty - q.y, dz = tz - q.z;
    return dx * dx + dy * dy + dz * dz;
}
__device__ __forceinline__ double exact_d2(const MapView &m, uint32_t gidx, const Query &q) {
    const double *t = m.pool + static_cast<size_t>(gidx) * 3;
    return exact_d2_of(t[0], t[1], t[2], q);
}

// ------------------------------------------------------------------------------------------------------------
// per-correspondence terms (Registration.cpp:86-93,108-113)
// ------------------------------------------------------------------------------------------------------------
// The reference forms J = [R UnitX | R (-s.y, s.x, 0)] and r = T s - t per correspondence and adds J^T J and J^T r.  With R
// orthogonal those products have a closed form, which is what is evaluated here:
//   JTJ(0,0) = |c0|^2 (the pose's alone: PassBasis::jtj00)     JTJ(0,1) = -s.y     JTJ(1,1) = s.x^2 + s.y^2
//   JTr(0)   = c0 . r                                           JTr(1)   = s.x (c1 . r) - s.y (c0 . r)
// It differs from the literal products by their own rounding (a few 1e-16 relative: tests/test_closed_form.py holds the two to
// 1e-12 on random data, the parity suite holds the sums and poses to the oracle's, which keeps the literal form); nothing here
// takes part in a decision - every term is rounded to 2^-40 right afterwards - so the multiply-adds are fused.
// One function for every pass kernel: their terms are the same doubles.
__device__ __forceinline__ void correspondence_terms(const PassBasis &B, double sx, double sy, double qx, double qy, double qz, double tx, double ty, double tz,
                                                     double (&term)[5]) {
    const double rx = qx - tx, ry = qy - ty, rz = qz - tz;  // residual = T*source - target (Registration.cpp:88)
    const double a = fma(B.c0x, rx, fma(B.c0y, ry, B.c0z * rz));
    const double b = fma(B.c1x, rx, fma(B.c1y, ry, B.c1z * rz));
    term[0] = -sy;
    term[1] = fma(sx, sx, sy * sy);
    term[2] = a;
    term[3] = fma(sx, b, -(sy * a));
    term[4] = fma(rx, rx, fma(ry, ry, rz * rz));
}
__device__ __forceinline__ void accumulate(Acc &a, const PassBasis &B, double sx, double sy, double qx, double qy, double qz, double tx, double ty, double tz) {
    double term[5];
    correspondence_terms(B, sx, sy, qx, qy, qz, tx, ty, tz, term);
#pragma unroll
    for (int k = 0; k < kTermLimbs; ++k) a.limb[k] = B.jtj00[k];
#pragma unroll
    for (int i = 0; i < 5; ++i) to_fixed(term[i], a.limb + (i + 1) * kTermLimbs, a.range_error);
    a.limb[6 * kTermLimbs + 1] = 1 << 19;  // the count: 1.0 = 2^40 = 2^19 * 2^21 (the other limbs stay 0)
}

// (The solve + pose update - Registration.cpp:119-125, 159-167, 181-184 - is the host's: kicp_reg_internal.hpp HostLoop::step.  The device-side
//  twin that round 1 to 5 carried - the launch's last workgroup solving on one lane - lost every A/B and went in round 6.)
// Workgroup epilogue of every pass kernel: exact workgroup sum, then a last-arriver tree.  Default (mode 4): ONE level -
// the last workgroup of every group of kGroup sends the group's row, tagged, to the host.  Other modes: two levels, the
// last workgroup of the launch finishes the iteration.  Inter-workgroup traffic follows cdna_hip_programming.md
// Guideline 16 (form R1): payload as 8-byte write-through (sc1) stores, ONE relaxed agent-scope ticket atomic per
// workgroup, readers use sc1 loads; no fences, no same-line atomic fan-in (tickets live on separate 128-B lines).  The
// two-level modes order payload before ticket with `s_waitcnt vmcnt(0)`; mode 4 needs no ordering (tags).
constexpr int kGroup = 32;        // workgroups per first-level group
constexpr int kTicketStride = 32; // uint32 words between group tickets (128 B)
constexpr int kAccStride = 32;                // 64-bit words between the groups' counting accumulators (256 B: one memory channel)
constexpr int kAccCountShift = 56;            // a word of an accumulator: contributions so far << 56 | sum of the biased words
constexpr long long kAccBias = 1ll << 41;     // makes a row word (|value| < 2^40) non-negative; 32 of them stay below 2^47

__device__ __forceinline__ unsigned long long ld_sc1(const unsigned long long *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_sc1(unsigned long long *p, unsigned long long v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// lanes 0..47 cooperatively sum `count` <= kGroup rows of kReduceWords words; the totals end up in lanes 0..23.
// All loads of a lane are issued before the first is consumed (one round trip, not count/2 of them).
__device__ __forceinline__ long long sum_rows(const unsigned long long *rows, uint32_t count, int lane) {
    long long v = 0;
    if (lane < 2 * kReduceWords) {
        const int word = lane % kReduceWords;
        const uint32_t first = lane / kReduceWords;
        unsigned long long t[kGroup / 2];
#pragma unroll
        for (int u = 0; u < kGroup / 2; ++u) {
            const uint32_t j = first + 2 * u;
            t[u] = (j < count) ? ld_sc1(rows + static_cast<size_t>(j) * kReduceWords + word) : 0ull;
        }
#pragma unroll
        for (int u = 0; u < kGroup / 2; ++u) v += static_cast<long long>(t[u]);
    }
    return v + __shfl_down(v, kReduceWords, 64);
}

// mode 4: the same fold over tagged rows (word = value << 16 | tag).  Returns the sum of the values; `ok` tells whether
// every word carried `tag`, i.e. whether every row had landed (the caller retries otherwise).
__device__ __forceinline__ long long sum_rows_tagged(const unsigned long long *rows, uint32_t count, int lane, uint32_t tag, bool &ok) {
    long long v = 0;
    ok = true;
    if (lane < 2 * kReduceWords) {
        const int word = lane % kReduceWords;
        const uint32_t first = lane / kReduceWords;
        unsigned long long t[kGroup / 2];
#pragma unroll
        for (int u = 0; u < kGroup / 2; ++u) {
            const uint32_t j = first + 2 * u;
            t[u] = (j < count) ? ld_sc1(rows + static_cast<size_t>(j) * kReduceWords + word) : static_cast<unsigned long long>(tag);
        }
#pragma unroll
        for (int u = 0; u < kGroup / 2; ++u) {
            ok = ok && (static_cast<uint32_t>(t[u]) & 0xFFFFu) == tag;
            v += static_cast<long long>(t[u]) >> 16;
        }
    }
    return v + __shfl_down(v, kReduceWords, 64);
}

// mode 6, the last workgroup of group 0 (wave 0): add every rank's group rows out of this rank's mailbox.  Lanes r < nranks first
// learn rank r's group count (word kP2pCountWord of its row 0); then all rows are fetched the way sum_rows_tagged does it - 48
// lanes, 16 loads each in flight, 32 rows per round over the flattened (rank, group) index - and re-fetched until every word
// carries the tag.  Returns the node-wide totals in lanes 0..23; lane kNumLimbs + 1 is set to 1 when a row did not arrive in time.
__device__ __forceinline__ long long p2p_collect_rows(const SolveParams &f, int lane) {
    const uint32_t nr = static_cast<uint32_t>(f.p2p_nranks), tag = f.p2p_tag;
    const unsigned long long *box = f.p2p_peers[f.p2p_rank] + p2p_rows_offset(f.p2p_nranks) +
                                    static_cast<size_t>(f.p2p_parity) * nr * kP2pMaxGroups * kReduceWords;
    const long long t0 = wall_clock64();
    int late = 0;
    uint32_t count = 0;
    if (static_cast<uint32_t>(lane) < nr) {
        for (;;) {
            const unsigned long long got = __hip_atomic_load(box + static_cast<size_t>(lane) * kP2pMaxGroups * kReduceWords + kP2pCountWord, __ATOMIC_RELAXED,
                                                             __HIP_MEMORY_SCOPE_SYSTEM);
            if ((static_cast<uint32_t>(got) & 0xFFFFu) == tag) {
                count = min(static_cast<uint32_t>(got >> 16), static_cast<uint32_t>(kP2pMaxGroups));
                break;
            }
            if (wall_clock64() - t0 > f.p2p_timeout_ticks) {
                late = 1;
                break;
            }
            __builtin_amdgcn_s_sleep(2);
        }
    }
    late = __any(late) ? 1 : 0;
    uint32_t widest = count;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) widest = max(widest, static_cast<uint32_t>(__shfl_xor(static_cast<int>(widest), off, 64)));
    long long sum = 0;
    const int word = lane % kReduceWords;
    const uint32_t phase = static_cast<uint32_t>(lane / kReduceWords);  // lanes 0..23 take the even rows of a round, 24..47 the odd ones
    for (uint32_t v0 = 0; v0 < nr * widest && !late; v0 += kGroup) {
        for (;;) {
            bool ok = true;
            long long part = 0;
            unsigned long long t[kGroup / 2];
#pragma unroll
            for (int u = 0; u < kGroup / 2; ++u) {
                const uint32_t v = v0 + phase + 2 * u, r = v / widest, j = v % widest;  // (widest >= 1 here)
                const uint32_t rows_of_r = static_cast<uint32_t>(__shfl(static_cast<int>(count), static_cast<int>(min(r, nr - 1u)), 64));  // (every lane takes part)
                const bool want = lane < 2 * kReduceWords && r < nr && j < rows_of_r;
                t[u] = want ? __hip_atomic_load(box + (static_cast<size_t>(r) * kP2pMaxGroups + j) * kReduceWords + word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)
                            : static_cast<unsigned long long>(tag);
            }
#pragma unroll
            for (int u = 0; u < kGroup / 2; ++u) {
                ok = ok && (static_cast<uint32_t>(t[u]) & 0xFFFFu) == tag;
                part += static_cast<long long>(t[u]) >> 16;
            }
            if (__all(ok)) {
                sum += part;
                break;
            }
            if (wall_clock64() - t0 > f.p2p_timeout_ticks) {
                late = 1;
                break;
            }
            __builtin_amdgcn_s_sleep(2);
        }
    }
    sum += __shfl_down(sum, kReduceWords, 64);
    if (lane == kP2pCountWord) sum = 0;  // (the counts are not part of the payload)
    if (lane == kNumLimbs + 1) sum = late;
    return sum;
}

// Sum of a 32-bit value over the wave with DPP adds only (no LDS crossbar): inclusive scan inside each row of 16 lanes
// (row_shr 1, 2, 4, 8; lanes shifted in from outside the row read 0), then lane 15 of row 0 / 2 is added to every lane of
// row 1 / 3 (row_bcast:15) and lane 31 to rows 2 and 3 (row_bcast:31).  LANE 63 holds the wave total.
__device__ __forceinline__ int wave_sum_to_lane63(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, true);
    return v;
}

// `row_tag` (tagged-row modes): the tag of this pass - p.sol.tag for a kernel that runs one pass, tag0 + pass for the resident
// kernel.  `gave_up` (resident kernel): this workgroup saw no command in time and hands over empty sums; the group row's flag
// word counts such workgroups in its upper half (kGaveUpUnit each) so that the host repeats the pass in a fresh launch.
constexpr unsigned long long kGaveUpUnit = 1ull << 16;
// A group's reader gives a row this long to land (100 MHz ticks = 2 s: rows arrive within microseconds of the ticket that
// announced them; anything longer is a lost workgroup).  It then hands over what it has, marked kLostRowUnit in the flag word, and
// the host repeats the pass (resident kernel) or fails the call - instead of a wave spinning on the device for ever.
constexpr unsigned long long kLostRowUnit = 1ull << 8;
constexpr long long kRowWaitTicks = 200000000ll;
// The hand-over of a workgroup's exact sums through its group's COUNTING ACCUMULATORS (wave 0; lanes 0..6 hold the seven 128-bit
// totals `t`): ONE round trip to the L2 and no reader.  Lane w < kReduceWords adds word w of the workgroup's row - biased to be
// non-negative, with a 1 in the count field above it - to word w of the group's accumulator.  The addition that finds the count at
// group size - 1 is the last one for that word: old value + own = the group's sum, which that lane hands to the host (and clears
// the word for the set's next turn).  Every word is completed by whichever workgroup happened to add to it last - not necessarily
// the same one for all 24 - and carries the pass tag, which is how the host tells a complete row anyway (wait_rows).
// `acc_set` / `row_set`: which of the launch's sets of accumulators / of host rows the pass uses (resident kernels: the pass's slot
// for both; an ordinary launch: the tag's parity for the accumulators, row set 0).  Round 4: the resident generic kernel; round 5:
// every launch whose group rows go to the host, and the small-scan kernels (kicp_small.hpp), whose every workgroup used to send a
// row of its own across PCIe.
__device__ __forceinline__ void counting_hand_over(const I128 &t, int range_error, int gave_up, const PassParams &p, uint32_t row_tag, uint32_t acc_set,
                                                   uint32_t row_set, int lane) {
    const uint32_t nblocks = gridDim.x, b = blockIdx.x, g = b / kGroup, ngroups = (nblocks + kGroup - 1) / kGroup;
    long long l[3] = {0ll, 0ll, 0ll};
    if (lane < kNumSums) i128_to_limbs(t, l);
    const int from = min(lane, kNumLimbs - 1) / 3;
    const long long a0 = __shfl(l[0], from, 64), a1 = __shfl(l[1], from, 64), a2 = __shfl(l[2], from, 64);
    long long word = lane % 3 == 0 ? a0 : (lane % 3 == 1 ? a1 : a2);
    if (lane >= kNumLimbs) word = lane == kNumLimbs ? static_cast<long long>(range_error) + (gave_up ? static_cast<long long>(kGaveUpUnit) : 0ll) : 0ll;
    if (lane < kReduceWords) {
        const uint32_t group_size = min(static_cast<uint32_t>(kGroup), nblocks - g * kGroup);
        unsigned long long *acc = p.group_acc + (static_cast<size_t>(acc_set) * ngroups + g) * kAccStride + lane;
        const unsigned long long mine = static_cast<unsigned long long>(word + kAccBias);  // |word| < 2^40: (0, 2^42)
        const unsigned long long old = __hip_atomic_fetch_add(acc, (1ull << kAccCountShift) + mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((old >> kAccCountShift) == group_size - 1u) {
            const long long total = static_cast<long long>((old & ((1ull << kAccCountShift) - 1ull)) + mine) - static_cast<long long>(group_size) * kAccBias;
            __hip_atomic_store(acc, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(p.sol.pub_rows + (static_cast<size_t>(row_set) * ngroups + g) * kReduceWords + lane,
                               (static_cast<unsigned long long>(total) << 16) | static_cast<unsigned long long>(row_tag), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// ROWS_ONLY: the caller only ever runs mode 4 (the resident kernel): none of the other hand-offs is compiled in.
// `parity` (resident kernel: pass & 1): workgroup rows, tickets and the groups' host rows are double-buffered by pass parity, so
// that what a workgroup writes for pass k + 1 - in particular the marked empty row of a workgroup that gave up waiting for the
// command of pass k + 1 - can never land on a row of pass k that its reader (the group's last workgroup, the host) has not
// consumed yet.  The buffer of parity (k + 1) & 1 last held pass k - 1, whose rows the host had added before it sent the command
// that started pass k.
template <int BLOCK, bool ROWS_ONLY = false>
__device__ __forceinline__ void finish_pass(Acc &a, const PassParams &p, int (*s_red)[kWaveLimbs], int *s_flag, uint32_t row_tag, int gave_up = 0,
                                            uint32_t parity = 0u) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    IcpState *st = p.st;
    if (!ROWS_ONLY && dbg_is(p, 8)) {  // ablation (tools/gpu_dbg.py): no reduction at all, workgroup 0 hands over zeros
        if (p.sol.mode == 4 && wave == 0 && blockIdx.x % kGroup == 0 && lane < kReduceWords)
            __hip_atomic_store(p.sol.pub_rows + static_cast<size_t>(blockIdx.x / kGroup) * kReduceWords + lane, static_cast<unsigned long long>(p.sol.tag),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (blockIdx.x == 0 && wave == 0 && p.sol.mode == 2) {
            if (lane < kReduceWords) __hip_atomic_store(p.sol.pub_words + lane, 0ll, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) __hip_atomic_store(p.sol.pub_seq, p.sol.pub_value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        return;
    }
    // Every lane contributes at most ONE correspondence per pass, so its seven sums are single terms (to_fixed): four limbs
    // of 21 bits each, and a wave's 64 values of a limb add up in int32.
    int range_error = a.range_error;
    int limb[kWaveLimbs];
    if (dbg_is(p, 10) || dbg_is(p, 12)) {  // (10, the census: the count's limbs carry something else; 12: the in-process A/B switch of what follows)
#pragma unroll
        for (int k = 0; k < kWaveLimbs; ++k) limb[k] = wave_sum_to_lane63(a.limb[k]);
    } else {
        // Two of the seven sums need no reduction.  The count: every correspondence adds 2^40 = limb 1 at 2^19, so the wave's sum is
        // the number of lanes that hold one.  |J.col(0)|^2 = |R UnitX|^2 depends on the pose alone: every correspondence adds the SAME
        // four limbs, so the wave's sum is that number times the limbs of any lane that holds one.  (Sums of identical integers:
        // exactly what the additions would give.)
        const unsigned long long holders = __ballot(a.limb[6 * kTermLimbs + 1] != 0);
        const int n_holders = __popcll(holders);
        const int some = holders ? static_cast<int>(__ffsll(static_cast<long long>(holders))) - 1 : 0;
#pragma unroll
        for (int k = 0; k < kTermLimbs; ++k) limb[k] = n_holders * __builtin_amdgcn_readlane(a.limb[k], some);
        limb[6 * kTermLimbs] = 0, limb[6 * kTermLimbs + 1] = n_holders << 19, limb[6 * kTermLimbs + 2] = 0, limb[6 * kTermLimbs + 3] = 0;
        // The other twenty limbs: a reduction that HALVES the number of values a lane carries with every step it can - the lanes of a
        // pair keep one half each and take the partner's share of it (quad_perm), twice; then the quads of a row (row_shr 4, 8) and the
        // four rows (ds_bpermute) are added for the five values a lane is left with.  ~65 instead of 120 DPP additions; lane 60 + q
        // ends up with the wave's sums of limbs 4 j + q (j = 0 .. 4).
        int w10[10], x5[5];
        const bool p0 = (lane & 1) != 0, p1 = (lane & 2) != 0;
#pragma unroll
        for (int j = 0; j < 10; ++j) {
            const int lo = a.limb[kTermLimbs + 2 * j], hi = a.limb[kTermLimbs + 2 * j + 1];
            w10[j] = (p0 ? hi : lo) + __builtin_amdgcn_update_dpp(0, p0 ? lo : hi, 0xB1, 0xF, 0xF, true);  // quad_perm [1, 0, 3, 2]
        }
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const int lo = w10[2 * j], hi = w10[2 * j + 1];
            x5[j] = (p1 ? hi : lo) + __builtin_amdgcn_update_dpp(0, p1 ? lo : hi, 0x4E, 0xF, 0xF, true);  // quad_perm [2, 3, 0, 1]
        }
#pragma unroll
        for (int j = 0; j < 5; ++j) x5[j] += __builtin_amdgcn_update_dpp(0, x5[j], 0x114, 0xF, 0xF, true);  // row_shr 4
#pragma unroll
        for (int j = 0; j < 5; ++j) x5[j] += __builtin_amdgcn_update_dpp(0, x5[j], 0x118, 0xF, 0xF, true);  // row_shr 8: lanes 12 .. 15 of a row hold its sums
#pragma unroll
        for (int j = 0; j < 5; ++j) x5[j] += __shfl_up(x5[j], 16, 64);
#pragma unroll
        for (int j = 0; j < 5; ++j) x5[j] += __shfl_up(x5[j], 32, 64);
        range_error = __any(range_error) ? 1 : 0;
        if (lane >= 60) {
#pragma unroll
            for (int j = 0; j < 5; ++j) s_red[wave][kTermLimbs + 4 * j + (lane & 3)] = x5[j];
        }
        if (lane == 63) {
#pragma unroll
            for (int k = 0; k < kTermLimbs; ++k) s_red[wave][k] = limb[k], s_red[wave][6 * kTermLimbs + k] = limb[6 * kTermLimbs + k];
            if (range_error) atomicOr(s_flag, 2);
        }
    }
    if (dbg_is(p, 10) || dbg_is(p, 12)) {
        range_error = __any(range_error) ? 1 : 0;
        if (lane == 63) {
#pragma unroll
            for (int k = 0; k < kWaveLimbs; ++k) s_red[wave][k] = limb[k];
            if (range_error) atomicOr(s_flag, 2);
        }
    }
    __syncthreads();
    if (wave != 0) return;
    range_error = (*s_flag & 2) ? 1 : 0;
    // wave 0 only from here.  Lane i < 7 takes the workgroup total of sum i (as a 128-bit integer) and publishes its three
    // 40-bit limbs.
    I128 t{0ull, 0ll};
    if (lane < kNumSums) {
        for (int w = 0; w < BLOCK / 64; ++w)
            i128_add_limb_sums(t, s_red[w][kTermLimbs * lane], s_red[w][kTermLimbs * lane + 1], s_red[w][kTermLimbs * lane + 2], s_red[w][kTermLimbs * lane + 3]);
    }
    const uint32_t nblocks = gridDim.x, b = blockIdx.x, g = b / kGroup, ngroups = (nblocks + kGroup - 1) / kGroup;
    // (an ordinary launch in mode 4 - round 5: the accumulators alternate with the pass tag's parity, the host's row of group g stays
    //  where it was; dbg 14: round 4's rows -> ticket -> reload, for the in-process A/B)
    const bool counting = ROWS_ONLY || (p.sol.mode == 4 && dbg_not(p, 14));
    if (!ROWS_ONLY && counting) parity = row_tag & 1u;
    if (counting) {
        counting_hand_over(t, range_error, gave_up, p, row_tag, parity, ROWS_ONLY ? parity : 0u, lane);
        return;
    }
    unsigned long long *const rows0 = p.partials + (ROWS_ONLY ? static_cast<size_t>(parity) * nblocks * kReduceWords : 0u);
    unsigned int *const tickets0 = p.tickets + (ROWS_ONLY ? static_cast<size_t>(parity) * ngroups * kTicketStride : 0u);
    unsigned long long *row = rows0 + static_cast<size_t>(b) * kReduceWords;
    if (ROWS_ONLY || p.sol.mode == 4 || p.sol.mode == 6) {
        // Tagged rows: no store acknowledgement is awaited anywhere.  The ticket only elects the group's reader; whether
        // a row has landed is visible in the row itself.  Values: limbs < 2^40 (the top limb is a small signed number
        // within the documented range), so value << 16 | tag fits a word, and so does the sum of a group's 32 rows.
        const unsigned long long tag = row_tag;
        if (lane < kNumSums) {
            long long l[3];
            i128_to_limbs(t, l);
#pragma unroll
            for (int j = 0; j < 3; ++j) st_sc1(row + 3 * lane + j, (static_cast<unsigned long long>(l[j]) << 16) | tag);
        } else if (lane < kNumSums + 3) {
            st_sc1(row + kNumLimbs + (lane - kNumSums),
                   ((lane == kNumSums ? static_cast<unsigned long long>(range_error) + (gave_up ? kGaveUpUnit : 0ull) : 0ull) << 16) | tag);
        }
        unsigned int ticket = 0;
        if (lane == 0) ticket = __hip_atomic_fetch_add(tickets0 + static_cast<size_t>(g) * kTicketStride, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ticket = __shfl(ticket, 0, 64);
        const uint32_t group_size = min(static_cast<uint32_t>(kGroup), nblocks - g * kGroup);
        if (ticket != group_size - 1) return;
        if (lane == 0) __hip_atomic_store(tickets0 + static_cast<size_t>(g) * kTicketStride, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        long long total;
        bool ok;
        total = sum_rows_tagged(rows0 + static_cast<size_t>(g) * kGroup * kReduceWords, group_size, lane, static_cast<uint32_t>(tag), ok);
        if (!__all(ok)) {  // (rare: a row still on its way) re-read, bounded by wall-clock time
            const long long t0 = wall_clock64();
            bool lost = false;
            do {
                __builtin_amdgcn_s_sleep(1);
                total = sum_rows_tagged(rows0 + static_cast<size_t>(g) * kGroup * kReduceWords, group_size, lane, static_cast<uint32_t>(tag), ok);
                lost = wall_clock64() - t0 > kRowWaitTicks;
            } while (!__all(ok) && !lost);
            if (!__all(ok)) total = lane == kNumLimbs ? static_cast<long long>(kLostRowUnit) : 0ll;  // nothing of an incomplete sum is handed on
        }
        if (ROWS_ONLY || p.sol.mode == 4) {
            if (lane < kReduceWords)
                __hip_atomic_store(p.sol.pub_rows + (static_cast<size_t>(ROWS_ONLY ? parity * ngroups : 0u) + g) * kReduceWords + lane,
                                   (static_cast<unsigned long long>(total) << 16) | tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            return;
        }
        // mode 6: the group's row goes into EVERY rank's mailbox (over xGMI for the peers'), tagged with the step
        const SolveParams &f = p.sol;
        if (lane < kReduceWords) {
            const unsigned long long value = lane == kP2pCountWord ? static_cast<unsigned long long>(ngroups) : static_cast<unsigned long long>(total);
            const unsigned long long w = (value << 16) | f.p2p_tag;
            const size_t at = p2p_rows_offset(f.p2p_nranks) + ((static_cast<size_t>(f.p2p_parity) * f.p2p_nranks + f.p2p_rank) * kP2pMaxGroups + g) * kReduceWords + lane;
            for (int r = 0; r < f.p2p_nranks; ++r) __hip_atomic_store(f.p2p_peers[r] + at, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        if (g != 0) return;
        // group 0's last workgroup collects for this rank and hands the node-wide totals to its host (as mode 5 does)
        const long long all = p2p_collect_rows(f, lane);
        if (lane < kReduceWords) __hip_atomic_store(f.pub_words + lane, all, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_store(f.pub_seq, f.pub_value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        return;
    }
    if (lane < kNumSums) {
        long long l[3];
        i128_to_limbs(t, l);
#pragma unroll
        for (int j = 0; j < 3; ++j) st_sc1(row + 3 * lane + j, static_cast<unsigned long long>(l[j]));
    } else if (lane < kNumSums + 3) {
        st_sc1(row + kNumLimbs + (lane - kNumSums), lane == kNumSums ? static_cast<unsigned long long>(range_error) : 0ull);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned int ticket = 0;
    if (lane == 0) ticket = __hip_atomic_fetch_add(p.tickets + static_cast<size_t>(g) * kTicketStride, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    ticket = __shfl(ticket, 0, 64);
    const uint32_t group_size = min(static_cast<uint32_t>(kGroup), nblocks - g * kGroup);
    if (ticket != group_size - 1) return;
    // ---- last workgroup of its group: fold the group's rows into one ---------------------------------------
    if (lane == 0) __hip_atomic_store(p.tickets + static_cast<size_t>(g) * kTicketStride, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    long long total = sum_rows(p.partials + static_cast<size_t>(g) * kGroup * kReduceWords, group_size, lane);
    if (ngroups > 1) {
        unsigned long long *grow = p.partials + (static_cast<size_t>(nblocks) + g) * kReduceWords;
        if (lane < kReduceWords) st_sc1(grow + lane, static_cast<unsigned long long>(total));
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) ticket = __hip_atomic_fetch_add(&st->ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ticket = __shfl(ticket, 0, 64);
        if (ticket != ngroups - 1) return;
        if (lane == 0) __hip_atomic_store(&st->ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        total = 0;
        for (uint32_t base = 0; base < ngroups; base += kGroup)
            total += sum_rows(p.partials + (static_cast<size_t>(nblocks) + base) * kReduceWords, min(static_cast<uint32_t>(kGroup), ngroups - base), lane);
    }
    // ---- last workgroup of the launch ------------------------------------------------------------------------
    if (p.sol.mode == 7) {
        // Group rows are what the ranks exchange (mode 6), but this launch has more groups than a rank's share of the mailbox
        // holds: its total travels as ONE row.  A row's words carry 48 bits, so the limb sums are brought back into limb range
        // first (lane 3i + j: limb j of sum i).
        const SolveParams &f = p.sol;
        const int i3 = 3 * (min(lane, kNumLimbs - 1) / 3);
        const unsigned long long a = static_cast<unsigned long long>(__shfl(total, i3, 64)), b = static_cast<unsigned long long>(__shfl(total, i3 + 1, 64));
        const long long c = __shfl(total, i3 + 2, 64);
        I128 t{a, 0ll};
        i128_add(t, I128{b << 40, static_cast<long long>(b >> 24)});  // (the two lower limb sums are non-negative)
        t.hi += c << 16;
        long long l[3];
        i128_to_limbs(t, l);
        const int j = lane % 3;
        long long word = j == 0 ? l[0] : (j == 1 ? l[1] : l[2]);
        if (lane >= kNumLimbs) word = lane == kNumLimbs ? (total != 0 ? 1 : 0) : 0;  // the range flag as 0 / 1
        if (lane < kReduceWords) {
            const unsigned long long value = lane == kP2pCountWord ? 1ull : static_cast<unsigned long long>(word);
            const unsigned long long w = (value << 16) | f.p2p_tag;
            const size_t at = p2p_rows_offset(f.p2p_nranks) + ((static_cast<size_t>(f.p2p_parity) * f.p2p_nranks + f.p2p_rank) * kP2pMaxGroups) * kReduceWords + lane;
            for (int r = 0; r < f.p2p_nranks; ++r) __hip_atomic_store(f.p2p_peers[r] + at, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        total = p2p_collect_rows(f, lane);
    }
    if (lane < kReduceWords) st->reduce[lane] = total;
    if (p.sol.mode == 7) {  // hand the totals to the host: write-through stores, one wait, then the sequence word
        if (lane < kReduceWords) __hip_atomic_store(p.sol.pub_words + lane, total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_store(p.sol.pub_seq, p.sol.pub_value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// wave-uniform values belong in SGPRs: tell the compiler explicitly
__device__ __forceinline__ int uniform_i(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ double uniform_d(double v) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readfirstlane(static_cast<int>(b)), hi = __builtin_amdgcn_readfirstlane(static_cast<int>(b >> 32));
    return __longlong_as_double((static_cast<long long>(hi) << 32) | static_cast<unsigned int>(lo));
}
__device__ __forceinline__ Pose load_pose(const PassParams &p) {
    if (p.sol.pass == 0 || p.sol.mode >= 2) return p.sol.pose0;
    const Pose T = p.st->T;
    return Pose{uniform_d(T.qx), uniform_d(T.qy), uniform_d(T.qz), uniform_d(T.qw), uniform_d(T.tx), uniform_d(T.ty), uniform_d(T.tz)};
}
#define KICP_PASS_SHARED(BLOCK)                       \
    __shared__ int s_red[(BLOCK) / 64][kWaveLimbs];   \
    __shared__ int s_flag;                            \
    if (threadIdx.x == 0) s_flag = 0;

// ------------------------------------------------------------------------------------------------------------
// variant 3: thread-per-query gather over the 16-bit mirror, per-lane work lists
//   * candidates are read from the compact mirror (kicp_common.hpp::MirrorPoint: 8 bytes per point, offsets from the voxel
//     corner in units of voxel_size / 65536; two points per 16-byte load) and compared in fp32 IN THOSE UNITS with packed
//     fp32 arithmetic (two points per instruction); the three smallest squared distances are tracked with the
//     indices / visiting order of the two smallest.  The mirror only PRE-SELECTS: the winner (and the runner-up when it
//     lies within the error margin) is re-evaluated in fp64 from the fp64 pool, ties resolved by the reference's visiting
//     order; if even the third smallest is within the margin the lane falls back to the exact fp64 search.  The chosen
//     neighbour and its distance are therefore exactly the fp64 reference's.
//   * each lane walks ITS OWN list of neighbour voxels (bit mask over the 27 shifts, in the reference's order): culled
//     voxels cost ALU only, so a wave iterates max-over-lanes(#voxels actually visited) times instead of over the union
//     of the lanes' shifts; a bucket is scanned 20 points per trip (ten 16-byte loads in flight).
// ------------------------------------------------------------------------------------------------------------
constexpr int kTrip = kMirrorTrip;  // bucket points in flight per lane and trip (even: two points per load)
typedef float v2f __attribute__((ext_vector_type(2)));
struct Best3 {
    float b1, b2, b3;
    uint32_t i1, i2, o1, o2;  // pool index and visiting order (shift * kOrdStride + k) of the two smallest
};
__device__ __forceinline__ void best3_update(Best3 &t, float d, uint32_t idx, uint32_t ord) {
    const bool lt1 = d < t.b1, lt2 = d < t.b2, lt3 = d < t.b3;
    t.b3 = lt2 ? t.b2 : (lt3 ? d : t.b3);
    t.b2 = lt1 ? t.b1 : (lt2 ? d : t.b2);
    t.i2 = lt1 ? t.i1 : (lt2 ? idx : t.i2);
    t.o2 = lt1 ? t.o1 : (lt2 ? ord : t.o2);
    t.b1 = lt1 ? d : t.b1;
    t.i1 = lt1 ? idx : t.i1;
    t.o1 = lt1 ? ord : t.o1;
}

// minimum of N register-resident unsigned keys (v_min3_u32 friendly reduction)
constexpr uint32_t kFarKey = 0x7F000000u;  // 1.7e38 as a float: beyond every real squared distance
template <int N>
__device__ __forceinline__ uint32_t tree_min_u32(const uint32_t (&v)[N]) {
    uint32_t a[N];
#pragma unroll
    for (int u = 0; u < N; ++u) a[u] = v[u];
#pragma unroll
    for (int width = N; width > 1; width = (width + 2) / 3) {
#pragma unroll
        for (int u = 0; u < (width + 2) / 3; ++u) {
            const int i0 = 3 * u, i1 = min(3 * u + 1, width - 1), i2 = min(3 * u + 2, width - 1);
            a[u] = min(a[i0], min(a[i1], a[i2]));
        }
    }
    return a[0];
}

// minimum AND runner-up of N register-resident unsigned keys in one tournament: every node carries (min, second).  For three
// nodes a, b, c:  min = min3(a.m, b.m, c.m);  second = min(med3(a.m, b.m, c.m), min3(a.s, b.s, c.s)) - the seconds of the two
// losing nodes are never smaller than the median of the minima, so they may take part unharmed (no selects).  20 keys: 26
// instructions, against 11 + 20 + 11 for a minimum, a re-keying subtraction and a second minimum.
__device__ __forceinline__ uint32_t umed3(uint32_t a, uint32_t b, uint32_t c) { return max(min(a, b), min(max(a, b), c)); }  // -> v_med3_u32
template <int N>
__device__ __forceinline__ void tree_min2_u32(const uint32_t (&v)[N], uint32_t &first, uint32_t &second) {
    constexpr int L1 = (N + 2) / 3;
    uint32_t m[L1], s[L1];
#pragma unroll
    for (int u = 0; u < L1; ++u) {  // leaves: triples (the last node may hold one or two keys)
        const int i0 = 3 * u, i1 = 3 * u + 1, i2 = 3 * u + 2;
        if (i2 < N) m[u] = min(v[i0], min(v[i1], v[i2])), s[u] = umed3(v[i0], v[i1], v[i2]);
        else if (i1 < N) m[u] = min(v[i0], v[i1]), s[u] = max(v[i0], v[i1]);
        else m[u] = v[i0], s[u] = 0xFFFFFFFFu;
    }
#pragma unroll
    for (int width = L1; width > 1; width = (width + 2) / 3) {
#pragma unroll
        for (int u = 0; u < (width + 2) / 3; ++u) {
            const int i0 = 3 * u, i1 = 3 * u + 1, i2 = 3 * u + 2;
            if (i2 < width) {
                const uint32_t mm = min(m[i0], min(m[i1], m[i2])), md = umed3(m[i0], m[i1], m[i2]), ss = min(s[i0], min(s[i1], s[i2]));
                m[u] = mm, s[u] = min(md, ss);
            } else if (i1 < width) {
                const uint32_t mm = min(m[i0], m[i1]), mx = max(m[i0], m[i1]), ss = min(s[i0], s[i1]);
                m[u] = mm, s[u] = min(mx, ss);
            } else {
                m[u] = m[i0], s[u] = s[i0];
            }
        }
    }
    first = m[0], second = s[0];
}

// merge the three-smallest record of another lane into `t`
__device__ __forceinline__ void best3_merge(Best3 &t, const Best3 &o) {
    best3_update(t, o.b1, o.i1, o.o1);
    best3_update(t, o.b2, o.i2, o.o2);
    t.b3 = fminf(t.b3, o.b3);  // o.b3 can never undercut the runner-up of a set that already holds o.b1 <= o.b2
}

// PointToVoxel component: static_cast<int>(floor(c / vs)) as the reference evaluates it (kiss-icp v1.2.0 core/VoxelUtils.hpp),
// without paying an fp64 division for every coordinate: t = c * (1 / vs) agrees with fl(c / vs) to a few ulps, so the two
// floors can only differ when t lies within that distance of an integer - and only then is the division carried out.
__device__ __forceinline__ int32_t voxel_coord(double c, double vs, double inv_vs) {
    const double t = c * inv_vs;
    const double f = floor(t);
    const double frac = t - f;
    const double guard = 4.0e-16 * fabs(t) + 1.0e-300;
    if (__builtin_expect(frac < guard || 1.0 - frac < guard, 0)) return static_cast<int32_t>(floor(c / vs));
    return static_cast<int32_t>(f);
}

// What a bucket visit needs to know about the query it serves (a lane may visit on behalf of another lane's query).
struct Probe {
    float lx, ly, lz;  // the query's offset inside its own voxel, mirror units
    uint32_t slot0;    // table slot of the own voxel's entry (its record lists the 27 neighbours' buckets)
};
// One query's search state.
struct Lane {
    uint32_t i;  // query index, kNoIndex32 = none
    Probe q;
    uint32_t todo;  // neighbour voxels still to visit (bit s = shift s of the reference's order)
    Best3 t;
};
constexpr float kCell = 65536.f;  // one voxel in mirror units

// the transformed point T * source[i] and its voxel (PointToVoxel); recomputed where needed rather than kept in registers
// (`kept`: the latency-oriented build has the registers to keep the query and the two source coordinates the Jacobian needs
// through the search, and so starts its exact phase without re-reading the source point: -0.5 us per cfg2 scan)
struct KeptQuery {
    Query q;
    double sx, sy;
    bool voxel;  // q.vx .. q.vz are valid (a query parked in LDS comes back without them: only the rare exact search needs them)
};
// (`src`: the scan's points - p.src, or the scan a resident kernel was told to take up next, kicp_small.hpp)
__device__ __forceinline__ void make_query_of(Query &q, const PassParams &p, const double *__restrict__ src, const Pose &T, uint32_t i, KeptQuery *kept = nullptr) {
    const double sx = src[3 * i], sy = src[3 * i + 1], sz = src[3 * i + 2];
    if (kept) kept->sx = sx, kept->sy = sy;
    double rx, ry, rz;
    quat_rotate(T, sx, sy, sz, rx, ry, rz);
    q.x = rx + T.tx, q.y = ry + T.ty, q.z = rz + T.tz;
    const double vs = p.map.voxel_size;
    q.vx = voxel_coord(q.x, vs, p.search.inv_vs), q.vy = voxel_coord(q.y, vs, p.search.inv_vs), q.vz = voxel_coord(q.z, vs, p.search.inv_vs);
}
// Everything the search does is in MIRROR UNITS (voxel_size / 65536) until the exact phase.  Error model (units): a mirror
// coordinate is within 1.0 of the true offset (rounding 0.5; 1.0 where the top of the range is clamped), the query offset and
// the difference add < 0.05 (fp32 roundings of numbers < 2^18), so a squared distance D is off by <= 2 sqrt(3 D) 1.05 + 3.3,
// plus 4.2e-6 D for the fp32 squares / sums and the 5 mantissa bits dropped for the integer tournament.  sp.margin_u covers
// the errors of BOTH candidates of a decision with 10 % to spare, for D up to the acceptance bound (and never farther than
// the 27-voxel neighbourhood reaches, D <= 12 voxel sizes^2): computed on the host (search_params()).
__device__ __forceinline__ void start_lane(Lane &L, const PassParams &p, const double *__restrict__ src, const Pose &T, uint32_t i, bool valid, KeptQuery *kept = nullptr) {
    const SearchParams &sp = p.search;
    L.i = valid ? i : kNoIndex32;
    L.q.slot0 = 0u, L.todo = 0u;
    L.t = Best3{sp.bound_u, sp.bound_u, sp.bound_u, kNoIndex32, kNoIndex32, 0u, 0u};
    Query q;
    make_query_of(q, p, src, T, valid ? i : 0u, kept);
    if (kept) kept->q = q, kept->voxel = true;
    const double vs = p.map.voxel_size;
    L.q.lx = static_cast<float>((q.x - q.vx * vs) * sp.upm), L.q.ly = static_cast<float>((q.y - q.vy * vs) * sp.upm),
    L.q.lz = static_cast<float>((q.z - q.vz * vs) * sp.upm);
    // ONE probe at the own voxel: the occupancy mask of the 27 neighbours (bit s = shift s of the reference's order) and
    // the record of their buckets.  Only voxels that hold points are ever visited; empty space costs nothing.
    if (valid && dbg_not(p, 2)) table_lookup_entry(p.map, q.vx, q.vy, q.vz, L.q.slot0, L.todo);
    if (dbg_is(p, 3)) L.todo &= 1u;     // experiments: own voxel only
    if (dbg_is(p, 5)) L.todo &= 0x7Fu;  // own + faces
    if (dbg_is(p, 4)) L.todo = 0u;      // probe only, no bucket visit
}
// drop the neighbour voxels that cannot hold anything within the margin of `best` (units^2).  A voxel's lower bound is the sum of
// the squared distances to the faces crossed on the way to it (one term for the six face neighbours, two for the twelve edge
// neighbours, three for the eight corners); it is alive while  sum <= (best + 4 margin) 1.00001  - the slack covers the fp32
// roundings of the sum and, per face crossed, the margin by which a mirror distance may undercut the truth (round 4 subtracted the
// margin per face: this bound is the same for corners and a hair more generous for faces and edges).
// Branch-free, and without a compare-and-select per voxel: the 26 differences `limit - sum` are formed two per instruction (packed
// fp32), and their SIGN BITS are shifted into the mask one v_alignbit_b32 each, last shift first.  ~50 instructions per round
// where the compare / select / or form took 125.
__device__ __forceinline__ uint32_t push_sign(uint32_t mask, float d) { return __builtin_amdgcn_alignbit(mask, __float_as_uint(d), 31u); }  // (mask << 1) | (d < 0)
__device__ __forceinline__ uint32_t cull_todo(const Lane &L, float best, float margin) {
    const Probe &P = L.q;
    const v2f x = {kCell - P.lx, P.lx}, y = {kCell - P.ly, P.ly}, z = {kCell - P.lz, P.lz};  // distances to the {+, -} faces of the own voxel
    const v2f fx = x * x, fy = y * y, fz = z * z;
    const float lim = (best + 4.0f * margin) * 1.00001f;
    const v2f A = {lim, lim};
    // reference order (kShiftTable): 0 own | 1 +x 2 -x 3 +y 4 -y 5 +z 6 -z | 7 ++0 8 +-0 9 -+0 10 --0 | 11 +0+ 12 +0- 13 -0+ 14 -0-
    // | 15 0++ 16 0+- 17 0-+ 18 0-- | 19 +++ 20 ++- 21 +-+ 22 +-- 23 -++ 24 -+- 25 --+ 26 ---
    const v2f dx = A - fx, dy = A - fy, dz = A - fz;                    // faces {1, 2} {3, 4} {5, 6}
    const v2f dxp = {dx.x, dx.x}, dxm = {dx.y, dx.y}, dyp = {dy.x, dy.x}, dym = {dy.y, dy.y};
    const v2f exy_p = dxp - fy, exy_m = dxm - fy;                        // xy edges {7, 8} {9, 10}
    const v2f exz_p = dxp - fz, exz_m = dxm - fz;                        // xz edges {11, 12} {13, 14}
    const v2f eyz_p = dyp - fz, eyz_m = dym - fz;                        // yz edges {15, 16} {17, 18}
    const v2f c_pp = v2f{exy_p.x, exy_p.x} - fz, c_pm = v2f{exy_p.y, exy_p.y} - fz;  // corners {19, 20} {21, 22}
    const v2f c_mp = v2f{exy_m.x, exy_m.x} - fz, c_mm = v2f{exy_m.y, exy_m.y} - fz;  //         {23, 24} {25, 26}
    uint32_t dead = 0u;
    dead = push_sign(push_sign(dead, c_mm.y), c_mm.x), dead = push_sign(push_sign(dead, c_mp.y), c_mp.x);
    dead = push_sign(push_sign(dead, c_pm.y), c_pm.x), dead = push_sign(push_sign(dead, c_pp.y), c_pp.x);
    dead = push_sign(push_sign(dead, eyz_m.y), eyz_m.x), dead = push_sign(push_sign(dead, eyz_p.y), eyz_p.x);
    dead = push_sign(push_sign(dead, exz_m.y), exz_m.x), dead = push_sign(push_sign(dead, exz_p.y), exz_p.x);
    dead = push_sign(push_sign(dead, exy_m.y), exy_m.x), dead = push_sign(push_sign(dead, exy_p.y), exy_p.x);
    dead = push_sign(push_sign(dead, dz.y), dz.x), dead = push_sign(push_sign(dead, dy.y), dy.x), dead = push_sign(push_sign(dead, dx.y), dx.x);
    return L.todo & ~(dead << 1);  // (bit 0, the own voxel, is always alive)
}
// One trip over N consecutive points of a mirror bucket, in two halves so that a caller can keep several trips in flight:
// load_trip issues the N / 2 16-byte loads at immediate offsets; process_trip turns the loaded words into distances in mirror
// units, finds the two smallest by an integer-key tournament and merges them into t.  `pos0` = position of the trip's first
// point inside the bucket.  process_trip returns whether the bucket goes on behind this trip (its last slot holds a point).
// PARTS = 2: two neighbouring lanes share the trip (the odd lane holds its last slot).
template <int N>
__device__ __forceinline__ void load_trip(const uint4 *bt, uint4 (&c)[N / 2]) {
#pragma unroll
    for (int u = 0; u < N / 2; ++u) c[u] = bt[u];
}
template <int N, int PARTS>
__device__ __forceinline__ bool process_trip(const uint4 (&c)[N / 2], uint32_t pos0, uint32_t base, int s, const v2f qx, const v2f qy, const v2f qz, Best3 &t,
                                             float margin) {
    // All distances first (independent), then the minimum as a tournament over integer keys: a non-negative
    // float orders like its bit pattern, so (bits & ~31) | position is one v_min3_u32 per three candidates
    // and yields value and position at once (unique keys: the lower position wins a tie, like the
    // reference's first minimum).  The 5 dropped mantissa bits are part of the margin's error model.
    // Empty slots need no test: their second word converts to 4.3e9 units (kicp_common.hpp).
    uint32_t key[N];
#pragma unroll
    for (int u = 0; u < N / 2; ++u) {
        const v2f px = {static_cast<float>(c[u].x & 0xffffu), static_cast<float>(c[u].z & 0xffffu)};
        const v2f py = {static_cast<float>(c[u].x >> 16), static_cast<float>(c[u].z >> 16)};
        const v2f pz = {static_cast<float>(c[u].y), static_cast<float>(c[u].w)};  // (the whole word: z, or far away for an empty slot)
        const v2f ddx = px - qx, ddy = py - qy, ddz = pz - qz;
        const v2f d = __builtin_elementwise_fma(ddz, ddz, __builtin_elementwise_fma(ddy, ddy, ddx * ddx));
        key[2 * u] = (__float_as_uint(d.x) & ~31u) | static_cast<uint32_t>(2 * u);
        key[2 * u + 1] = (__float_as_uint(d.y) & ~31u) | static_cast<uint32_t>(2 * u + 1);
    }
    // the bucket ends where a trip's last slot is empty (the pair of lanes of a shared bucket must agree: the odd lane
    // holds that slot - quad_perm [1, 1, 3, 3])
    uint32_t last_word = c[N / 2 - 1].w;
    if (PARTS == 2) last_word = static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(last_word), 0xF5, 0xf, 0xf, false));
    const bool more = (last_word >> 16) == 0u;
    // winner and runner-up in one tournament (the keys are unique, so the runner-up is a different candidate)
    uint32_t key1, key2;
    tree_min2_u32<N>(key, key1, key2);
    const float m1 = __uint_as_float(key1 & ~31u);
    if (m1 <= t.b1 + margin) {  // something here can come within the margin of the running minimum
        // The third place is only worth a tournament when the runner-up lies within the margin of the winner; otherwise the
        // runner-up's value stands in for it (a lower bound that can never look like a near tie).  key - (key2 + 1) wraps to
        // the top of the range for the winner and the runner-up alone and keeps the order of all the others.
        uint32_t key3 = key2;
        if (__uint_as_float(min(key2, kFarKey) & ~31u) - m1 <= margin) {
            const uint32_t after2 = key2 + 1u;
#pragma unroll
            for (int u = 0; u < N; ++u) key[u] -= after2;
            key3 = after2 + tree_min_u32<N>(key);
        }
        const uint32_t k1 = pos0 + (key1 & 31u), k2 = pos0 + (key2 & 31u);  // positions within the bucket
        // with fewer than three points the far key stands in (finite, beyond every real distance)
        Best3 o{m1, __uint_as_float(min(key2, kFarKey) & ~31u), __uint_as_float(min(key3, kFarKey) & ~31u), base + k1,
                base + k2, static_cast<uint32_t>(s) * kOrdStride + k1, static_cast<uint32_t>(s) * kOrdStride + k2};
        best3_merge(t, o);
    }
    return more;
}
// the query as seen from the corner of neighbour voxel `s`, both lanes of the packed arithmetic
struct QueryFrom {
    v2f x, y, z;
};
__device__ __forceinline__ QueryFrom query_from(const Probe &P, int s) {
    const int dx = shift_component(kShiftX, s), dy = shift_component(kShiftY, s), dz = shift_component(kShiftZ, s);
    return QueryFrom{{P.lx - dx * kCell, P.lx - dx * kCell}, {P.ly - dy * kCell, P.ly - dy * kCell}, {P.lz - dz * kCell, P.lz - dz * kCell}};
}
// scan the bucket of neighbour voxel `s` of query P and merge what it finds into t.  PARTS = 2: two neighbouring lanes serve the
// same query and share every bucket - this lane takes points [10 part, 10 part + 10) of each 20-point trip (five 16-byte loads),
// the records are merged by the caller.
template <int PARTS>
__device__ __forceinline__ void visit_bucket(const Probe &P, Best3 &t, const MapView &m, int s, float margin, int part = 0) {
    static_assert(PARTS == 1 || PARTS == 2, "a trip is dealt to one lane or to two");
    constexpr int kMine = kTrip / PARTS;  // points of a trip this lane looks at
    // the neighbour's bucket comes out of the own voxel's record (same cache line as the probe): no second probe
    const uint32_t bucket = m.table[P.slot0].nb[s];
    const uint32_t base = bucket * m.cap;  // index into the fp64 pool
    const uint32_t stride16 = m.cap16;
    const uint4 *b = reinterpret_cast<const uint4 *>(m.pool16 + static_cast<size_t>(bucket) * stride16);
    const QueryFrom q = query_from(P, s);
    const uint32_t first = static_cast<uint32_t>(part) * kMine;  // this lane's first point within a trip
    for (uint32_t k0 = 0; k0 < stride16; k0 += kTrip) {  // (the mirror's bucket stride is a multiple of kTrip: a trip never leaves the bucket)
        uint4 c[kMine / 2];
        load_trip<kMine>(b + (k0 + first) / 2, c);
        if (!process_trip<kMine, PARTS>(c, k0 + first, base, s, q.x, q.y, q.z, t, margin)) break;
    }
}
// Latency-oriented round (scans that leave the machine two waves per SIMD: every dependent memory access is ~0.9 us of a wave's
// ~12): TWO neighbour voxels per round.  Both bucket records and both buckets' first trips are in flight together; the first is
// decided, and the second is only looked at if its voxel can still hold something within the margin of the new minimum (its
// lower bound `lb2`, the same test cull_todo applies) - so the arithmetic saved by culling stays saved, only the wait is shared.
__device__ __forceinline__ void visit_two(const Probe &P, Best3 &t, const MapView &m, int s1, int s2, bool has2, float lb2, float margin) {
    const uint32_t bucket1 = m.table[P.slot0].nb[s1], bucket2 = has2 ? m.table[P.slot0].nb[s2] : 0u;
    const uint32_t stride16 = m.cap16;
    const uint4 *b1 = reinterpret_cast<const uint4 *>(m.pool16 + static_cast<size_t>(bucket1) * stride16);
    const uint4 *b2 = reinterpret_cast<const uint4 *>(m.pool16 + static_cast<size_t>(bucket2) * stride16);
    uint4 c1[kTrip / 2], c2[kTrip / 2];
    load_trip<kTrip>(b1, c1);
    if (has2) load_trip<kTrip>(b2, c2);
    const QueryFrom q1 = query_from(P, s1);
    bool more = process_trip<kTrip, 1>(c1, 0u, bucket1 * m.cap, s1, q1.x, q1.y, q1.z, t, margin);
    for (uint32_t k0 = kTrip; more && k0 < stride16; k0 += kTrip) {  // (deeper buckets than one trip: max_points_per_voxel > 20)
        load_trip<kTrip>(b1 + k0 / 2, c1);
        more = process_trip<kTrip, 1>(c1, k0, bucket1 * m.cap, s1, q1.x, q1.y, q1.z, t, margin);
    }
    if (has2 && lb2 <= t.b1 + margin) {
        const QueryFrom q2 = query_from(P, s2);
        more = process_trip<kTrip, 1>(c2, 0u, bucket2 * m.cap, s2, q2.x, q2.y, q2.z, t, margin);
        for (uint32_t k0 = kTrip; more && k0 < stride16; k0 += kTrip) {
            load_trip<kTrip>(b2 + k0 / 2, c2);
            more = process_trip<kTrip, 1>(c2, k0, bucket2 * m.cap, s2, q2.x, q2.y, q2.z, t, margin);
        }
    }
}
// lower bound (units^2, with cull_todo's slack) of the squared distance from query P to anything in neighbour voxel s
__device__ __forceinline__ float voxel_lower_bound(const Probe &P, int s, float margin) {
    const int dx = shift_component(kShiftX, s), dy = shift_component(kShiftY, s), dz = shift_component(kShiftZ, s);
    const float lx = dx > 0 ? kCell - P.lx : P.lx, ly = dy > 0 ? kCell - P.ly : P.ly, lz = dz > 0 ? kCell - P.lz : P.lz;
    float lb = 0.f;
    if (dx != 0) lb += lx * lx * 0.99999f - margin;
    if (dy != 0) lb += ly * ly * 0.99999f - margin;
    if (dz != 0) lb += lz * lz * 0.99999f - margin;
    return lb;
}

// The kernel arguments as they lie in the kernarg segment, through an opaque copy of its address: what is read through this view is
// fetched (scalar loads, scalar cache) WHERE it is used instead of being loaded at the kernel's start and kept in scalar registers
// through the search - the pass kernels' sixteen words of basis pushed as many other values out into vector registers, and the
// four-waves build spilled.  Every pass kernel's first (or only) argument is its PassParams, at offset 0.
typedef const PassParams __attribute__((address_space(4))) *PassKernarg;
__device__ __forceinline__ const PassParams &args_at_point_of_use() {
    PassKernarg q = (PassKernarg)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("; kernel arguments, re-read at the point of use" : "+s"(q));
    return *(const PassParams *)q;
}
// exact resolution of a finished search: the winner (and whatever lies within the margin of it) re-evaluated in fp64, the
// reference's tie rule, the acceptance test and the per-correspondence terms (Registration.cpp:74-77, 86-93)
// `host_pose`: T is the host's pose (the kernel arguments carry its basis); otherwise the basis is formed here, from T
// EXPORT (kicp_pass_correspondences): the decision is also written out per query - index, squared distance, coordinates
__device__ __forceinline__ void export_correspondence(const PassParams &p, uint32_t i, uint32_t idx, double d2, double x, double y, double z) {
    p.corr_index[i] = idx == kNoIndex32 ? -1 : static_cast<int32_t>(idx);
    p.corr_d2[i] = idx == kNoIndex32 ? DBL_MAX : d2;
    p.corr_nn[3 * i] = x, p.corr_nn[3 * i + 1] = y, p.corr_nn[3 * i + 2] = z;
}
template <bool EXPORT = false>
__device__ __forceinline__ void resolve_and_accumulate(Acc &acc, const PassParams &p, bool host_pose, const double *__restrict__ src, const Pose &T, uint32_t i,
                                                       const Best3 &t, const KeptQuery *kept = nullptr) {
    if (EXPORT && i != kNoIndex32) export_correspondence(p, i, kNoIndex32, 0.0, 0.0, 0.0, 0.0);  // (overwritten below when the query has a correspondence)
    if (i == kNoIndex32 || t.i1 == kNoIndex32 || (kDbgBuild && p.dbg != 0 && p.dbg != 9 && p.dbg != 11 && p.dbg != 12 && p.dbg != 13 && p.dbg != 14)) return;
    const MapView &m = p.map;
    const float margin = p.search.margin_u;
    Query q;
    if (kept) q = kept->q;
    else make_query_of(q, p, src, T, i);
    const double bound = p.search.bound;
    double best = bound;
    uint32_t best_idx = kNoIndex32;
    double wx = 0.0, wy = 0.0, wz = 0.0;  // the winner's coordinates, as far as they have passed through registers already
    bool have_winner = false;
    if (t.b3 - t.b1 <= margin && dbg_not(p, 9)) {  // three near-equal candidates: leave it to the exact fp64 search (dbg 9: experiment without it)
        // (which needs to look no farther than the pre-selected winner's exact distance)
        if (kept && !kept->voxel) {
            const double vs = m.voxel_size;
            q.vx = voxel_coord(q.x, vs, p.search.inv_vs), q.vy = voxel_coord(q.y, vs, p.search.inv_vs), q.vz = voxel_coord(q.z, vs, p.search.inv_vs);
        }
        const double *c1 = m.pool + static_cast<size_t>(t.i1) * 3;
        search_global(m, q, best, best_idx, exact_d2_of(c1[0], c1[1], c1[2], q));
    } else {
        const double *c1 = m.pool + static_cast<size_t>(t.i1) * 3;
        const double x1 = c1[0], y1 = c1[1], z1 = c1[2];
        const double d1 = exact_d2_of(x1, y1, z1, q);
        if (d1 < best) best = d1, best_idx = t.i1, wx = x1, wy = y1, wz = z1, have_winner = true;
        if (t.b2 - t.b1 <= margin && t.i2 != kNoIndex32) {
            const double *c2 = m.pool + static_cast<size_t>(t.i2) * 3;
            const double x2 = c2[0], y2 = c2[1], z2 = c2[2];
            const double d2 = exact_d2_of(x2, y2, z2, q);
            // the reference keeps the FIRST candidate (in visiting order) whose NORM attains the strict minimum: the later of
            // the two only replaces the earlier when its rounded square root is smaller (closer_by_norm)
            if (d2 < bound) {
                if (best_idx == kNoIndex32 || (t.o2 < t.o1 ? !closer_by_norm(d1, d2) : closer_by_norm(d2, d1)))
                    best = d2, best_idx = t.i2, wx = x2, wy = y2, wz = z2, have_winner = true;
            }
        }
    }
    if (best_idx != kNoIndex32 && accepted_by_norm(best, p.tau, p.search)) {  // `distance < max_correspondance_distance`, Registration.cpp:75
        if (!have_winner) {  // (found by the exact search)
            const double *tp = m.pool + static_cast<size_t>(best_idx) * 3;
            wx = tp[0], wy = tp[1], wz = tp[2];
        }
        if (EXPORT) export_correspondence(p, i, best_idx, best, wx, wy, wz);
        // The basis is taken up HERE - behind an opaque copy of the flag, so that the compiler cannot merge its two sources ahead of
        // the exact phase and carry sixteen registers through it (the four-waves build spilled them) - and, where it is the host's,
        // read from the kernarg segment here rather than at the kernel's start (args_at_point_of_use).
        if (dbg_is(p, 13)) return;  // (attribution, tools/valu_attribution.sh: the exact phase without the terms)
        int from_args = __builtin_amdgcn_readfirstlane(host_pose ? 1 : 0);
        asm volatile("; the basis is taken up here" : "+s"(from_args));
        PassBasis B;
        if (from_args) B = args_at_point_of_use().sol.basis;
        else B = basis_of(T);
        // the untransformed source point again (L1 / L2 hit) unless the build kept it: cheaper than registers kept live through the search
        accumulate(acc, B, kept ? kept->sx : src[3 * i], kept ? kept->sy : src[3 * i + 1], q.x, q.y, q.z, wx, wy, wz);
    }
}

// G consecutive lanes serve one query: the occupied neighbour voxels are dealt round-robin to the G sub-lanes, which
// halves/quarters each lane's dependent chain and multiplies the resident waves; the sub-lanes share their running
// minimum for culling and merge their records with shuffles at the end.  (Small scans: few waves, latency bound.)
// OCC = waves per SIMD the register allocation aims for: 4 (<= 128 VGPRs; what scans larger than the machine want) or 3
// (<= 168: the compiler keeps more values instead of recomputing them; for scans that do not fill three waves per SIMD).
// SPLIT (G == 2): the two sub-lanes of a query do not deal the neighbour voxels between them but share every bucket, ten
// points each.
// LAT (G == 1, built at two waves per SIMD): the latency-oriented build for scans that do not fill the machine beyond that
// (<= 131 072 points on 256 CUs) - two neighbour voxels per round (visit_two).
// the search and the exact phase of one pass for lane `tid` of workgroup blockIdx.x; `acc` receives the lane's terms
// (`src`, `n`: the scan - p.src / p.n for a kernel that serves one call, the current scan of a resident kernel that serves a batch)
// `park` (the four-waves build): BLOCK x kParkWords doubles of LDS in which a lane parks its transformed point and the two source
// coordinates the terms need while it searches - that build has no registers to keep them (128 VGPRs) and used to transform the
// source point a second time for the exact phase (~70 fp64 instructions and three loads per lane).
constexpr int kParkWords = 5;
// `host_pose`: T is the host's pose (kernel arguments) and p.sol.basis its basis; where the kernel got T from the device the basis
// is formed right before the exact phase, not kept through the search.
template <int BLOCK, int G, bool SPLIT, bool LAT, bool PARK = false, bool EXPORT = false>
__device__ __forceinline__ void gather32_pass(const PassParams &p, const Pose &T, bool host_pose, uint32_t tid, Acc &acc, const double *__restrict__ src, uint32_t n,
                                              uint32_t block, int *lend = nullptr, double *park = nullptr) {
    const MapView &m = p.map;
    const float margin = p.search.margin_u;
    const uint32_t gt = block * BLOCK + tid;  // (`block`: which BLOCK points of the scan this workgroup takes - blockIdx.x, or a resident kernel's turn)
    const uint32_t i = gt / G;
    const int sub = static_cast<int>(gt % G);
    const bool valid = i < n && dbg_not(p, 7) && dbg_not(p, 8);
    Lane L;
    KeptQuery kept;
    static_assert(!(LAT && PARK), "the latency-oriented build keeps the query in registers");
    start_lane(L, p, src, T, i, valid, (LAT || PARK) ? &kept : nullptr);
    if (PARK) {
        park[0 * BLOCK + tid] = kept.q.x, park[1 * BLOCK + tid] = kept.q.y, park[2 * BLOCK + tid] = kept.q.z;
        park[3 * BLOCK + tid] = kept.sx, park[4 * BLOCK + tid] = kept.sy;
    }
    if (G > 1 && !SPLIT) {  // deal the set bits round-robin: the r-th occupied voxel goes to sub-lane r % G
        uint32_t rest = L.todo, mine = 0u;
        for (int r = 0; rest; ++r) {
            const uint32_t low = rest & (0u - rest);
            rest ^= low;
            if (r % G == sub) mine |= low;
        }
        L.todo = mine;
    }
    float cull = L.t.b1;  // running minimum shared by the G sub-lanes (culling only)
    uint32_t rounds = 0u;  // (wave-uniform; read by the dbg 10 census only)
    while (__any(L.todo != 0u)) {
        ++rounds;
        // which of the 27 neighbours could still hold something within the margin of the current minimum
        L.todo = cull_todo(L, cull, margin);
        if (LAT) {
            if (L.todo) {
                const int s1 = __ffs(L.todo) - 1;
                L.todo &= L.todo - 1u;
                const bool has2 = L.todo != 0u;
                const int s2 = has2 ? __ffs(L.todo) - 1 : s1;
                L.todo &= L.todo - 1u;  // (0 stays 0; a second voxel that the first one's result rules out is dropped here - cull_todo would)
                visit_two(L.q, L.t, m, s1, s2, has2, has2 ? voxel_lower_bound(L.q, s2, margin) : 0.f, margin);
            }
        } else if (G == 1 && lend != nullptr && dbg_not(p, 11)) {  // (dbg 11: the in-process A/B switch of this, tools/ab_option.py)
            // One lane per query, one neighbour voxel per round - and every round costs the WAVE its ~430 instructions however few
            // lanes still have a voxel to see (cfg2: 100 % of the lanes in round 1, 42 % in round 2, 7 % in round 3, 5 % in round 4;
            // 3.0 rounds per wave).  So from the second round on, lanes with nothing to do take over voxels of the queries that
            // have more than one left: such a query gives away up to two voxels beyond the one it visits itself, job j goes to the
            // j-th idle lane (dealt through a few words of LDS), which fetches the query's probe by ds_bpermute, parks its own
            // record in LDS, visits the voxel with a fresh record and hands that record back through LDS to be merged.  The visiting
            // order travels with every candidate (Best3::o1 / o2), so the reference's first-minimum rule holds whoever did the
            // visiting; a voxel that a sharper minimum would have culled is visited needlessly but cannot change the outcome.
            const bool busy = L.todo != 0u;
            int s = 0;
            if (busy) s = __ffs(L.todo) - 1, L.todo &= L.todo - 1u;
            bool helps = false;
            int jobs = 0, job0 = 0, slot = 0;
            if (rounds > 1u) {
                const int spare = min(2, __popc(L.todo));
                const unsigned long long give1 = __ballot(spare >= 1), give2 = __ballot(spare >= 2), idle = __ballot(!busy);
                if (give1 != 0ull && idle != 0ull) {  // (wave-uniform)
                    const int lane = static_cast<int>(tid & 63u);
                    const unsigned long long below = (1ull << lane) - 1ull;
                    job0 = __popcll(give1 & below) + __popcll(give2 & below);
                    const int pairs = min(min(__popcll(give1) + __popcll(give2), __popcll(idle)), kLendJobs);
                    slot = __popcll(idle & below);
                    helps = !busy && slot < pairs;
                    jobs = max(0, min(spare, pairs - job0));
                    for (int j = 0; j < jobs; ++j) {
                        const int sj = __ffs(L.todo) - 1;
                        L.todo &= L.todo - 1u;
                        lend[job0 + j] = lane | (sj << 8);
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    int from = lane;
                    if (helps) {
                        const int e = lend[slot];
                        from = e & 63, s = e >> 8;
                    }
                    const float glx = __shfl(L.q.lx, from, 64), gly = __shfl(L.q.ly, from, 64), glz = __shfl(L.q.lz, from, 64);
                    const uint32_t gslot = __shfl(L.q.slot0, from, 64);
                    if (helps) {  // (a lane with nothing to do never needs its own probe again; its record waits in LDS)
                        int *keep = lend + kLendJobs + 7 * slot;
                        keep[0] = __float_as_int(L.t.b1), keep[1] = __float_as_int(L.t.b2), keep[2] = __float_as_int(L.t.b3);
                        keep[3] = static_cast<int>(L.t.i1), keep[4] = static_cast<int>(L.t.i2), keep[5] = static_cast<int>(L.t.o1), keep[6] = static_cast<int>(L.t.o2);
                        L.q = Probe{glx, gly, glz, gslot};
                        L.t = Best3{p.search.bound_u, p.search.bound_u, p.search.bound_u, kNoIndex32, kNoIndex32, 0u, 0u};
                    }
                }
            }
            if (busy || helps) visit_bucket<1>(L.q, L.t, m, s, margin, 0);
            if (__any(helps)) {  // (wave-uniform) the lent lanes' records go home
                if (helps) {
                    int *back = lend + kLendJobs + 7 * kLendJobs + 7 * slot;
                    back[0] = __float_as_int(L.t.b1), back[1] = __float_as_int(L.t.b2), back[2] = __float_as_int(L.t.b3);
                    back[3] = static_cast<int>(L.t.i1), back[4] = static_cast<int>(L.t.i2), back[5] = static_cast<int>(L.t.o1), back[6] = static_cast<int>(L.t.o2);
                    const int *keep = lend + kLendJobs + 7 * slot;
                    L.t = Best3{__int_as_float(keep[0]), __int_as_float(keep[1]), __int_as_float(keep[2]), static_cast<uint32_t>(keep[3]), static_cast<uint32_t>(keep[4]),
                                static_cast<uint32_t>(keep[5]), static_cast<uint32_t>(keep[6])};
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                for (int j = 0; j < jobs; ++j) {
                    const int *back = lend + kLendJobs + 7 * kLendJobs + 7 * (job0 + j);
                    const Best3 o{__int_as_float(back[0]), __int_as_float(back[1]), __int_as_float(back[2]), static_cast<uint32_t>(back[3]), static_cast<uint32_t>(back[4]),
                                  static_cast<uint32_t>(back[5]), static_cast<uint32_t>(back[6])};
                    if (o.i1 != kNoIndex32) best3_merge(L.t, o);
                }
            }
        } else if (L.todo) {
            const int s = __ffs(L.todo) - 1;
            L.todo &= L.todo - 1u;
            visit_bucket<SPLIT ? 2 : 1>(L.q, L.t, m, s, margin, SPLIT ? sub : 0);
        }
        cull = L.t.b1;
#pragma unroll
        for (int off = 1; off < G; off <<= 1) cull = fminf(cull, __shfl_xor(cull, off, 64));
    }
    // ---- merge the sub-lanes' records (butterfly: every sub-lane ends up with the query's record) ------------------------
#pragma unroll
    for (int off = 1; off < G; off <<= 1) {
        Best3 o;
        o.b1 = __shfl_xor(L.t.b1, off, 64), o.b2 = __shfl_xor(L.t.b2, off, 64), o.b3 = __shfl_xor(L.t.b3, off, 64);
        o.i1 = __shfl_xor(L.t.i1, off, 64), o.i2 = __shfl_xor(L.t.i2, off, 64), o.o1 = __shfl_xor(L.t.o1, off, 64), o.o2 = __shfl_xor(L.t.o2, off, 64);
        best3_merge(L.t, o);
    }
    // ---- exact resolution (one sub-lane per query) ----------------------------------------------------------------------
    if (PARK) {  // (a lane reads back what it wrote itself: no barrier)
        kept.q.x = park[0 * BLOCK + tid], kept.q.y = park[1 * BLOCK + tid], kept.q.z = park[2 * BLOCK + tid];
        kept.sx = park[3 * BLOCK + tid], kept.sy = park[4 * BLOCK + tid], kept.voxel = false;
    }
    if (sub == 0) {
        resolve_and_accumulate<EXPORT>(acc, p, host_pose, src, T, L.i, L.t, (LAT || PARK) ? &kept : nullptr);
    }
    // dbg 10 (bench.py's latency model): no correspondences are formed; the "count" sum carries the number of visiting rounds
    // this WAVE ran - its chain of dependent bucket visits - from lane 0 (as rounds x 2^40: limb 1 holds bits 21..41, limb 2 the rest)
    if (dbg_is(p, 10) && (tid & 63u) == 0u) acc.limb[6 * kTermLimbs + 1] = static_cast<int>(rounds & 3u) << 19, acc.limb[6 * kTermLimbs + 2] = static_cast<int>(rounds >> 2);
}
template <int BLOCK, int G, int OCC, bool SPLIT, bool LAT = false, bool EXPORT = false>
__global__ __launch_bounds__(BLOCK, OCC) void k_pass_gather32(const PassParams p) {
    static_assert(!SPLIT || G == 2, "bucket sharing is written for pairs of lanes");
    static_assert(!LAT || G == 1, "the latency-oriented build serves one lane per query");
    KICP_PASS_SHARED(BLOCK)
    constexpr bool kLends = G == 1 && !SPLIT && !LAT;  // idle lanes take over voxels of loaded queries (gather32_pass)
    __shared__ int s_lend[kLends ? BLOCK / 64 : 1][kLendWords];
    __shared__ double s_park[kLends ? BLOCK * kParkWords : 1];
    const Pose T = load_pose(p);
    const bool host_pose = p.sol.pass == 0 || p.sol.mode >= 2;  // (load_pose)
    Acc acc{};
    gather32_pass<BLOCK, G, SPLIT, LAT, kLends, EXPORT>(p, T, host_pose, threadIdx.x, acc, p.src, p.n, blockIdx.x, kLends ? &s_lend[kLends ? threadIdx.x / 64 : 0][0] : nullptr,
                                                s_park);
    if (BLOCK > 64) __syncthreads();
    finish_pass<BLOCK>(acc, p, s_red, &s_flag, p.sol.tag);
}

// multi-GPU with host-side solve: hand the all-reduced limb totals to the host
static __global__ __launch_bounds__(64) void k_publish_words(IcpState *st, HostRecord *rec, unsigned long long call_id, int pass) {
    const int lane = threadIdx.x;
    if (lane < kReduceWords) __hip_atomic_store(&rec->words[lane], st->reduce[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) __hip_atomic_store(&rec->seq, (call_id << 16) | static_cast<unsigned long long>(pass + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// publish the raw sums of the last pass (kicp_pass_sums)
static __global__ __launch_bounds__(64) void k_publish_sums(IcpState *st, HostRecord *rec, unsigned long long call_id) {
    if (threadIdx.x != 0) return;
    for (int i = 0; i < kNumSums; ++i)
        __hip_atomic_store(reinterpret_cast<unsigned long long *>(&rec->sums[i]),
                           static_cast<unsigned long long>(__double_as_longlong(limbs_to_double(st->reduce + 3 * i))), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_SYSTEM);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_store(&rec->seq, (call_id << 16) | 0x8001ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ------------------------------------------------------------------------------------------------------------
// delta upload of the map mirror: scatter staged rows (8-byte words) to their places
//   row r of `staged` (row_words uint2 each) goes to dst + index[r] * row_words
// ------------------------------------------------------------------------------------------------------------
static __global__ __launch_bounds__(256) void k_scatter_rows(const uint2 *__restrict__ staged, const uint32_t *__restrict__ index, uint32_t rows,
                                                      uint32_t row_words, uint2 *__restrict__ dst) {
    const size_t total = static_cast<size_t>(rows) * row_words;
    for (size_t e = static_cast<size_t>(blockIdx.x) * 256 + threadIdx.x; e < total; e += static_cast<size_t>(gridDim.x) * 256) {
        const uint32_t r = static_cast<uint32_t>(e / row_words), w = static_cast<uint32_t>(e % row_words);
        dst[static_cast<size_t>(index[r]) * row_words + w] = staged[e];
    }
}

// ------------------------------------------------------------------------------------------------------------
// GetClosestNeighbor for a batch (API parity; same search code as the fused kernel, no acceptance bound)
// ------------------------------------------------------------------------------------------------------------
static __global__ __launch_bounds__(256) void k_closest(const double *__restrict__ queries, uint32_t n, const MapView m,
                                                 double *__restrict__ out_nn, double *__restrict__ out_dist) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    Query q;
    make_query(q, queries[3 * i], queries[3 * i + 1], queries[3 * i + 2], m.voxel_size);
    double best = DBL_MAX;
    uint32_t best_idx = 0xFFFFFFFFu;
    search_global(m, q, best, best_idx);
    if (best_idx == 0xFFFFFFFFu) {
        out_nn[3 * i] = out_nn[3 * i + 1] = out_nn[3 * i + 2] = 0.0;
        out_dist[i] = DBL_MAX;
    } else {
        const double *t = m.pool + static_cast<size_t>(best_idx) * 3;
        out_nn[3 * i] = t[0], out_nn[3 * i + 1] = t[1], out_nn[3 * i + 2] = t[2];
        out_dist[i] = sqrt(best);
    }
}

}  // namespace kicp
