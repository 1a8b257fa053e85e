// kicp_pre.hpp -- the pipeline's pre-steps on the GPU (reference call sites
// pipeline/KinematicICP.cpp:54-62): kiss_icp::Preprocessor::Preprocess (constant-velocity deskew + range crop, kiss-icp
// v1.2.0 core/Preprocessing.cpp) fused with transform_points (KinematicICP.cpp:31-36), and kiss_icp::VoxelDownsample
// (core/VoxelUtils.cpp: first point of every voxel wins, output in the hash table's iteration order).  All of them are
// order-sensitive on the CPU; here the order is made explicit - the crop keeps input order, "first" means lowest input
// index (atomicMin), and the downsample's output order is the reference table's, replayed per cluster - so the results
// are deterministic and equal to the sequential reference's, order included.
// HBM bound streaming kernels: 24-32 B read + <= 24 B written per point, fp64 throughout.
#pragma once
#include <cmath>
#include "kicp_common.hpp"
#include "kicp_se3.hpp"
#include "kicp_table_order.hpp"

namespace kicp {

struct PreprocessParams {
    const double *in;          // raw scan, sensor frame
    const double *timestamps;  // normalised to [0,1], or nullptr
    // ts_normalise != 0: `timestamps` still holds the stamps in seconds as the ingest decoded them; (t - ts_lo) / (ts_hi - ts_lo) -
    // TimeStampHandler.cpp:121-128, the same two fp64 operations - happens here, where the stamp is consumed (round 6: the
    // separate normalisation kernel is gone; the extrema come back from the ingest and travel by value)
    double ts_lo, ts_hi;
    int32_t ts_normalise;
    uint32_t n;
    int32_t deskew;
    double omega[6];           // log(relative_motion)
    Pose motion_inverse;       // relative_motion^-1
    Pose lidar_to_base;
    double max_range, min_range;
    uint32_t *flags;           // 1 = survives the crop
    double *staged;            // transformed point of every input (base frame), compacted afterwards
    uint32_t *block_counts;    // survivors per 256-thread block
};

// block-wide count of set predicates -> *out (lane 0 of wave 0 writes)
__device__ __forceinline__ void block_count_store_at(bool pred, uint32_t *out) {
    __shared__ uint32_t s_cnt[4];
    __syncthreads();  // (a caller that loops: the previous turn's reader is done with s_cnt)
    const unsigned long long ballot = __ballot(pred);
    if ((threadIdx.x & 63) == 0) s_cnt[threadIdx.x >> 6] = static_cast<uint32_t>(__popcll(ballot));
    __syncthreads();
    if (threadIdx.x == 0) *out = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
}
__device__ __forceinline__ void block_count_store(bool pred, uint32_t *block_counts) { block_count_store_at(pred, block_counts + blockIdx.x); }

// Preprocess + transform_points of point i: deskew to the scan end (when asked), crop decision in the sensor frame, the point in
// the base frame.  Returns whether the point survives the crop.
__device__ __forceinline__ bool preprocess_point(const PreprocessParams &p, uint32_t i, double &bx, double &by, double &bz) {
    double x = p.in[3 * i], y = p.in[3 * i + 1], z = p.in[3 * i + 2];
    if (p.deskew) {  // p' = (relative_motion^-1 * exp(t_i * omega)) * p_i : deskew to the scan end
        double t = p.timestamps[i];
        if (p.ts_normalise) t = (t - p.ts_lo) / (p.ts_hi - p.ts_lo);
        const double xi[6] = {t * p.omega[0], t * p.omega[1], t * p.omega[2], t * p.omega[3], t * p.omega[4], t * p.omega[5]};
        const Pose M = pose_mul(p.motion_inverse, pose_exp(xi));
        double rx, ry, rz;
        quat_rotate(M, x, y, z, rx, ry, rz);
        x = rx + M.tx, y = ry + M.ty, z = rz + M.tz;
    }
    const double r = sqrt(x * x + y * y + z * z);
    quat_rotate(p.lidar_to_base, x, y, z, bx, by, bz);  // transform_points: into the base frame
    bx += p.lidar_to_base.tx, by += p.lidar_to_base.ty, bz += p.lidar_to_base.tz;
    return r < p.max_range && r > p.min_range;  // strict on both sides
}
static __global__ __launch_bounds__(256) void k_preprocess(const PreprocessParams p) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    bool keep = false;
    if (i < p.n) {
        double bx, by, bz;
        keep = preprocess_point(p, i, bx, by, bz);
        p.staged[3 * i] = bx, p.staged[3 * i + 1] = by, p.staged[3 * i + 2] = bz;
        p.flags[i] = keep ? 1u : 0u;
    }
    block_count_store(keep, p.block_counts);
}

// exclusive scan of the per-block counts (one workgroup; up to 1024 * 64 blocks = 16.7M points)
static __global__ __launch_bounds__(1024) void k_scan_blocks(uint32_t *block_counts, uint32_t nblocks, uint32_t *total) {
    __shared__ uint32_t s_wave[16];
    __shared__ uint32_t s_carry;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    for (uint32_t b0 = 0; b0 < nblocks; b0 += 1024) {
        const uint32_t b = b0 + threadIdx.x;
        const uint32_t c = b < nblocks ? block_counts[b] : 0u;
        uint32_t incl = c;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t t = __shfl_up(incl, off, 64);
            if (lane >= off) incl += t;
        }
        if (lane == 63) s_wave[wave] = incl;
        __syncthreads();
        uint32_t before = s_carry;
        for (int w = 0; w < wave; ++w) before += s_wave[w];
        if (b < nblocks) block_counts[b] = before + incl - c;
        __syncthreads();
        if (threadIdx.x == 1023) s_carry = before + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) *total = s_carry;
}

// Offset of this workgroup's survivors = the sum of the counts of the workgroups before it, added up by the workgroup itself
// (256 threads, a strided share each): for the grids of a frame (<= kFusedScanBlocks workgroups) that is a few loads per thread
// and saves the separate scan kernel - a launch of one workgroup whose ~5 us were pure latency, three times per frame.  The last
// workgroup also leaves the grand total in *total.  `raw` == 0: the counts have been scanned already (k_scan_blocks: larger grids).
constexpr uint32_t kFusedScanBlocks = 4096;
__device__ __forceinline__ uint32_t block_offset(const uint32_t *counts, int raw, uint32_t *total) {
    __shared__ uint32_t s_part[4];
    __shared__ uint32_t s_offset;
    const uint32_t b = blockIdx.x;
    if (!raw) return counts[b];
    uint32_t sum = 0u;
    for (uint32_t i = threadIdx.x; i < b; i += 256u) sum += counts[i];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off, 64);
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = sum;
    __syncthreads();
    if (threadIdx.x == 0) {
        s_offset = s_part[0] + s_part[1] + s_part[2] + s_part[3];
        if (total && b == gridDim.x - 1u) *total = s_offset + counts[b];
    }
    __syncthreads();
    return s_offset;
}
// order-preserving compaction: survivor i goes to block_offset + (number of survivors before it in its block)
static __global__ __launch_bounds__(256) void k_compact(const double *staged, const uint32_t *flags, const uint32_t *block_counts, int raw, uint32_t *total,
                                                 uint32_t n, double *out) {
    __shared__ uint32_t s_wave[4];
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool keep = i < n && flags[i] != 0u;
    const unsigned long long ballot = __ballot(keep);
    if (lane == 0) s_wave[wave] = static_cast<uint32_t>(__popcll(ballot));
    const uint32_t offset = block_offset(block_counts, raw, total);  // (its barriers also publish s_wave)
    __syncthreads();
    if (!keep) return;
    uint32_t pos = offset + static_cast<uint32_t>(__popcll(ballot & ((1ull << lane) - 1ull)));
    for (int w = 0; w < wave; ++w) pos += s_wave[w];
    out[3 * pos] = staged[3 * i], out[3 * pos + 1] = staged[3 * i + 1], out[3 * pos + 2] = staged[3 * i + 2];
}

// ---- VoxelDownsample ----------------------------------------------------------------------------------------------------
// kiss_icp::VoxelDownsample (kiss-icp v1.2.0 core/VoxelUtils.cpp; call sites
// pipeline/KinematicICP.cpp:40,42) keeps the first point of every voxel and returns the survivors in the ITERATION ORDER of
// its tsl::robin_map<Voxel, Vector3d> after reserve(frame.size()).  That order decides which point of a coarser voxel the
// second downsample keeps and the order in which local_map_.Update inserts, so it is reproduced here, in parallel:
//   1. the device table has the reference's bucket count and the reference's ideal bucket (std::hash<Voxel> & mask).
//      Linear probing and robin-hood probing occupy the SAME set of buckets (an insertion always ends in the first free
//      bucket at or after the ideal one), so after k_downsample_claim the occupied slots are the reference's occupied
//      buckets, and every maximal run of occupied slots ("cluster") holds exactly the keys the reference holds there -
//      only their arrangement inside the run differs;
//   2. "first" = lowest input index (atomicMin per slot), which is also the order in which the reference inserted;
//   3. k_downsample_replay: the thread at the head of a run replays the reference's robin-hood insertions of that
//      run's keys, in insertion order, inside the run (insertions never leave their final cluster, so clusters are
//      independent).  Runs are short (load factor <= 0.5 by construction, typically < 0.2);
//   4. the survivors are gathered in ascending bucket index = the reference's iteration order.
constexpr unsigned long long kEmptyVoxelKey = ~0ull;
struct DownsampleParams {
    const double *in;
    uint32_t n;
    double voxel_size;
    unsigned long long *keys;  // [mask+1] packed voxel of the slot, kEmptyVoxelKey when free
    uint32_t *min_index;       // [mask+1] lowest input index seen for the slot's voxel, 0xFFFFFFFF initially
    uint32_t *order;           // [mask+1] after the replay: input index of the point the reference keeps in this bucket
    uint32_t *home_at;         // [mask+1] scratch of the replay: ideal bucket of order[]'s resident
    uint32_t mask;             // the reference's bucket count - 1
    uint32_t *block_counts;    // occupied buckets per 256-slot block
    uint32_t *error;           // set when a voxel coordinate leaves the 21-bit packable range
    uint32_t *probe_max;       // largest robin-hood displacement the replay saw (atomicMax; see replay_cluster)
    // chained pre-steps (kicp_pre_frame_*): the input count is the previous step's survivor count, still on the device; n and mask
    // above are then unused, the launch is sized for an upper bound and every kernel derives the reference's bucket count itself
    const uint32_t *n_dev;
    uint32_t *probe_max_sticky;  // nullable: the chain's second downsample must not reset the first one's figure
};
__device__ __forceinline__ uint32_t ds_count(const DownsampleParams &p) { return p.n_dev ? *p.n_dev : p.n; }
__device__ __forceinline__ uint32_t ds_mask(const DownsampleParams &p, uint32_t n) {
    if (!p.n_dev) return p.mask;
    const uint32_t buckets = reference_bucket_count_u32(n);
    return buckets ? buckets - 1u : 0u;
}

__device__ __forceinline__ unsigned long long pack_voxel21(int32_t x, int32_t y, int32_t z, bool &ok) {
    const int lim = 1 << 20;
    ok = x >= -lim && x < lim && y >= -lim && y < lim && z >= -lim && z < lim;
    return (static_cast<unsigned long long>(static_cast<uint32_t>(z + lim) & 0x1FFFFFu) << 42) |
           (static_cast<unsigned long long>(static_cast<uint32_t>(y + lim) & 0x1FFFFFu) << 21) |
           static_cast<unsigned long long>(static_cast<uint32_t>(x + lim) & 0x1FFFFFu);
}
// One point's claim: `key` (kEmptyVoxelKey: this lane has no point) finds / takes its voxel's slot by linear probing from the
// reference's ideal bucket and lowers the slot's winner to `index`.  The lanes of a wave hold CONSECUTIVE, ASCENDING indices.
// `overflow` / `overflow_tag` (the fused chain's first level, whose table size is a guess): a probe longer than kClaimProbeLimit
// gives up and leaves the tag there - a table sized for fewer points than it gets would otherwise be probed for ever.
constexpr uint32_t kClaimProbeLimit = 1024;
__device__ __forceinline__ void claim_voxel(unsigned long long *keys, uint32_t *min_index, uint32_t mask, unsigned long long key, bool ok, int32_t vx, int32_t vy,
                                            int32_t vz, uint32_t index, uint32_t *error, uint32_t *overflow = nullptr, uint32_t overflow_tag = 0u) {
    // Consecutive points of a scan ring fall into the same voxel in long runs (hundreds of points per voxel close to the sensor),
    // and every one of them would hammer the same two words with atomics - the kernel's tail, up to 0.6 ms on some frames.  Only
    // the FIRST lane of a run inside a wave goes on: it has the lowest index of the run, which is all atomicMin would keep.
    const unsigned long long prev = __shfl_up(key, 1, 64);
    if (key == kEmptyVoxelKey || ((threadIdx.x & 63) != 0 && prev == key)) return;
    if (!ok) {
        *error = 1u;
        return;
    }
    uint32_t slot = reference_voxel_hash(vx, vy, vz) & mask;
    for (uint32_t probes = 0;; ++probes) {
        const unsigned long long seen = atomicCAS(keys + slot, kEmptyVoxelKey, key);
        if (seen == kEmptyVoxelKey || seen == key) break;
        if (overflow && probes >= kClaimProbeLimit) {
            *overflow = overflow_tag;
            return;
        }
        slot = (slot + 1) & mask;
    }
    atomicMin(min_index + slot, index);
}
// pass 1: every point claims / finds its voxel's slot (linear probing from the reference's ideal bucket) and lowers the
// slot's winner to its own index
static __global__ __launch_bounds__(256) void k_downsample_claim(const DownsampleParams p) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i == 0 && !p.probe_max_sticky) *p.probe_max = 0u;  // (the replay kernel, which raises it, runs after this one: no separate memset per call)
    const uint32_t n = ds_count(p), mask = ds_mask(p, n);
    const bool in_range = i < n;
    const double vs = p.voxel_size;
    int32_t vx = 0, vy = 0, vz = 0;
    if (in_range)
        vx = static_cast<int32_t>(floor(p.in[3 * i] / vs)), vy = static_cast<int32_t>(floor(p.in[3 * i + 1] / vs)), vz = static_cast<int32_t>(floor(p.in[3 * i + 2] / vs));
    bool ok = true;
    const unsigned long long key = in_range ? pack_voxel21(vx, vy, vz, ok) : kEmptyVoxelKey;
    claim_voxel(p.keys, p.min_index, mask, key, ok, vx, vy, vz, i, p.error);
}

// pass 2: one thread per bucket; heads of clusters replay them; every thread counts its bucket for the compaction
static __global__ __launch_bounds__(256) void k_downsample_replay(const DownsampleParams p) {
    const uint32_t s = blockIdx.x * 256 + threadIdx.x;
    const uint32_t n = ds_count(p), mask = ds_mask(p, n);
    bool occupied = false;
    if (n != 0u && s <= mask) {
        occupied = p.keys[s] != kEmptyVoxelKey;
        if (occupied && p.keys[(s - 1u) & mask] == kEmptyVoxelKey) {
            uint32_t len = 1u;
            while (p.keys[(s + len) & mask] != kEmptyVoxelKey) ++len;  // ends: at least half of the buckets are free
            const uint32_t probe = replay_cluster(p.keys, p.min_index, p.order, p.home_at, mask, s, len);
            if (probe >= 32u) atomicMax(p.probe_max, probe);  // (short probes are the rule: only the rare long one touches the counter)
        }
    }
    block_count_store(occupied, p.block_counts);
}
// pass 3: survivors in ascending bucket index (the reference's iteration order)
static __global__ __launch_bounds__(256) void k_downsample_gather(const DownsampleParams p, const uint32_t *block_counts, int raw, uint32_t *total, double *out) {
    __shared__ uint32_t s_wave[4];
    const uint32_t s = blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t n = ds_count(p), mask = ds_mask(p, n);
    const bool occupied = n != 0u && s <= mask && p.keys[s] != kEmptyVoxelKey;
    const unsigned long long ballot = __ballot(occupied);
    if (lane == 0) s_wave[wave] = static_cast<uint32_t>(__popcll(ballot));
    const uint32_t offset = block_offset(block_counts, raw, total);
    __syncthreads();
    if (!occupied) return;
    uint32_t pos = offset + static_cast<uint32_t>(__popcll(ballot & ((1ull << lane) - 1ull)));
    for (int w = 0; w < wave; ++w) pos += s_wave[w];
    const uint32_t i = p.order[s];
    out[3 * pos] = p.in[3 * i], out[3 * pos + 1] = p.in[3 * i + 1], out[3 * pos + 2] = p.in[3 * i + 2];
    // leave the bucket as the next call must find it (free slots were never written): the host clears the table only once
    p.keys[s] = kEmptyVoxelKey, p.min_index[s] = 0xFFFFFFFFu, p.order[s] = kFreeBucket, p.home_at[s] = 0xFFFFFFFFu;
}

// ---- the pre-steps of one frame in FIVE launches (round 6) ----------------------------------------------------------------
// pipeline/KinematicICP.cpp:54-62: Preprocess + transform_points, VoxelDownsample(0.5 vs), VoxelDownsample(1.5 vs).  The steps
// above as one chain were eight launches (preprocess, compact, 2 x {claim, replay, gather}) of 5-12 us each for a few MB of
// traffic: every one of them latency, not bandwidth (a dependent kernel boundary costs ~1.5 us, a kernel's fill and drain ~4).
// What really orders them are five grid-wide facts, and each launch below ends where the next such fact is needed:
//   k_frame_pre        point -> base frame, crop flag, tile counts; the survivor claims its voxel in table A
//                      [needs: nothing.  The table's size depends on the survivor count, which no workgroup knows yet: it is
//                       SPECULATED (f.spec_mask: what the previous frame had) and verified by the next launch]
//   k_frame_l1_replay  survivor count n0 (every workgroup adds the tile counts up itself) -> speculation right?; compaction into
//                      buffer 0; the clusters of table A replayed in the reference's insertion order; occupied-bucket counts
//                      [needs: all claims]
//   k_frame_l1_gather  n1; survivors of level 1 in table order -> buffer 1; each one claims its voxel in table B
//                      [needs: all replays, all counts]
//   k_frame_l2_replay  the clusters of table B                      [needs: all claims of level 2]
//   k_frame_l2_gather  n2; survivors of level 2 -> buffer 2; the LAST workgroup to finish hands the three counts to the host
//                      (tagged words in host memory the caller polls: no copy, no stream synchronisation)
// The claim uses the point's index in the raw frame instead of its position after the crop: the order is the same, which is all
// "first point of the voxel" and the replay's insertion order look at.  A wrong guess of the table size (the survivor count
// crossed a power of two since the last frame) is found by k_frame_l1_replay before anything depends on it: the later launches
// return at once and the host runs the unfused steps from buffer 0 on.  Results: those of the unfused chain, bit for bit.
// A store that goes THROUGH the L2 to memory (relaxed, device scope: an sc1 store on gfx950).  What a workgroup hands to kernels of
// other queues - or to the host - before its launch has ended is written this way, followed by `s_waitcnt vmcnt(0)` and the
// workgroup's ticket: no release fence.  A fence writes the whole L2 back, and a few hundred of them per frame - one per workgroup
// of the ingest, the push and the last gather - slowed every kernel running beside them two- to fourfold (profiles/r06: the
// second-level replay 14 -> 48 us).
__device__ __forceinline__ void store_through(double *p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void store_through(unsigned long long *p, unsigned long long v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
struct DsTable {
    unsigned long long *keys;
    uint32_t *min_index, *order, *home_at;
};
// misc words of the chain (device): [1] range error (sticky until reported) [2] longest probe [3] ticket of the last launch
// [4] n0 [5] n1 [6] n2 [7] speculation failed [8] == seq: a claim of k_frame_pre gave up (table A too small for the frame)
struct FrameParams {
    PreprocessParams pre;  // block_counts: survivors per 256-point tile
    DsTable A, B;
    double voxel_a, voxel_b;
    uint32_t spec_mask;   // table A's mask (bucket count - 1) this chain assumes
    uint32_t tiles_pts;   // ceil(n / 256)
    uint32_t tiles_spec;  // 256-slot tiles of table A under spec_mask
    uint32_t *counts1, *counts2;  // occupied buckets per tile of table A / B
    uint32_t *misc;
    double *buf0, *buf1, *buf2;
    double *host_buf2;    // nullable: host-mapped pinned memory that receives buffer 2 as well (the registration source the pipeline returns)
    unsigned long long *host_rec;  // 5 words, each (seq << 32) | value: n0, n1, n2, longest probe, flags (1 range error, 2 speculation failed)
    uint32_t seq;
};
__device__ __forceinline__ uint32_t expected_mask(uint32_t n) {
    const uint32_t buckets = reference_bucket_count_u32(n);
    return buckets ? buckets - 1u : 0u;
}
// Every workgroup adds the per-tile counts up itself (a few loads per thread for the grids of a frame): offset = the tiles before
// `tile`, total = all `ntiles`.
__device__ __forceinline__ void tile_offset_total(const uint32_t *counts, uint32_t tile, uint32_t ntiles, uint32_t &offset, uint32_t &total) {
    __shared__ uint32_t s_before[4], s_all[4];
    uint32_t before = 0u, all = 0u;
    for (uint32_t i = threadIdx.x; i < ntiles; i += 256u) {
        const uint32_t c = counts[i];
        all += c, before += i < tile ? c : 0u;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) before += __shfl_xor(before, off, 64), all += __shfl_xor(all, off, 64);
    __syncthreads();  // (a caller that loops: the previous turn's readers are done)
    if ((threadIdx.x & 63) == 0) s_before[threadIdx.x >> 6] = before, s_all[threadIdx.x >> 6] = all;
    __syncthreads();
    offset = s_before[0] + s_before[1] + s_before[2] + s_before[3], total = s_all[0] + s_all[1] + s_all[2] + s_all[3];
}
// position of this thread's element among the workgroup's flagged ones, behind `offset` (all 256 threads call)
__device__ __forceinline__ uint32_t tile_position(bool flagged, uint32_t offset) {
    __shared__ uint32_t s_wave[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long ballot = __ballot(flagged);
    __syncthreads();
    if (lane == 0) s_wave[wave] = static_cast<uint32_t>(__popcll(ballot));
    __syncthreads();
    uint32_t pos = offset + static_cast<uint32_t>(__popcll(ballot & ((1ull << lane) - 1ull)));
    for (int w = 0; w < wave; ++w) pos += s_wave[w];
    return pos;
}
__device__ __forceinline__ unsigned long long voxel_key_of(double x, double y, double z, double vs, bool &ok, int32_t &vx, int32_t &vy, int32_t &vz) {
    vx = static_cast<int32_t>(floor(x / vs)), vy = static_cast<int32_t>(floor(y / vs)), vz = static_cast<int32_t>(floor(z / vs));
    return pack_voxel21(vx, vy, vz, ok);
}

static __global__ __launch_bounds__(256) void k_frame_pre(const FrameParams f) {
    const PreprocessParams &p = f.pre;
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i == 0) f.misc[2] = 0u, f.misc[3] = 0u, f.misc[7] = 0u, f.misc[9] = 0u;  // (raised / drawn / set by the launches behind this one)
    bool keep = false;
    double bx = 0.0, by = 0.0, bz = 0.0;
    if (i < p.n) {
        keep = preprocess_point(p, i, bx, by, bz);
        p.staged[3 * i] = bx, p.staged[3 * i + 1] = by, p.staged[3 * i + 2] = bz;
        p.flags[i] = keep ? 1u : 0u;
    }
    block_count_store(keep, p.block_counts);
    bool ok = true;
    int32_t vx = 0, vy = 0, vz = 0;
    const unsigned long long key = keep ? voxel_key_of(bx, by, bz, f.voxel_a, ok, vx, vy, vz) : kEmptyVoxelKey;
    claim_voxel(f.A.keys, f.A.min_index, f.spec_mask, key, ok, vx, vy, vz, i, f.misc + 1, f.misc + 8, f.seq);
}

// One 256-bucket tile of a claimed table: every cluster whose HEAD lies in the tile is put into the reference's order (replay_cluster's
// result, bit for bit) - with TWO round trips to memory instead of a chain of them.  The first version walked global memory: this
// bucket's key, then the one before it, then the ones behind it one by one, then replay_cluster's O(len^2) reloads of min_index and
// its read-modify-writes of order / home_at - 5 dependent accesses for a cluster of one, dozens for a cluster of four, each 2-4 us
// while the frame's push and the look-ahead decode load the memory system (the second-level replay: 10 us alone, 44 us in the drive).
// Here the tile's keys and input indices, kReplayHalo buckets behind it and the one before it are fetched at once into LDS; heads,
// lengths and the robin-hood replay work there (a cluster's LDS positions are its head thread's alone); the order goes back with one
// store per bucket.  home_at is not written (the gather resets it, nobody else reads it).  A cluster that leaves the window, and
// tables smaller than the window, take the global path.  Returns whether this thread's bucket is occupied.
constexpr uint32_t kReplayHalo = 32, kReplayWindow = 256 + kReplayHalo;
struct ReplayTile {
    unsigned long long key[kReplayWindow];
    uint32_t min[kReplayWindow], ord[kReplayWindow], home[kReplayWindow];
    uint32_t prev_occupied;
};
__device__ __forceinline__ bool replay_tile(const DsTable &T, uint32_t mask, uint32_t tile, ReplayTile &L, uint32_t *probe_max) {
    const uint32_t base = tile * 256u, i = threadIdx.x, s = base + i;
    if (mask + 1u < 2u * kReplayWindow) {  // (a small table: window positions would alias buckets)
        bool occupied = false;
        if (s <= mask) {
            occupied = T.keys[s] != kEmptyVoxelKey;
            if (occupied && T.keys[(s - 1u) & mask] == kEmptyVoxelKey) {
                uint32_t len = 1u;
                while (T.keys[(s + len) & mask] != kEmptyVoxelKey) ++len;  // ends: at least half of the buckets are free
                const uint32_t probe = replay_cluster(T.keys, T.min_index, T.order, T.home_at, mask, s, len);
                if (probe >= 32u) atomicMax(probe_max, probe);
            }
        }
        return occupied;
    }
    unsigned long long k = kEmptyVoxelKey, kh = kEmptyVoxelKey, kp = kEmptyVoxelKey;
    uint32_t m = 0xFFFFFFFFu, mh = 0xFFFFFFFFu;
    if (s <= mask) k = T.keys[s], m = T.min_index[s];
    if (i < kReplayHalo) kh = T.keys[(base + 256u + i) & mask], mh = T.min_index[(base + 256u + i) & mask];
    if (i == 255u) kp = T.keys[(base - 1u) & mask];
    __syncthreads();  // (a caller that loops over tiles: the previous tile's clusters are done with the window)
    L.key[i] = k, L.min[i] = m, L.ord[i] = kFreeBucket;
    if (i < kReplayHalo) L.key[256u + i] = kh, L.min[256u + i] = mh, L.ord[256u + i] = kFreeBucket;
    if (i == 255u) L.prev_occupied = kp != kEmptyVoxelKey ? 1u : 0u;
    __syncthreads();
    const bool occupied = k != kEmptyVoxelKey;
    const bool head = occupied && !(i ? L.key[i - 1u] != kEmptyVoxelKey : L.prev_occupied != 0u);
    if (!head) return occupied;
    uint32_t end = i + 1u;
    while (end < kReplayWindow && L.key[end] != kEmptyVoxelKey) ++end;
    uint32_t len = end - i;
    if (end == kReplayWindow) {  // the cluster leaves the window: the global walk (order / home_at of its buckets are still untouched)
        while (T.keys[(s + len) & mask] != kEmptyVoxelKey) ++len;
        const uint32_t probe = replay_cluster(T.keys, T.min_index, T.order, T.home_at, mask, s, len);
        if (probe >= 32u) atomicMax(probe_max, probe);
        return occupied;
    }
    if (len == 1u) {
        T.order[s] = m;
        return occupied;
    }
    // replay_cluster's result on window positions (kicp_table_order.hpp replay_window, CPU-tested against it): bucket (s + j) & mask <-> position i + j
    const uint32_t max_probe = replay_window(L.key, L.min, L.ord, L.home, i, end, s, mask);
    for (uint32_t j = i; j < end; ++j) T.order[(base + j) & mask] = L.ord[j];
    if (max_probe >= 32u) atomicMax(probe_max, max_probe);
    return occupied;
}

static __global__ __launch_bounds__(256) void k_frame_l1_replay(const FrameParams f) {
    const PreprocessParams &p = f.pre;
    // (this thread's point and its crop flag: fetched before the tile counts, not behind them - one round trip less)
    const uint32_t point = blockIdx.x * 256 + threadIdx.x;
    uint32_t flag = 0u;
    double sx = 0.0, sy = 0.0, sz = 0.0;
    if (blockIdx.x < f.tiles_pts && point < p.n) flag = p.flags[point], sx = p.staged[3 * point], sy = p.staged[3 * point + 1], sz = p.staged[3 * point + 2];
    uint32_t offset, n0;
    tile_offset_total(p.block_counts, blockIdx.x, f.tiles_pts, offset, n0);
    const bool spec_ok = n0 == 0u || (expected_mask(n0) == f.spec_mask && f.misc[8] != f.seq);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        f.misc[4] = n0;
        if (!spec_ok) f.misc[7] = 1u;
    }
    if (blockIdx.x < f.tiles_pts) {  // order-preserving compaction of this tile's survivors: buffer 0, the frame the pipeline returns
        const bool keep = flag != 0u;
        const uint32_t pos = tile_position(keep, offset);
        if (keep) f.buf0[3 * pos] = sx, f.buf0[3 * pos + 1] = sy, f.buf0[3 * pos + 2] = sz;
    }
    if (!spec_ok || blockIdx.x >= f.tiles_spec) return;
    __shared__ ReplayTile s_tile;
    const bool occupied = n0 != 0u && replay_tile(f.A, f.spec_mask, blockIdx.x, s_tile, f.misc + 2);  // (n0: the same for every thread)
    block_count_store(occupied, f.counts1);
}

static __global__ __launch_bounds__(256) void k_frame_l1_gather(const FrameParams f) {
    // (the bucket's key and its place in the reference's order are fetched BEFORE the tile counts, not behind them: one round trip to
    //  memory less in a chain of five, each 2-8 us while the push and the look-ahead decode run)
    const uint32_t s = blockIdx.x * 256 + threadIdx.x;
    unsigned long long bucket_key = kEmptyVoxelKey;
    uint32_t bucket_order = kFreeBucket;
    if (s <= f.spec_mask) bucket_key = f.A.keys[s], bucket_order = f.A.order[s];
    if (f.misc[7]) return;
    uint32_t offset, n1;
    tile_offset_total(f.counts1, blockIdx.x, f.tiles_spec, offset, n1);
    if (blockIdx.x == 0 && threadIdx.x == 0) f.misc[5] = n1;
    const bool occupied = f.misc[4] != 0u && bucket_key != kEmptyVoxelKey;
    const uint32_t pos = tile_position(occupied, offset);
    double x = 0.0, y = 0.0, z = 0.0;
    if (occupied) {
        const uint32_t i = bucket_order;
        x = f.pre.staged[3 * i], y = f.pre.staged[3 * i + 1], z = f.pre.staged[3 * i + 2];
        f.buf1[3 * pos] = x, f.buf1[3 * pos + 1] = y, f.buf1[3 * pos + 2] = z;
        // leave the bucket as the next frame must find it (free slots were never written)
        f.A.keys[s] = kEmptyVoxelKey, f.A.min_index[s] = 0xFFFFFFFFu, f.A.order[s] = kFreeBucket, f.A.home_at[s] = 0xFFFFFFFFu;
    }
    bool ok = true;
    int32_t vx = 0, vy = 0, vz = 0;
    const unsigned long long key = occupied ? voxel_key_of(x, y, z, f.voxel_b, ok, vx, vy, vz) : kEmptyVoxelKey;
    claim_voxel(f.B.keys, f.B.min_index, expected_mask(n1), key, ok, vx, vy, vz, pos, f.misc + 1);
}

// (table B's size is known on the device only: the launch covers an upper bound and walks the tiles there are)
static __global__ __launch_bounds__(256) void k_frame_l2_replay(const FrameParams f) {
    if (f.misc[7]) return;
    const uint32_t n1 = f.misc[5], mask = expected_mask(n1), tiles = n1 ? (mask >> 8) + 1u : 0u;
    __shared__ ReplayTile s_tile;
    for (uint32_t t = blockIdx.x; t < tiles; t += gridDim.x) {
        const bool occupied = replay_tile(f.B, mask, t, s_tile, f.misc + 2);
        block_count_store_at(occupied, f.counts2 + t);
    }
}

static __global__ __launch_bounds__(256) void k_frame_l2_gather(const FrameParams f) {
    __shared__ uint32_t s_last;
    const unsigned long long tag = static_cast<unsigned long long>(f.seq) << 32;
    auto publish = [&](int word, uint32_t v) { __hip_atomic_store(f.host_rec + word, tag | v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); };
    // (the first tile's bucket and its place in the order: fetched before anything that depends on the counts - k_frame_l1_gather;
    //  table B is never larger than table A, whose size the launch knows)
    unsigned long long first_key = kEmptyVoxelKey;
    uint32_t first_order = kFreeBucket;
    if (blockIdx.x * 256u + threadIdx.x <= f.spec_mask) first_key = f.B.keys[blockIdx.x * 256u + threadIdx.x], first_order = f.B.order[blockIdx.x * 256u + threadIdx.x];
    if (f.misc[7]) {  // the table-size guess was wrong: nothing behind buffer 0 was done; the host takes the unfused steps from there
        if (blockIdx.x == 0 && threadIdx.x == 0) publish(0, f.misc[4]), publish(1, 0u), publish(2, 0u), publish(3, 0u), publish(4, 2u | (f.misc[1] ? 1u : 0u));
        return;
    }
    const uint32_t n1 = f.misc[5], mask = expected_mask(n1), tiles = n1 ? (mask >> 8) + 1u : 0u;
    uint32_t n2_seen = 0xFFFFFFFFu;
    for (uint32_t t = blockIdx.x; t < tiles; t += gridDim.x) {
        uint32_t offset, n2;
        const uint32_t s = t * 256 + threadIdx.x;
        const bool mine = t == blockIdx.x;  // (this workgroup's first tile: fetched at the top of the kernel)
        const unsigned long long bucket_key = mine ? first_key : (s <= mask ? f.B.keys[s] : kEmptyVoxelKey);
        const uint32_t bucket_order = mine ? first_order : (s <= mask ? f.B.order[s] : kFreeBucket);
        tile_offset_total(f.counts2, t, tiles, offset, n2);
        n2_seen = n2;
        const bool occupied = s <= mask && bucket_key != kEmptyVoxelKey;
        const uint32_t pos = tile_position(occupied, offset);
        if (occupied) {
            const uint32_t i = bucket_order;
            const double x = f.buf1[3 * i], y = f.buf1[3 * i + 1], z = f.buf1[3 * i + 2];
            store_through(f.buf2 + 3 * pos, x), store_through(f.buf2 + 3 * pos + 1, y), store_through(f.buf2 + 3 * pos + 2, z);
            f.B.keys[s] = kEmptyVoxelKey, f.B.min_index[s] = 0xFFFFFFFFu, f.B.order[s] = kFreeBucket, f.B.home_at[s] = 0xFFFFFFFFu;
        }
    }
    // The workgroup that finishes LAST tells the host (which polls the record instead of synchronising the stream).  Buffer 2 goes
    // to a kernel on another queue next: its stores went through to memory (store_through) and every wave waits for them, then the
    // workgroup draws its ticket.  (The copy of buffer 2 in host memory is k_frame_src_host's: written here, its PCIe writes queued
    // up behind the frame push's and a system-scope release held the record back by ~10 us.)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        s_last = __hip_atomic_fetch_add(f.misc + 3, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1u ? 1u : 0u;
    }
    __syncthreads();
    if (!s_last) return;
    uint32_t offset, n2 = n2_seen;
    if (n2_seen == 0xFFFFFFFFu) tile_offset_total(f.counts2, 0u, tiles, offset, n2);  // (a workgroup without a tile of its own; n2_seen: the same for every thread)
    if (threadIdx.x == 0) {
        f.misc[6] = n2;
        publish(0, f.misc[4]), publish(1, n1), publish(2, n2), publish(3, f.misc[2]), publish(4, f.misc[1] ? 1u : 0u);
    }
}

// Buffer 2 - the registration source, ~1 300 points, the second cloud RegisterFrame returns - into host memory: a launch of its own
// behind the chain, announced by word 5 of the host record ((seq << 32) | points), which only kicp_pre_download(2) waits for.
// misc[9]: its ticket (reset by k_frame_pre).
static __global__ __launch_bounds__(256) void k_frame_src_host(const FrameParams f) {
    __shared__ uint32_t s_last;
    if (f.misc[7]) return;  // (the guess was wrong: the host takes the unfused steps and downloads buffer 2 itself)
    const uint32_t n2 = f.misc[6];
    for (uint32_t o = blockIdx.x * 256u + threadIdx.x; o < 3u * n2; o += gridDim.x * 256u) f.host_buf2[o] = f.buf2[o];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        // system scope: this workgroup's bytes are in host memory before its ticket - and so before the flag, whoever writes it
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        s_last = __hip_atomic_fetch_add(f.misc + 9, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1u ? 1u : 0u;
        if (s_last) __hip_atomic_store(f.host_rec + 5, (static_cast<unsigned long long>(f.seq) << 32) | n2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// (round 6 tried the second level in ONE workgroup - replay, ballot bitmap + scan in LDS, gather, as a fourth and last launch: its chains
//  of dependent loads - cluster walk, order -> point -> store - take as long on one CU as the two launches do on thirty, 100 vs 140 us
//  per frame with a contiguous bucket range per thread, no gain with coalesced rows; removed.)

// ---- the preprocessed frame on its way back to the caller (pipeline/KinematicICP.cpp:84 returns it) ------------------------
// 3 MB that nothing on the device waits for.  A kernel on a stream of its own PUSHES buffer 0 into host-mapped pinned memory, 16 bytes
// per lane (PCIe writes: ~46 GB/s with 64 workgroups, tools/micro/d2h.hip; the DMA engine reaches that only in ONE piece, and every
// piece it is cut into costs ~10 us of API calls and engine start-up), piece by piece: the workgroup that completes a piece says so
// in host memory - (seq << 32) | bytes of the piece -, and the handle's helper thread copies that piece into the caller's vector
// while the next ones are still crossing PCIe.  The point count is read on the device (the launch is queued before the host knows it).
constexpr int kPushPieces = 8;
struct PushParams {
    const unsigned char *src;       // buffer 0
    unsigned char *dst;             // pinned landing area as the device sees it
    const uint32_t *n_points;       // the chain's survivor count (misc[4])
    uint32_t piece_bytes;           // a multiple of 16; kPushPieces pieces cover the largest frame the handle holds
    unsigned long long *tickets;    // [kPushPieces] device counters, never reset
    unsigned long long ticket_done; // their value once this launch's last workgroup has drawn
    unsigned long long *host_flags; // [kPushPieces] pinned
    uint32_t seq;
};
static __global__ __launch_bounds__(256) void k_push_frame(const PushParams q) {
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    __shared__ uint32_t s_last;
    const size_t bytes = static_cast<size_t>(*q.n_points) * 24u;
    for (int piece = 0; piece < kPushPieces; ++piece) {
        const size_t lo = static_cast<size_t>(piece) * q.piece_bytes, hi = lo + q.piece_bytes < bytes ? lo + q.piece_bytes : bytes;
        const size_t len = hi > lo ? hi - lo : 0u;
        for (size_t o = (static_cast<size_t>(blockIdx.x) * 256u + threadIdx.x) * 16u; o < len; o += static_cast<size_t>(gridDim.x) * 4096u) {
            if (o + 16u <= len) *reinterpret_cast<u32x4 *>(q.dst + lo + o) = *reinterpret_cast<const u32x4 *>(q.src + lo + o);
            else
                for (size_t k = o; k < len; ++k) q.dst[lo + k] = q.src[lo + k];
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
            // system scope: this workgroup's bytes are in host memory before its ticket - and so before the flag, whoever writes it
            // (without the fence the flag of another workgroup overtook the bytes: measured)
            // (a timing run WITHOUT this fence left the launches beside the push as slow as with it: the fence is not what they pay for)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            s_last = __hip_atomic_fetch_add(q.tickets + piece, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1ull == q.ticket_done ? 1u : 0u;
            if (s_last) __hip_atomic_store(q.host_flags + piece, (static_cast<unsigned long long>(q.seq) << 32) | static_cast<unsigned long long>(len), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        __syncthreads();
    }
}

// ---- PointCloud2 wire-format ingest ----------------------------------------------------
// ros/src/kinematic_icp_ros/utils/RosUtils.cpp:30-39 (PointCloud2ToEigen: float32 x,y,z at point_step stride -> fp64,
// T * p) and TimeStampHandler.cpp:57-104 (per-point stamp of type uint32 / float32 / float64 -> double seconds, values
// with more than 10 integer digits are nanoseconds) + :106,:121-128 (min/max, normalisation to [0,1]).
// The raw message bytes are what crosses PCIe (point_step bytes per point instead of 32 B of inflated fp64).
struct IngestParams {
    const unsigned char *raw;  // first record of THIS launch's piece: pinned host memory as the device sees it (the records are decoded
                               // straight out of the staging buffer, round 6: no copy of the bytes in HBM), or HBM where that is unavailable
    uint32_t first;            // index of that record in the cloud (a multiple of 256)
    uint32_t n;                // records of the piece
    uint32_t point_step;
    uint32_t off_x, off_y, off_z, off_t;
    int32_t stamp_type;  // 0 none, 6 UINT32, 7 FLOAT32, 8 FLOAT64 (sensor_msgs::msg::PointField datatype codes)
    int32_t transform;   // 0: identity (what LidarOdometryServer.cpp:203 passes)
    int32_t aligned;     // every field sits at a multiple of its size (base pointer, point_step and offsets): plain loads instead of byte-wise ones
    Pose T;
    double *out_xyz;     // the whole cloud's arrays
    double *out_stamps;  // seconds, NOT normalised: the extrema go to the host, which hands them to whoever consumes the stamps
    unsigned long long *block_minmax;  // [workgroups of the cloud][2] the stamps' extrema per workgroup as order-preserving integer keys
    // the cloud's pieces are launches of their own (each behind the CPU's copy of its bytes into the staging buffer); the workgroup
    // that finishes LAST - over all pieces - folds the extrema and tells the host, which polls `host_rec`
    unsigned long long *ticket;      // device counter, never reset
    unsigned long long ticket_done;  // its value once the cloud's last workgroup has drawn
    uint32_t total_blocks;
    unsigned long long *host_rec;    // pinned: [0] min key [1] max key [2] seq
    unsigned long long seq;
};

template <typename T>
__device__ __forceinline__ T load_unaligned(const unsigned char *p) {
    T v;
    __builtin_memcpy(&v, p, sizeof(T));
    return v;
}
// order-preserving map double -> uint64 (and back), so the block extrema can be merged as integers
__device__ __forceinline__ unsigned long long ordered_key(double v) {
    const unsigned long long b = static_cast<unsigned long long>(__double_as_longlong(v));
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
KICP_HD double ordered_value(unsigned long long k) {
    const unsigned long long b = (k >> 63) ? (k & 0x7FFFFFFFFFFFFFFFull) : ~k;
    double v;
    __builtin_memcpy(&v, &b, 8);
    return v;
}

template <typename T>
__device__ __forceinline__ T load_field(const unsigned char *p, bool aligned) {
    return aligned ? *reinterpret_cast<const T *>(p) : load_unaligned<T>(p);
}
static __global__ __launch_bounds__(256) void k_ingest(const IngestParams p) {
    __shared__ unsigned long long s_min[4], s_max[4];
    __shared__ uint32_t s_last;
    unsigned long long kmin = ~0ull, kmax = 0ull;
    const bool al = p.aligned != 0;  // (wave-uniform: the usual PointCloud2 layouts are naturally aligned)
    // A workgroup's 256 records are contiguous bytes: they cross PCIe as full 16-byte loads per lane (one request per 64 bytes
    // whatever the field layout is; four 4-byte field loads per record were four times the requests and 17 GB/s) into LDS and are
    // picked apart there.  Records longer than kLdsStep bytes are read field by field from where they are.
    // The launch's workgroups go round its tiles of 256 records: this call's message has one workgroup per tile; a look-ahead message
    // gets a few dozen (kicp_prestep.hip ingest_run) - with all of its 2 MB requested at once, every kernel the frame dispatched in
    // the meantime waited ~50 us for ITS packet and arguments to cross the same PCIe read queue.
    constexpr uint32_t kLdsStep = 128;
    __shared__ __attribute__((aligned(16))) unsigned char s_rec[256 * kLdsStep];
    const bool via_lds = p.point_step <= kLdsStep;
    const uint32_t tiles = (p.n + 255u) / 256u;
    for (uint32_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const uint32_t first_local = tile * 256u, local = first_local + threadIdx.x, i = p.first + local;
        if (via_lds) {
            typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
            const uint32_t wg_bytes = (p.n - first_local < 256u ? p.n - first_local : 256u) * p.point_step;
            const unsigned char *src = p.raw + static_cast<size_t>(first_local) * p.point_step;
            if (tile != blockIdx.x) __syncthreads();  // (the previous tile's records have been picked apart)
            for (uint32_t o = threadIdx.x * 16u; o < wg_bytes; o += 4096u) {
                if (o + 16u <= wg_bytes) *reinterpret_cast<u32x4 *>(s_rec + o) = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(src + o));
                else
                    for (uint32_t k = o; k < wg_bytes; ++k) s_rec[k] = src[k];
            }
            __syncthreads();
        }
        kmin = ~0ull, kmax = 0ull;
        if (local < p.n) {
            const unsigned char *rec = via_lds ? s_rec + threadIdx.x * p.point_step : p.raw + static_cast<size_t>(local) * p.point_step;
            double x = static_cast<double>(load_field<float>(rec + p.off_x, al));
            double y = static_cast<double>(load_field<float>(rec + p.off_y, al));
            double z = static_cast<double>(load_field<float>(rec + p.off_z, al));
            if (p.transform) {
                double rx, ry, rz;
                quat_rotate(p.T, x, y, z, rx, ry, rz);
                x = rx + p.T.tx, y = ry + p.T.ty, z = rz + p.T.tz;
            }
            store_through(p.out_xyz + 3 * i, x), store_through(p.out_xyz + 3 * i + 1, y), store_through(p.out_xyz + 3 * i + 2, z);
            if (p.stamp_type) {
                double stamp;
                if (p.stamp_type == 6) stamp = static_cast<double>(load_field<uint32_t>(rec + p.off_t, al));
                else if (p.stamp_type == 7) stamp = static_cast<double>(load_field<float>(rec + p.off_t, al));
                else stamp = load_field<double>(rec + p.off_t, al);
                // TimeStampHandler.cpp:60-63,73-78: floor(log10(uint64(round(stamp))) + 1) > 10  <=>  round(stamp) >= 1e10
                if (round(stamp) >= 1e10) stamp *= 1e-9;
                store_through(p.out_stamps + i, stamp);
                kmin = kmax = ordered_key(stamp);
            }
        }
        if (p.stamp_type) {
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                const unsigned long long a = __shfl_xor(kmin, off, 64), b = __shfl_xor(kmax, off, 64);
                kmin = a < kmin ? a : kmin, kmax = b > kmax ? b : kmax;
            }
            if (tile != blockIdx.x) __syncthreads();  // (the previous tile's extrema have been folded)
            if ((threadIdx.x & 63) == 0) s_min[threadIdx.x >> 6] = kmin, s_max[threadIdx.x >> 6] = kmax;
            __syncthreads();
            if (threadIdx.x == 0) {
                for (int w = 1; w < 4; ++w) kmin = s_min[w] < kmin ? s_min[w] : kmin, kmax = s_max[w] > kmax ? s_max[w] : kmax;
                const uint32_t gb = p.first / 256u + tile;
                store_through(p.block_minmax + 2 * gb, kmin), store_through(p.block_minmax + 2 * gb + 1, kmax);
            }
        }
    }
    // The decoded cloud is read by kernels of other launches (and, after a look-ahead upload, of another stream) once the host has
    // seen the record: every store above went through to memory (store_through); every wave waits for its own, then the workgroup
    // draws its ticket.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) s_last = __hip_atomic_fetch_add(p.ticket, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1ull == p.ticket_done ? 1u : 0u;
    __syncthreads();
    if (!s_last) return;
    // TimeStampHandler.cpp:106: the extrema over the whole cloud
    kmin = ~0ull, kmax = 0ull;
    if (p.stamp_type) {
        for (uint32_t b = threadIdx.x; b < p.total_blocks; b += 256u) {
            const unsigned long long a = __hip_atomic_load(p.block_minmax + 2 * b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned long long c = __hip_atomic_load(p.block_minmax + 2 * b + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            kmin = a < kmin ? a : kmin, kmax = c > kmax ? c : kmax;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const unsigned long long a = __shfl_xor(kmin, off, 64), c = __shfl_xor(kmax, off, 64);
            kmin = a < kmin ? a : kmin, kmax = c > kmax ? c : kmax;
        }
        __syncthreads();
        if ((threadIdx.x & 63) == 0) s_min[threadIdx.x >> 6] = kmin, s_max[threadIdx.x >> 6] = kmax;
        __syncthreads();
        for (int w = 0; w < 4; ++w) kmin = s_min[w] < kmin ? s_min[w] : kmin, kmax = s_max[w] > kmax ? s_max[w] : kmax;
    }
    if (threadIdx.x == 0) {
        __hip_atomic_store(p.host_rec + 0, kmin, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(p.host_rec + 1, kmax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __threadfence_system();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(p.host_rec + 2, p.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

}  // namespace kicp
