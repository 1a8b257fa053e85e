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

// block-wide count of set predicates -> block_counts[blockIdx.x] (lane 0 of wave 0 writes)
__device__ __forceinline__ void block_count_store(bool pred, uint32_t *block_counts) {
    __shared__ uint32_t s_cnt[4];
    const unsigned long long ballot = __ballot(pred);
    if ((threadIdx.x & 63) == 0) s_cnt[threadIdx.x >> 6] = static_cast<uint32_t>(__popcll(ballot));
    __syncthreads();
    if (threadIdx.x == 0) block_counts[blockIdx.x] = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
}

static __global__ __launch_bounds__(256) void k_preprocess(const PreprocessParams p) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    bool keep = false;
    if (i < p.n) {
        double x = p.in[3 * i], y = p.in[3 * i + 1], z = p.in[3 * i + 2];
        if (p.deskew) {  // p' = (relative_motion^-1 * exp(t_i * omega)) * p_i : deskew to the scan end
            const double t = p.timestamps[i];
            const double xi[6] = {t * p.omega[0], t * p.omega[1], t * p.omega[2], t * p.omega[3], t * p.omega[4], t * p.omega[5]};
            const Pose M = pose_mul(p.motion_inverse, pose_exp(xi));
            double rx, ry, rz;
            quat_rotate(M, x, y, z, rx, ry, rz);
            x = rx + M.tx, y = ry + M.ty, z = rz + M.tz;
        }
        const double r = sqrt(x * x + y * y + z * z);
        keep = r < p.max_range && r > p.min_range;  // strict on both sides
        double bx, by, bz;
        quat_rotate(p.lidar_to_base, x, y, z, bx, by, bz);  // transform_points: into the base frame
        p.staged[3 * i] = bx + p.lidar_to_base.tx, p.staged[3 * i + 1] = by + p.lidar_to_base.ty, p.staged[3 * i + 2] = bz + p.lidar_to_base.tz;
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
    bool ok;
    const unsigned long long key = in_range ? pack_voxel21(vx, vy, vz, ok) : kEmptyVoxelKey;
    // Consecutive points of a scan ring fall into the same voxel in long runs (hundreds of points per voxel close to the sensor),
    // and every one of them would hammer the same two words with atomics - the kernel's tail, up to 0.6 ms on some frames.  Only
    // the FIRST lane of a run inside a wave goes on: it has the lowest index of the run, which is all atomicMin would keep.
    const unsigned long long prev = __shfl_up(key, 1, 64);
    if (!in_range || ((threadIdx.x & 63) != 0 && prev == key)) return;
    if (!ok) {
        *p.error = 1u;
        return;
    }
    uint32_t slot = reference_voxel_hash(vx, vy, vz) & mask;
    for (;;) {
        const unsigned long long seen = atomicCAS(p.keys + slot, kEmptyVoxelKey, key);
        if (seen == kEmptyVoxelKey || seen == key) break;
        slot = (slot + 1) & mask;
    }
    atomicMin(p.min_index + slot, i);
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

// ---- PointCloud2 wire-format ingest ----------------------------------------------------
// ros/src/kinematic_icp_ros/utils/RosUtils.cpp:30-39 (PointCloud2ToEigen: float32 x,y,z at point_step stride -> fp64,
// T * p) and TimeStampHandler.cpp:57-104 (per-point stamp of type uint32 / float32 / float64 -> double seconds, values
// with more than 10 integer digits are nanoseconds) + :106,:121-128 (min/max, normalisation to [0,1]).
// The raw message bytes are what crosses PCIe (point_step bytes per point instead of 32 B of inflated fp64).
struct IngestParams {
    const unsigned char *raw;
    uint32_t n, point_step;
    uint32_t off_x, off_y, off_z, off_t;
    int32_t stamp_type;  // 0 none, 6 UINT32, 7 FLOAT32, 8 FLOAT64 (sensor_msgs::msg::PointField datatype codes)
    int32_t transform;   // 0: identity (what LidarOdometryServer.cpp:203 passes)
    int32_t aligned;     // every field sits at a multiple of its size (base pointer, point_step and offsets): plain loads instead of byte-wise ones
    Pose T;
    double *out_xyz;
    double *out_stamps;
    unsigned long long *minmax;  // [0] min, [1] max of the stamps as order-preserving integer keys
    unsigned long long *block_minmax;  // nullable: [gridDim.x][2] extrema per workgroup instead of two atomics per workgroup on ONE pair of words
                                       // (512 workgroups queueing on them were 10 of this kernel's 15 us); k_normalize_stamps folds them
};

template <typename T>
__device__ __forceinline__ T load_unaligned(const unsigned char *p) {
    T v;
    __builtin_memcpy(&v, p, sizeof(T));
    return v;
}
// order-preserving map double -> uint64 (and back), so the block extrema can be merged with integer atomics
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
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    unsigned long long kmin = ~0ull, kmax = 0ull;
    const bool al = p.aligned != 0;  // (wave-uniform: the usual PointCloud2 layouts are naturally aligned)
    if (i < p.n) {
        const unsigned char *rec = p.raw + static_cast<size_t>(i) * p.point_step;
        double x = static_cast<double>(load_field<float>(rec + p.off_x, al));
        double y = static_cast<double>(load_field<float>(rec + p.off_y, al));
        double z = static_cast<double>(load_field<float>(rec + p.off_z, al));
        if (p.transform) {
            double rx, ry, rz;
            quat_rotate(p.T, x, y, z, rx, ry, rz);
            x = rx + p.T.tx, y = ry + p.T.ty, z = rz + p.T.tz;
        }
        p.out_xyz[3 * i] = x, p.out_xyz[3 * i + 1] = y, p.out_xyz[3 * i + 2] = z;
        if (p.stamp_type) {
            double stamp;
            if (p.stamp_type == 6) stamp = static_cast<double>(load_field<uint32_t>(rec + p.off_t, al));
            else if (p.stamp_type == 7) stamp = static_cast<double>(load_field<float>(rec + p.off_t, al));
            else stamp = load_field<double>(rec + p.off_t, al);
            // TimeStampHandler.cpp:60-63,73-78: floor(log10(uint64(round(stamp))) + 1) > 10  <=>  round(stamp) >= 1e10
            if (round(stamp) >= 1e10) stamp *= 1e-9;
            p.out_stamps[i] = stamp;
            kmin = kmax = ordered_key(stamp);
        }
    }
    if (!p.stamp_type) return;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned long long a = __shfl_xor(kmin, off, 64), b = __shfl_xor(kmax, off, 64);
        kmin = a < kmin ? a : kmin, kmax = b > kmax ? b : kmax;
    }
    if ((threadIdx.x & 63) == 0) s_min[threadIdx.x >> 6] = kmin, s_max[threadIdx.x >> 6] = kmax;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) kmin = s_min[w] < kmin ? s_min[w] : kmin, kmax = s_max[w] > kmax ? s_max[w] : kmax;
        if (p.block_minmax) p.block_minmax[2 * blockIdx.x] = kmin, p.block_minmax[2 * blockIdx.x + 1] = kmax;
        else atomicMin(p.minmax, kmin), atomicMax(p.minmax + 1, kmax);
    }
}
// TimeStampHandler.cpp:121-128: (t - min) / (max - min), the same two fp64 operations
// (`block_minmax` != nullptr: the extrema are still spread over k_ingest's workgroups - every workgroup folds the `nblocks` pairs
//  itself, workgroup 0 leaves the result in minmax[] for the host)
static __global__ __launch_bounds__(256) void k_normalize_stamps(double *stamps, uint32_t n, unsigned long long *minmax, const unsigned long long *block_minmax,
                                                          uint32_t nblocks) {
    __shared__ unsigned long long s_lo[4], s_hi[4];
    unsigned long long klo, khi;
    if (block_minmax) {
        klo = ~0ull, khi = 0ull;
        for (uint32_t b = threadIdx.x; b < nblocks; b += 256u) {
            const unsigned long long a = block_minmax[2 * b], c = block_minmax[2 * b + 1];
            klo = a < klo ? a : klo, khi = c > khi ? c : khi;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const unsigned long long a = __shfl_xor(klo, off, 64), c = __shfl_xor(khi, off, 64);
            klo = a < klo ? a : klo, khi = c > khi ? c : khi;
        }
        if ((threadIdx.x & 63) == 0) s_lo[threadIdx.x >> 6] = klo, s_hi[threadIdx.x >> 6] = khi;
        __syncthreads();
        for (int w = 0; w < 4; ++w) klo = s_lo[w] < klo ? s_lo[w] : klo, khi = s_hi[w] > khi ? s_hi[w] : khi;
        if (blockIdx.x == 0 && threadIdx.x == 0) minmax[0] = klo, minmax[1] = khi;
    } else {
        klo = minmax[0], khi = minmax[1];
    }
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double lo = ordered_value(klo), hi = ordered_value(khi);
    stamps[i] = (stamps[i] - lo) / (hi - lo);
}

}  // namespace kicp
