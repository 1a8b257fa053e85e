// kicp_common.hpp -- layouts and small helpers shared by the host map and the gfx950 kernels.
#pragma once
#include <cstdint>
#include <hip/hip_runtime.h>

#define KICP_HD __host__ __device__ __forceinline__

namespace kicp {

// One slot of the open-addressing voxel table: 128 B = one cache line.
//   key : the reference's Voxel = Eigen::Vector3i (kiss-icp v1.2.0 core/VoxelUtils.hpp)
//   val : (bucket_index << cb) | point_count, cb = the map's count bits (count_bits_for: 8 for max_points_per_voxel <= 255,
//         12 up to 4 095, 16 up to 65 535); kEmptyVal marks a free slot; halo_val(cb) an entry without points.
//   nbr : bit s set <=> voxel key + shift[s] holds points (s in the reference's visiting order, bit 0 = this voxel).
//   nb  : bucket index of voxel key + shift[s] for every set bit of nbr (that bucket's point count sits in the
//         bucket itself, see MapView::pool16).
// Besides the occupied voxels the table holds "halo" entries for every empty voxel that has an occupied neighbour,
// so ONE probe at a query's own voxel yields the buckets of all 27 neighbours: empty space is never probed, and
// neighbour voxels need no probe of their own.
struct alignas(128) Slot {
    int32_t x, y, z;
    uint32_t val;
    uint32_t nbr;
    uint32_t nb[27];
};
static_assert(sizeof(Slot) == 128, "Slot must be one cache line");
constexpr uint32_t kEmptyVal = 0xFFFFFFFFu;
constexpr uint32_t kMaxPointsPerVoxel = 65535;  // (the reference's max_points_per_voxel is a plain unsigned int, KinematicICP.hpp:43)
// The split of `val` follows the map's max_points_per_voxel: the default 20 (any value <= 255) leaves 24 bits = 16.7M voxels;
// larger buckets trade voxels for points (4 095 points: 1M voxels; 65 535 points: 65 534 voxels).
KICP_HD uint32_t count_bits_for(uint32_t cap) { return cap <= 255u ? 8u : (cap <= 4095u ? 12u : 16u); }
KICP_HD uint32_t val_count(uint32_t val, uint32_t cb) { return val & ((1u << cb) - 1u); }
KICP_HD uint32_t val_bucket(uint32_t val, uint32_t cb) { return val >> cb; }
KICP_HD uint32_t make_val(uint32_t bucket, uint32_t count, uint32_t cb) { return (bucket << cb) | count; }
KICP_HD uint32_t halo_val(uint32_t cb) { return 0xFFFFFFFFu << cb; }            // no bucket, zero points
KICP_HD uint32_t max_buckets(uint32_t cb) { return (1u << (32u - cb)) - 2u; }   // (the all-ones bucket index belongs to halo / free slots)
constexpr uint32_t kOrdStride = 65536;  // visiting-order key of a candidate = shift * kOrdStride + position inside its bucket

// Table hash.  The reference's std::hash<Voxel> (three-prime XOR, App. A.1) followed by a murmur3
// finaliser so that a power-of-two mask sees well mixed low bits.  The hash only decides slot positions;
// it can never change a query result.
KICP_HD uint32_t voxel_hash(int32_t x, int32_t y, int32_t z) {
    uint32_t h = (static_cast<uint32_t>(x) * 73856093u) ^ (static_cast<uint32_t>(y) * 19349669u) ^
                 (static_cast<uint32_t>(z) * 83492791u);
    h ^= h >> 16;
    h *= 0x85ebca6bu;
    h ^= h >> 13;
    h *= 0xc2b2ae35u;
    h ^= h >> 16;
    return h;
}

// One point of the compact mirror the pre-selection pass reads: the offset from the voxel corner, per axis, in units of
// voxel_size / 65536 (so the quantisation error is <= 1 unit = 1.5e-5 voxel sizes wherever the map is), 8 bytes:
//   x = qx | qy << 16,  y = qz | aux << 16;  aux = 0 for a point, 0xFFFF for an empty slot of the bucket.
// The pass kernel converts the whole second word to float: a point yields its z offset exactly, an empty slot 4.3e9 units -
// farther than anything real, so empty slots lose every comparison without being tested for.  A bucket's mirror is reset to
// empty slots (mirror_empty) when the bucket receives its first point.
// It only PRE-SELECTS: the winner, and anything within the error margin of it, is re-evaluated from the fp64 pool.
using MirrorPoint = uint2;
KICP_HD MirrorPoint mirror_empty() {
    MirrorPoint m;
    m.x = 0u, m.y = 0xFFFF0000u;
    return m;
}
KICP_HD uint32_t mirror_quant(double offset, double units_per_metre) {  // round to nearest unit, clamped into 16 bits
    const double q = floor(offset * units_per_metre + 0.5);
    return q <= 0.0 ? 0u : (q >= 65535.0 ? 65535u : static_cast<uint32_t>(q));
}
KICP_HD MirrorPoint mirror_point(double ox, double oy, double oz, double units_per_metre) {
    MirrorPoint m;
    m.x = mirror_quant(ox, units_per_metre) | (mirror_quant(oy, units_per_metre) << 16);
    m.y = mirror_quant(oz, units_per_metre);
    return m;
}
KICP_HD double mirror_units_per_metre(double voxel_size) { return 65536.0 / voxel_size; }

// Device view of a voxel map mirror (all pointers in HBM).
struct MapView {
    const Slot *table;    // capacity = mask + 1 (power of two), linear probing, no tombstones
    uint32_t mask;
    const double *pool;   // bucket b holds <= cap points at pool + b * cap * 3 (AoS xyz, insertion order)
    const MirrorPoint *pool16; // 16-bit mirror: point k of bucket b at pool16[b * cap16 + k] (see MirrorPoint); slots beyond
                               // the bucket's point count are marked empty
    uint32_t cap;         // max_points_per_voxel = bucket stride of `pool` in points
    uint32_t cap16;       // bucket stride of `pool16` in points: cap rounded up to a multiple of kMirrorTrip (mirror_stride)
    double voxel_size;
    uint32_t cbits;       // count bits of Slot::val (count_bits_for(cap))
};
// The pass kernel reads a mirror bucket kMirrorTrip points (= kMirrorTrip / 2 16-byte loads at immediate offsets) at a time
// without ever looking at the count first; the stride is padded so that such a trip never leaves the bucket.
constexpr uint32_t kMirrorTrip = 20;
KICP_HD uint32_t mirror_stride(uint32_t cap) { return (cap + kMirrorTrip - 1u) / kMirrorTrip * kMirrorTrip; }

// The reference's neighbour visiting order (kiss-icp v1.2.0 core/VoxelHashMap.cpp `voxel_shifts`; App. A.3),
// packed 2 bits per axis (value+1) so it lives in three 64-bit immediates instead of a constant-memory table.
//   shift s: dx = ((kShiftX >> 2s) & 3) - 1, ...
constexpr int kShiftTable[27][3] = {{0, 0, 0},   {1, 0, 0},   {-1, 0, 0},  {0, 1, 0},   {0, -1, 0},  {0, 0, 1},   {0, 0, -1},
                                    {1, 1, 0},   {1, -1, 0},  {-1, 1, 0},  {-1, -1, 0}, {1, 0, 1},   {1, 0, -1},  {-1, 0, 1},
                                    {-1, 0, -1}, {0, 1, 1},   {0, 1, -1},  {0, -1, 1},  {0, -1, -1}, {1, 1, 1},   {1, 1, -1},
                                    {1, -1, 1},  {1, -1, -1}, {-1, 1, 1},  {-1, 1, -1}, {-1, -1, 1}, {-1, -1, -1}};
constexpr uint64_t pack_axis(int axis) {
    uint64_t v = 0;
    for (int s = 0; s < 27; ++s) v |= static_cast<uint64_t>(kShiftTable[s][axis] + 1) << (2 * s);
    return v;
}
constexpr uint64_t kShiftX = pack_axis(0), kShiftY = pack_axis(1), kShiftZ = pack_axis(2);
KICP_HD int shift_component(uint64_t packed, int s) { return static_cast<int>((packed >> (2 * s)) & 3u) - 1; }

// counters of the device-maintained map (kicp_mapdev.hpp), mirrored on the host after every device-side update
struct DevMapCounters {
    unsigned long long n_points;
    uint32_t n_voxels, n_entries, n_buckets_hi, free_count;
    uint32_t touched, error, may_occupy, pad1;  // may_occupy: touched voxels without points (k_up_scan)
};

}  // namespace kicp
