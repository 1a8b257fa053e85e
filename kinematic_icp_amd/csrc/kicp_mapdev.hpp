// kicp_mapdev.hpp -- VoxelHashMap::Update(points, pose) on the GPU (kiss-icp v1.2.0
// core/VoxelHashMap.cpp: transform, AddPoints, RemovePointsFarFromLocation; call site pipeline/KinematicICP.cpp:79).
//
// AddPoints is order dependent (first come first kept: a point is dropped when its voxel is full or an earlier point
// of the same voxel lies closer than map_resolution), but the dependence never crosses a voxel.  So the new points are
// grouped by voxel and every touched voxel is processed by ONE thread that walks its group in input-index order with
// the reference's fp64 arithmetic: the accepted set and the order inside each bucket are exactly the sequential
// reference's, while voxels proceed in parallel.  Table slots are claimed with a 64-bit CAS on a parallel array of
// packed voxel keys (voxel coordinates within +-2^20: beyond that the host map takes the update over - the reference has no
// such limit); first-time occupation updates the 27 neighbour masks / bucket
// records with atomics, creating halo entries on demand.  Table growth and dead-entry cleanup are a device-side re-hash
// (bottom of this file); the pools grow with device-to-device copies (kicp_map.hip).
#pragma once
#include "kicp_common.hpp"
#include "kicp_se3.hpp"

namespace kicp {

constexpr unsigned long long kEmptyKey64 = ~0ull;

struct DevMap {
    Slot *table;
    unsigned long long *keys64;  // packed key of every live entry, kEmptyKey64 otherwise (find-or-insert by CAS)
    uint32_t mask;
    double *pool;
    MirrorPoint *pool16;
    uint32_t cap;
    uint32_t cbits;              // count bits of Slot::val (count_bits_for(cap))
    uint32_t bucket_capacity;    // buckets the pools can hold
    double voxel_size, max_distance;
    uint32_t *free_list;         // stack of reusable bucket ids
    uint32_t *cnt;               // per slot: new points of the running update (zero between updates)
    uint32_t *seg_start;         // per slot: start of the voxel's group in `order`
    DevMapCounters *ctr;
};

struct UpdateParams {
    DevMap m;
    const double *in;  // points in the local frame (device)
    uint32_t n;
    Pose pose;
    double *world;     // [n*3] transformed points
    uint32_t *slot_of; // [n]
    uint32_t *order;   // [n] point indices grouped by voxel
    uint32_t *touched; // [n] slots that received points
};

__device__ __forceinline__ unsigned long long pack_key64(int32_t x, int32_t y, int32_t z, bool &ok) {
    const int lim = 1 << 20;
    ok = x >= -lim && x < lim && y >= -lim && y < lim && z >= -lim && z < lim;
    return (static_cast<unsigned long long>(static_cast<uint32_t>(z + lim) & 0x1FFFFFu) << 42) |
           (static_cast<unsigned long long>(static_cast<uint32_t>(y + lim) & 0x1FFFFFu) << 21) |
           static_cast<unsigned long long>(static_cast<uint32_t>(x + lim) & 0x1FFFFFu);
}

// slot of voxel (x,y,z); creates a halo entry when absent.  Same probe sequence as the host map (voxel_hash, linear).
// The host keeps the table below a load factor of 0.75 (head-room check in map_update_device), so a probe sequence is
// short; should the table ever be full all the same, the walk stops after one lap, raises ctr->error = 3 and returns
// kNoSlot (callers skip their writes; the host reports KICP_ERR_CAPACITY) instead of spinning for ever.
constexpr uint32_t kNoSlot = 0xFFFFFFFFu;
__device__ __forceinline__ uint32_t dev_find_or_insert(const DevMap &m, int32_t x, int32_t y, int32_t z) {
    bool ok;
    const unsigned long long key = pack_key64(x, y, z, ok);
    if (!ok) {  // outside the packable range: nothing is inserted; the host sees the flag and takes the update over (kicp_map.hip)
        m.ctr->error = 1u;
        return kNoSlot;
    }
    uint32_t h = voxel_hash(x, y, z) & m.mask;
    for (uint32_t probes = 0;; ++probes) {
        if (probes > m.mask) {
            m.ctr->error = 3u;
            return kNoSlot;
        }
        const unsigned long long seen = atomicCAS(m.keys64 + h, kEmptyKey64, key);
        if (seen == kEmptyKey64) {  // this thread owns the new entry: key fields and the halo marker (nbr is already 0)
            m.table[h].x = x, m.table[h].y = y, m.table[h].z = z;
            m.table[h].val = halo_val(m.cbits);
            atomicAdd(&m.ctr->n_entries, 1u);
            return h;
        }
        if (seen == key) return h;
        h = (h + 1) & m.mask;
    }
}
// read-only lookup; 0xFFFFFFFF when the voxel has no entry
__device__ __forceinline__ uint32_t dev_find(const DevMap &m, int32_t x, int32_t y, int32_t z) {
    bool ok;
    const unsigned long long key = pack_key64(x, y, z, ok);
    uint32_t h = voxel_hash(x, y, z) & m.mask;
    for (uint32_t probes = 0; probes <= m.mask; ++probes) {
        const unsigned long long seen = m.keys64[h];
        if (seen == kEmptyKey64) return kNoSlot;
        if (seen == key) return h;
        h = (h + 1) & m.mask;
    }
    return kNoSlot;
}

// keys64 from the table (after every host -> device upload)
static __global__ __launch_bounds__(256) void k_build_keys64(const Slot *table, uint32_t slots, unsigned long long *keys64) {
    for (uint32_t h = blockIdx.x * 256 + threadIdx.x; h < slots; h += gridDim.x * 256) {
        bool ok;
        keys64[h] = table[h].val == kEmptyVal ? kEmptyKey64 : pack_key64(table[h].x, table[h].y, table[h].z, ok);
    }
}

// The later steps of an update that the claim step gave up on (a voxel coordinate out of the packable range: error 1; table
// full: 3) do nothing - with a single host synchronisation per update the host only learns of it at the end.  (Error 2 is
// raised inside step 4 itself and is not a reason for its other waves to stop.)
__device__ __forceinline__ bool claim_failed(const DevMap &m) {
    const uint32_t err = m.ctr->error;
    return err == 1u || err == 3u;
}

// 1. transform, find/claim the voxel's slot, count the voxel's new points
static __global__ __launch_bounds__(256) void k_up_claim(const UpdateParams p) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= p.n) return;
    double rx, ry, rz;
    quat_rotate(p.pose, p.in[3 * i], p.in[3 * i + 1], p.in[3 * i + 2], rx, ry, rz);
    const double wx = rx + p.pose.tx, wy = ry + p.pose.ty, wz = rz + p.pose.tz;  // pose * point
    p.world[3 * i] = wx, p.world[3 * i + 1] = wy, p.world[3 * i + 2] = wz;
    const double vs = p.m.voxel_size;
    const int32_t vx = static_cast<int32_t>(floor(wx / vs)), vy = static_cast<int32_t>(floor(wy / vs)), vz = static_cast<int32_t>(floor(wz / vs));
    // (one voxel of head-room: the 26 neighbours of an accepted voxel must be packable too)
    const int lim = (1 << 20) - 1;
    uint32_t h = kNoSlot;
    if (vx > -lim && vx < lim && vy > -lim && vy < lim && vz > -lim && vz < lim) h = dev_find_or_insert(p.m, vx, vy, vz);
    else p.m.ctr->error = 1u;
    p.slot_of[i] = h;
    if (h == kNoSlot) return;  // out of the packable range / table full (error raised): the host takes over / aborts the update
    if (atomicAdd(p.m.cnt + h, 1u) == 0u) p.touched[atomicAdd(&p.m.ctr->touched, 1u)] = h;
}

// 2. exclusive scan of the touched voxels' counts -> group starts; counts are zeroed to serve as fill cursors.  Also counts
//    the touched voxels that hold no point yet (fresh entries and halo entries alike): each of them may become occupied
//    in step 4 and then insert up to 26 halo entries of its own - the head-room the host checks before going on.
//    One workgroup; a frame's worth of touched voxels (<= 8 192) is ONE chunk: every thread fetches its eight voxels' counts
//    up front - the loads of a chunk are independent of the carry, so their latency is paid once, not once per 1 024 voxels
//    (12 us -> ~4 us per frame-sized update).
constexpr int kScanPerThread = 8;
static __global__ __launch_bounds__(1024) void k_up_scan(const UpdateParams p) {
    __shared__ uint32_t s_wave[16];
    __shared__ uint32_t s_carry;
    // (runs even when the claim step gave up: it leaves the per-slot counters clean)
    const uint32_t n_touched = p.m.ctr->touched;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_carry = 0;
    uint32_t may_become_occupied = 0;
    __syncthreads();
    for (uint32_t j0 = 0; j0 < n_touched; j0 += 1024 * kScanPerThread) {
        // thread t owns the kScanPerThread consecutive voxels from j0 + t * kScanPerThread
        const uint32_t first = j0 + threadIdx.x * kScanPerThread;
        uint32_t h[kScanPerThread], c[kScanPerThread];
#pragma unroll
        for (int u = 0; u < kScanPerThread; ++u) h[u] = first + u < n_touched ? p.touched[first + u] : kNoSlot;
        uint32_t mine = 0;
#pragma unroll
        for (int u = 0; u < kScanPerThread; ++u) {
            c[u] = 0;
            if (h[u] != kNoSlot) c[u] = p.m.cnt[h[u]], may_become_occupied += val_count(p.m.table[h[u]].val, p.m.cbits) == 0u ? 1u : 0u;
            mine += c[u];
        }
        uint32_t incl = mine;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t t = __shfl_up(incl, off, 64);
            if (lane >= off) incl += t;
        }
        if (lane == 63) s_wave[wave] = incl;
        __syncthreads();
        uint32_t before = s_carry;
        for (int w = 0; w < wave; ++w) before += s_wave[w];
        uint32_t at = before + incl - mine;
#pragma unroll
        for (int u = 0; u < kScanPerThread; ++u) {
            if (h[u] != kNoSlot) p.m.seg_start[h[u]] = at, p.m.cnt[h[u]] = 0;
            at += c[u];
        }
        __syncthreads();
        if (threadIdx.x == 1023) s_carry = before + incl;
        __syncthreads();
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) may_become_occupied += __shfl_down(may_become_occupied, off, 64);
    if (lane == 0 && may_become_occupied) atomicAdd(&p.m.ctr->may_occupy, may_become_occupied);
}

// 2b. the same scan over MANY workgroups, for bulk insertions (tens of thousands of touched voxels and more): one workgroup
//     gathering the counts of every touched voxel is bound by its CU's address path (~400 us per 70 000 voxels, half of a bulk
//     insertion's kernel time).  k_up_scan_local: workgroup b scans its kScanSpan voxels (offsets local to the workgroup) and leaves
//     its total in sums[b]; k_up_scan_sums: one workgroup turns the totals into exclusive prefixes; k_up_scan_add: every voxel's
//     offset gets its workgroup's prefix.  `sums` borrows the head of `order`, which step 3 only fills afterwards.
constexpr uint32_t kScanSpan = 256 * kScanPerThread;  // voxels per workgroup of k_up_scan_local
static __global__ __launch_bounds__(256) void k_up_scan_local(const UpdateParams p, uint32_t *sums) {
    __shared__ uint32_t s_wave[4];
    const uint32_t n_touched = p.m.ctr->touched;
    const uint32_t base = blockIdx.x * kScanSpan;
    if (base >= n_touched) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t first = base + threadIdx.x * kScanPerThread;
    uint32_t h[kScanPerThread], c[kScanPerThread], may_become_occupied = 0, mine = 0;
#pragma unroll
    for (int u = 0; u < kScanPerThread; ++u) h[u] = first + u < n_touched ? p.touched[first + u] : kNoSlot;
#pragma unroll
    for (int u = 0; u < kScanPerThread; ++u) {
        c[u] = 0;
        if (h[u] != kNoSlot) c[u] = p.m.cnt[h[u]], may_become_occupied += val_count(p.m.table[h[u]].val, p.m.cbits) == 0u ? 1u : 0u;
        mine += c[u];
    }
    uint32_t incl = mine;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t t = __shfl_up(incl, off, 64);
        if (lane >= off) incl += t;
    }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    uint32_t at = incl - mine;
    for (int w = 0; w < wave; ++w) at += s_wave[w];
#pragma unroll
    for (int u = 0; u < kScanPerThread; ++u) {
        if (h[u] != kNoSlot) p.m.seg_start[h[u]] = at, p.m.cnt[h[u]] = 0;
        at += c[u];
    }
    if (threadIdx.x == 255) sums[blockIdx.x] = at;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) may_become_occupied += __shfl_down(may_become_occupied, off, 64);
    if (lane == 0 && may_become_occupied) atomicAdd(&p.m.ctr->may_occupy, may_become_occupied);
}
static __global__ __launch_bounds__(1024) void k_up_scan_sums(const UpdateParams p, uint32_t *sums) {
    __shared__ uint32_t s_wave[16];
    __shared__ uint32_t s_carry;
    const uint32_t nblocks = (p.m.ctr->touched + kScanSpan - 1) / kScanSpan;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    for (uint32_t b0 = 0; b0 < nblocks; b0 += 1024) {
        const uint32_t b = b0 + threadIdx.x;
        const uint32_t c = b < nblocks ? sums[b] : 0u;
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
        if (b < nblocks) sums[b] = before + incl - c;
        __syncthreads();
        if (threadIdx.x == 1023) s_carry = before + incl;
        __syncthreads();
    }
}
static __global__ __launch_bounds__(256) void k_up_scan_add(const UpdateParams p, const uint32_t *sums) {
    const uint32_t j = blockIdx.x * 256 + threadIdx.x;
    if (j >= p.m.ctr->touched) return;
    const uint32_t prefix = sums[j / kScanSpan];
    if (prefix) p.m.seg_start[p.touched[j]] += prefix;
}

// 3. scatter the point indices into their voxel's group (any order; the apply step walks a group by ascending index)
static __global__ __launch_bounds__(256) void k_up_scatter(const UpdateParams p) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= p.n || claim_failed(p.m)) return;
    const uint32_t h = p.slot_of[i];
    if (h == kNoSlot) return;
    p.order[p.m.seg_start[h] + atomicAdd(p.m.cnt + h, 1u)] = i;
}

// 4. one WAVE per touched voxel: the reference's AddPoints rule over the voxel's new points in input order.  The order
//    dependence (a point is judged against everything accepted before it) stays sequential; what the 64 lanes share is the
//    work inside a step: finding the next input index of the group, the distance test against the bucket's points (held in
//    LDS, one point per lane and trip, same fp64 expression as the reference), and - when the voxel becomes occupied - the
//    27 neighbour records, one neighbour per lane.  Used for the updates a pipeline issues (a few thousand touched voxels:
//    one THREAD per voxel leaves the machine empty there and took 80 us per frame; this takes ~15); see 4b for large ones.
constexpr int kApplyWaves = 4;  // voxels per workgroup
constexpr uint32_t kApplyMaxPoints = 255;  // deepest bucket the wave-per-voxel kernel holds in LDS
static __global__ __launch_bounds__(64 * kApplyWaves) void k_up_apply(const UpdateParams p) {
    __shared__ double s_pts[kApplyWaves][kApplyMaxPoints * 3];  // (the host sends voxels with deeper buckets to k_up_apply_thread)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t t = blockIdx.x * kApplyWaves + wave;
    const DevMap &m = p.m;
    if (t >= m.ctr->touched || claim_failed(m)) return;  // (wave-uniform)
    const uint32_t h = p.touched[t];
    const uint32_t g = m.cnt[h], start = m.seg_start[h];
    Slot &e = m.table[h];
    const uint32_t old_val = e.val;
    const uint32_t old_count = val_count(old_val, m.cbits);
    uint32_t count = old_count, bucket = val_bucket(old_val, m.cbits);
    if (lane == 0) {
        if (g != 0xFFFFFFFFu) m.cnt[h] = 0;  // (after g has arrived) leave the per-slot scratch clean for the next update
        if (old_count == 0) {  // first points of this voxel: take a bucket (re-use a freed one if any)
            const uint32_t f = atomicSub(&m.ctr->free_count, 1u);
            if (f != 0u && f <= m.bucket_capacity) {
                bucket = m.free_list[f - 1];
            } else {
                atomicAdd(&m.ctr->free_count, 1u);
                bucket = atomicAdd(&m.ctr->n_buckets_hi, 1u);
                if (bucket >= m.bucket_capacity) m.ctr->error = 2u, bucket = kNoSlot;  // cannot happen: the host checked the capacity beforehand
            }
        }
    }
    bucket = __shfl(bucket, 0, 64);
    if (bucket == kNoSlot) return;
    const double vs = m.voxel_size;
    const double map_resolution = sqrt(vs * vs / m.cap);
    double *b = m.pool + static_cast<size_t>(bucket) * m.cap * 3;
    MirrorPoint *b16 = m.pool16 + static_cast<size_t>(bucket) * mirror_stride(m.cap);
    const double upm = mirror_units_per_metre(vs);
    double *sp = s_pts[wave];
    if (old_count == 0) {  // a fresh or re-used bucket: every slot of its mirror is empty until a point is stored there
        for (uint32_t k = lane; k < mirror_stride(m.cap); k += 64) b16[k] = mirror_empty();
    } else {
        for (uint32_t k = lane; k < 3 * old_count; k += 64) sp[k] = b[k];  // the bucket as it stands
    }
    // walk the group in ascending input index
    uint32_t last = 0;
    bool first = true;
    for (uint32_t step = 0; step < g && count < m.cap; ++step) {
        uint32_t idx = 0xFFFFFFFFu;
        for (uint32_t k = lane; k < g; k += 64) {
            const uint32_t c = p.order[start + k];
            if ((first || c > last) && c < idx) idx = c;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) idx = min(idx, static_cast<uint32_t>(__shfl_xor(idx, off, 64)));
        first = false, last = idx;
        const double px = p.world[3 * idx], py = p.world[3 * idx + 1], pz = p.world[3 * idx + 2];
        bool too_close = false;
        for (uint32_t k = lane; k < count; k += 64) {
            const double dx = sp[3 * k] - px, dy = sp[3 * k + 1] - py, dz = sp[3 * k + 2] - pz;
            too_close = too_close || sqrt(dx * dx + dy * dy + dz * dz) < map_resolution;
        }
        if (__any(too_close)) continue;
        if (lane == 0) {
            sp[3 * count] = px, sp[3 * count + 1] = py, sp[3 * count + 2] = pz;
            b[3 * count] = px, b[3 * count + 1] = py, b[3 * count + 2] = pz;
            b16[count] = mirror_point(px - e.x * vs, py - e.y * vs, pz - e.z * vs, upm);
        }
        ++count;
    }
    if (count == old_count) return;
    if (lane == 0) {
        e.val = make_val(bucket, count, m.cbits);
        atomicAdd(&m.ctr->n_points, static_cast<unsigned long long>(count - old_count));
        if (old_count == 0) atomicAdd(&m.ctr->n_voxels, 1u);
    }
    if (old_count == 0 && lane < 27) {  // newly occupied: tell the 27 voxels that see this one (U + shift[s] == this  <=>  U = this - shift[s])
        const int s = lane;
        const uint32_t u = dev_find_or_insert(m, e.x - shift_component(kShiftX, s), e.y - shift_component(kShiftY, s), e.z - shift_component(kShiftZ, s));
        if (u != kNoSlot) {  // (kNoSlot: table full, error raised)
            m.table[u].nb[s] = bucket;
            atomicOr(&m.table[u].nbr, 1u << s);
        }
    }
}

// 4b. the same step with one THREAD per touched voxel: what large updates use (tens of thousands of touched voxels - bulk
//     insertions, map building - fill the machine with independent voxels; a wave per voxel only adds overhead there)
static __global__ __launch_bounds__(64) void k_up_apply_thread(const UpdateParams p) {
    const uint32_t t = blockIdx.x * 64 + threadIdx.x;
    const DevMap &m = p.m;
    if (t >= m.ctr->touched || claim_failed(m)) return;
    const uint32_t h = p.touched[t];
    const uint32_t g = m.cnt[h], start = m.seg_start[h];
    m.cnt[h] = 0;  // leave the per-slot scratch clean for the next update
    Slot &e = m.table[h];
    const uint32_t old_count = val_count(e.val, m.cbits);
    uint32_t count = old_count, bucket = val_bucket(e.val, m.cbits);
    if (old_count == 0) {  // first points of this voxel: take a bucket (re-use a freed one if any)
        const uint32_t f = atomicSub(&m.ctr->free_count, 1u);
        if (f != 0u && f <= m.bucket_capacity) {
            bucket = m.free_list[f - 1];
        } else {
            atomicAdd(&m.ctr->free_count, 1u);
            bucket = atomicAdd(&m.ctr->n_buckets_hi, 1u);
            if (bucket >= m.bucket_capacity) {  // cannot happen: the host checked the capacity beforehand
                m.ctr->error = 2u;
                return;
            }
        }
    }
    const double vs = m.voxel_size;
    const double map_resolution = sqrt(vs * vs / m.cap);
    double *b = m.pool + static_cast<size_t>(bucket) * m.cap * 3;
    MirrorPoint *b16 = m.pool16 + static_cast<size_t>(bucket) * mirror_stride(m.cap);
    const double upm = mirror_units_per_metre(vs);
    if (old_count == 0)  // a fresh or re-used bucket: every slot of its mirror is empty until a point is stored there
        for (uint32_t k = 0; k < mirror_stride(m.cap); ++k) b16[k] = mirror_empty();
    // walk the group in ascending input index (selection; groups are small: the pipeline feeds <= 8 points per voxel)
    uint32_t last = 0;
    bool first = true;
    for (uint32_t step = 0; step < g && count < m.cap; ++step) {
        uint32_t idx = 0xFFFFFFFFu;
        for (uint32_t k = 0; k < g; ++k) {
            const uint32_t c = p.order[start + k];
            if ((first || c > last) && c < idx) idx = c;
        }
        first = false, last = idx;
        const double px = p.world[3 * idx], py = p.world[3 * idx + 1], pz = p.world[3 * idx + 2];
        bool too_close = false;
        for (uint32_t k = 0; k < count; ++k) {
            const double dx = b[3 * k] - px, dy = b[3 * k + 1] - py, dz = b[3 * k + 2] - pz;
            if (sqrt(dx * dx + dy * dy + dz * dz) < map_resolution) {
                too_close = true;
                break;
            }
        }
        if (too_close) continue;
        b[3 * count] = px, b[3 * count + 1] = py, b[3 * count + 2] = pz;
        b16[count] = mirror_point(px - e.x * vs, py - e.y * vs, pz - e.z * vs, upm);
        ++count;
    }
    if (count == old_count) return;
    e.val = make_val(bucket, count, m.cbits);
    atomicAdd(&m.ctr->n_points, static_cast<unsigned long long>(count - old_count));
    if (old_count == 0) {  // newly occupied: tell the 27 voxels that see this one (U + shift[s] == this  <=>  U = this - shift[s])
        atomicAdd(&m.ctr->n_voxels, 1u);
        for (int s = 0; s < 27; ++s) {
            const uint32_t u = dev_find_or_insert(m, e.x - kShiftTable[s][0], e.y - kShiftTable[s][1], e.z - kShiftTable[s][2]);
            if (u == kNoSlot) continue;  // table full (error raised)
            m.table[u].nb[s] = bucket;
            atomicOr(&m.table[u].nbr, 1u << s);
        }
    }
}

// 5. RemovePointsFarFromLocation(origin): a voxel goes when its FIRST point is >= max_distance away.
//    The sweep runs over the packed-key side array (8 B per slot, four slots per lane and trip), not over the 128-byte
//    slots: most of the table is free (load factor <= 0.25 after a re-hash) and most live entries are halo entries, so only
//    the entries that exist are looked at, and only the occupied ones touch the point pool.  (Sweeping the slots themselves
//    took 0.7 ms per update on cfg2's 2M-slot table - the largest single item of a frame's map update.)
static __global__ __launch_bounds__(256) void k_up_remove(const DevMap m, double ox, double oy, double oz) {
    if (claim_failed(m)) return;
    const uint32_t slots = m.mask + 1;
    const double max_distance2 = m.max_distance * m.max_distance;
    const ulonglong2 *keys2 = reinterpret_cast<const ulonglong2 *>(m.keys64);
    for (uint32_t q = blockIdx.x * 256 + threadIdx.x; q < slots / 4; q += gridDim.x * 256) {  // (slots is a power of two >= 1024)
        const ulonglong2 ka = keys2[2 * q], kb = keys2[2 * q + 1];
        const unsigned long long key[4] = {ka.x, ka.y, kb.x, kb.y};
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (key[u] == kEmptyKey64) continue;
            const uint32_t h = 4 * q + u;
            Slot &e = m.table[h];
            const uint32_t val = e.val;
            if (val == kEmptyVal || val_count(val, m.cbits) == 0u) continue;
            const uint32_t bucket = val_bucket(val, m.cbits);
            const double *b = m.pool + static_cast<size_t>(bucket) * m.cap * 3;
            const double dx = b[0] - ox, dy = b[1] - oy, dz = b[2] - oz;
            if (!(dx * dx + dy * dy + dz * dz >= max_distance2)) continue;
            e.val = halo_val(m.cbits);
            m.free_list[atomicAdd(&m.ctr->free_count, 1u)] = bucket;
            atomicSub(&m.ctr->n_voxels, 1u);
            atomicAdd(&m.ctr->n_points, ~static_cast<unsigned long long>(val_count(val, m.cbits)) + 1ull);
            for (int s = 0; s < 27; ++s) {
                const uint32_t w = dev_find(m, e.x - kShiftTable[s][0], e.y - kShiftTable[s][1], e.z - kShiftTable[s][2]);
                if (w != kNoSlot) atomicAnd(&m.table[w].nbr, ~(1u << s));
            }
        }
    }
}

// ---- Pointcloud() from the device copy (KinematicICP.hpp:92, published by the ROS node when someone listens) ----------
// All points, voxel by voxel in table order - the order HostMap::Pointcloud emits - without bringing the table and the
// pools back to the host: count per 256-slot block, scan the block totals, then every slot copies its bucket's points.
// (Free slots are recognised in the packed-key side array, 8 B per slot: the 128-byte slots of a mostly empty table are never read.)
// The end of an update whose caller collects it later (kicp_map_update_pose_device_begin): the counters go into host memory as
// this one-wave launch's stores, the sequence number behind them - the host polls that word (map_finish_pending) instead of
// synchronising the stream, which took ~55 us of a 200 us frame while other streams of the process were busy.
// It also leaves the per-update counters (touched, error, may_occupy) at zero for the next update: two fill launches less per frame.
static __global__ __launch_bounds__(64) void k_up_publish(DevMapCounters *ctr, unsigned long long *host_words, unsigned long long seq) {
    static_assert(sizeof(DevMapCounters) == 40, "five words");
    if (threadIdx.x < 5u) __hip_atomic_store(host_words + threadIdx.x, reinterpret_cast<const unsigned long long *>(ctr)[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (threadIdx.x == 0u) ctr->touched = 0u, ctr->error = 0u, ctr->may_occupy = 0u;
    if (threadIdx.x == 0u) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
        __hip_atomic_store(host_words + 7, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

static __global__ __launch_bounds__(256) void k_pc_count(const Slot *table, const unsigned long long *keys64, uint32_t slots, uint32_t cbits, uint32_t *block_counts) {
    __shared__ uint32_t s_sum[4];
    const uint32_t h = blockIdx.x * 256 + threadIdx.x;
    uint32_t c = 0;
    if (h < slots && keys64[h] != kEmptyKey64 && table[h].val != kEmptyVal) c = val_count(table[h].val, cbits);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off, 64);
    if ((threadIdx.x & 63) == 0) s_sum[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) block_counts[blockIdx.x] = s_sum[0] + s_sum[1] + s_sum[2] + s_sum[3];
}
static __global__ __launch_bounds__(256) void k_pc_gather(const Slot *table, const unsigned long long *keys64, uint32_t slots, const double *pool,
                                                   uint32_t cap, uint32_t cbits, const uint32_t *block_offsets, double *out) {
    __shared__ uint32_t s_wave[4];
    const uint32_t h = blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t c = 0, bucket = 0;
    if (h < slots && keys64[h] != kEmptyKey64 && table[h].val != kEmptyVal) c = val_count(table[h].val, cbits), bucket = val_bucket(table[h].val, cbits);
    uint32_t incl = c;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t t = __shfl_up(incl, off, 64);
        if (lane >= off) incl += t;
    }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    uint32_t pos = block_offsets[blockIdx.x] + incl - c;
    for (int w = 0; w < wave; ++w) pos += s_wave[w];
    const double *b = pool + static_cast<size_t>(bucket) * cap * 3;
    double *o = out + static_cast<size_t>(pos) * 3;
    for (uint32_t k = 0; k < 3 * c; ++k) o[k] = b[k];
}

// ---- re-hash on the device -------------------------------------------------------------------------------------------------
// The table only ever gains entries between re-hashes (erased voxels turn into halo entries, halo entries nobody needs any
// more stay behind as dead weight), so every now and then the live entries - occupied voxels and halo entries that still
// see an occupied neighbour, the host map's rule - move into a fresh (possibly larger) table.  Entries refer to buckets,
// never to slots, so they can move freely.
static __global__ __launch_bounds__(256) void k_rehash_count(const Slot *table, uint32_t slots, uint32_t cbits, uint32_t *live) {
    uint32_t c = 0;
    for (uint32_t h = blockIdx.x * 256 + threadIdx.x; h < slots; h += gridDim.x * 256) {
        const uint32_t val = table[h].val;
        c += (val != kEmptyVal && (val_count(val, cbits) != 0u || table[h].nbr != 0u)) ? 1u : 0u;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off, 64);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(live, c);
}
static __global__ __launch_bounds__(256) void k_table_clear(Slot *table, uint32_t slots) {
    int4 *w = reinterpret_cast<int4 *>(table);
    const size_t words = static_cast<size_t>(slots) * (sizeof(Slot) / 16);
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < words; i += gridDim.x * 256ull)
        w[i] = (i % (sizeof(Slot) / 16) == 0) ? make_int4(0, 0, 0, static_cast<int>(kEmptyVal)) : make_int4(0, 0, 0, 0);
}
static __global__ __launch_bounds__(256) void k_rehash_move(const Slot *old_table, uint32_t old_slots, Slot *table, unsigned long long *keys64, uint32_t mask,
                                                     uint32_t cbits, uint32_t *error) {
    for (uint32_t o = blockIdx.x * 256 + threadIdx.x; o < old_slots; o += gridDim.x * 256) {
        const Slot &e = old_table[o];
        if (e.val == kEmptyVal || (val_count(e.val, cbits) == 0u && e.nbr == 0u)) continue;
        bool ok;
        const unsigned long long key = pack_key64(e.x, e.y, e.z, ok);
        if (!ok) *error = 1u;
        uint32_t h = voxel_hash(e.x, e.y, e.z) & mask;
        while (atomicCAS(keys64 + h, kEmptyKey64, key) != kEmptyKey64) h = (h + 1) & mask;  // keys are unique: a taken slot is someone else's
        const int4 *src = reinterpret_cast<const int4 *>(&e);
        int4 *dst = reinterpret_cast<int4 *>(table + h);
#pragma unroll
        for (int u = 0; u < static_cast<int>(sizeof(Slot) / 16); ++u) dst[u] = src[u];
    }
}

}  // namespace kicp
