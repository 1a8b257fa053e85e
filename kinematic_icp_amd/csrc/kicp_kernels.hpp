// kicp_kernels.hpp -- gfx950 (CDNA4, wave64) kernels of the registration hot path.
//
//   k_pass_gather / k_pass_lds : fused DataAssociation + ComputePerturbation reduction (+ the pass-0 sum of
//                                ComputeOdometryRegularization): registration/Registration.cpp:62-81, 83-118, 48-55,
//                                with kiss_icp::VoxelHashMap::GetClosestNeighbor (kiss-icp v1.2.0)
//                                inlined as the 27-voxel probe + bucket scan.  No correspondence list is materialised.
//   k_finalize                 : fixed-order sum of the block partials, then Registration.cpp:119-125 (2x2 solve),
//                                :159-167 (motion model), :181-184 (pose update + stop test) on one lane.
//   k_closest                  : GetClosestNeighbor for a batch of queries (API parity / tests).
//
// Roofline: gather + reduction, ~0.02 flop/B -> memory bound, no MFMA.  All arithmetic is fp64,
// evaluated in the reference's operation order (TU is compiled with -ffp-contract=off), so NN decisions and
// per-point terms are those of the fp64 reference; only the summation order differs.
#pragma once
#include <cfloat>
#include <climits>

#include "kicp_common.hpp"
#include "kicp_se3.hpp"

namespace kicp {

constexpr int kMaxLog = 32;
constexpr int kNumSums = 8;  // JTJ00 JTJ01 JTJ11 JTr0 JTr1 ssq count pad

// Device-resident loop state of one ComputeRobotMotion call.
struct IcpState {
    Pose T;  // current_estimate
    double beta;
    int32_t done, iter, converged, nan_flag;
    double log_ncorr[kMaxLog];
    double log_sums[kMaxLog][6];
    double log_dx[kMaxLog][2];
    double reduced[kNumSums];  // all-reduce buffer (multi-GPU) / last pass sums
};

struct PassParams {
    const double *src;  // scan points, base frame, AoS xyz fp64 (device)
    uint32_t n;
    MapView map;
    double tau;
    IcpState *st;
    double *partials;  // [gridDim.x][kNumSums]
    Pose pose0;        // used when pass == 0 (the initial guess travels as a kernel argument)
    int32_t pass;
};

struct FinalizeParams {
    IcpState *st;
    const double *partials;
    uint32_t nblocks;
    Pose pose0;
    int32_t pass;
    int32_t max_iterations;
    double convergence_criterion;
    int32_t adaptive;
    double fixed_regularization;
    int32_t stage;  // 0 = reduce + solve (single GPU); 1 = reduce only -> st->reduced; 2 = solve from st->reduced
};

// ------------------------------------------------------------------------------------------------------------
// nearest-neighbour search helpers
// ------------------------------------------------------------------------------------------------------------
struct Query {
    double x, y, z;      // transformed point T*p
    int32_t vx, vy, vz;  // PointToVoxel(T*p)
    double fm[3], fp[3]; // squared distance to the -/+ faces of the own voxel, per axis
    double slack;
};

__device__ __forceinline__ void make_query(Query &q, double x, double y, double z, double vs) {
    q.x = x, q.y = y, q.z = z;
    const double fx = floor(x / vs), fy = floor(y / vs), fz = floor(z / vs);
    q.vx = static_cast<int32_t>(fx), q.vy = static_cast<int32_t>(fy), q.vz = static_cast<int32_t>(fz);
    const double lx = x - fx * vs, ly = y - fy * vs, lz = z - fz * vs;
    q.fm[0] = lx * lx, q.fm[1] = ly * ly, q.fm[2] = lz * lz;
    q.fp[0] = (vs - lx) * (vs - lx), q.fp[1] = (vs - ly) * (vs - ly), q.fp[2] = (vs - lz) * (vs - lz);
    // culling slack: far above fp64 rounding of the face distances, far below anything that matters
    q.slack = 4.0 * vs * 9.1e-13 * (fabs(x) + fabs(y) + fabs(z) + vs);
}

// lower bound of the squared distance from the query to any point of neighbour voxel (dx,dy,dz)
__device__ __forceinline__ double box_d2(const Query &q, int dx, int dy, int dz) {
    double d = 0.0;
    d += dx > 0 ? q.fp[0] : (dx < 0 ? q.fm[0] : 0.0);
    d += dy > 0 ? q.fp[1] : (dy < 0 ? q.fm[1] : 0.0);
    d += dz > 0 ? q.fp[2] : (dz < 0 ? q.fm[2] : 0.0);
    return d;
}

__device__ __forceinline__ uint32_t table_lookup(const MapView &m, int32_t x, int32_t y, int32_t z) {
    uint32_t h = voxel_hash(x, y, z) & m.mask;
    for (;;) {
        const int4 e = *reinterpret_cast<const int4 *>(m.table + h);
        if (static_cast<uint32_t>(e.w) == kEmptyVal) return kEmptyVal;
        if (e.x == x && e.y == y && e.z == z) return static_cast<uint32_t>(e.w);
        h = (h + 1) & m.mask;
    }
}

// scan one bucket in insertion order; strict '<' keeps the first minimum (std::min_element + `distance < closest`)
__device__ __forceinline__ void scan_points(const double *__restrict__ p, uint32_t count, uint32_t base_index, const Query &q,
                                            double &best, uint32_t &best_idx) {
    for (uint32_t k = 0; k < count; ++k) {
        const double dx = p[3 * k] - q.x, dy = p[3 * k + 1] - q.y, dz = p[3 * k + 2] - q.z;
        const double d2 = dx * dx + dy * dy + dz * dz;
        if (d2 < best) best = d2, best_idx = base_index + k;
    }
}

// 27-voxel 1-NN straight from HBM/L2.  `best` enters as the acceptance bound (or DBL_MAX).
__device__ __forceinline__ void search_global(const MapView &m, const Query &q, double &best, uint32_t &best_idx) {
#pragma unroll 1
    for (int s = 0; s < 27; ++s) {
        const int dx = shift_component(kShiftX, s), dy = shift_component(kShiftY, s), dz = shift_component(kShiftZ, s);
        if (box_d2(q, dx, dy, dz) > best + q.slack) continue;  // no point in there can beat `best`
        const uint32_t val = table_lookup(m, q.vx + dx, q.vy + dy, q.vz + dz);
        if (val == kEmptyVal) continue;
        const uint32_t bucket = val >> 8;
        scan_points(m.pool + static_cast<size_t>(bucket) * m.cap * 3, val & 0xffu, bucket * m.cap, q, best, best_idx);
    }
}

// ------------------------------------------------------------------------------------------------------------
// per-correspondence terms (Registration.cpp:86-93,108-113) and the block reduction
// ------------------------------------------------------------------------------------------------------------
struct Acc {
    double v[7];  // JTJ00 JTJ01 JTJ11 JTr0 JTr1 ssq count
};

__device__ __forceinline__ void accumulate(Acc &a, const Pose &T, double sx, double sy, double qx, double qy, double qz, double tx,
                                           double ty, double tz) {
    const double rx = qx - tx, ry = qy - ty, rz = qz - tz;  // residual = T*source - target
    double j0x, j0y, j0z, j1x, j1y, j1z;
    quat_rotate(T, 1.0, 0.0, 0.0, j0x, j0y, j0z);  // J.col(0) = R * UnitX
    quat_rotate(T, -sy, sx, 0.0, j1x, j1y, j1z);   // J.col(1) = R * (-s.y, s.x, 0)
    a.v[0] += j0x * j0x + j0y * j0y + j0z * j0z;
    a.v[1] += j0x * j1x + j0y * j1y + j0z * j1z;
    a.v[2] += j1x * j1x + j1y * j1y + j1z * j1z;
    a.v[3] += j0x * rx + j0y * ry + j0z * rz;
    a.v[4] += j1x * rx + j1y * ry + j1z * rz;
    a.v[5] += rx * rx + ry * ry + rz * rz;
    a.v[6] += 1.0;
}

template <int BLOCK>
__device__ __forceinline__ void block_reduce_store(const Acc &a, double *__restrict__ out, double (*s_red)[8]) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double v[7];
#pragma unroll
    for (int i = 0; i < 7; ++i) {
        double x = a.v[i];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
        v[i] = x;
    }
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < 7; ++i) s_red[wave][i] = v[i];
    }
    __syncthreads();
    if (threadIdx.x < 7) {
        double x = 0.0;
        for (int w = 0; w < BLOCK / 64; ++w) x += s_red[w][threadIdx.x];
        out[threadIdx.x] = x;
    }
}

__device__ __forceinline__ Pose load_pose(const PassParams &p) { return p.pass == 0 ? p.pose0 : p.st->T; }

// ------------------------------------------------------------------------------------------------------------
// K1+K2 variant A: thread-per-query gather from HBM/L2
// ------------------------------------------------------------------------------------------------------------
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_pass_gather(const PassParams p) {
    __shared__ double s_red[BLOCK / 64][8];
    if (p.pass != 0 && p.st->done) return;
    const Pose T = load_pose(p);
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    Acc acc{};
    if (i < p.n) {
        const double sx = p.src[3 * i], sy = p.src[3 * i + 1], sz = p.src[3 * i + 2];
        double rx, ry, rz;
        quat_rotate(T, sx, sy, sz, rx, ry, rz);
        Query q;
        make_query(q, rx + T.tx, ry + T.ty, rz + T.tz, p.map.voxel_size);
        double best = p.tau * p.tau * (1.0 + 9.1e-13);
        uint32_t best_idx = 0xFFFFFFFFu;
        search_global(p.map, q, best, best_idx);
        if (best_idx != 0xFFFFFFFFu && sqrt(best) < p.tau) {  // `distance < max_correspondance_distance`, Registration.cpp:75
            const double *t = p.map.pool + static_cast<size_t>(best_idx) * 3;
            accumulate(acc, T, sx, sy, q.x, q.y, q.z, t[0], t[1], t[2]);
        }
    }
    block_reduce_store<BLOCK>(acc, p.partials + static_cast<size_t>(blockIdx.x) * kNumSums, s_red);
}

// ------------------------------------------------------------------------------------------------------------
// K1+K2 variant B: the block's voxel neighbourhood is staged in LDS once, then scanned by every query
// ------------------------------------------------------------------------------------------------------------
// LDS budget scales with the block: BLOCK*4 region voxels, BLOCK*6 staged points (256 threads: 1024 voxels,
// 1536 points = 36 KiB of fp64 xyz, ~49 KiB per block -> 3 blocks/CU; 128 threads: ~25 KiB -> 6 blocks/CU).
constexpr uint32_t kNotStaged = 0xFFFFFFFFu;

template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_pass_lds(const PassParams p) {
    __shared__ double s_red[BLOCK / 64][8];
    __shared__ int s_bb[6];
    __shared__ uint32_t s_used, s_nlist;
    constexpr int kLdsSlots = BLOCK * 4;   // voxels of the (bbox + 1 halo) region a block may stage
    constexpr int kLdsPoints = BLOCK * 6;  // staged map points
    __shared__ uint32_t s_val[kLdsSlots];   // table value (bucket<<8 | count) of each region voxel, or kEmptyVal
    __shared__ uint32_t s_off[kLdsSlots];   // first staged point of that voxel in s_pts, or kNotStaged
    __shared__ uint32_t s_list[kLdsSlots];  // region slots whose bucket must be copied
    __shared__ double s_pts[kLdsPoints * 3];

    if (p.pass != 0 && p.st->done) return;
    const Pose T = load_pose(p);
    const MapView &m = p.map;
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const bool valid = i < p.n;

    double sx = 0, sy = 0, sz = 0;
    Query q;
    if (valid) {
        sx = p.src[3 * i], sy = p.src[3 * i + 1], sz = p.src[3 * i + 2];
        double rx, ry, rz;
        quat_rotate(T, sx, sy, sz, rx, ry, rz);
        make_query(q, rx + T.tx, ry + T.ty, rz + T.tz, m.voxel_size);
    } else {
        make_query(q, 0.0, 0.0, 0.0, m.voxel_size);
    }

    // ---- block bounding box of the query voxels -------------------------------------------------------------
    if (threadIdx.x == 0) {
        s_bb[0] = s_bb[1] = s_bb[2] = INT_MAX;
        s_bb[3] = s_bb[4] = s_bb[5] = INT_MIN;
        s_used = 0, s_nlist = 0;
    }
    __syncthreads();
    {
        int lo[3] = {valid ? q.vx : INT_MAX, valid ? q.vy : INT_MAX, valid ? q.vz : INT_MAX};
        int hi[3] = {valid ? q.vx : INT_MIN, valid ? q.vy : INT_MIN, valid ? q.vz : INT_MIN};
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                lo[a] = min(lo[a], __shfl_xor(lo[a], off, 64));
                hi[a] = max(hi[a], __shfl_xor(hi[a], off, 64));
            }
        }
        if (lane == 0) {
#pragma unroll
            for (int a = 0; a < 3; ++a) atomicMin(&s_bb[a], lo[a]), atomicMax(&s_bb[3 + a], hi[a]);
        }
    }
    __syncthreads();
    const int bx = s_bb[0] - 1, by = s_bb[1] - 1, bz = s_bb[2] - 1;  // region origin (one voxel of halo)
    const long long ex = static_cast<long long>(s_bb[3]) - s_bb[0] + 3, ey = static_cast<long long>(s_bb[4]) - s_bb[1] + 3,
                    ez = static_cast<long long>(s_bb[5]) - s_bb[2] + 3;
    const bool staged = s_bb[3] >= s_bb[0] && ex * ey * ez <= kLdsSlots;  // block-uniform

    if (staged) {
        const int nslots = static_cast<int>(ex * ey * ez), iex = static_cast<int>(ex), iey = static_cast<int>(ey);
        // ---- one table probe per region voxel; reserve LDS space for the occupied ones -----------------------
        for (int s = threadIdx.x; s < nslots; s += BLOCK) {
            const int ix = s % iex, iy = (s / iex) % iey, iz = s / (iex * iey);
            const uint32_t val = table_lookup(m, bx + ix, by + iy, bz + iz);
            uint32_t off = kNotStaged;
            if (val != kEmptyVal) {
                const uint32_t cnt = val & 0xffu;
                const uint32_t o = atomicAdd(&s_used, cnt);
                if (o + cnt <= kLdsPoints) {
                    off = o;
                    s_list[atomicAdd(&s_nlist, 1u)] = static_cast<uint32_t>(s);
                }
            }
            s_val[s] = val, s_off[s] = off;
        }
        __syncthreads();
        // ---- copy the buckets: one wave per bucket, 8 B per lane, coalesced ---------------------------------
        const uint32_t nlist = s_nlist;
        for (uint32_t j = threadIdx.x >> 6; j < nlist; j += BLOCK / 64) {
            const uint32_t s = s_list[j];
            const uint32_t val = s_val[s], cnt3 = (val & 0xffu) * 3;
            const double *src = m.pool + static_cast<size_t>(val >> 8) * m.cap * 3;
            double *dst = s_pts + static_cast<size_t>(s_off[s]) * 3;
            for (uint32_t d = lane; d < cnt3; d += 64) dst[d] = src[d];
        }
        __syncthreads();
    }

    Acc acc{};
    if (valid) {
        double best = p.tau * p.tau * (1.0 + 9.1e-13);
        uint32_t best_idx = 0xFFFFFFFFu;  // bit 31 set: index into s_pts; clear: index into the global pool
        if (staged) {
            const int iex = static_cast<int>(ex), iey = static_cast<int>(ey);
            const int cx = q.vx - bx, cy = q.vy - by, cz = q.vz - bz;
#pragma unroll 1
            for (int s = 0; s < 27; ++s) {
                const int dx = shift_component(kShiftX, s), dy = shift_component(kShiftY, s), dz = shift_component(kShiftZ, s);
                if (box_d2(q, dx, dy, dz) > best + q.slack) continue;
                const int slot = ((cz + dz) * iey + (cy + dy)) * iex + (cx + dx);
                const uint32_t val = s_val[slot];
                if (val == kEmptyVal) continue;
                const uint32_t off = s_off[slot], cnt = val & 0xffu;
                if (off != kNotStaged) {
                    scan_points(s_pts + static_cast<size_t>(off) * 3, cnt, 0x80000000u | off, q, best, best_idx);
                } else {  // LDS pool overflow: this bucket stays in HBM
                    const uint32_t bucket = val >> 8;
                    scan_points(m.pool + static_cast<size_t>(bucket) * m.cap * 3, cnt, bucket * m.cap, q, best, best_idx);
                }
            }
        } else {
            search_global(m, q, best, best_idx);
        }
        if (best_idx != 0xFFFFFFFFu && sqrt(best) < p.tau) {
            const double *t = (best_idx & 0x80000000u) ? s_pts + static_cast<size_t>(best_idx & 0x7FFFFFFFu) * 3
                                                         : m.pool + static_cast<size_t>(best_idx) * 3;
            accumulate(acc, T, sx, sy, q.x, q.y, q.z, t[0], t[1], t[2]);
        }
    }
    block_reduce_store<BLOCK>(acc, p.partials + static_cast<size_t>(blockIdx.x) * kNumSums, s_red);
}

// ------------------------------------------------------------------------------------------------------------
// K3: fixed-order reduction of the partials + solve + pose update (one 256-thread block)
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_finalize(const FinalizeParams f) {
    __shared__ double s_red[4][8];
    IcpState *st = f.st;
    if (f.pass != 0 && st->done) return;
    double sums[7];
    if (f.stage != 2) {
        Acc a{};
        for (uint32_t b = threadIdx.x; b < f.nblocks; b += 256) {
#pragma unroll
            for (int i = 0; i < 7; ++i) a.v[i] += f.partials[static_cast<size_t>(b) * kNumSums + i];
        }
        block_reduce_store<256>(a, st->reduced, s_red);
        __syncthreads();
        if (f.stage == 1) return;
    }
    if (threadIdx.x != 0) return;
#pragma unroll
    for (int i = 0; i < 7; ++i) sums[i] = st->reduced[i];
    const double n = sums[6];
    Pose T = f.pass == 0 ? f.pose0 : st->T;
    double beta;
    if (f.pass == 0) {  // ComputeOdometryRegularization at the predicted pose (Registration.cpp:48-60,171-177)
        beta = f.adaptive ? 1.0 / (sums[5] / n + DBL_MIN) : f.fixed_regularization;
        st->beta = beta;
        st->converged = 0, st->nan_flag = 0;
    } else {
        beta = st->beta;
    }
    double dx0, dx1;
    solve_perturbation(sums, n, beta, dx0, dx1);
    T = pose_mul(T, motion_model(dx0, dx1));  // current_estimate * delta_motion (Registration.cpp:181-182)
    st->T = T;
    if (f.pass < kMaxLog) {
        st->log_ncorr[f.pass] = n;
#pragma unroll
        for (int i = 0; i < 6; ++i) st->log_sums[f.pass][i] = sums[i];
        st->log_dx[f.pass][0] = dx0, st->log_dx[f.pass][1] = dx1;
    }
    st->iter = f.pass + 1;
    int done = 0;
    if (sqrt(dx0 * dx0 + dx1 * dx1) < f.convergence_criterion) done = 1, st->converged = 1;  // Registration.cpp:184
    if (f.pass + 1 >= f.max_iterations) done = 1;
    if (!(n > 0.0)) done = 1, st->nan_flag = 1;  // 0/0: the pose is NaN from here on, exactly as in the reference
    st->done = done;
}

// ------------------------------------------------------------------------------------------------------------
// GetClosestNeighbor for a batch (API parity; same search code as the fused kernel, no acceptance bound)
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_closest(const double *__restrict__ queries, uint32_t n, const MapView m,
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
