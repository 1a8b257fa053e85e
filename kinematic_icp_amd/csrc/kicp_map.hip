// kicp_map.hip -- kiss_icp::VoxelHashMap behind include/kicp.h: the host map (kicp_host_map.hpp), its HBM mirror (full and
// delta upload, lazy refresh of the host copy), the device-side Update / re-hash / Pointcloud (kicp_mapdev.hpp) and the
// batch GetClosestNeighbor.
#include "kicp_internal.hpp"
#include "kicp_kernels.hpp"
#include "kicp_mapdev.hpp"
#include "kicp_pre.hpp"  // k_scan_blocks

using namespace kicp;
using namespace kicp::host;

namespace {
// points of the 16-bit mirror that go with `pool_doubles` doubles of the fp64 pool (bucket strides cap / cap16)
size_t mirror_points(size_t pool_doubles, uint32_t cap) { return pool_doubles / (static_cast<size_t>(cap) * 3) * mirror_stride(cap); }
// release every device buffer of a mirror (on its own device) and reset it
void free_mirror(DeviceMirror &mr) {
    if (mr.device >= 0) {
        hipSetDevice(mr.device);
        hipFree(mr.d_table), hipFree(mr.d_pool), hipFree(mr.d_pool16), hipFree(mr.d_stage), hipFree(mr.d_index);
        hipFree(mr.d_keys64), hipFree(mr.d_cnt), hipFree(mr.d_seg_start), hipFree(mr.d_free_list), hipFree(mr.d_ctr);
        hipFree(mr.d_world), hipFree(mr.d_slot_of), hipFree(mr.d_order), hipFree(mr.d_touched);
        hipFree(mr.d_pc), hipFree(mr.d_pc_blocks);
        if (mr.h_ctr) hipHostFree(mr.h_ctr);
        mr.stage.release();
    }
    mr = DeviceMirror{};
}

// per-slot helper arrays, free list and counters of the device-side maintenance, rebuilt after every upload
int sync_aux(kicp_map *map, hipStream_t stream) {
    DeviceMirror &mr = map->mirror;
    const HostMap &h = map->host;
    const size_t slots = h.table().size();
    if (slots != mr.aux_slots) {
        hipFree(mr.d_keys64), hipFree(mr.d_cnt), hipFree(mr.d_seg_start);
        mr.d_keys64 = nullptr, mr.d_cnt = nullptr, mr.d_seg_start = nullptr;
        HIP_TRY(hipMalloc(&mr.d_keys64, slots * 8));
        HIP_TRY(hipMalloc(&mr.d_cnt, slots * 4));
        HIP_TRY(hipMalloc(&mr.d_seg_start, slots * 4));
        HIP_TRY(hipMemsetAsync(mr.d_cnt, 0, slots * 4, stream));
        mr.aux_slots = slots;
    }
    const size_t bucket_cap = mr.pool_doubles / (static_cast<size_t>(h.cap()) * 3);
    if (bucket_cap > mr.free_cap) {
        hipFree(mr.d_free_list);
        mr.d_free_list = nullptr;
        HIP_TRY(hipMalloc(&mr.d_free_list, (bucket_cap + 1) * 4));
        mr.free_cap = bucket_cap;
    }
    if (!mr.d_ctr) HIP_TRY(hipMalloc(&mr.d_ctr, sizeof(DevMapCounters)));
    hipLaunchKernelGGL(k_build_keys64, dim3(static_cast<uint32_t>(std::min<size_t>((slots + 255) / 256, 4096))), dim3(256), 0, stream, mr.d_table,
                       static_cast<uint32_t>(slots), mr.d_keys64);
    DevMapCounters c{};
    c.n_points = h.num_points(), c.n_voxels = static_cast<uint32_t>(h.num_voxels()), c.n_entries = static_cast<uint32_t>(h.num_entries());
    c.n_buckets_hi = static_cast<uint32_t>(h.buckets_in_use_hi()), c.free_count = static_cast<uint32_t>(h.free_list().size());
    if (c.free_count) HIP_TRY(hipMemcpyAsync(mr.d_free_list, h.free_list().data(), c.free_count * 4, hipMemcpyHostToDevice, stream));
    HIP_TRY(hipMemcpyAsync(mr.d_ctr, &c, sizeof c, hipMemcpyHostToDevice, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    map->dev = c;
    return KICP_OK;
}

// scatter `rows` staged rows of `row_words` 8-byte words each into dst at the given row indices
int upload_rows(DeviceMirror &mr, const std::vector<uint2> &staged, const std::vector<uint32_t> &index, uint32_t row_words, void *dst,
                hipStream_t stream) {
    if (index.empty()) return KICP_OK;
    if (staged.size() > mr.stage_words) {
        if (mr.d_stage) HIP_TRY(hipFree(mr.d_stage));
        mr.d_stage = nullptr;
        mr.stage_words = staged.size() + staged.size() / 2;
        HIP_TRY(hipMalloc(&mr.d_stage, mr.stage_words * sizeof(uint2)));
    }
    if (index.size() > mr.index_cap) {
        if (mr.d_index) HIP_TRY(hipFree(mr.d_index));
        mr.d_index = nullptr;
        mr.index_cap = index.size() + index.size() / 2;
        HIP_TRY(hipMalloc(&mr.d_index, mr.index_cap * sizeof(uint32_t)));
    }
    HIP_TRY(hipMemcpyAsync(mr.d_stage, staged.data(), staged.size() * sizeof(uint2), hipMemcpyHostToDevice, stream));
    HIP_TRY(hipMemcpyAsync(mr.d_index, index.data(), index.size() * sizeof(uint32_t), hipMemcpyHostToDevice, stream));
    const size_t total = staged.size();
    const uint32_t grid = static_cast<uint32_t>(std::min<size_t>((total + 255) / 256, 4096));
    hipLaunchKernelGGL(k_scatter_rows, dim3(grid), dim3(256), 0, stream, mr.d_stage, mr.d_index, static_cast<uint32_t>(index.size()), row_words,
                       static_cast<uint2 *>(dst));
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(stream));  // the staging vectors are reused by the caller
    mr.last_upload_bytes += staged.size() * sizeof(uint2) + index.size() * sizeof(uint32_t);
    return KICP_OK;
}

}  // namespace

namespace kicp {
namespace host {

int map_sync(kicp_map *map, int device, hipStream_t stream) {
    if (int rc = map_finish_pending(map)) return rc;
    DeviceMirror &mr = map->mirror;
    HostMap &h = map->host;
    if (map->device_ahead) {
        if (mr.device == device) return KICP_OK;  // the HBM copy is the current one
        if (int rc = ensure_host_current(map)) return rc;
    }
    if (mr.device == device && mr.synced_epoch == h.epoch()) return KICP_OK;
    if (int rc = set_device(device)) return rc;
    if (mr.device != device && mr.device >= 0) {  // mirror lives on another GPU: drop it
        free_mirror(mr);
        hipSetDevice(device);
    }
    mr.device = device;
    const size_t slots = h.table().size();
    const uint32_t cap = h.cap();
    const size_t pool_doubles = h.buckets_in_use_hi() * static_cast<size_t>(cap) * 3;
    bool full = mr.synced_generation != h.generation() || mr.live_slots != slots;
    if (slots > mr.table_slots) {
        if (mr.d_table) HIP_TRY(hipFree(mr.d_table));
        mr.d_table = nullptr;
        HIP_TRY(hipMalloc(&mr.d_table, slots * sizeof(Slot)));
        mr.table_slots = slots;
        full = true;
    }
    if (pool_doubles > mr.pool_doubles) {  // grow the pools, keeping what is already there (device-side copy)
        const size_t want = pool_doubles + pool_doubles / 2 + 3 * 1024;
        double *np = nullptr;
        MirrorPoint *np32 = nullptr;
        HIP_TRY(hipMalloc(&np, want * sizeof(double)));
        HIP_TRY(hipMalloc(&np32, mirror_points(want, cap) * sizeof(MirrorPoint)));
        if (mr.d_pool && !full) {
            HIP_TRY(hipMemcpyAsync(np, mr.d_pool, mr.pool_doubles * sizeof(double), hipMemcpyDeviceToDevice, stream));
            HIP_TRY(hipMemcpyAsync(np32, mr.d_pool16, mirror_points(mr.pool_doubles, cap) * sizeof(MirrorPoint), hipMemcpyDeviceToDevice, stream));
            HIP_TRY(hipStreamSynchronize(stream));
        }
        if (mr.d_pool) HIP_TRY(hipFree(mr.d_pool));
        if (mr.d_pool16) HIP_TRY(hipFree(mr.d_pool16));
        mr.d_pool = np, mr.d_pool16 = np32, mr.pool_doubles = want;
    }
    mr.last_upload_bytes = 0;
    // delta only pays off while the changed part is small
    if (!full && (h.dirty_slots().size() * 4 > slots || h.dirty_buckets().size() * 2 > h.buckets_in_use_hi())) full = true;
    if (full) {
        HIP_TRY(hipMemcpyAsync(mr.d_table, h.table().data(), slots * sizeof(Slot), hipMemcpyHostToDevice, stream));
        if (pool_doubles) HIP_TRY(hipMemcpyAsync(mr.d_pool, h.pool().data(), pool_doubles * sizeof(double), hipMemcpyHostToDevice, stream));
        if (pool_doubles) HIP_TRY(hipMemcpyAsync(mr.d_pool16, h.pool16().data(), mirror_points(pool_doubles, cap) * sizeof(MirrorPoint), hipMemcpyHostToDevice, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        mr.last_upload_bytes = slots * sizeof(Slot) + pool_doubles * sizeof(double) + mirror_points(pool_doubles, cap) * sizeof(MirrorPoint);
    } else {
        // gather the changed rows on the host, ship them with their indices, scatter on the device
        std::vector<uint2> staged;
        const std::vector<uint32_t> &ds = h.dirty_slots(), &db = h.dirty_buckets();
        staged.resize(ds.size() * (sizeof(Slot) / 8));
        for (size_t i = 0; i < ds.size(); ++i) std::memcpy(&staged[i * (sizeof(Slot) / 8)], &h.table()[ds[i]], sizeof(Slot));
        if (int rc = upload_rows(mr, staged, ds, sizeof(Slot) / 8, mr.d_table, stream)) return rc;
        staged.resize(db.size() * static_cast<size_t>(cap) * 3);
        for (size_t i = 0; i < db.size(); ++i)
            std::memcpy(&staged[i * static_cast<size_t>(cap) * 3], &h.pool()[static_cast<size_t>(db[i]) * cap * 3], static_cast<size_t>(cap) * 24);
        if (int rc = upload_rows(mr, staged, db, cap * 3, mr.d_pool, stream)) return rc;
        const size_t cap16 = h.cap16();
        staged.resize(db.size() * cap16);  // one 8-byte word per mirror point
        for (size_t i = 0; i < db.size(); ++i)
            std::memcpy(&staged[i * cap16], &h.pool16()[static_cast<size_t>(db[i]) * cap16], cap16 * sizeof(MirrorPoint));
        if (int rc = upload_rows(mr, staged, db, static_cast<uint32_t>(cap16), mr.d_pool16, stream)) return rc;
    }
    mr.last_upload_full = full ? 1 : 0;
    if (int rc = sync_aux(map, stream)) return rc;
    h.mark_synced(full);
    mr.view = MapView{mr.d_table, static_cast<uint32_t>(slots - 1), mr.d_pool, mr.d_pool16, cap, h.cap16(), h.voxel_size(), h.count_bits()};
    mr.synced_epoch = h.epoch(), mr.synced_generation = h.generation(), mr.live_slots = slots;
    return KICP_OK;
}

// bring the host copy up to date after device-side updates: same layouts, so this is a plain download
int ensure_host_current(kicp_map *map) {
    if (int rc = map_finish_pending(map)) return rc;
    if (!map->device_ahead) return KICP_OK;
    TraceScope trace_scope_("  (host copy refreshed from HBM)");
    DeviceMirror &mr = map->mirror;
    if (int rc = set_device(mr.device)) return rc;
    HIP_TRY(hipDeviceSynchronize());
    DevMapCounters c{};
    HIP_TRY(hipMemcpy(&c, mr.d_ctr, sizeof c, hipMemcpyDeviceToHost));
    const uint32_t cap = map->host.cap();
    std::vector<Slot> table(mr.live_slots);
    std::vector<double> pool(static_cast<size_t>(c.n_buckets_hi) * cap * 3);
    std::vector<MirrorPoint> pool16(static_cast<size_t>(c.n_buckets_hi) * map->host.cap16());
    std::vector<uint32_t> free_list(c.free_count);
    HIP_TRY(hipMemcpy(table.data(), mr.d_table, table.size() * sizeof(Slot), hipMemcpyDeviceToHost));
    if (!pool.empty()) HIP_TRY(hipMemcpy(pool.data(), mr.d_pool, pool.size() * 8, hipMemcpyDeviceToHost));
    if (!pool16.empty()) HIP_TRY(hipMemcpy(pool16.data(), mr.d_pool16, pool16.size() * sizeof(MirrorPoint), hipMemcpyDeviceToHost));
    if (!free_list.empty()) HIP_TRY(hipMemcpy(free_list.data(), mr.d_free_list, free_list.size() * 4, hipMemcpyDeviceToHost));
    map->host.Adopt(std::move(table), std::move(pool), std::move(pool16), c.n_buckets_hi, std::move(free_list));
    map->host.mark_synced(true);
    mr.synced_epoch = map->host.epoch(), mr.synced_generation = map->host.generation();  // the mirror already holds this state
    map->device_ahead = false;
    return KICP_OK;
}

}  // namespace host
}  // namespace kicp

namespace {

// make the device pools (and the free-list stack) hold at least `want_buckets` buckets, keeping their contents
int grow_pools(kicp_map *map, size_t want_buckets) {
    DeviceMirror &mr = map->mirror;
    const uint32_t cap = map->host.cap();
    const size_t have = mr.pool_doubles / (static_cast<size_t>(cap) * 3);
    if (want_buckets <= have && have <= mr.free_cap) return KICP_OK;
    const size_t buckets = std::max(want_buckets + want_buckets / 2 + 1024, have);
    const size_t doubles = buckets * cap * 3;
    double *np = nullptr;
    MirrorPoint *np32 = nullptr;
    uint32_t *nf = nullptr;
    HIP_TRY(hipMalloc(&np, doubles * sizeof(double)));
    HIP_TRY(hipMalloc(&np32, mirror_points(doubles, cap) * sizeof(MirrorPoint)));
    HIP_TRY(hipMalloc(&nf, (buckets + 1) * 4));
    if (mr.d_pool) HIP_TRY(hipMemcpy(np, mr.d_pool, mr.pool_doubles * sizeof(double), hipMemcpyDeviceToDevice));
    if (mr.d_pool16) HIP_TRY(hipMemcpy(np32, mr.d_pool16, mirror_points(mr.pool_doubles, cap) * sizeof(MirrorPoint), hipMemcpyDeviceToDevice));
    if (mr.d_free_list && mr.free_cap) HIP_TRY(hipMemcpy(nf, mr.d_free_list, std::min(mr.free_cap, buckets) * 4, hipMemcpyDeviceToDevice));
    hipFree(mr.d_pool), hipFree(mr.d_pool16), hipFree(mr.d_free_list);
    mr.d_pool = np, mr.d_pool16 = np32, mr.d_free_list = nf, mr.pool_doubles = doubles, mr.free_cap = buckets;
    mr.view.pool = mr.d_pool, mr.view.pool16 = mr.d_pool16;
    return KICP_OK;
}

int ensure_update_scratch(DeviceMirror &mr, size_t n) {
    if (n <= mr.upd_cap) return KICP_OK;
    hipFree(mr.d_world), hipFree(mr.d_slot_of), hipFree(mr.d_order), hipFree(mr.d_touched);
    mr.d_world = nullptr, mr.d_slot_of = nullptr, mr.d_order = nullptr, mr.d_touched = nullptr;
    const size_t cap = n + n / 4 + 1024;
    HIP_TRY(hipMalloc(&mr.d_world, cap * 24));
    HIP_TRY(hipMalloc(&mr.d_slot_of, cap * 4));
    HIP_TRY(hipMalloc(&mr.d_order, cap * 4));
    HIP_TRY(hipMalloc(&mr.d_touched, cap * 4));
    mr.upd_cap = cap;
    return KICP_OK;
}

// Move the live entries of the device table into a fresh table with room for `extra_entries` more at a load factor of
// at most 0.25 (the host map's ReserveEntries rule).  The HBM copy becomes the authoritative one.
int device_rehash(kicp_map *map, size_t extra_entries) {
    DeviceMirror &mr = map->mirror;
    hipStream_t st = nullptr;
    const size_t old_slots = mr.live_slots;
    HIP_TRY(hipMemsetAsync(&mr.d_ctr->touched, 0, 8, st));  // touched (borrowed as the live counter) + error
    mr.ctr_clean = false;
    const uint32_t grid_old = static_cast<uint32_t>(std::min<size_t>((old_slots + 255) / 256, 8192));
    hipLaunchKernelGGL(k_rehash_count, dim3(grid_old), dim3(256), 0, st, mr.d_table, static_cast<uint32_t>(old_slots), map->host.count_bits(), &mr.d_ctr->touched);
    uint32_t live = 0;
    HIP_TRY(hipMemcpy(&live, &mr.d_ctr->touched, 4, hipMemcpyDeviceToHost));
    size_t want = 1024;
    while ((live + extra_entries) * 4 > want) want *= 2;
    if (want > (1ull << 31)) return fail(KICP_ERR_CAPACITY, "voxel table would exceed 2^31 slots");
    Slot *nt = nullptr;
    unsigned long long *nk = nullptr;
    uint32_t *nc = nullptr, *ns = nullptr;
    HIP_TRY(hipMalloc(&nt, want * sizeof(Slot)));
    HIP_TRY(hipMalloc(&nk, want * 8));
    HIP_TRY(hipMalloc(&nc, want * 4));
    HIP_TRY(hipMalloc(&ns, want * 4));
    const uint32_t grid_new = static_cast<uint32_t>(std::min<size_t>((want + 255) / 256, 8192));
    hipLaunchKernelGGL(k_table_clear, dim3(grid_new), dim3(256), 0, st, nt, static_cast<uint32_t>(want));
    HIP_TRY(hipMemsetAsync(nk, 0xFF, want * 8, st));
    HIP_TRY(hipMemsetAsync(nc, 0, want * 4, st));
    hipLaunchKernelGGL(k_rehash_move, dim3(grid_old), dim3(256), 0, st, mr.d_table, static_cast<uint32_t>(old_slots), nt, nk,
                       static_cast<uint32_t>(want - 1), map->host.count_bits(), &mr.d_ctr->error);
    HIP_TRY(hipGetLastError());
    map->dev.n_entries = live, map->dev.touched = 0;
    HIP_TRY(hipMemcpyAsync(&mr.d_ctr->n_entries, &map->dev.n_entries, 4, hipMemcpyHostToDevice, st));
    uint32_t err = 0;
    HIP_TRY(hipMemcpy(&err, &mr.d_ctr->error, 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipStreamSynchronize(st));
    hipFree(mr.d_table), hipFree(mr.d_keys64), hipFree(mr.d_cnt), hipFree(mr.d_seg_start);
    mr.d_table = nt, mr.d_keys64 = nk, mr.d_cnt = nc, mr.d_seg_start = ns;
    mr.table_slots = mr.live_slots = mr.aux_slots = want;
    mr.view.table = nt, mr.view.mask = static_cast<uint32_t>(want - 1);
    map->device_ahead = true;
    if (err) return fail(KICP_ERR_CAPACITY, "a voxel coordinate left the +-2^20 range of the device-side map update");
    return KICP_OK;
}

// VoxelHashMap::Update(points, pose) = transform + AddPoints + RemovePointsFarFromLocation(pose.translation) with the points
// already in HBM.  Runs on the device, table growth and pool growth included; only degenerate calls (no points) take the
// host path.  `remove_origin` == nullptr: AddPoints only (no pruning); otherwise the origin of the pruning step.
int map_update_device(kicp_map *map, int device, const double *d_points, size_t n, const Pose &pose, const double *remove_origin, bool defer = false) {
    if (int rc = map_finish_pending(map)) return rc;
    DeviceMirror &mr = map->mirror;
    map->last_update_on_device = 0;
    auto host_fallback = [&]() -> int {
        std::vector<double> pts(3 * n);
        if (n) HIP_TRY(hipMemcpy(pts.data(), d_points, n * 24, hipMemcpyDeviceToHost));
        if (int rc = ensure_host_current(map)) return rc;
        map->host.ReserveEntries(32 * n + 1024);
        std::vector<double> w(3 * n);
        for (size_t i = 0; i < n; ++i) {
            double rx, ry, rz;
            quat_rotate(pose, pts[3 * i], pts[3 * i + 1], pts[3 * i + 2], rx, ry, rz);
            w[3 * i] = rx + pose.tx, w[3 * i + 1] = ry + pose.ty, w[3 * i + 2] = rz + pose.tz;
        }
        if (!map->host.AddPoints(w.data(), n)) return fail(KICP_ERR_CAPACITY, "more than 2^24-2 voxels");
        if (remove_origin) map->host.RemovePointsFarFromLocation(remove_origin);
        return KICP_OK;
    };
    // (a map that holds a voxel the packed keys cannot express - through host-side AddPoints, or inherited by a clone - stays on
    //  the host: the device kernels would alias its key)
    if (n == 0 || n > 0x7FFFFFF0ull / 3 || map->host_updates_only || map->host.has_far_voxel()) return host_fallback();
    if (int rc = set_device(device)) return rc;
    if (int rc = map_sync(map, device, nullptr)) return rc;
    const uint32_t cap = map->host.cap();
    // room for the points' own voxels: <= n new buckets (the pools grow on the device) ...
    if (map->dev.n_buckets_hi + n > max_buckets(map->host.count_bits())) return fail(KICP_ERR_CAPACITY, "more voxels than the table's bucket index can address (2^24-2 for max_points_per_voxel <= 255)");
    if (int rc = grow_pools(map, map->dev.n_buckets_hi + n)) return rc;
    if (int rc = ensure_update_scratch(mr, n)) return rc;
    const uint32_t grid = static_cast<uint32_t>((n + 255) / 256);
    hipStream_t st = nullptr;
    UpdateParams up{};
    DevMapCounters c{};
    auto bind = [&]() {
        const size_t slots = mr.live_slots, bucket_cap = mr.pool_doubles / (static_cast<size_t>(cap) * 3);
        up.m = DevMap{mr.d_table, mr.d_keys64, static_cast<uint32_t>(slots - 1), mr.d_pool, mr.d_pool16, cap, map->host.count_bits(), static_cast<uint32_t>(bucket_cap),
                      map->host.voxel_size(), map->host.max_distance(), mr.d_free_list, mr.d_cnt, mr.d_seg_start, mr.d_ctr};
        up.in = d_points, up.n = static_cast<uint32_t>(n), up.pose = pose, up.world = mr.d_world, up.slot_of = mr.d_slot_of, up.order = mr.d_order;
        up.touched = mr.d_touched;
    };
    // steps 3, 4 and the far-voxel sweep; `touched_bound`: the number of touched voxels, or an upper bound of it (the kernels
    // read the exact number on the device)
    auto enqueue_apply = [&](size_t touched_bound) {
        hipLaunchKernelGGL(k_up_scatter, dim3(grid), dim3(256), 0, st, up);
        if (touched_bound <= 16384 && cap <= kApplyMaxPoints)  // a frame's worth of voxels: one wave each; bulk insertions: one thread each (kicp_mapdev.hpp steps 4 / 4b)
            hipLaunchKernelGGL(k_up_apply, dim3(static_cast<uint32_t>((touched_bound + kApplyWaves - 1) / kApplyWaves)), dim3(64 * kApplyWaves), 0, st, up);
        else
            hipLaunchKernelGGL(k_up_apply_thread, dim3(static_cast<uint32_t>((touched_bound + 63) / 64)), dim3(64), 0, st, up);
        if (remove_origin)
            hipLaunchKernelGGL(k_up_remove, dim3(static_cast<uint32_t>(std::min<size_t>((mr.live_slots / 4 + 255) / 256, 8192))), dim3(256), 0, st, up.m,
                               remove_origin[0], remove_origin[1], remove_origin[2]);
    };
    // ... and <= n new table entries without leaving the probing regime: re-hash (on the device) when the table is short
    if ((map->dev.n_entries + n) * 2 > mr.live_slots)
        if (int rc = device_rehash(map, 32 * n + 1024)) return rc;
    // Every touched voxel that holds no point yet - a fresh entry or a halo entry that existed before - may become occupied in
    // k_up_apply and then adds up to 26 halo entries of its own; the update only goes ahead with that head-room (load factor
    // <= 0.75 in the worst case, so that no probe sequence can run away).  With room for the worst case - every point a new voxel
    // with 26 new neighbours - nothing has to be asked of the device in between and the update is ONE queue of kernels behind
    // one synchronisation; a claim step that gives up (voxel coordinate out of range) turns the later steps into no-ops.
    if (n <= 16384 && (map->dev.n_entries + 27ull * n) * 4 <= mr.live_slots * 3ull) {
        bind();
        if (!mr.ctr_clean) HIP_TRY(hipMemsetAsync(&mr.d_ctr->touched, 0, 12, st));  // touched + error + may_occupy (k_up_publish left them at zero otherwise)
        mr.ctr_clean = false;
        hipLaunchKernelGGL(k_up_claim, dim3(grid), dim3(256), 0, st, up);
        hipLaunchKernelGGL(k_up_scan, dim3(1), dim3(1024), 0, st, up);
        enqueue_apply(n);
        HIP_TRY(hipGetLastError());
        if (defer) {
            // the caller collects the end of this update later (kicp_map_update_finish, or whatever it calls on the map next):
            // the counters land in pinned memory, nothing is waited for here
            if (!mr.h_ctr) {
                HIP_TRY(pinned_alloc(reinterpret_cast<void **>(&mr.h_ctr), 8 * sizeof(unsigned long long), hipHostMallocMapped | hipHostMallocCoherent));
                std::memset(mr.h_ctr, 0, 8 * sizeof(unsigned long long));
                HIP_TRY(hipHostGetDevicePointer(reinterpret_cast<void **>(&mr.h_ctr_dev), mr.h_ctr, 0));
            }
            hipLaunchKernelGGL(k_up_publish, dim3(1), dim3(64), 0, st, mr.d_ctr, mr.h_ctr_dev, ++mr.ctr_seq);
            HIP_TRY(hipGetLastError());
            mr.ctr_clean = true;
            map->device_ahead = true;
            map->pending_update = true, map->pending_points = d_points, map->pending_n = n, map->pending_pose = pose;
            map->pending_has_origin = remove_origin != nullptr;
            if (remove_origin) map->pending_origin[0] = remove_origin[0], map->pending_origin[1] = remove_origin[1], map->pending_origin[2] = remove_origin[2];
            map->last_update_on_device = 1;  // (corrected by the finish step should the host have to take the update over)
            return KICP_OK;
        }
        HIP_TRY(hipMemcpyAsync(&c, mr.d_ctr, sizeof c, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        map->device_ahead = true;
        map->dev = c;
        if (c.error == 3) return fail(KICP_ERR_CAPACITY, "device-side map update: voxel table full");
        if (c.error == 1) {  // (see below: the host map takes over)
            map->host_updates_only = true;
            return host_fallback();
        }
        if (c.error) return fail(KICP_ERR_CAPACITY, "device-side map update ran out of room");
        map->last_update_on_device = 1, ++map->device_updates;
        return KICP_OK;
    }
    for (int attempt = 0;; ++attempt) {
        if (attempt == 0 && (map->dev.n_entries + n) * 2 > mr.live_slots)
            if (int rc = device_rehash(map, 32 * n + 1024)) return rc;
        bind();
        HIP_TRY(hipMemsetAsync(&mr.d_ctr->touched, 0, 12, st));  // touched + error + may_occupy
        mr.ctr_clean = false;
        hipLaunchKernelGGL(k_up_claim, dim3(grid), dim3(256), 0, st, up);
        if (n > 16384) {  // bulk: the scan over many workgroups (the number of touched voxels is only known on the device: <= n)
            const uint32_t spans = static_cast<uint32_t>((n + kScanSpan - 1) / kScanSpan);
            hipLaunchKernelGGL(k_up_scan_local, dim3(spans), dim3(256), 0, st, up, mr.d_order);
            hipLaunchKernelGGL(k_up_scan_sums, dim3(1), dim3(1024), 0, st, up, mr.d_order);
            hipLaunchKernelGGL(k_up_scan_add, dim3(grid), dim3(256), 0, st, up, mr.d_order);
        } else {
            hipLaunchKernelGGL(k_up_scan, dim3(1), dim3(1024), 0, st, up);
        }
        HIP_TRY(hipMemcpyAsync(&c, mr.d_ctr, sizeof c, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        map->device_ahead = true;  // the table now carries the new voxels' (still empty) entries
        if (c.error == 3) return fail(KICP_ERR_CAPACITY, "device-side map update: voxel table full");
        if (c.error) {
            // A voxel coordinate beyond +-2^20 (the packed keys' range; the reference has no such limit): nothing was inserted for
            // such points, and what the claim step did insert are plain halo entries.  The host map takes this update - and every
            // later one of this map - over.
            map->host_updates_only = true;
            map->dev = c;
            return host_fallback();
        }
        map->dev = c;
        if ((c.n_entries + 26ull * c.may_occupy) * 4 <= mr.live_slots * 3ull) break;
        if (attempt) return fail(KICP_ERR_CAPACITY, "device-side map update found no room after a re-hash");
        // Too tight.  The entries just claimed are still plain halo entries without an occupied neighbour, so a re-hash drops
        // them together with the per-slot counters of this attempt; then claim again in the larger table.
        if (int rc = device_rehash(map, 32 * n + 1024)) return rc;
    }
    enqueue_apply(c.touched);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(&c, mr.d_ctr, sizeof c, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (c.error) return fail(KICP_ERR_CAPACITY, c.error == 3 ? "device-side map update: voxel table full" : "device-side map update ran out of room");
    map->dev = c;
    map->last_update_on_device = 1, ++map->device_updates;
    return KICP_OK;
}

}  // namespace
namespace kicp {
namespace host {
// The end of an update begun with defer = true: wait for its kernels, take the counters over, and - should the claim step have met
// a voxel the packed keys cannot express - let the host map redo the update from the (still borrowed) points.
int map_finish_pending(kicp_map *map) {
    if (!map->pending_update) return KICP_OK;
    map->pending_update = false;
    DeviceMirror &mr = map->mirror;
    if (int rc = set_device(mr.device)) return rc;
    // the update's last launch said so in host memory (k_up_publish); a word that is not there after 2 ms: synchronise, look again
    const volatile unsigned long long *words = mr.h_ctr;
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spins = 0; words[7] != mr.ctr_seq; ++spins) {
        if ((spins & 255u) == 255u && std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() > 2.0) {
            HIP_TRY(hipStreamSynchronize(nullptr));
            if (words[7] != mr.ctr_seq) return fail(KICP_ERR_HIP, "the map update finished without handing its counters over");
            break;
        }
        __builtin_ia32_pause();
    }
    DevMapCounters c;
    unsigned long long raw[5];
    for (int w = 0; w < 5; ++w) raw[w] = words[w];
    std::memcpy(&c, raw, sizeof c);
    map->dev = c;
    if (c.error == 3) return fail(KICP_ERR_CAPACITY, "device-side map update: voxel table full");
    if (c.error == 1) {
        map->host_updates_only = true;
        return map_update_device(map, mr.device, map->pending_points, map->pending_n, map->pending_pose, map->pending_has_origin ? map->pending_origin : nullptr);
    }
    if (c.error) return fail(KICP_ERR_CAPACITY, "device-side map update ran out of room");
    ++map->device_updates;
    return KICP_OK;
}
}  // namespace host
}  // namespace kicp
namespace {
// Host-side AddPoints / Update calls with many points go through the device path when the map has a preferred device
// (kicp_map_set_device): the points are staged into HBM and inserted there - the same map as the host insertion builds
// (tests/test_gpu_mapdev.py), an order of magnitude faster (the host table's 128-byte slots and 27 neighbour records per
// newly occupied voxel make the host insertion cache-miss bound: 9 us per point at cfg5's 2.9M voxels).
constexpr size_t kBulkThreshold = 4096;
int bulk_insert(kicp_map *map, const double *xyz, size_t n, const Pose &pose, const double *remove_origin) {
    DeviceMirror &mr = map->mirror;
    const int device = map->bulk_device;
    if (int rc = set_device(device)) return rc;
    if (n > map->bulk_cap) {
        hipFree(map->d_bulk);
        map->d_bulk = nullptr, map->bulk_cap = 0;
        HIP_TRY(hipMalloc(&map->d_bulk, (n + n / 4 + 1024) * 24));
        map->bulk_cap = n + n / 4 + 1024;
    }
    if (int rc = staged_upload(mr.stage, 0, map->d_bulk, xyz, n * 24, nullptr)) return rc;
    return map_update_device(map, device, map->d_bulk, n, pose, remove_origin);
}
bool use_bulk(const kicp_map *map, size_t n) { return map->bulk_device >= 0 && n >= kBulkThreshold; }
const Pose kIdentityPose{0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0};

}  // namespace

extern "C" {

// ---- map ------------------------------------------------------------------------------------------------------------
int kicp_map_create(double voxel_size, double max_distance, unsigned int max_points_per_voxel, kicp_map **out) {
    if (!out || !(voxel_size > 0.0) || max_points_per_voxel == 0) return fail(KICP_ERR_ARG, "bad map parameters");
    if (max_points_per_voxel > kMaxPointsPerVoxel) return fail(KICP_ERR_CAPACITY, "max_points_per_voxel > 65535");
    *out = new kicp_map(voxel_size, max_distance, max_points_per_voxel);
    return KICP_OK;
}
void kicp_map_destroy(kicp_map *map) {
    if (!map) return;
    (void)map_finish_pending(map);
    if (map->d_bulk) {
        hipSetDevice(map->bulk_device >= 0 ? map->bulk_device : 0);
        hipFree(map->d_bulk);
    }
    free_mirror(map->mirror);
    delete map;
}
int kicp_map_clone(const kicp_map *map, kicp_map **out) {
    if (!map || !out) return fail(KICP_ERR_ARG, "null argument");
    if (int rc = ensure_host_current(const_cast<kicp_map *>(map))) return rc;  // (refreshing the host copy does not change the map's value)
    kicp_map *c = new kicp_map(map->host.voxel_size(), map->host.max_distance(), map->host.cap());
    c->host = map->host;  // table, pools, free list, counters; the copy's mirror starts empty and uploads on first use
    c->bulk_device = map->bulk_device;
    c->host_updates_only = map->host_updates_only;
    *out = c;
    return KICP_OK;
}
int kicp_map_set_device(kicp_map *map, int device) {
    if (!map) return fail(KICP_ERR_ARG, "null map");
    if (device >= 0) {
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || device >= ndev) return fail(KICP_ERR_ARG, "device index out of range");
    }
    if (map->d_bulk && device != map->bulk_device) {
        hipSetDevice(map->bulk_device);
        hipFree(map->d_bulk);
        map->d_bulk = nullptr, map->bulk_cap = 0;
    }
    map->bulk_device = device < 0 ? -1 : device;
    return KICP_OK;
}
int kicp_map_clear(kicp_map *map) {
    KICP_TRACE_CALL();
    if (!map) return fail(KICP_ERR_ARG, "null map");
    (void)map_finish_pending(map);  // (its kernels must have left the table before anything else is queued on it)
    map->device_ahead = false;  // whatever the device holds is obsolete now
    map->host_updates_only = false;
    map->host.Clear();
    return KICP_OK;
}
int kicp_map_empty(const kicp_map *map) {
    if (!map) return 1;
    (void)map_finish_pending(const_cast<kicp_map *>(map));
    return (map->device_ahead ? map->dev.n_voxels == 0 : map->host.Empty()) ? 1 : 0;
}
int kicp_map_add_points(kicp_map *map, const double *xyz, size_t n) {
    KICP_TRACE_CALL();
    if (!map || (!xyz && n)) return fail(KICP_ERR_ARG, "null argument");
    if (use_bulk(map, n)) return bulk_insert(map, xyz, n, kIdentityPose, nullptr);
    if (int rc = ensure_host_current(map)) return rc;
    return map->host.AddPoints(xyz, n) ? KICP_OK : fail(KICP_ERR_CAPACITY, "more than 2^24-2 voxels");
}
int kicp_map_remove_far(kicp_map *map, const double origin[3]) {
    KICP_TRACE_CALL();
    if (!map || !origin) return fail(KICP_ERR_ARG, "null argument");
    if (int rc = ensure_host_current(map)) return rc;
    map->host.RemovePointsFarFromLocation(origin);
    return KICP_OK;
}
int kicp_map_update_origin(kicp_map *map, const double *xyz, size_t n, const double origin[3]) {
    KICP_TRACE_CALL();
    if (!map || (!xyz && n) || !origin) return fail(KICP_ERR_ARG, "null argument");
    if (use_bulk(map, n)) return bulk_insert(map, xyz, n, kIdentityPose, origin);
    if (int rc = ensure_host_current(map)) return rc;
    return map->host.Update(xyz, n, origin) ? KICP_OK : fail(KICP_ERR_CAPACITY, "more than 2^24-2 voxels");
}
int kicp_map_update_pose(kicp_map *map, const double *xyz, size_t n, const double pose_qt[7]) {
    KICP_TRACE_CALL();
    if (!map || (!xyz && n) || !pose_qt) return fail(KICP_ERR_ARG, "null argument");
    if (use_bulk(map, n)) {
        const Pose pose = pose_from(pose_qt);
        const double origin[3] = {pose.tx, pose.ty, pose.tz};
        return bulk_insert(map, xyz, n, pose, origin);
    }
    if (int rc = ensure_host_current(map)) return rc;
    return map->host.Update(xyz, n, pose_from(pose_qt)) ? KICP_OK : fail(KICP_ERR_CAPACITY, "more than 2^24-2 voxels");
}
int kicp_map_update_pose_device(kicp_map *map, int device, const double *d_points_xyz, size_t n, const double pose_qt[7]) {
    KICP_TRACE_CALL();
    if (!map || (!d_points_xyz && n) || !pose_qt) return fail(KICP_ERR_ARG, "null argument");
    const Pose pose = pose_from(pose_qt);
    const double origin[3] = {pose.tx, pose.ty, pose.tz};
    return map_update_device(map, device, d_points_xyz, n, pose, origin);
}
// The same update in two halves: _begin queues its kernels (and the copy of its counters) and returns without waiting - the caller
// goes on with host-side work that does not touch the map (the pipeline collects the frame's returned clouds) -, kicp_map_update_finish
// waits and takes the result over.  `d_points_xyz` stays borrowed until then.  Frame-sized updates into a table with room for the
// worst case take this route; anything else (table growth, bulk insertions, a map that lives on the host) runs to completion inside
// _begin.  Any other call on the map finishes a pending update first.
int kicp_map_update_pose_device_begin(kicp_map *map, int device, const double *d_points_xyz, size_t n, const double pose_qt[7]) {
    KICP_TRACE_CALL();
    if (!map || (!d_points_xyz && n) || !pose_qt) return fail(KICP_ERR_ARG, "null argument");
    const Pose pose = pose_from(pose_qt);
    // (queuing the update's six launches from a thread of the map's own - ~20 us of API time off the caller's thread - was measured:
    //  the frame then waits that much longer for its way back; no gain)
    const double origin[3] = {pose.tx, pose.ty, pose.tz};
    return map_update_device(map, device, d_points_xyz, n, pose, origin, true);
}
int kicp_map_update_finish(kicp_map *map) {
    KICP_TRACE_CALL();
    if (!map) return fail(KICP_ERR_ARG, "null map");
    return map_finish_pending(map);
}
unsigned long long kicp_map_device_updates(const kicp_map *map) {
    if (map) (void)map_finish_pending(const_cast<kicp_map *>(map));
    return map ? map->device_updates : 0ull;
}
int kicp_map_last_update_on_device(const kicp_map *map) {
    if (map) (void)map_finish_pending(const_cast<kicp_map *>(map));
    return map ? map->last_update_on_device : 0;
}
size_t kicp_map_num_points(const kicp_map *map) {
    if (!map) return 0;
    (void)map_finish_pending(const_cast<kicp_map *>(map));
    return map->device_ahead ? static_cast<size_t>(map->dev.n_points) : map->host.num_points();
}
size_t kicp_map_num_voxels(const kicp_map *map) {
    if (!map) return 0;
    (void)map_finish_pending(const_cast<kicp_map *>(map));
    return map->device_ahead ? map->dev.n_voxels : map->host.num_voxels();
}
size_t kicp_map_pointcloud(const kicp_map *cmap, double *out_xyz, size_t cap_points) {
    KICP_TRACE_CALL();
    if (!cmap) return 0;
    kicp_map *map = const_cast<kicp_map *>(cmap);  // logically const: only scratch buffers / the host copy are touched
    if (!map->device_ahead) return map->host.Pointcloud(out_xyz, out_xyz ? cap_points : 0);
    // the HBM copy is the current one: gather the points there (same table order) and download just them
    const size_t total = static_cast<size_t>(map->dev.n_points);
    const size_t want = out_xyz ? std::min(total, cap_points) : 0;
    if (want == 0) return total;
    DeviceMirror &mr = map->mirror;
    auto gather = [&]() -> int {
        if (int rc = set_device(mr.device)) return rc;
        const size_t slots = mr.live_slots, blocks = (slots + 255) / 256;
        if (blocks + 1 > mr.pc_blocks) {
            hipFree(mr.d_pc_blocks);
            mr.d_pc_blocks = nullptr, mr.pc_blocks = 0;
            HIP_TRY(hipMalloc(&mr.d_pc_blocks, (blocks + 1) * 4));
            mr.pc_blocks = blocks + 1;
        }
        if (total > mr.pc_points) {
            hipFree(mr.d_pc);
            mr.d_pc = nullptr, mr.pc_points = 0;
            HIP_TRY(hipMalloc(&mr.d_pc, (total + total / 4 + 1024) * 24));
            mr.pc_points = total + total / 4 + 1024;
        }
        hipStream_t st = nullptr;
        hipLaunchKernelGGL(k_pc_count, dim3(static_cast<uint32_t>(blocks)), dim3(256), 0, st, mr.d_table, mr.d_keys64, static_cast<uint32_t>(slots), map->host.count_bits(), mr.d_pc_blocks);
        hipLaunchKernelGGL(k_scan_blocks, dim3(1), dim3(1024), 0, st, mr.d_pc_blocks, static_cast<uint32_t>(blocks), mr.d_pc_blocks + blocks);
        hipLaunchKernelGGL(k_pc_gather, dim3(static_cast<uint32_t>(blocks)), dim3(256), 0, st, mr.d_table, mr.d_keys64, static_cast<uint32_t>(slots),
                           mr.d_pool, map->host.cap(), map->host.count_bits(), mr.d_pc_blocks, mr.d_pc);
        HIP_TRY(hipGetLastError());
        uint32_t counted = 0;
        HIP_TRY(hipMemcpy(&counted, mr.d_pc_blocks + blocks, 4, hipMemcpyDeviceToHost));
        if (counted != total) return fail(KICP_ERR_HIP, "device map counters disagree with the table");
        if (int rc = staged_download(mr.stage, out_xyz, mr.d_pc, want * 24, st)) return rc;
        return KICP_OK;
    };
    if (gather() != KICP_OK) {  // fall back to refreshing the host copy
        if (ensure_host_current(map) != KICP_OK) return 0;
        return map->host.Pointcloud(out_xyz, cap_points);
    }
    return total;
}
size_t kicp_map_check(const kicp_map *map) {
    if (!map || ensure_host_current(const_cast<kicp_map *>(map)) != KICP_OK) return 1;
    return map->host.CheckInvariants();
}
int kicp_map_sync(kicp_map *map, int device) {
    KICP_TRACE_CALL();
    if (!map) return fail(KICP_ERR_ARG, "null map");
    if (int rc = set_device(device)) return rc;
    return map_sync(map, device, nullptr);
}
size_t kicp_map_device_bytes(const kicp_map *map) {
    if (!map || !map->mirror.d_table) return 0;
    const DeviceMirror &mr = map->mirror;
    // what a query can touch: the table's entries (occupied + halo) and the buckets of the occupied voxels, not the head-room
    // around them
    const size_t entries = map->device_ahead ? map->dev.n_entries : map->host.num_entries();
    const size_t voxels = map->device_ahead ? map->dev.n_voxels : map->host.num_voxels();
    (void)mr;
    return entries * sizeof(Slot) + voxels * (static_cast<size_t>(map->host.cap()) * 24 + static_cast<size_t>(map->host.cap16()) * sizeof(MirrorPoint));
}
int kicp_map_last_upload(const kicp_map *map, size_t *bytes, int *was_full) {
    if (!map || !bytes || !was_full) return fail(KICP_ERR_ARG, "null argument");
    *bytes = map->mirror.last_upload_bytes, *was_full = map->mirror.last_upload_full;
    return KICP_OK;
}
int kicp_map_closest(kicp_map *map, int device, const double *queries_xyz, size_t n, double *out_nn_xyz, double *out_dist) {
    KICP_TRACE_CALL();
    if (!map || (!queries_xyz && n) || !out_nn_xyz || !out_dist) return fail(KICP_ERR_ARG, "null argument");
    if (n == 0) return KICP_OK;
    if (kicp_map_empty(map)) {
        for (size_t i = 0; i < n; ++i) out_nn_xyz[3 * i] = out_nn_xyz[3 * i + 1] = out_nn_xyz[3 * i + 2] = 0.0, out_dist[i] = DBL_MAX;
        return KICP_OK;
    }
    if (int rc = kicp_map_sync(map, device)) return rc;
    double *d_q = nullptr, *d_nn = nullptr, *d_d = nullptr;
    HIP_TRY(hipMalloc(&d_q, n * 24));
    HIP_TRY(hipMalloc(&d_nn, n * 24));
    HIP_TRY(hipMalloc(&d_d, n * 8));
    if (int rc = staged_upload(map->mirror.stage, 0, d_q, queries_xyz, n * 24, nullptr)) return rc;
    hipLaunchKernelGGL(k_closest, dim3(static_cast<uint32_t>((n + 255) / 256)), dim3(256), 0, nullptr, d_q, static_cast<uint32_t>(n),
                       map->mirror.view, d_nn, d_d);
    HIP_TRY(hipGetLastError());
    if (int rc = staged_download(map->mirror.stage, out_nn_xyz, d_nn, n * 24, nullptr)) return rc;
    if (int rc = staged_download(map->mirror.stage, out_dist, d_d, n * 8, nullptr)) return rc;
    hipFree(d_q), hipFree(d_nn), hipFree(d_d);
    return KICP_OK;
}

}  // extern "C"
