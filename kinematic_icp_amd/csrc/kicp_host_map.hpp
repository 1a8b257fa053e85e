// kicp_host_map.hpp -- host side of the voxel map: the authoritative container behind the
// kiss_icp::VoxelHashMap API (kiss-icp v1.2.0 core/VoxelHashMap.{hpp,cpp}), stored
// directly in the layout the gfx950 kernels read (kicp_common.hpp: Slot table + fixed-stride bucket pool), so
// that "upload" is two plain copies and no per-scan re-packing happens.
//
// Besides the occupied voxels the table carries halo entries and per-entry 27-bit neighbour-occupancy masks
// (kicp_common.hpp::Slot), maintained here whenever a voxel gains its first / loses its last point.
//
// Insertion keeps the reference's order-dependent semantics (first come first kept, <= max_points_per_voxel,
// min spacing = voxel_size / sqrt(max_points_per_voxel) inside one voxel), which is why it stays sequential on
// the host for now (device-side maintenance is a section 8f "next" row).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "kicp_common.hpp"
#include "kicp_se3.hpp"

namespace kicp {

class HostMap {
public:
    HostMap(double voxel_size, double max_distance, uint32_t max_points_per_voxel)
        : voxel_size_(voxel_size), max_distance_(max_distance), cap_(max_points_per_voxel), cap16_(mirror_stride(max_points_per_voxel)),
          cb_(count_bits_for(max_points_per_voxel)) {
        Clear();
    }

    double voxel_size() const { return voxel_size_; }
    double max_distance() const { return max_distance_; }
    uint32_t cap() const { return cap_; }
    uint32_t count_bits() const { return cb_; }  // split of Slot::val (kicp_common.hpp::count_bits_for)
    uint32_t cap16() const { return cap16_; }  // bucket stride of the 16-bit mirror (kicp_common.hpp::mirror_stride)
    size_t num_voxels() const { return n_voxels_; }
    size_t num_points() const { return n_points_; }
    uint64_t epoch() const { return epoch_; }
    bool Empty() const { return n_voxels_ == 0; }
    const std::vector<Slot> &table() const { return table_; }
    const std::vector<double> &pool() const { return pool_; }
    const std::vector<MirrorPoint> &pool16() const { return pool16_; }  // 16-bit offsets from the voxel corner, stride cap
    size_t buckets_in_use_hi() const { return n_buckets_hi_; }  // pool prefix that may hold live buckets

    // ---- hand-over with the device-side maintenance (kicp_mapdev.hpp) ---------------------------------------------------
    const std::vector<uint32_t> &free_list() const { return free_; }
    size_t num_entries() const { return n_entries_; }
    // a voxel beyond the range the device-side maintenance can pack (+-(2^20 - 1), one voxel of head-room for the neighbours) has
    // been occupied since the last Clear(): updates of this map belong on the host (kicp_map.hip::map_update_device)
    bool has_far_voxel() const { return has_far_voxel_; }
    // make room so that `extra_entries` more table entries keep the load factor <= 0.25 (re-hashes if necessary)
    void ReserveEntries(size_t extra_entries) {
        size_t want = table_.size();
        while ((n_entries_ + extra_entries) * 4 > want) want *= 2;
        if (want != table_.size()) rebuild(want), ++epoch_;
    }
    // take over the state the device produced (same layouts): table, pools, free list; counters are recounted
    void Adopt(std::vector<Slot> &&table, std::vector<double> &&pool, std::vector<MirrorPoint> &&pool16, size_t n_buckets_hi,
               std::vector<uint32_t> &&free_list) {
        table_ = std::move(table), pool_ = std::move(pool), pool16_ = std::move(pool16), free_ = std::move(free_list);
        n_buckets_hi_ = n_buckets_hi;
        n_voxels_ = n_points_ = n_entries_ = n_dead_ = 0;
        for (const Slot &e : table_) {
            if (e.val == kEmptyVal) continue;
            ++n_entries_;
            const uint32_t c = val_count(e.val, cb_);
            if (c) ++n_voxels_, n_points_ += c;
            else if (e.nbr == 0) ++n_dead_;
        }
        slot_flag_.assign(table_.size(), 0), bucket_flag_.assign(n_buckets_hi_ + 1024, 0), dirty_slots_.clear(), dirty_buckets_.clear();
        ++generation_, ++epoch_;
    }

    // ---- change tracking for the HBM mirror (delta upload) ----------------------------------------------------------
    // generation() changes whenever slot positions change wholesale (Clear, re-hash): the mirror must then be re-sent
    // in full.  Otherwise dirty_slots()/dirty_buckets() list what changed since the last mark_synced().
    uint64_t generation() const { return generation_; }
    const std::vector<uint32_t> &dirty_slots() const { return dirty_slots_; }
    const std::vector<uint32_t> &dirty_buckets() const { return dirty_buckets_; }
    void mark_synced(bool was_full) {
        if (was_full) {  // after a re-hash every flag was raised without listing the slots
            std::fill(slot_flag_.begin(), slot_flag_.end(), 0);
            std::fill(bucket_flag_.begin(), bucket_flag_.end(), 0);
        } else {
            for (uint32_t i : dirty_slots_) slot_flag_[i] = 0;
            for (uint32_t b : dirty_buckets_) bucket_flag_[b] = 0;
        }
        dirty_slots_.clear(), dirty_buckets_.clear();
    }

    void Clear() {
        table_.assign(kMinTable, empty_slot());
        n_entries_ = 0, n_dead_ = 0;
        slot_flag_.assign(kMinTable, 0), bucket_flag_.clear(), dirty_slots_.clear(), dirty_buckets_.clear();
        ++generation_;
        pool_.clear();
        pool16_.clear();
        free_.clear();
        n_buckets_hi_ = 0, n_voxels_ = 0, n_points_ = 0;
        has_far_voxel_ = false;
        ++epoch_;
    }

    // AddPoints(points) -- App. A.4.  Returns false if a documented capacity limit would be exceeded.
    bool AddPoints(const double *xyz, size_t n) {
        const double map_resolution = std::sqrt(voxel_size_ * voxel_size_ / cap_);
        for (size_t i = 0; i < n; ++i) {
            const double px = xyz[3 * i], py = xyz[3 * i + 1], pz = xyz[3 * i + 2];
            const int32_t vx = to_voxel(px), vy = to_voxel(py), vz = to_voxel(pz);
            const int64_t s = find(vx, vy, vz);
            if (s >= 0 && (val_count(table_[static_cast<size_t>(s)].val, cb_)) != 0) {
                Slot &slot = table_[static_cast<size_t>(s)];
                const uint32_t count = val_count(slot.val, cb_), bucket = val_bucket(slot.val, cb_);
                if (count == cap_) continue;
                double *b = &pool_[static_cast<size_t>(bucket) * cap_ * 3];
                bool too_close = false;
                for (uint32_t k = 0; k < count; ++k) {
                    const double dx = b[3 * k] - px, dy = b[3 * k + 1] - py, dz = b[3 * k + 2] - pz;
                    if (std::sqrt(dx * dx + dy * dy + dz * dz) < map_resolution) {
                        too_close = true;
                        break;
                    }
                }
                if (too_close) continue;
                b[3 * count] = px, b[3 * count + 1] = py, b[3 * count + 2] = pz;
                store32(bucket, count, px, py, pz, vx, vy, vz);
                slot.val = make_val(bucket, count + 1, cb_);
                touch_slot(static_cast<size_t>(s)), touch_bucket(bucket);
            } else {
                if (n_voxels_ + 1 > max_buckets(cb_)) return false;
                constexpr int32_t kFar = (1 << 20) - 1;
                if (vx <= -kFar || vx >= kFar || vy <= -kFar || vy >= kFar || vz <= -kFar || vz >= kFar) has_far_voxel_ = true;
                const uint32_t bucket = alloc_bucket();
                double *b = &pool_[static_cast<size_t>(bucket) * cap_ * 3];
                b[0] = px, b[1] = py, b[2] = pz;
                store32(bucket, 0, px, py, pz, vx, vy, vz);
                touch_bucket(bucket);
                occupy(vx, vy, vz, make_val(bucket, 1u, cb_));
            }
            ++n_points_;
        }
        ++epoch_;
        return true;
    }

    // RemovePointsFarFromLocation(origin) -- App. A.6: drop a voxel when its FIRST point is >= max_distance away.
    void RemovePointsFarFromLocation(const double origin[3]) {
        const double max_distance2 = max_distance_ * max_distance_;
        for (size_t i = 0; i < table_.size(); ++i) {  // no entry moves during the sweep (erase leaves a halo/dead entry)
            const Slot &slot = table_[i];
            if (slot.val == kEmptyVal || (val_count(slot.val, cb_)) == 0) continue;
            const double *b = &pool_[static_cast<size_t>(val_bucket(slot.val, cb_)) * cap_ * 3];
            const double dx = b[0] - origin[0], dy = b[1] - origin[1], dz = b[2] - origin[2];
            if (dx * dx + dy * dy + dz * dz >= max_distance2) erase_voxel(i);
        }
        if (n_dead_ * 4 > n_entries_) rebuild(table_.size());
        ++epoch_;
    }

    // Update(points, origin) / Update(points, pose) -- App. A.5
    bool Update(const double *xyz, size_t n, const double origin[3]) {
        const bool ok = AddPoints(xyz, n);
        RemovePointsFarFromLocation(origin);
        return ok;
    }
    bool Update(const double *xyz, size_t n, const Pose &pose) {
        std::vector<double> w(3 * n);
        for (size_t i = 0; i < n; ++i) {
            double rx, ry, rz;
            quat_rotate(pose, xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2], rx, ry, rz);
            w[3 * i] = rx + pose.tx, w[3 * i + 1] = ry + pose.ty, w[3 * i + 2] = rz + pose.tz;
        }
        const double origin[3] = {pose.tx, pose.ty, pose.tz};
        return Update(w.data(), n, origin);
    }

    // Pointcloud(): all points, voxel by voxel in table order (the reference's order is robin_map's, unspecified).
    size_t Pointcloud(double *out, size_t cap_points) const {
        size_t w = 0;
        for (const Slot &slot : table_) {
            if (slot.val == kEmptyVal) continue;
            const uint32_t count = val_count(slot.val, cb_);  // 0 for halo entries
            const double *b = &pool_[static_cast<size_t>(val_bucket(slot.val, cb_)) * cap_ * 3];
            for (uint32_t k = 0; k < count && w < cap_points; ++k, ++w) std::memcpy(out + 3 * w, b + 3 * k, 24);
        }
        return n_points_;
    }

    // Self-check of the table invariants the kernels rely on; returns the number of violations (0 = consistent):
    //  * every occupied voxel has a live bucket whose fp32 header carries its count and whose fp32 points are the
    //    rounded offsets of the fp64 points from the voxel corner;
    //  * for every entry E and every shift s: bit s of E.nbr is set  <=>  voxel E.key + shift[s] is occupied, and then
    //    E.nb[s] is that voxel's bucket;
    //  * every voxel within one step of an occupied voxel has an entry (halo completeness);
    //  * the counters agree with the table.
    size_t CheckInvariants() const {
        size_t bad = 0, occupied = 0, points = 0, entries = 0, dead = 0;
        for (const Slot &e : table_) {
            if (e.val == kEmptyVal) continue;
            ++entries;
            const uint32_t count = val_count(e.val, cb_);
            if (count == 0 && e.nbr == 0) ++dead;
            if (count) {
                ++occupied, points += count;
                const uint32_t b = val_bucket(e.val, cb_);
                if (b >= n_buckets_hi_ || count > cap_) ++bad;
                for (uint32_t k = count; k < cap16_; ++k)  // empty slots are marked (what lets the kernel skip testing for them)
                    if ((pool16_[static_cast<size_t>(b) * cap16_ + k].y >> 16) != 0xFFFFu) ++bad;
                const double upm = mirror_units_per_metre(voxel_size_);
                for (uint32_t k = 0; k < count; ++k) {
                    const double *p = &pool_[(static_cast<size_t>(b) * cap_ + k) * 3];
                    const MirrorPoint f = pool16_[static_cast<size_t>(b) * cap16_ + k];
                    if (to_voxel(p[0]) != e.x || to_voxel(p[1]) != e.y || to_voxel(p[2]) != e.z) ++bad;
                    const MirrorPoint want = mirror_point(p[0] - e.x * voxel_size_, p[1] - e.y * voxel_size_, p[2] - e.z * voxel_size_, upm);
                    if (f.x != want.x || f.y != want.y) ++bad;
                    // the decoded offset is within one unit of the true one on every axis (what the kernel's margin assumes)
                    const double o[3] = {p[0] - e.x * voxel_size_, p[1] - e.y * voxel_size_, p[2] - e.z * voxel_size_};
                    const uint32_t q[3] = {f.x & 0xffffu, f.x >> 16, f.y & 0xffffu};
                    for (int a = 0; a < 3; ++a)
                        if (std::fabs(q[a] / upm - o[a]) > 1.0 / upm) ++bad;
                }
            } else if (e.val != halo_val(cb_)) {
                ++bad;
            }
            for (int s = 0; s < 27; ++s) {
                const int64_t n = find(e.x + kShiftTable[s][0], e.y + kShiftTable[s][1], e.z + kShiftTable[s][2]);
                const bool occ = n >= 0 && (val_count(table_[static_cast<size_t>(n)].val, cb_)) != 0;
                if (occ != (((e.nbr >> s) & 1u) != 0)) ++bad;
                if (occ && e.nb[s] != (val_bucket(table_[static_cast<size_t>(n)].val, cb_))) ++bad;
                if (count && n < 0) ++bad;  // halo completeness around occupied voxels
            }
        }
        if (occupied != n_voxels_ || points != n_points_ || entries != n_entries_ || dead != n_dead_) ++bad;
        return bad;
    }

private:
    static constexpr size_t kMinTable = 1024;
    int32_t to_voxel(double c) const { return static_cast<int32_t>(std::floor(c / voxel_size_)); }  // PointToVoxel, App. A.1

    // compact mirror used by the pre-selection pass (kicp_common.hpp::MirrorPoint): 16-bit offsets from the voxel corner,
    // so the error is <= voxel_size / 65536 per axis regardless of how far the map is from the origin
    void store32(uint32_t bucket, uint32_t k, double px, double py, double pz, int32_t vx, int32_t vy, int32_t vz) {
        MirrorPoint *row = &pool16_[static_cast<size_t>(bucket) * cap16_];
        if (k == 0)  // first point of a (possibly re-used) bucket: every other slot is empty from now on
            for (uint32_t j = 1; j < cap16_; ++j) row[j] = mirror_empty();
        row[k] = mirror_point(px - vx * voxel_size_, py - vy * voxel_size_, pz - vz * voxel_size_, mirror_units_per_metre(voxel_size_));
    }
    int64_t find(int32_t x, int32_t y, int32_t z) const {
        const size_t mask = table_.size() - 1;
        for (size_t i = voxel_hash(x, y, z) & mask;; i = (i + 1) & mask) {
            const Slot &s = table_[i];
            if (s.val == kEmptyVal) return -1;
            if (s.x == x && s.y == y && s.z == z) return static_cast<int64_t>(i);
        }
    }
    static Slot empty_slot() {
        Slot e{};
        e.val = kEmptyVal;
        return e;
    }
    size_t place(const Slot &e) {
        const size_t mask = table_.size() - 1;
        size_t i = voxel_hash(e.x, e.y, e.z) & mask;
        while (table_[i].val != kEmptyVal) i = (i + 1) & mask;
        table_[i] = e;
        touch_slot(i);
        return i;
    }
    void touch_slot(size_t i) {
        if (!slot_flag_[i]) slot_flag_[i] = 1, dirty_slots_.push_back(static_cast<uint32_t>(i));
    }
    void touch_bucket(uint32_t b) {
        if (bucket_flag_.size() <= b) bucket_flag_.resize(std::max<size_t>(2 * bucket_flag_.size(), b + 1024), 0);
        if (!bucket_flag_[b]) bucket_flag_[b] = 1, dirty_buckets_.push_back(b);
    }
    // re-hash the live entries (occupied voxels and halo entries that still have an occupied neighbour)
    void rebuild(size_t slots) {
        std::vector<Slot> old(slots, empty_slot());
        old.swap(table_);
        n_entries_ = 0, n_dead_ = 0;
        slot_flag_.assign(slots, 1), dirty_slots_.clear();  // every slot moved: the mirror is re-sent in full
        ++generation_;
        for (const Slot &e : old)
            if (e.val != kEmptyVal && ((val_count(e.val, cb_)) != 0 || e.nbr != 0)) place(e), ++n_entries_;
    }
    bool is_dead(const Slot &e) const { return val_count(e.val, cb_) == 0 && e.nbr == 0; }  // halo entry nobody needs any more
    template <typename F>
    void mutate(size_t i, F f) {
        const bool before = is_dead(table_[i]);
        f(table_[i]);
        touch_slot(i);
        n_dead_ += static_cast<size_t>(is_dead(table_[i])) - static_cast<size_t>(before);
    }
    // slot index of the entry for voxel (x,y,z), created as a (for now dead) halo entry if absent; never re-hashes
    size_t find_or_insert(int32_t x, int32_t y, int32_t z) {
        const int64_t s = find(x, y, z);
        if (s >= 0) return static_cast<size_t>(s);
        ++n_entries_, ++n_dead_;
        Slot e{};
        e.x = x, e.y = y, e.z = z, e.val = halo_val(cb_);
        return place(e);
    }
    // voxel (x,y,z) receives its first point: give it a bucket and tell the 27 voxels that see it
    void occupy(int32_t x, int32_t y, int32_t z, uint32_t val) {
        while ((n_entries_ + 27) * 2 > table_.size()) rebuild(table_.size() * 2);  // load factor <= 0.5, no re-hash below
        mutate(find_or_insert(x, y, z), [&](Slot &e) { e.val = val; });
        ++n_voxels_;
        for (int s = 0; s < 27; ++s)  // U + shift[s] == this voxel  <=>  U = this - shift[s]
            mutate(find_or_insert(x - kShiftTable[s][0], y - kShiftTable[s][1], z - kShiftTable[s][2]),
                   [&](Slot &e) { e.nbr |= 1u << s, e.nb[s] = val_bucket(val, cb_); });
    }
    // the voxel at slot i loses all its points (entries never move here)
    void erase_voxel(size_t i) {
        const int32_t x = table_[i].x, y = table_[i].y, z = table_[i].z;
        n_points_ -= val_count(table_[i].val, cb_);
        free_.push_back(val_bucket(table_[i].val, cb_));
        mutate(i, [&](Slot &e) { e.val = halo_val(cb_); });
        --n_voxels_;
        for (int s = 0; s < 27; ++s) {
            const int64_t u = find(x - kShiftTable[s][0], y - kShiftTable[s][1], z - kShiftTable[s][2]);
            if (u >= 0) mutate(static_cast<size_t>(u), [&](Slot &e) { e.nbr &= ~(1u << s); });
        }
    }
    uint32_t alloc_bucket() {
        if (!free_.empty()) {
            const uint32_t b = free_.back();
            free_.pop_back();
            return b;
        }
        const uint32_t b = static_cast<uint32_t>(n_buckets_hi_++);
        if (pool_.size() < n_buckets_hi_ * cap_ * 3) {
            const size_t buckets = std::max<size_t>(pool_.size() / (cap_ * 3) * 2, n_buckets_hi_ + 1024);
            pool_.resize(buckets * cap_ * 3);
            pool16_.resize(buckets * cap16_);
        }
        return b;
    }
    double voxel_size_, max_distance_;
    uint32_t cap_, cap16_, cb_;
    std::vector<Slot> table_;
    std::vector<double> pool_;
    std::vector<MirrorPoint> pool16_;
    std::vector<uint32_t> free_;
    size_t n_buckets_hi_ = 0, n_voxels_ = 0, n_points_ = 0;
    bool has_far_voxel_ = false;
    size_t n_entries_ = 0, n_dead_ = 0;  // table entries (occupied + halo); halo entries with no occupied neighbour left
    std::vector<uint8_t> slot_flag_, bucket_flag_;
    std::vector<uint32_t> dirty_slots_, dirty_buckets_;
    uint64_t generation_ = 0;
    uint64_t epoch_ = 0;
};

}  // namespace kicp
