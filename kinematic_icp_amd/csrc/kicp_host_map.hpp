// kicp_host_map.hpp -- host side of the voxel map: the authoritative container behind the
// kiss_icp::VoxelHashMap API (kiss-icp v1.2.0 core/VoxelHashMap.{hpp,cpp}), stored
// directly in the layout the gfx950 kernels read (kicp_common.hpp: Slot table + fixed-stride bucket pool), so
// that "upload" is two plain copies and no per-scan re-packing happens.
//
// Insertion keeps the reference's order-dependent semantics (first come first kept, <= max_points_per_voxel,
// min spacing = voxel_size / sqrt(max_points_per_voxel) inside one voxel), which is why it stays sequential on
// the host for now (device-side maintenance is a section 8f "next" row).
#pragma once
#include <cmath>
#include <cstring>
#include <vector>

#include "kicp_common.hpp"
#include "kicp_se3.hpp"

namespace kicp {

class HostMap {
public:
    HostMap(double voxel_size, double max_distance, uint32_t max_points_per_voxel)
        : voxel_size_(voxel_size), max_distance_(max_distance), cap_(max_points_per_voxel) {
        Clear();
    }

    double voxel_size() const { return voxel_size_; }
    double max_distance() const { return max_distance_; }
    uint32_t cap() const { return cap_; }
    size_t num_voxels() const { return n_voxels_; }
    size_t num_points() const { return n_points_; }
    uint64_t epoch() const { return epoch_; }
    bool Empty() const { return n_voxels_ == 0; }
    const std::vector<Slot> &table() const { return table_; }
    const std::vector<double> &pool() const { return pool_; }
    size_t buckets_in_use_hi() const { return n_buckets_hi_; }  // pool prefix that may hold live buckets

    void Clear() {
        table_.assign(kMinTable, Slot{0, 0, 0, kEmptyVal});
        pool_.clear();
        free_.clear();
        n_buckets_hi_ = 0, n_voxels_ = 0, n_points_ = 0;
        ++epoch_;
    }

    // AddPoints(points) -- App. A.4.  Returns false if a documented capacity limit would be exceeded.
    bool AddPoints(const double *xyz, size_t n) {
        const double map_resolution = std::sqrt(voxel_size_ * voxel_size_ / cap_);
        for (size_t i = 0; i < n; ++i) {
            const double px = xyz[3 * i], py = xyz[3 * i + 1], pz = xyz[3 * i + 2];
            const int32_t vx = to_voxel(px), vy = to_voxel(py), vz = to_voxel(pz);
            const int64_t s = find(vx, vy, vz);
            if (s >= 0) {
                Slot &slot = table_[static_cast<size_t>(s)];
                const uint32_t count = slot.val & 0xffu, bucket = slot.val >> 8;
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
                slot.val = (bucket << 8) | (count + 1);
            } else {
                if (n_voxels_ + 1 > kMaxBuckets) return false;
                const uint32_t bucket = alloc_bucket();
                double *b = &pool_[static_cast<size_t>(bucket) * cap_ * 3];
                b[0] = px, b[1] = py, b[2] = pz;
                insert_new(vx, vy, vz, (bucket << 8) | 1u);
            }
            ++n_points_;
        }
        ++epoch_;
        return true;
    }

    // RemovePointsFarFromLocation(origin) -- App. A.6: drop a voxel when its FIRST point is >= max_distance away.
    void RemovePointsFarFromLocation(const double origin[3]) {
        const double max_distance2 = max_distance_ * max_distance_;
        for (size_t i = 0; i < table_.size();) {
            const Slot &slot = table_[i];
            if (slot.val != kEmptyVal) {
                const double *b = &pool_[static_cast<size_t>(slot.val >> 8) * cap_ * 3];
                const double dx = b[0] - origin[0], dy = b[1] - origin[1], dz = b[2] - origin[2];
                if (dx * dx + dy * dy + dz * dz >= max_distance2) {
                    if (erase_at(i)) continue;  // an element was shifted into i: test it too
                }
            }
            ++i;
        }
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
            const uint32_t count = slot.val & 0xffu;
            const double *b = &pool_[static_cast<size_t>(slot.val >> 8) * cap_ * 3];
            for (uint32_t k = 0; k < count && w < cap_points; ++k, ++w) std::memcpy(out + 3 * w, b + 3 * k, 24);
        }
        return n_points_;
    }

private:
    static constexpr size_t kMinTable = 1024;
    int32_t to_voxel(double c) const { return static_cast<int32_t>(std::floor(c / voxel_size_)); }  // PointToVoxel, App. A.1

    int64_t find(int32_t x, int32_t y, int32_t z) const {
        const size_t mask = table_.size() - 1;
        for (size_t i = voxel_hash(x, y, z) & mask;; i = (i + 1) & mask) {
            const Slot &s = table_[i];
            if (s.val == kEmptyVal) return -1;
            if (s.x == x && s.y == y && s.z == z) return static_cast<int64_t>(i);
        }
    }
    void place(int32_t x, int32_t y, int32_t z, uint32_t val) {
        const size_t mask = table_.size() - 1;
        size_t i = voxel_hash(x, y, z) & mask;
        while (table_[i].val != kEmptyVal) i = (i + 1) & mask;
        table_[i] = Slot{x, y, z, val};
    }
    void insert_new(int32_t x, int32_t y, int32_t z, uint32_t val) {
        if ((n_voxels_ + 1) * 4 > table_.size()) {  // keep load factor <= 0.25: misses end after ~1.2 probes
            std::vector<Slot> old(table_.size() * 2, Slot{0, 0, 0, kEmptyVal});
            old.swap(table_);
            for (const Slot &s : old)
                if (s.val != kEmptyVal) place(s.x, s.y, s.z, s.val);
        }
        place(x, y, z, val);
        ++n_voxels_;
    }
    uint32_t alloc_bucket() {
        if (!free_.empty()) {
            const uint32_t b = free_.back();
            free_.pop_back();
            return b;
        }
        const uint32_t b = static_cast<uint32_t>(n_buckets_hi_++);
        if (pool_.size() < n_buckets_hi_ * cap_ * 3) pool_.resize(std::max<size_t>(pool_.size() * 2, n_buckets_hi_ * cap_ * 3 + 1024 * cap_ * 3));
        return b;
    }
    // backward-shift deletion (keeps the table tombstone-free so device probes stop at the first empty slot)
    bool erase_at(size_t i) {
        const size_t mask = table_.size() - 1;
        n_points_ -= table_[i].val & 0xffu;
        free_.push_back(table_[i].val >> 8);
        table_[i].val = kEmptyVal;
        --n_voxels_;
        bool moved_into_i = false;
        size_t hole = i;
        for (size_t j = (i + 1) & mask; table_[j].val != kEmptyVal; j = (j + 1) & mask) {
            const size_t home = voxel_hash(table_[j].x, table_[j].y, table_[j].z) & mask;
            const bool stays = (hole <= j) ? (home > hole && home <= j) : (home > hole || home <= j);
            if (!stays) {
                table_[hole] = table_[j];
                table_[j].val = kEmptyVal;
                if (hole == i) moved_into_i = true;
                hole = j;
            }
        }
        return moved_into_i;
    }

    double voxel_size_, max_distance_;
    uint32_t cap_;
    std::vector<Slot> table_;
    std::vector<double> pool_;
    std::vector<uint32_t> free_;
    size_t n_buckets_hi_ = 0, n_voxels_ = 0, n_points_ = 0;
    uint64_t epoch_ = 0;
};

}  // namespace kicp
