// kiss_icp/core/VoxelUtils.hpp -- host pre-step of the pipeline (kiss-icp v1.2.0 core/VoxelUtils.{hpp,cpp};
// SURVEY.md App. A.1/A.7; call sites pipeline/KinematicICP.cpp:40,42).  First point per voxel wins and the survivors come
// out in the iteration order of the reference's tsl::robin_map<Voxel, Vector3d> after reserve(frame.size()): that order
// decides which point the second downsample keeps and the order of the map update, so it is part of the contract.  The
// container is not available here; its published insertion rule (power-of-two bucket count at load factor 0.5, ideal
// bucket = hash & mask, robin-hood displacement of the first resident that is strictly closer to home, iteration in
// ascending bucket index) is restated on two flat arrays.  This is the host twin of the device path
// (kicp_pre_voxel_downsample, csrc/kicp_pre.hpp), used when KICP_HOST_PRESTEPS is defined.
#pragma once
#include <Eigen/Core>
#include <cmath>
#include <cstdint>
#include <utility>
#include <vector>

namespace kiss_icp {
struct VoxelKey {
    int32_t x, y, z;
    bool operator==(const VoxelKey &o) const { return x == o.x && y == o.y && z == o.z; }
};
struct VoxelKeyHash {
    size_t operator()(const VoxelKey &v) const {
        return (static_cast<uint32_t>(v.x) * 73856093u) ^ (static_cast<uint32_t>(v.y) * 19349669u) ^ (static_cast<uint32_t>(v.z) * 83492791u);
    }
};
inline VoxelKey PointToVoxel(const Eigen::Vector3d &p, double voxel_size) {
    return {static_cast<int>(std::floor(p.x() / voxel_size)), static_cast<int>(std::floor(p.y() / voxel_size)),
            static_cast<int>(std::floor(p.z() / voxel_size))};
}
inline std::vector<Eigen::Vector3d> VoxelDownsample(const std::vector<Eigen::Vector3d> &frame, double voxel_size) {
    std::vector<Eigen::Vector3d> out;
    if (frame.empty()) return out;
    size_t buckets = 1;  // reserve(n): ceil(float(n) / 0.5f) rounded up to a power of two
    while (buckets < static_cast<size_t>(std::ceil(static_cast<float>(frame.size()) / 0.5f))) buckets <<= 1;
    const size_t mask = buckets - 1;
    constexpr size_t kFree = ~size_t(0);
    std::vector<size_t> resident(buckets, kFree);  // input index of the point kept in a bucket
    std::vector<size_t> travelled(buckets, 0);     // how far that point sits from its ideal bucket
    std::vector<VoxelKey> voxel_of(frame.size());
    for (size_t i = 0; i < frame.size(); ++i) {
        const VoxelKey v = voxel_of[i] = PointToVoxel(frame[i], voxel_size);
        size_t b = VoxelKeyHash()(v) & mask, far = 0;
        bool seen = false;
        for (; resident[b] != kFree && far <= travelled[b]; b = (b + 1) & mask, ++far)
            if (voxel_of[resident[b]] == v) {
                seen = true;
                break;
            }
        if (seen) continue;  // the first point of a voxel stays
        for (size_t who = i; who != kFree; b = (b + 1) & mask, ++far) {
            if (resident[b] == kFree) {
                resident[b] = who, travelled[b] = far;
                break;
            }
            if (far > travelled[b]) std::swap(who, resident[b]), std::swap(far, travelled[b]);
        }
    }
    for (size_t b = 0; b < buckets; ++b)
        if (resident[b] != kFree) out.emplace_back(frame[resident[b]]);
    return out;
}
}  // namespace kiss_icp
