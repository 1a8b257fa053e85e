// kiss_icp/core/VoxelUtils.hpp -- host pre-step of the pipeline (kiss-icp v1.2.0 core/VoxelUtils.{hpp,cpp};
// SURVEY.md App. A.1/A.7; call sites pipeline/KinematicICP.cpp:40,42).  First point per voxel wins; the output
// order here is first-seen order (the reference's is its hash table's iteration order - only the order of later sums
// depends on it).  SURVEY.md section 8f row 2: an on-device version is a "next" item.
#pragma once
#include <Eigen/Core>
#include <cmath>
#include <cstdint>
#include <unordered_set>
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
    std::unordered_set<VoxelKey, VoxelKeyHash> seen;
    seen.reserve(frame.size());
    std::vector<Eigen::Vector3d> out;
    out.reserve(frame.size());
    for (const auto &p : frame)
        if (seen.insert(PointToVoxel(p, voxel_size)).second) out.emplace_back(p);
    out.shrink_to_fit();
    return out;
}
}  // namespace kiss_icp
