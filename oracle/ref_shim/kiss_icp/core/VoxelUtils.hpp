// kiss_icp/core/VoxelUtils.hpp -- STAND-IN (test infrastructure, see oracle/ref_shim/README.md).
// kiss-icp v1.2.0 (the version /root/reference/cpp/kinematic_icp/kiss_icp/kiss-icp.cmake:28-31 fetches) is not in this
// image; this header restates its published interface and algorithm so that the reference's
// own sources compile against it.  RECALLED, not verifiable offline.
#pragma once
#include <Eigen/Core>
#include <cmath>
#include <cstdint>
#include <functional>
#include <vector>

namespace kiss_icp {
using Voxel = Eigen::Vector3i;
inline Voxel PointToVoxel(const Eigen::Vector3d &point, const double voxel_size) {
    return Voxel(static_cast<int>(std::floor(point.x() / voxel_size)), static_cast<int>(std::floor(point.y() / voxel_size)),
                 static_cast<int>(std::floor(point.z() / voxel_size)));
}
// first point of every voxel, in the grid's (tsl::robin_map) iteration order
std::vector<Eigen::Vector3d> VoxelDownsample(const std::vector<Eigen::Vector3d> &frame, const double voxel_size);
}  // namespace kiss_icp

template <>
struct std::hash<kiss_icp::Voxel> {
    std::size_t operator()(const kiss_icp::Voxel &voxel) const {
        const uint32_t *vec = reinterpret_cast<const uint32_t *>(voxel.data());
        return (vec[0] * 73856093 ^ vec[1] * 19349669 ^ vec[2] * 83492791);  // uint32 wrap-around products, then XOR
    }
};
