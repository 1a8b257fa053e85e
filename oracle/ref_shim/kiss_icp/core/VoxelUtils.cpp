// kiss_icp/core/VoxelUtils.cpp -- STAND-IN for kiss-icp v1.2.0 (test infrastructure, see oracle/ref_shim/README.md).
#include "VoxelUtils.hpp"

#include <tsl/robin_map.h>

namespace kiss_icp {
std::vector<Eigen::Vector3d> VoxelDownsample(const std::vector<Eigen::Vector3d> &frame, const double voxel_size) {
    tsl::robin_map<Voxel, Eigen::Vector3d> grid;
    grid.reserve(frame.size());
    for (const auto &point : frame) {
        const Voxel voxel = PointToVoxel(point, voxel_size);
        if (!grid.contains(voxel)) grid.insert({voxel, point});  // the first point of a voxel stays
    }
    std::vector<Eigen::Vector3d> frame_downsampled;
    frame_downsampled.reserve(grid.size());
    for (auto it = grid.cbegin(); it != grid.cend(); ++it) frame_downsampled.emplace_back(it->second);
    return frame_downsampled;
}
}  // namespace kiss_icp
