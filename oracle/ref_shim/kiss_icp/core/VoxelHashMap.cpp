// kiss_icp/core/VoxelHashMap.cpp -- STAND-IN for kiss-icp v1.2.0 (test infrastructure, see oracle/ref_shim/README.md).
// Restates the published algorithm; written independently of oracle/kicp_oracle.cpp's
// VoxelMap so that the two can be checked against each other.
#include "VoxelHashMap.hpp"

#include <algorithm>
#include <array>
#include <cmath>
#include <limits>

namespace {
using kiss_icp::Voxel;
// visiting order of the 27 neighbour voxels: own voxel, the 6 face, the 12 edge, the 8 corner neighbours
const std::array<Voxel, 27> shifts{
    Voxel{0, 0, 0},   Voxel{1, 0, 0},   Voxel{-1, 0, 0},  Voxel{0, 1, 0},   Voxel{0, -1, 0},  Voxel{0, 0, 1},   Voxel{0, 0, -1},
    Voxel{1, 1, 0},   Voxel{1, -1, 0},  Voxel{-1, 1, 0},  Voxel{-1, -1, 0}, Voxel{1, 0, 1},   Voxel{1, 0, -1},  Voxel{-1, 0, 1},
    Voxel{-1, 0, -1}, Voxel{0, 1, 1},   Voxel{0, 1, -1},  Voxel{0, -1, 1},  Voxel{0, -1, -1}, Voxel{1, 1, 1},   Voxel{1, 1, -1},
    Voxel{1, -1, 1},  Voxel{1, -1, -1}, Voxel{-1, 1, 1},  Voxel{-1, 1, -1}, Voxel{-1, -1, 1}, Voxel{-1, -1, -1}};
}  // namespace

namespace kiss_icp {

std::tuple<Eigen::Vector3d, double> VoxelHashMap::GetClosestNeighbor(const Eigen::Vector3d &query) const {
    const Voxel voxel = PointToVoxel(query, voxel_size_);
    Eigen::Vector3d closest_neighbor = Eigen::Vector3d::Zero();
    double closest_distance = std::numeric_limits<double>::max();
    for (const Voxel &shift : shifts) {
        const auto search = map_.find(voxel + shift);
        if (search == map_.end()) continue;
        const std::vector<Eigen::Vector3d> &points = search.value();
        const Eigen::Vector3d &neighbor = *std::min_element(
            points.cbegin(), points.cend(),
            [&](const Eigen::Vector3d &lhs, const Eigen::Vector3d &rhs) { return (lhs - query).norm() < (rhs - query).norm(); });
        const double distance = (neighbor - query).norm();
        if (distance < closest_distance) {
            closest_neighbor = neighbor;
            closest_distance = distance;
        }
    }
    return std::make_tuple(closest_neighbor, closest_distance);
}

std::vector<Eigen::Vector3d> VoxelHashMap::Pointcloud() const {
    std::vector<Eigen::Vector3d> points;
    points.reserve(map_.size() * static_cast<size_t>(max_points_per_voxel_));
    for (auto it = map_.cbegin(); it != map_.cend(); ++it) points.insert(points.end(), it->second.cbegin(), it->second.cend());
    points.shrink_to_fit();
    return points;
}

void VoxelHashMap::Update(const std::vector<Eigen::Vector3d> &points, const Eigen::Vector3d &origin) {
    AddPoints(points);
    RemovePointsFarFromLocation(origin);
}

void VoxelHashMap::Update(const std::vector<Eigen::Vector3d> &points, const Sophus::SE3d &pose) {
    std::vector<Eigen::Vector3d> points_transformed(points.size());
    std::transform(points.cbegin(), points.cend(), points_transformed.begin(), [&](const Eigen::Vector3d &point) { return pose * point; });
    const Eigen::Vector3d &origin = pose.translation();
    Update(points_transformed, origin);
}

void VoxelHashMap::AddPoints(const std::vector<Eigen::Vector3d> &points) {
    const double map_resolution = std::sqrt(voxel_size_ * voxel_size_ / max_points_per_voxel_);
    for (const Eigen::Vector3d &point : points) {
        const Voxel voxel = PointToVoxel(point, voxel_size_);
        auto search = map_.find(voxel);
        if (search != map_.end()) {
            std::vector<Eigen::Vector3d> &voxel_points = search.value();
            const bool full = voxel_points.size() == max_points_per_voxel_;
            if (full || std::any_of(voxel_points.cbegin(), voxel_points.cend(), [&](const Eigen::Vector3d &voxel_point) {
                    return (voxel_point - point).norm() < map_resolution;
                }))
                continue;
            voxel_points.emplace_back(point);
        } else {
            std::vector<Eigen::Vector3d> voxel_points;
            voxel_points.reserve(max_points_per_voxel_);
            voxel_points.emplace_back(point);
            map_.insert({voxel, std::move(voxel_points)});
        }
    }
}

void VoxelHashMap::RemovePointsFarFromLocation(const Eigen::Vector3d &origin) {
    const double max_distance2 = max_distance_ * max_distance_;
    for (auto it = map_.begin(); it != map_.end();) {
        const Eigen::Vector3d &pt = it->second.front();
        if ((pt - origin).squaredNorm() >= max_distance2) {
            it = map_.erase(it);
        } else {
            ++it;
        }
    }
}

}  // namespace kiss_icp
