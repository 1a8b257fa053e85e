// kiss_icp/core/VoxelHashMap.hpp -- drop-in for kiss-icp v1.2.0's header of the same path, backed by the
// MI355X library (host map + HBM mirror, either side may hold the newest state).  Same struct name, constructor, methods and public
// configuration fields (reference call sites: registration/Registration.cpp:63,74,157,
// pipeline/KinematicICP.hpp:79,88,92,94-95, pipeline/KinematicICP.cpp:79).
// Copyable and movable like the reference's struct (a copy is a deep copy of the newest state, wherever it lives).
// Not reproduced: the public `map_` member (a tsl::robin_map; an implementation detail no caller in the reference touches -
// the container lives behind the C-ABI as a flat table + bucket pools).
#pragma once
#include <Eigen/Core>
#include <limits>
#include <sophus/se3.hpp>
#include <tuple>
#include <utility>
#include <vector>

#include "kicp_bridge.hpp"

namespace kiss_icp {
struct VoxelHashMap {
    explicit VoxelHashMap(double voxel_size, double max_distance, unsigned int max_points_per_voxel)
        : voxel_size_(voxel_size), max_distance_(max_distance), max_points_per_voxel_(max_points_per_voxel) {
        kicp_bridge::check(kicp_map_create(voxel_size, max_distance, max_points_per_voxel, &handle_), "VoxelHashMap");
        kicp_map_set_device(handle_, device_);  // bulk AddPoints / Update calls from host vectors insert on the GPU
    }
    ~VoxelHashMap() { kicp_map_destroy(handle_); }
    VoxelHashMap(const VoxelHashMap &o)
        : voxel_size_(o.voxel_size_), max_distance_(o.max_distance_), max_points_per_voxel_(o.max_points_per_voxel_), device_(o.device_) {
        kicp_bridge::check(kicp_map_clone(o.handle_, &handle_), "VoxelHashMap(const VoxelHashMap&)");
    }
    VoxelHashMap(VoxelHashMap &&o) noexcept
        : voxel_size_(o.voxel_size_), max_distance_(o.max_distance_), max_points_per_voxel_(o.max_points_per_voxel_), device_(o.device_),
          handle_(o.handle_) {
        o.handle_ = nullptr;
    }
    VoxelHashMap &operator=(VoxelHashMap o) noexcept {  // copy / move and swap
        std::swap(voxel_size_, o.voxel_size_), std::swap(max_distance_, o.max_distance_), std::swap(max_points_per_voxel_, o.max_points_per_voxel_);
        std::swap(device_, o.device_), std::swap(handle_, o.handle_);
        return *this;
    }

    inline void Clear() { kicp_map_clear(handle_); }
    inline bool Empty() const { return kicp_map_empty(handle_) != 0; }
    void Update(const std::vector<Eigen::Vector3d> &points, const Eigen::Vector3d &origin) {
        kicp_bridge::check(kicp_map_update_origin(handle_, kicp_bridge::xyz(points), points.size(), origin.data()), "VoxelHashMap::Update");
    }
    void Update(const std::vector<Eigen::Vector3d> &points, const Sophus::SE3d &pose) {
        double p[7];
        kicp_bridge::to_params(pose, p);
        kicp_bridge::check(kicp_map_update_pose(handle_, kicp_bridge::xyz(points), points.size(), p), "VoxelHashMap::Update");
    }
    // Update(points, pose) with the points already in HBM on device_ (e.g. a kicp_pre buffer); backend extension
    void UpdateDevice(const double *d_points_xyz, size_t n, const Sophus::SE3d &pose) {
        double p[7];
        kicp_bridge::to_params(pose, p);
        kicp_bridge::check(kicp_map_update_pose_device(handle_, device_, d_points_xyz, n, p), "VoxelHashMap::Update");
    }
    void AddPoints(const std::vector<Eigen::Vector3d> &points) {
        kicp_bridge::check(kicp_map_add_points(handle_, kicp_bridge::xyz(points), points.size()), "VoxelHashMap::AddPoints");
    }
    void RemovePointsFarFromLocation(const Eigen::Vector3d &origin) { kicp_map_remove_far(handle_, origin.data()); }
    std::vector<Eigen::Vector3d> Pointcloud() const {
        std::vector<Eigen::Vector3d> points(kicp_map_num_points(handle_));
        if (!points.empty()) kicp_map_pointcloud(handle_, points.front().data(), points.size());
        return points;
    }
    // One query -> (closest point, distance); (0, DBL_MAX) when the 27 voxels hold nothing.  Runs the device search.
    std::tuple<Eigen::Vector3d, double> GetClosestNeighbor(const Eigen::Vector3d &query) const {
        Eigen::Vector3d nn;
        double d = std::numeric_limits<double>::max();
        kicp_bridge::check(kicp_map_closest(handle_, device_, query.data(), 1, nn.data(), &d), "VoxelHashMap::GetClosestNeighbor");
        return std::make_tuple(nn, d);
    }

    double voxel_size_;
    double max_distance_;
    unsigned int max_points_per_voxel_;

    // backend access (not part of the reference API)
    kicp_map *handle() const { return handle_; }
    int device_ = kicp_bridge::default_device();  // KICP_DEVICE

private:
    kicp_map *handle_ = nullptr;
};
}  // namespace kiss_icp
