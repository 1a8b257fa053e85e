// kinematic_icp/pipeline/KinematicICP.hpp -- drop-in for the reference header of the same path
// (/root/reference/cpp/kinematic_icp/pipeline/KinematicICP.{hpp,cpp}): same Config fields and defaults, same class
// surface (RegisterFrame, SetPose, LocalMap, VoxelMap, pose), so ros/src/.../LidarOdometryServer.cpp compiles
// against it unchanged.  The ICP (registration_ + local_map_) runs on the MI355X; deskew/crop/voxelize and the
// threshold bookkeeping are host pre/post steps exactly as in the reference.
#pragma once
#include <Eigen/Core>
#include <cmath>
#include <kiss_icp/core/Preprocessing.hpp>
#include <kiss_icp/core/VoxelHashMap.hpp>
#include <kiss_icp/core/VoxelUtils.hpp>
#include <sophus/se3.hpp>
#include <tuple>
#include <vector>

#include "kinematic_icp/correspondence_threshold/CorrespondenceThreshold.hpp"
#include "kinematic_icp/registration/Registration.hpp"

namespace kinematic_icp::pipeline {

struct Config {
    // Preprocessing
    double max_range = 100.0;
    double min_range = 0.0;
    // Mapping parameters
    double voxel_size = 1.0;
    unsigned int max_points_per_voxel = 20;
    // Derived parameter
    double map_resolution() const { return voxel_size / std::sqrt(max_points_per_voxel); }
    // Correspondence threshold parameters
    bool use_adaptive_threshold = true;
    double fixed_threshold = 1.0;
    // Registration Parameters
    int max_num_iterations = 10;
    double convergence_criterion = 0.001;
    int max_num_threads = 1;
    bool use_adaptive_odometry_regularization = true;
    double fixed_regularization = 0.0;
    // Motion compensation
    bool deskew = false;
};

class KinematicICP {
public:
    using Vector3dVector = std::vector<Eigen::Vector3d>;
    using Vector3dVectorTuple = std::tuple<Vector3dVector, Vector3dVector>;

    explicit KinematicICP(const Config &config)
        : registration_(config.max_num_iterations, config.convergence_criterion, config.max_num_threads,
                        config.use_adaptive_odometry_regularization, config.fixed_regularization),
          correspondence_threshold_(config.map_resolution(), config.max_range, config.use_adaptive_threshold, config.fixed_threshold),
          config_(config),
          preprocessor_(config.max_range, config.min_range, config.deskew, config.max_num_threads),
          local_map_(config.voxel_size, config.max_range, config.max_points_per_voxel) {}

    // pipeline/KinematicICP.cpp:48-85
    Vector3dVectorTuple RegisterFrame(const std::vector<Eigen::Vector3d> &frame, const std::vector<double> &timestamps,
                                      const Sophus::SE3d &lidar_to_base, const Sophus::SE3d &relative_odometry) {
        const Sophus::SE3d relative_odometry_in_lidar = lidar_to_base.inverse() * relative_odometry * lidar_to_base;
        const auto preprocessed_frame = preprocessor_.Preprocess(frame, timestamps, relative_odometry_in_lidar);
        Vector3dVector preprocessed_frame_in_base(preprocessed_frame.size());
        for (size_t i = 0; i < preprocessed_frame.size(); ++i) preprocessed_frame_in_base[i] = lidar_to_base * preprocessed_frame[i];
        const auto frame_downsample = kiss_icp::VoxelDownsample(preprocessed_frame_in_base, config_.voxel_size * 0.5);
        const auto source = kiss_icp::VoxelDownsample(frame_downsample, config_.voxel_size * 1.5);
        const double tau = correspondence_threshold_.ComputeThreshold();
        const auto new_pose = registration_.ComputeRobotMotion(source, local_map_, last_pose_, relative_odometry, tau);
        const auto odometry_error = (last_pose_ * relative_odometry).inverse() * new_pose;
        correspondence_threshold_.UpdateOdometryError(odometry_error);
        local_map_.Update(frame_downsample, new_pose);
        last_pose_ = new_pose;
        return {preprocessed_frame_in_base, source};
    }

    inline void SetPose(const Sophus::SE3d &pose) {
        last_pose_ = pose;
        local_map_.Clear();
        correspondence_threshold_.Reset();
    }

    std::vector<Eigen::Vector3d> LocalMap() const { return local_map_.Pointcloud(); }
    const kiss_icp::VoxelHashMap &VoxelMap() const { return local_map_; }
    kiss_icp::VoxelHashMap &VoxelMap() { return local_map_; }
    const Sophus::SE3d &pose() const { return last_pose_; }
    Sophus::SE3d &pose() { return last_pose_; }

protected:
    Sophus::SE3d last_pose_;
    KinematicRegistration registration_;
    CorrespondenceThreshold correspondence_threshold_;
    Config config_;
    kiss_icp::Preprocessor preprocessor_;
    kiss_icp::VoxelHashMap local_map_;
};

}  // namespace kinematic_icp::pipeline
