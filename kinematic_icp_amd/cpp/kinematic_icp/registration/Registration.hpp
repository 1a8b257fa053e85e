// kinematic_icp/registration/Registration.hpp -- drop-in for the reference header of the same path
// (/root/reference/cpp/kinematic_icp/registration/Registration.hpp:23-51): same struct, constructor signature,
// ComputeRobotMotion signature and public fields; the body runs on the MI355X through include/kicp.h.
#pragma once
#include <Eigen/Core>
#include <kiss_icp/core/VoxelHashMap.hpp>
#include <sophus/se3.hpp>
#include <utility>
#include <vector>

#include "kicp_bridge.hpp"

namespace kinematic_icp {
struct KinematicRegistration {
    explicit KinematicRegistration(const int max_num_iteration, const double convergence_criterion, const int max_num_threads,
                                   const bool use_adaptive_odometry_regularization, const double fixed_regularization)
        : max_num_iterations_(max_num_iteration),
          convergence_criterion_(convergence_criterion),
          // Registration.cpp:141-142: "only manipulate the number of threads if the user specifies something greater than 0", else
          // tbb::this_task_arena::max_concurrency() = the hardware threads this process may use.  The field reads like the
          // reference's; the GPU path itself has no host thread pool to size with it.
          max_num_threads_(max_num_threads > 0 ? max_num_threads : kicp_bridge::hardware_threads()),
          use_adaptive_odometry_regularization_(use_adaptive_odometry_regularization),
          fixed_regularization_(fixed_regularization) {
        const kicp_reg_config c = config();
        kicp_bridge::check(kicp_reg_create(&c, device_, &handle_), "KinematicRegistration");
    }
    ~KinematicRegistration() { kicp_reg_destroy(handle_); }
    // copyable and movable like the reference's struct (a plain aggregate of five parameters, Registration.hpp:32-50): a copy has
    // the same parameters and backend options and device workspaces of its own (kicp_reg_clone)
    KinematicRegistration(const KinematicRegistration &o)
        : max_num_iterations_(o.max_num_iterations_),
          convergence_criterion_(o.convergence_criterion_),
          max_num_threads_(o.max_num_threads_),
          use_adaptive_odometry_regularization_(o.use_adaptive_odometry_regularization_),
          fixed_regularization_(o.fixed_regularization_),
          device_(o.device_),
          last_stats_(o.last_stats_) {
        kicp_bridge::check(kicp_reg_clone(o.handle_, &handle_), "KinematicRegistration(const KinematicRegistration&)");
    }
    KinematicRegistration(KinematicRegistration &&o) noexcept
        : max_num_iterations_(o.max_num_iterations_),
          convergence_criterion_(o.convergence_criterion_),
          max_num_threads_(o.max_num_threads_),
          use_adaptive_odometry_regularization_(o.use_adaptive_odometry_regularization_),
          fixed_regularization_(o.fixed_regularization_),
          device_(o.device_),
          handle_(o.handle_),
          last_stats_(o.last_stats_) {
        o.handle_ = nullptr;
    }
    KinematicRegistration &operator=(KinematicRegistration o) noexcept {  // copy / move and swap
        std::swap(max_num_iterations_, o.max_num_iterations_), std::swap(convergence_criterion_, o.convergence_criterion_);
        std::swap(max_num_threads_, o.max_num_threads_), std::swap(use_adaptive_odometry_regularization_, o.use_adaptive_odometry_regularization_);
        std::swap(fixed_regularization_, o.fixed_regularization_), std::swap(device_, o.device_), std::swap(handle_, o.handle_);
        std::swap(last_stats_, o.last_stats_);
        return *this;
    }

    Sophus::SE3d ComputeRobotMotion(const std::vector<Eigen::Vector3d> &frame, const kiss_icp::VoxelHashMap &voxel_map,
                                    const Sophus::SE3d &last_robot_pose, const Sophus::SE3d &relative_wheel_odometry,
                                    const double max_correspondence_distance) {
        const kicp_reg_config c = config();  // the fields are public and mutable in the reference: honour edits
        kicp_bridge::check(kicp_reg_set_config(handle_, &c), "KinematicRegistration");
        double last[7], rel[7], out[7];
        kicp_bridge::to_params(last_robot_pose, last);
        kicp_bridge::to_params(relative_wheel_odometry, rel);
        kicp_bridge::check(kicp_register(handle_, voxel_map.handle(), kicp_bridge::xyz(frame), frame.size(), last, rel,
                                         max_correspondence_distance, out, &last_stats_),
                           "KinematicRegistration::ComputeRobotMotion");
        return kicp_bridge::from_params(out);
    }

    // backend extension: the frame as float32 xyz, the way a PointCloud2 carries it (RosUtils.cpp:30-39 widens it on the host);
    // half the bytes cross PCIe, the (exact) widening happens on the device
    Sophus::SE3d ComputeRobotMotion(const float *frame_xyz_f32, size_t n, const kiss_icp::VoxelHashMap &voxel_map, const Sophus::SE3d &last_robot_pose,
                                    const Sophus::SE3d &relative_wheel_odometry, const double max_correspondence_distance) {
        const kicp_reg_config c = config();
        kicp_bridge::check(kicp_reg_set_config(handle_, &c), "KinematicRegistration");
        double last[7], rel[7], out[7];
        kicp_bridge::to_params(last_robot_pose, last);
        kicp_bridge::to_params(relative_wheel_odometry, rel);
        kicp_bridge::check(kicp_register_f32(handle_, voxel_map.handle(), frame_xyz_f32, n, last, rel, max_correspondence_distance, out, &last_stats_),
                           "KinematicRegistration::ComputeRobotMotion(float)");
        return kicp_bridge::from_params(out);
    }

    // backend extension: the frame is already in HBM (output of the on-device pre-steps)
    Sophus::SE3d ComputeRobotMotionDevice(const double *d_frame_xyz, size_t n, const kiss_icp::VoxelHashMap &voxel_map,
                                          const Sophus::SE3d &last_robot_pose, const Sophus::SE3d &relative_wheel_odometry,
                                          const double max_correspondence_distance) {
        const kicp_reg_config c = config();
        kicp_bridge::check(kicp_reg_set_config(handle_, &c), "KinematicRegistration");
        double last[7], rel[7], out[7];
        kicp_bridge::to_params(last_robot_pose, last);
        kicp_bridge::to_params(relative_wheel_odometry, rel);
        kicp_bridge::check(kicp_register_device(handle_, voxel_map.handle(), d_frame_xyz, n, last, rel, max_correspondence_distance, out,
                                                &last_stats_),
                           "KinematicRegistration::ComputeRobotMotionDevice");
        return kicp_bridge::from_params(out);
    }

    int max_num_iterations_;
    double convergence_criterion_;
    int max_num_threads_;
    bool use_adaptive_odometry_regularization_;
    double fixed_regularization_;

    // backend access (not part of the reference API)
    kicp_reg *handle() const { return handle_; }
    const kicp_stats &last_stats() const { return last_stats_; }

private:
    kicp_reg_config config() const {
        return kicp_reg_config{max_num_iterations_, convergence_criterion_, max_num_threads_, use_adaptive_odometry_regularization_ ? 1 : 0,
                               fixed_regularization_};
    }
    int device_ = kicp_bridge::default_device();  // KICP_DEVICE
    kicp_reg *handle_ = nullptr;
    kicp_stats last_stats_{};
};
}  // namespace kinematic_icp
