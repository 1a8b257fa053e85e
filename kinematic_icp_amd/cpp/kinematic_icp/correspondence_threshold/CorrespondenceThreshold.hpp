// kinematic_icp/correspondence_threshold/CorrespondenceThreshold.hpp -- host restatement of the reference's
// adaptive threshold (correspondence_threshold/CorrespondenceThreshold.{hpp,cpp}:29-64): O(1) scalar state per frame
// that produces the tau fed to the GPU path.  Same struct, members and method names.
#pragma once
#include <cmath>
#include <sophus/se3.hpp>

#include "kicp_bridge.hpp"

namespace kinematic_icp {
struct CorrespondenceThreshold {
    explicit CorrespondenceThreshold(const double map_discretization_error, const double max_range, const bool use_adaptive_threshold,
                                     const double fixed_threshold)
        : map_discretization_error_(map_discretization_error),
          max_range_(max_range),
          use_adaptive_threshold_(use_adaptive_threshold),
          fixed_threshold_(fixed_threshold),
          odom_sse_(0.0),
          num_samples_(1e-8) {}

    void UpdateOdometryError(const Sophus::SE3d &odometry_error) {
        if (!use_adaptive_threshold_) return;
        double p[7];
        kicp_bridge::to_params(odometry_error, p);
        // theta of SO3::logAndTheta() from the unit quaternion (Sophus so3.hpp)
        const double sn = p[0] * p[0] + p[1] * p[1] + p[2] * p[2], w = p[3];
        double theta;
        if (sn < 1e-20) {
            theta = 2.0 * sn / w;
        } else {
            const double n = std::sqrt(sn);
            theta = 2.0 * ((w < 0.0) ? std::atan2(-n, -w) : std::atan2(n, w));
        }
        const double delta_rot = 2.0 * max_range_ * std::sin(theta / 2.0);
        const double delta_trans = std::sqrt(p[4] * p[4] + p[5] * p[5] + p[6] * p[6]);
        const double e = delta_trans + delta_rot;
        odom_sse_ += e * e;
        num_samples_ += 1.0;
    }
    double ComputeThreshold() const {
        if (!use_adaptive_threshold_) return fixed_threshold_;
        const double sigma_odom = std::sqrt(odom_sse_ / num_samples_);
        return 3.0 * (map_discretization_error_ + sigma_odom);
    }
    inline void Reset() {
        odom_sse_ = 0.0;
        num_samples_ = 1e-8;
    }

    double map_discretization_error_;
    double max_range_;
    bool use_adaptive_threshold_;
    double fixed_threshold_;
    double odom_sse_;
    double num_samples_;
};
}  // namespace kinematic_icp
