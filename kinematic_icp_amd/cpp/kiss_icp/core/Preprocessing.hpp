// kiss_icp/core/Preprocessing.hpp -- host pre-step of the pipeline (kiss-icp v1.2.0 core/Preprocessing.{hpp,cpp};
// SURVEY.md App. A.8; call sites pipeline/KinematicICP.hpp:78, KinematicICP.cpp:56-57): constant-velocity deskew to
// the scan end, then min/max range crop (strict on both sides), order preserved.
#pragma once
#include <Eigen/Core>
#include <sophus/se3.hpp>
#include <vector>

namespace kiss_icp {
struct Preprocessor {
    Preprocessor(double max_range, double min_range, bool deskew, int max_num_threads)
        : max_range_(max_range), min_range_(min_range), deskew_(deskew), max_num_threads_(max_num_threads) {}

    std::vector<Eigen::Vector3d> Preprocess(const std::vector<Eigen::Vector3d> &frame, const std::vector<double> &timestamps,
                                            const Sophus::SE3d &relative_motion) const {
        std::vector<Eigen::Vector3d> out;
        out.reserve(frame.size());
        const bool deskew = deskew_ && !timestamps.empty();
        Sophus::SE3d::Tangent omega{};
        Sophus::SE3d motion_inverse;
        if (deskew) omega = relative_motion.log(), motion_inverse = relative_motion.inverse();
        for (size_t i = 0; i < frame.size(); ++i) {
            Eigen::Vector3d p = frame[i];
            if (deskew) {
                Sophus::SE3d::Tangent xi = omega;
                for (int k = 0; k < 6; ++k) xi[k] = xi[k] * timestamps[i];
                p = (motion_inverse * Sophus::SE3d::exp(xi)) * frame[i];
            }
            const double r = p.norm();
            if (r < max_range_ && r > min_range_) out.emplace_back(p);
        }
        return out;
    }
    double max_range_, min_range_;
    bool deskew_;
    int max_num_threads_;
};
}  // namespace kiss_icp
