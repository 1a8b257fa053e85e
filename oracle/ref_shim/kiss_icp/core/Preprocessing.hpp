// kiss_icp/core/Preprocessing.hpp -- STAND-IN for kiss-icp v1.2.0 (test infrastructure, see oracle/ref_shim/README.md).
// ctor (max_range, min_range, deskew, max_num_threads): pipeline/KinematicICP.hpp:78; Preprocess(frame, timestamps,
// relative_motion): pipeline/KinematicICP.cpp:56-57.  RECALLED, not verifiable offline.
#pragma once
#include <Eigen/Core>
#include <sophus/se3.hpp>
#include <vector>

namespace kiss_icp {
struct Preprocessor {
    Preprocessor(const double max_range, const double min_range, const bool deskew, const int max_num_threads);
    std::vector<Eigen::Vector3d> Preprocess(const std::vector<Eigen::Vector3d> &frame, const std::vector<double> &timestamps,
                                            const Sophus::SE3d &relative_motion) const;
    double max_range_;
    double min_range_;
    bool deskew_;
    int max_num_threads_;
};
}  // namespace kiss_icp
