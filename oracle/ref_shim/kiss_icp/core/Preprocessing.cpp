// kiss_icp/core/Preprocessing.cpp -- STAND-IN for kiss-icp v1.2.0 (test infrastructure, see oracle/ref_shim/README.md).
#include "Preprocessing.hpp"

#include <tbb/blocked_range.h>
#include <tbb/parallel_for.h>

namespace kiss_icp {
Preprocessor::Preprocessor(const double max_range, const double min_range, const bool deskew, const int max_num_threads)
    : max_range_(max_range), min_range_(min_range), deskew_(deskew), max_num_threads_(max_num_threads > 0 ? max_num_threads : 1) {}

std::vector<Eigen::Vector3d> Preprocessor::Preprocess(const std::vector<Eigen::Vector3d> &frame, const std::vector<double> &timestamps,
                                                      const Sophus::SE3d &relative_motion) const {
    // constant-velocity deskew to the END of the scan: p' = (relative_motion^-1 * exp(t * log(relative_motion))) * p, t in [0, 1]
    std::vector<Eigen::Vector3d> deskewed_frame = frame;
    if (deskew_ && !timestamps.empty()) {
        const Sophus::SE3d::Tangent omega = relative_motion.log();
        const Sophus::SE3d inverse_motion = relative_motion.inverse();
        tbb::parallel_for(tbb::blocked_range<size_t>(0, frame.size()), [&](const tbb::blocked_range<size_t> &r) {
            for (size_t idx = r.begin(); idx < r.end(); ++idx) {
                const Sophus::SE3d pose = inverse_motion * Sophus::SE3d::exp(timestamps.at(idx) * omega);
                deskewed_frame.at(idx) = pose * frame.at(idx);
            }
        });
    }
    // range crop, strict on both sides, order preserved
    std::vector<Eigen::Vector3d> preprocessed_frame;
    preprocessed_frame.reserve(deskewed_frame.size());
    for (const Eigen::Vector3d &point : deskewed_frame) {
        const double point_range = point.norm();
        if (point_range < max_range_ && point_range > min_range_) preprocessed_frame.emplace_back(point);
    }
    preprocessed_frame.shrink_to_fit();
    return preprocessed_frame;
}
}  // namespace kiss_icp
