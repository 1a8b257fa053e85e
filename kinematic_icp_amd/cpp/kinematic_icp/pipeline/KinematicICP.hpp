// kinematic_icp/pipeline/KinematicICP.hpp -- drop-in for the reference header of the same path
// (/root/reference/cpp/kinematic_icp/pipeline/KinematicICP.{hpp,cpp}): same Config fields and defaults, same class
// surface (RegisterFrame, SetPose, LocalMap, VoxelMap, pose), so ros/src/.../LidarOdometryServer.cpp compiles
// against it unchanged.  The ICP (registration_ + local_map_) and - unless KICP_HOST_PRESTEPS is defined - the
// pre-steps (deskew + crop + transform, two-level voxel downsample: kicp_pre_*) run on the MI355X; the registration
// source and the frame that goes into the map never leave HBM, and the map update itself (kiss-icp Update: ordered
// insertion + far-voxel removal) runs there too.  The threshold bookkeeping stays on the host
// (SURVEY.md section 8f rows 1, 2 and 4).
#pragma once
#include <Eigen/Core>
#include <cmath>
#include <cstdlib>
#include <kiss_icp/core/Preprocessing.hpp>
#include <kiss_icp/core/VoxelHashMap.hpp>
#include <kiss_icp/core/VoxelUtils.hpp>
#include <sophus/se3.hpp>
#include <tuple>
#include <utility>
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
          local_map_(config.voxel_size, config.max_range, config.max_points_per_voxel) {
#ifndef KICP_HOST_PRESTEPS
        kicp_bridge::check(kicp_pre_create(kicp_bridge::default_device(), &pre_), "KinematicICP");
#endif
    }
    ~KinematicICP() { kicp_pre_destroy(pre_); }
    // copyable and movable like the reference's class (every member is held by value there, KinematicICP.hpp:100-108): a copy owns
    // a deep copy of the map, a registration handle of its own and its own pre-step workspace
    KinematicICP(const KinematicICP &o)
        : last_pose_(o.last_pose_),
          registration_(o.registration_),
          correspondence_threshold_(o.correspondence_threshold_),
          config_(o.config_),
          preprocessor_(o.preprocessor_),
          local_map_(o.local_map_) {
#ifndef KICP_HOST_PRESTEPS
        kicp_bridge::check(kicp_pre_create(kicp_bridge::default_device(), &pre_), "KinematicICP");
#endif
    }
    KinematicICP(KinematicICP &&o) noexcept
        : last_pose_(o.last_pose_),
          registration_(std::move(o.registration_)),
          correspondence_threshold_(o.correspondence_threshold_),
          config_(o.config_),
          preprocessor_(o.preprocessor_),
          local_map_(std::move(o.local_map_)),
          pre_(o.pre_) {
        o.pre_ = nullptr;
    }
    KinematicICP &operator=(KinematicICP o) noexcept {  // copy / move and swap
        std::swap(last_pose_, o.last_pose_);
        registration_ = std::move(o.registration_);
        std::swap(correspondence_threshold_, o.correspondence_threshold_), std::swap(config_, o.config_), std::swap(preprocessor_, o.preprocessor_);
        local_map_ = std::move(o.local_map_);
        std::swap(pre_, o.pre_);
        return *this;
    }

    // pipeline/KinematicICP.cpp:48-85
    Vector3dVectorTuple RegisterFrame(const std::vector<Eigen::Vector3d> &frame, const std::vector<double> &timestamps,
                                      const Sophus::SE3d &lidar_to_base, const Sophus::SE3d &relative_odometry) {
        const Sophus::SE3d relative_odometry_in_lidar = lidar_to_base.inverse() * relative_odometry * lidar_to_base;
#ifndef KICP_HOST_PRESTEPS
        double rel_lidar[7], ext[7];
        kicp_bridge::to_params(relative_odometry_in_lidar, rel_lidar);
        kicp_bridge::to_params(lidar_to_base, ext);
        // Preprocess + transform_points + the two VoxelDownsamples as ONE backend call behind one host synchronisation; the
        // preprocessed frame (a return value nothing on the device waits for) travels back while the downsamples run
        Vector3dVectorTuple result{Vector3dVector(frame.size()), Vector3dVector()};
        auto &out_frame = std::get<0>(result);
        DownloadGuard guard{pre_};
        size_t counts[3] = {0, 0, 0};
        const int order = kicp_bridge::check(
            kicp_pre_frame(pre_, kicp_bridge::xyz(frame), frame.size(), timestamps.data(), timestamps.size(), rel_lidar, ext, config_.max_range, config_.min_range,
                           config_.deskew ? 1 : 0, config_.voxel_size * 0.5, config_.voxel_size * 1.5, out_frame.empty() ? nullptr : out_frame.front().data(),
                           out_frame.size(), counts),
            "Preprocess + VoxelDownsample");
        if (order == KICP_WARN_TABLE_ORDER) kicp_bridge::warn_once(kicp_last_error());
        return RegisterChained(result, guard, counts, relative_odometry);
#else
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
#endif
    }

#ifndef KICP_HOST_PRESTEPS
    // ---- backend extension: feed the PointCloud2 bytes directly ----
    // IngestCloud replaces PointCloud2ToEigen(msg, {}) (RosUtils.cpp:30-39) and the per-point part of
    // TimeStampHandler::ProcessTimestamps (TimeStampHandler.cpp:57-106,121-128): it returns {cloud has stamps, min stamp,
    // max stamp} (seconds), which is all ProcessTimestamps' begin/end-stamp logic (:107-119) needs; the decoded points
    // and normalised stamps stay in HBM.  RegisterIngestedFrame is RegisterFrame on that cloud.
    std::tuple<bool, double, double> IngestCloud(const void *data, size_t n_points, const kicp_cloud_layout &layout) {
        double lo = 0.0, hi = 0.0;
        kicp_bridge::check(kicp_pre_ingest(pre_, data, n_points, &layout, nullptr, &lo, &hi), "IngestCloud");
        return {layout.stamp_datatype != 0 && n_points != 0, lo, hi};
    }
    // Look-ahead for callers that already hold the NEXT message (a bag replay; ros/src/kinematic_icp_ros/nodes/offline_node.cpp reads
    // its messages in a loop): announce it before RegisterIngestedFrame of the current one - it is then uploaded and decoded while the
    // current frame's pre-steps run, and its IngestCloud call returns at once.  The bytes must stay valid until that IngestCloud call.
    void AnnounceNextCloud(const void *data, size_t n_points, const kicp_cloud_layout &layout) {
        kicp_bridge::check(kicp_pre_ingest_ahead(pre_, data, n_points, &layout, nullptr), "AnnounceNextCloud");
    }
    Vector3dVectorTuple RegisterIngestedFrame(const Sophus::SE3d &lidar_to_base, const Sophus::SE3d &relative_odometry) {
        const Sophus::SE3d relative_odometry_in_lidar = lidar_to_base.inverse() * relative_odometry * lidar_to_base;
        double rel_lidar[7], ext[7];
        kicp_bridge::to_params(relative_odometry_in_lidar, rel_lidar);
        kicp_bridge::to_params(lidar_to_base, ext);
        Vector3dVectorTuple result{Vector3dVector(kicp_pre_ingested_count(pre_)), Vector3dVector()};
        auto &out_frame = std::get<0>(result);
        DownloadGuard guard{pre_};
        size_t counts[3] = {0, 0, 0};
        const int order = kicp_bridge::check(
            kicp_pre_frame_ingested(pre_, rel_lidar, ext, config_.max_range, config_.min_range, config_.deskew ? 1 : 0, config_.voxel_size * 0.5,
                                    config_.voxel_size * 1.5, out_frame.empty() ? nullptr : out_frame.front().data(), out_frame.size(), counts),
            "Preprocess + VoxelDownsample");
        if (order == KICP_WARN_TABLE_ORDER) kicp_bridge::warn_once(kicp_last_error());
        return RegisterChained(result, guard, counts, relative_odometry);
    }
#endif

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
#ifndef KICP_HOST_PRESTEPS
    // From the moment the backend's helper thread holds a pointer into the result's frame vector: should any later step throw,
    // the download is collected (and dropped) before the vector is destroyed, so nothing is ever copied into freed memory.
    struct DownloadGuard {
        kicp_pre *pre;
        bool armed = true;
        ~DownloadGuard() {
            if (armed) (void)kicp_pre_download_finish(pre, 0, nullptr, 0, nullptr);
        }
    };
    // pipeline/KinematicICP.cpp:65-84 from the pre-steps' three buffers on (0: preprocessed frame, 1: first downsample - what goes
    // into the map, 2: second downsample - the registration source): register, update the threshold and the map - all on the
    // device; only the two returned clouds come back to the host.
    Vector3dVectorTuple RegisterChained(Vector3dVectorTuple &result, DownloadGuard &guard, const size_t counts[3], const Sophus::SE3d &relative_odometry) {
        kicp_bridge::Trace trace("registration");
        auto &frame = std::get<0>(result);
        const size_t n_down = counts[1], n_source = counts[2];
        const double tau = correspondence_threshold_.ComputeThreshold();
        const auto new_pose = registration_.ComputeRobotMotionDevice(kicp_pre_device_ptr(pre_, 2, nullptr), n_source, local_map_, last_pose_,
                                                                     relative_odometry, tau);
        trace.lap("threshold + map update");
        correspondence_threshold_.UpdateOdometryError((last_pose_ * relative_odometry).inverse() * new_pose);
        // (the map update's kernels run while this thread collects the two returned clouds: nothing below touches the map or buffer 1)
        local_map_.UpdateDeviceBegin(kicp_pre_device_ptr(pre_, 1, nullptr), n_down, new_pose);
        last_pose_ = new_pose;
        trace.lap("collect results");
        auto &source = std::get<1>(result);
        source.resize(n_source);
        kicp_bridge::check(kicp_pre_download(pre_, 2, source.empty() ? nullptr : source.front().data(), source.size(), nullptr), "download");
        guard.armed = false;
        if (!frame.empty()) kicp_bridge::check(kicp_pre_download_finish(pre_, 0, frame.front().data(), frame.size(), nullptr), "download");
        frame.resize(counts[0]);  // (the landing area held every input point; shrinking costs nothing)
        // The map update's kernels are still running: whatever touches the map next - the next frame's registration, LocalMap(),
        // VoxelMap() - collects them first (kicp.h: kicp_map_update_pose_device_begin), so they overlap the caller's own work and the
        // next frame's pre-steps instead of this thread's idle wait (round 6; the points they read stay in the pre-step workspace's
        // spare buffer meanwhile).  KICP_SYNC_MAP_UPDATE=1: wait here, as round 5 did.
        static const bool sync_update = [] { const char *e = std::getenv("KICP_SYNC_MAP_UPDATE"); return e && *e && *e != '0'; }();
        if (sync_update) {
            trace.lap("map update: wait");
            local_map_.UpdateFinish();
        }
        return std::move(result);  // built in place: no copy of the clouds on the way out
    }
#endif
    Sophus::SE3d last_pose_;
    KinematicRegistration registration_;
    CorrespondenceThreshold correspondence_threshold_;
    Config config_;
    kiss_icp::Preprocessor preprocessor_;
    kiss_icp::VoxelHashMap local_map_;
    kicp_pre *pre_ = nullptr;  // device workspace of the pre-steps (backend detail)
};

}  // namespace kinematic_icp::pipeline
