// ref_capi.cpp -- C entry points over the REFERENCE'S OWN CLASSES (test infrastructure, NOT product code).
//
// This file is linked with the reference's sources, compiled unmodified from where they lie
//   /root/reference/cpp/kinematic_icp/registration/Registration.cpp
//   /root/reference/cpp/kinematic_icp/correspondence_threshold/CorrespondenceThreshold.cpp
//   /root/reference/cpp/kinematic_icp/pipeline/KinematicICP.cpp
// against the stand-in headers of oracle/ref_shim/ (Eigen, Sophus, oneTBB, tsl::robin_map, kiss-icp v1.2.0 - none of
// which exist in this image) into oracle/_ref/libkicp_ref.so (oracle/Makefile, target `ref`).  It is what pins
// oracle/kicp_oracle.cpp and the HIP path to the reference's own text: control flow, formulas, thresholds, stop rule.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load the library.
//
// Conventions as in include/kicp.h: points = packed fp64 xyz, poses = [qx qy qz qw tx ty tz].
#include <tbb/shim_threads.h>

#include <chrono>
#include <cmath>
#include <cstring>
#include <memory>
#include <thread>
#include <vector>

#include "kinematic_icp/correspondence_threshold/CorrespondenceThreshold.hpp"
#include "kinematic_icp/pipeline/KinematicICP.hpp"
#include "kinematic_icp/registration/Registration.hpp"

namespace {
using Vec3 = Eigen::Vector3d;
std::vector<Vec3> to_points(const double *xyz, size_t n) {
    std::vector<Vec3> v(n);
    for (size_t i = 0; i < n; ++i) v[i] = Vec3(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]);
    return v;
}
size_t from_points(const std::vector<Vec3> &v, double *out, size_t cap) {
    const size_t k = v.size() < cap ? v.size() : cap;
    for (size_t i = 0; i < k; ++i) out[3 * i] = v[i].x(), out[3 * i + 1] = v[i].y(), out[3 * i + 2] = v[i].z();
    return v.size();
}
// the 7 parameters are adopted as they are (what an Eigen::Map<Sophus::SE3d> over caller memory does): no re-normalisation,
// so that poses cross this boundary bit for bit, as they do at include/kicp.h
Sophus::SE3d to_se3(const double p[7]) { return Sophus::SE3d(Sophus::SO3d::fromParams(p[0], p[1], p[2], p[3]), Vec3(p[4], p[5], p[6])); }
void from_se3(const Sophus::SE3d &T, double p[7]) {
    const auto &q = T.unit_quaternion();
    p[0] = q.x(), p[1] = q.y(), p[2] = q.z(), p[3] = q.w();
    p[4] = T.translation().x(), p[5] = T.translation().y(), p[6] = T.translation().z();
}
struct ThreadScope {  // per-call thread count for the TBB stand-in (<= 0: every core)
    explicit ThreadScope(int n) { tbb::shim::override_threads().store(n > 0 ? n : tbb::shim::hardware_threads()); }
    ~ThreadScope() { tbb::shim::override_threads().store(0); }
};
struct RefPipeline : kinematic_icp::pipeline::KinematicICP {  // reaches the protected members for inspection
    using KinematicICP::KinematicICP;
    double tau() const { return correspondence_threshold_.ComputeThreshold(); }
};
kiss_icp::VoxelHashMap *as_map(void *m) { return static_cast<kiss_icp::VoxelHashMap *>(m); }
}  // namespace

extern "C" {
const char *rkicp_sources() {
    return "registration/Registration.cpp correspondence_threshold/CorrespondenceThreshold.cpp pipeline/KinematicICP.cpp "
           "(unmodified, /root/reference/cpp/kinematic_icp) + stand-ins oracle/ref_shim";
}
int rkicp_hardware_threads() { return tbb::shim::hardware_threads(); }

// ---- kiss_icp::VoxelHashMap (stand-in) -------------------------------------------------------------------------------
void *rkicp_map_create(double voxel_size, double max_distance, unsigned int max_points_per_voxel) {
    return new kiss_icp::VoxelHashMap(voxel_size, max_distance, max_points_per_voxel);
}
void rkicp_map_destroy(void *m) { delete as_map(m); }
void rkicp_map_clear(void *m) { as_map(m)->Clear(); }
int rkicp_map_empty(void *m) { return as_map(m)->Empty() ? 1 : 0; }
void rkicp_map_add_points(void *m, const double *xyz, size_t n) { as_map(m)->AddPoints(to_points(xyz, n)); }
void rkicp_map_remove_far(void *m, const double origin[3]) { as_map(m)->RemovePointsFarFromLocation(Vec3(origin[0], origin[1], origin[2])); }
void rkicp_map_update_origin(void *m, const double *xyz, size_t n, const double origin[3]) {
    as_map(m)->Update(to_points(xyz, n), Vec3(origin[0], origin[1], origin[2]));
}
void rkicp_map_update_pose(void *m, const double *xyz, size_t n, const double pose_qt[7]) { as_map(m)->Update(to_points(xyz, n), to_se3(pose_qt)); }
size_t rkicp_map_num_voxels(void *m) { return as_map(m)->map_.size(); }
size_t rkicp_map_num_points(void *m) {
    size_t n = 0;
    const auto &map = as_map(m)->map_;
    for (auto it = map.cbegin(); it != map.cend(); ++it) n += it->second.size();
    return n;
}
size_t rkicp_map_pointcloud(void *m, double *out_xyz, size_t cap_points) { return from_points(as_map(m)->Pointcloud(), out_xyz, cap_points); }
void rkicp_map_closest(void *m, const double *queries, size_t n, double *out_nn, double *out_dist) {
    const auto *map = as_map(m);
    for (size_t i = 0; i < n; ++i) {
        const auto [nn, d] = map->GetClosestNeighbor(Vec3(queries[3 * i], queries[3 * i + 1], queries[3 * i + 2]));
        out_nn[3 * i] = nn.x(), out_nn[3 * i + 1] = nn.y(), out_nn[3 * i + 2] = nn.z();
        out_dist[i] = d;
    }
}

// ---- kinematic_icp::KinematicRegistration (the reference's own translation unit) ------------------------------------------
// Returns 0, or 1 if the pose contains NaN (zero correspondences).  out_seconds (may be NULL) = wall time of
// ComputeRobotMotion alone.  num_threads <= 0: all cores, like the reference's constructor (Registration.cpp:140-141).
int rkicp_register(void *m, const double *frame_xyz, size_t n, const double last_pose_qt[7], const double rel_odom_qt[7], double tau,
                   int max_num_iterations, double convergence_criterion, int num_threads, int use_adaptive_odometry_regularization,
                   double fixed_regularization, double out_pose_qt[7], double *out_seconds) {
    const std::vector<Vec3> frame = to_points(frame_xyz, n);
    kinematic_icp::KinematicRegistration registration(max_num_iterations, convergence_criterion, num_threads,
                                                      use_adaptive_odometry_regularization != 0, fixed_regularization);
    ThreadScope scope(num_threads);
    const auto t0 = std::chrono::steady_clock::now();
    const Sophus::SE3d pose = registration.ComputeRobotMotion(frame, *as_map(m), to_se3(last_pose_qt), to_se3(rel_odom_qt), tau);
    const auto t1 = std::chrono::steady_clock::now();
    if (out_seconds) *out_seconds = std::chrono::duration<double>(t1 - t0).count();
    from_se3(pose, out_pose_qt);
    for (int i = 0; i < 7; ++i)
        if (std::isnan(out_pose_qt[i])) return 1;
    return 0;
}
// Same call repeated `repeats` times on a frame converted once (timing aid: excludes the array -> vector conversion).
// Returns the wall time of the `repeats` ComputeRobotMotion calls.
double rkicp_register_timed(void *m, const double *frame_xyz, size_t n, const double last_pose_qt[7], const double rel_odom_qt[7],
                            double tau, int max_num_iterations, double convergence_criterion, int num_threads,
                            int use_adaptive_odometry_regularization, double fixed_regularization, int repeats, double out_pose_qt[7]) {
    const std::vector<Vec3> frame = to_points(frame_xyz, n);
    kinematic_icp::KinematicRegistration registration(max_num_iterations, convergence_criterion, num_threads,
                                                      use_adaptive_odometry_regularization != 0, fixed_regularization);
    ThreadScope scope(num_threads);
    const Sophus::SE3d last = to_se3(last_pose_qt), odom = to_se3(rel_odom_qt);
    Sophus::SE3d pose;
    const auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < repeats; ++r) pose = registration.ComputeRobotMotion(frame, *as_map(m), last, odom, tau);
    const auto t1 = std::chrono::steady_clock::now();
    from_se3(pose, out_pose_qt);
    return std::chrono::duration<double>(t1 - t0).count();
}

// THROUGHPUT of independent registrations: `threads` host threads, each registering scans (dealt round-robin, frames converted
// once up front) with its OWN KinematicRegistration at ONE thread (the reference's default, ros/launch/offline_node.launch.py:60)
// against the shared read-only map, for about `seconds`.  Returns the wall time; *out_scans = registrations completed.  What a CPU
// does with independent scans - the twin of the GPU path's scans-in-flight mode (bench.py cpu_baseline.throughput).
double rkicp_register_throughput(void *m, const double *frames_xyz, const size_t *n, size_t count, const double *last_poses_qt, const double *rel_odoms_qt,
                                 double tau, int max_num_iterations, double convergence_criterion, int use_adaptive_odometry_regularization,
                                 double fixed_regularization, int threads, double seconds, size_t *out_scans) {
    std::vector<std::vector<Vec3>> frames(count);
    size_t at = 0;
    for (size_t k = 0; k < count; ++k) frames[k] = to_points(frames_xyz + 3 * at, n[k]), at += n[k];
    ThreadScope scope(1);
    std::vector<size_t> done(static_cast<size_t>(threads), 0);
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<std::thread> pool;
    for (int t = 0; t < threads; ++t)
        pool.emplace_back([&, t] {
            kinematic_icp::KinematicRegistration registration(max_num_iterations, convergence_criterion, 1, use_adaptive_odometry_regularization != 0, fixed_regularization);
            for (size_t i = static_cast<size_t>(t); std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds; i += static_cast<size_t>(threads)) {
                const size_t k = i % count;
                volatile double sink = registration.ComputeRobotMotion(frames[k], *as_map(m), to_se3(last_poses_qt + 7 * k), to_se3(rel_odoms_qt + 7 * k), tau).translation().x();
                (void)sink;
                ++done[static_cast<size_t>(t)];
            }
        });
    for (auto &th : pool) th.join();
    const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    size_t total = 0;
    for (size_t d : done) total += d;
    if (out_scans) *out_scans = total;
    return wall;
}

// ---- kinematic_icp::CorrespondenceThreshold (the reference's own translation unit) -----------------------------------
void *rkicp_threshold_create(double map_discretization_error, double max_range, int use_adaptive_threshold, double fixed_threshold) {
    return new kinematic_icp::CorrespondenceThreshold(map_discretization_error, max_range, use_adaptive_threshold != 0, fixed_threshold);
}
void rkicp_threshold_destroy(void *t) { delete static_cast<kinematic_icp::CorrespondenceThreshold *>(t); }
double rkicp_threshold_compute(void *t) { return static_cast<kinematic_icp::CorrespondenceThreshold *>(t)->ComputeThreshold(); }
void rkicp_threshold_update(void *t, const double odometry_error_qt[7]) {
    static_cast<kinematic_icp::CorrespondenceThreshold *>(t)->UpdateOdometryError(to_se3(odometry_error_qt));
}
void rkicp_threshold_reset(void *t) { static_cast<kinematic_icp::CorrespondenceThreshold *>(t)->Reset(); }

// ---- kinematic_icp::pipeline::KinematicICP (the reference's own translation unit) ------------------------------------
struct rkicp_config {  // field for field pipeline::Config (KinematicICP.hpp:38-60), C types
    double max_range, min_range, voxel_size;
    unsigned int max_points_per_voxel;
    int use_adaptive_threshold;
    double fixed_threshold;
    int max_num_iterations;
    double convergence_criterion;
    int max_num_threads;
    int use_adaptive_odometry_regularization;
    double fixed_regularization;
    int deskew;
};
void *rkicp_pipeline_create(const rkicp_config *c) {
    kinematic_icp::pipeline::Config config;
    config.max_range = c->max_range, config.min_range = c->min_range, config.voxel_size = c->voxel_size;
    config.max_points_per_voxel = c->max_points_per_voxel, config.use_adaptive_threshold = c->use_adaptive_threshold != 0;
    config.fixed_threshold = c->fixed_threshold, config.max_num_iterations = c->max_num_iterations;
    config.convergence_criterion = c->convergence_criterion, config.max_num_threads = c->max_num_threads;
    config.use_adaptive_odometry_regularization = c->use_adaptive_odometry_regularization != 0;
    config.fixed_regularization = c->fixed_regularization, config.deskew = c->deskew != 0;
    return new RefPipeline(config);
}
void rkicp_pipeline_destroy(void *p) { delete static_cast<RefPipeline *>(p); }
void rkicp_pipeline_set_pose(void *p, const double pose_qt[7]) { static_cast<RefPipeline *>(p)->SetPose(to_se3(pose_qt)); }
void rkicp_pipeline_pose(void *p, double out_pose_qt[7]) { from_se3(static_cast<RefPipeline *>(p)->pose(), out_pose_qt); }
double rkicp_pipeline_tau(void *p) { return static_cast<RefPipeline *>(p)->tau(); }
// RegisterFrame(frame, timestamps, lidar_to_base, relative_odometry) -> {preprocessed frame in base, source}.  The output
// arrays need room for n points each; the counts come back through out_n_frame / out_n_source.
void rkicp_pipeline_register_frame(void *p, const double *frame_xyz, size_t n, const double *timestamps, size_t n_timestamps,
                                   const double lidar_to_base_qt[7], const double relative_odometry_qt[7], int num_threads, double *out_frame_xyz,
                                   size_t *out_n_frame, double *out_source_xyz, size_t *out_n_source) {
    ThreadScope scope(num_threads > 0 ? num_threads : 1);
    const std::vector<double> stamps(timestamps, timestamps + n_timestamps);
    const auto [frame, source] =
        static_cast<RefPipeline *>(p)->RegisterFrame(to_points(frame_xyz, n), stamps, to_se3(lidar_to_base_qt), to_se3(relative_odometry_qt));
    *out_n_frame = from_points(frame, out_frame_xyz, n);
    *out_n_source = from_points(source, out_source_xyz, n);
}
// The same call with the clock around RegisterFrame ALONE (the conversions from / to the C arrays on either side are not the
// reference's work): seconds of the reference's own KinematicICP::RegisterFrame (pipeline/KinematicICP.cpp:48-85) - what
// tools/bench_pipeline.py quotes beside the GPU's frame time.
double rkicp_pipeline_register_frame_timed(void *p, const double *frame_xyz, size_t n, const double *timestamps, size_t n_timestamps,
                                           const double lidar_to_base_qt[7], const double relative_odometry_qt[7], int num_threads, double *out_frame_xyz,
                                           size_t *out_n_frame, double *out_source_xyz, size_t *out_n_source) {
    ThreadScope scope(num_threads > 0 ? num_threads : 1);
    const std::vector<double> stamps(timestamps, timestamps + n_timestamps);
    const auto points = to_points(frame_xyz, n);
    const Sophus::SE3d lidar_to_base = to_se3(lidar_to_base_qt), odometry = to_se3(relative_odometry_qt);
    const auto t0 = std::chrono::steady_clock::now();
    const auto [frame, source] = static_cast<RefPipeline *>(p)->RegisterFrame(points, stamps, lidar_to_base, odometry);
    const auto t1 = std::chrono::steady_clock::now();
    *out_n_frame = from_points(frame, out_frame_xyz, n);
    *out_n_source = from_points(source, out_source_xyz, n);
    return std::chrono::duration<double>(t1 - t0).count();
}
size_t rkicp_pipeline_local_map(void *p, double *out_xyz, size_t cap_points) {
    return from_points(static_cast<RefPipeline *>(p)->LocalMap(), out_xyz, cap_points);
}
size_t rkicp_pipeline_map_num_points(void *p) { return static_cast<RefPipeline *>(p)->LocalMap().size(); }

// ---- the kiss-icp stand-ins on their own (pins oracle/kicp_oracle.cpp's second restatement of the same algorithms) ----
size_t rkicp_voxel_downsample(const double *xyz, size_t n, double voxel_size, double *out_xyz) {
    return from_points(kiss_icp::VoxelDownsample(to_points(xyz, n), voxel_size), out_xyz, n);
}
size_t rkicp_preprocess(const double *xyz, size_t n, const double *timestamps, size_t n_ts, const double relative_motion_qt[7],
                        double max_range, double min_range, int deskew, double *out_xyz) {
    const kiss_icp::Preprocessor pre(max_range, min_range, deskew != 0, 1);
    ThreadScope scope(1);
    return from_points(pre.Preprocess(to_points(xyz, n), std::vector<double>(timestamps, timestamps + n_ts), to_se3(relative_motion_qt)), out_xyz, n);
}

// ---- the Sophus stand-in on its own (pinned against scipy in tests/test_ref.py) ----------------------------------------
void rkicp_se3_exp(const double xi[6], double out_qt[7]) {
    Sophus::SE3d::Tangent a;
    for (int i = 0; i < 6; ++i) a(i) = xi[i];
    from_se3(Sophus::SE3d::exp(a), out_qt);
}
void rkicp_se3_log(const double qt[7], double out_xi[6]) {
    const Sophus::SE3d::Tangent a = to_se3(qt).log();
    for (int i = 0; i < 6; ++i) out_xi[i] = a(i);
}
void rkicp_se3_mul(const double a[7], const double b[7], double out[7]) { from_se3(to_se3(a) * to_se3(b), out); }
void rkicp_se3_inverse(const double a[7], double out[7]) { from_se3(to_se3(a).inverse(), out); }
void rkicp_se3_act(const double a[7], const double *xyz, size_t n, double *out_xyz) {
    const Sophus::SE3d T = to_se3(a);
    for (size_t i = 0; i < n; ++i) {
        const Vec3 r = T * Vec3(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]);
        out_xyz[3 * i] = r.x(), out_xyz[3 * i + 1] = r.y(), out_xyz[3 * i + 2] = r.z();
    }
}
}  // extern "C"
