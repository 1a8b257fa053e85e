// facade_test.cpp -- exercises the drop-in C++ headers exactly the way the reference's callers do
// (registration: pipeline/KinematicICP.cpp:68-72; pipeline: ros/.../LidarOdometryServer.cpp:105,205-206).
// Input: a little binary file written by tests/test_facade.py; output: poses as text on stdout.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <utility>
#include <thread>
#include <vector>

#include "kinematic_icp/pipeline/KinematicICP.hpp"

static std::vector<double> read_doubles(FILE *f, size_t n) {
    std::vector<double> v(n);
    if (n && fread(v.data(), sizeof(double), n, f) != n) {
        fprintf(stderr, "short read\n");
        exit(2);
    }
    return v;
}
static std::vector<Eigen::Vector3d> to_points(const std::vector<double> &v) {
    std::vector<Eigen::Vector3d> p(v.size() / 3);
    if (!p.empty()) std::memcpy(p.front().data(), v.data(), v.size() * sizeof(double));
    return p;
}
struct FrameLog {  // one frame of a timed drive, printed after the drive
    double ms, ms_with_free;
    size_t n_deskewed, n_source;
    Sophus::SE3d pose;
};
static void print_pose(const char *tag, const Sophus::SE3d &T) {
    double p[7];
    kicp_bridge::to_params(T, p);
    printf("%s %.17g %.17g %.17g %.17g %.17g %.17g %.17g\n", tag, p[0], p[1], p[2], p[3], p[4], p[5], p[6]);
}

int main(int argc, char **argv) {
    if (argc < 3) return 1;
    const std::string mode = argv[1];
    FILE *f = fopen(argv[2], "rb");
    if (!f) return 1;
    try {
        if (mode == "reg") {
            const auto h = read_doubles(f, 5);  // n_map, n_frame, voxel, max_range, tau
            const auto map_pts = to_points(read_doubles(f, static_cast<size_t>(h[0]) * 3));
            const auto frame = to_points(read_doubles(f, static_cast<size_t>(h[1]) * 3));
            const auto last = read_doubles(f, 7), rel = read_doubles(f, 7);
            kiss_icp::VoxelHashMap map(h[2], h[3], 20);
            map.AddPoints(map_pts);
            kinematic_icp::KinematicRegistration reg(10, 1e-3, 1, true, 0.0);
            const Sophus::SE3d pose = reg.ComputeRobotMotion(frame, map, kicp_bridge::from_params(last.data()),
                                                             kicp_bridge::from_params(rel.data()), h[4]);
            print_pose("pose", pose);
            printf("iterations %d converged %d\n", reg.last_stats().iterations, reg.last_stats().converged);
            const auto [nn, d] = map.GetClosestNeighbor(frame[0]);
            printf("closest %.17g %.17g %.17g %.17g\n", nn.x(), nn.y(), nn.z(), d);
            reg.max_num_iterations_ = 1;  // public mutable field, as in the reference
            reg.ComputeRobotMotion(frame, map, kicp_bridge::from_params(last.data()), kicp_bridge::from_params(rel.data()), h[4]);
            printf("iterations_after_edit %d\n", reg.last_stats().iterations);
            // the map is copyable like the reference's struct: a copy registers to the same pose, and clearing the original
            // leaves the copy alone
            kiss_icp::VoxelHashMap copy(map);
            map.Clear();
            kiss_icp::VoxelHashMap assigned(h[2], h[3], 20);
            assigned = copy;
            reg.max_num_iterations_ = 10;
            print_pose("pose_on_copy", reg.ComputeRobotMotion(frame, assigned, kicp_bridge::from_params(last.data()),
                                                              kicp_bridge::from_params(rel.data()), h[4]));
            printf("original_empty %d copy_points %zu\n", map.Empty() ? 1 : 0, copy.Pointcloud().size());
            // the registration is copyable / movable like the reference's struct (Registration.hpp:32-50): a copy carries the
            // edited public fields, registers to the same pose on a handle of its own, and outlives the original
            kinematic_icp::KinematicRegistration *first = new kinematic_icp::KinematicRegistration(10, 1e-3, 1, true, 0.0);
            first->max_num_iterations_ = 7;
            kinematic_icp::KinematicRegistration copied(*first);
            kinematic_icp::KinematicRegistration moved(std::move(*first));
            delete first;
            kinematic_icp::KinematicRegistration target(3, 1e-2, 1, false, 0.5);
            target = copied;
            printf("copied_fields %d %d %d %d %d\n", copied.max_num_iterations_, moved.max_num_iterations_, target.max_num_iterations_,
                   target.use_adaptive_odometry_regularization_ ? 1 : 0, copied.handle() != moved.handle() && target.handle() != copied.handle() ? 1 : 0);
            copied.max_num_iterations_ = moved.max_num_iterations_ = target.max_num_iterations_ = 10;
            print_pose("pose_copied", copied.ComputeRobotMotion(frame, assigned, kicp_bridge::from_params(last.data()), kicp_bridge::from_params(rel.data()), h[4]));
            print_pose("pose_moved", moved.ComputeRobotMotion(frame, assigned, kicp_bridge::from_params(last.data()), kicp_bridge::from_params(rel.data()), h[4]));
            print_pose("pose_assigned", target.ComputeRobotMotion(frame, assigned, kicp_bridge::from_params(last.data()), kicp_bridge::from_params(rel.data()), h[4]));
            // float32 frame (the wire format): the same registration as on the widened doubles
            std::vector<float> f32(frame.size() * 3);
            std::vector<Eigen::Vector3d> widened(frame.size());
            for (size_t i = 0; i < frame.size(); ++i)
                for (int c = 0; c < 3; ++c) f32[3 * i + c] = static_cast<float>(frame[i][c]), widened[i][c] = static_cast<double>(f32[3 * i + c]);
            print_pose("pose_f32", copied.ComputeRobotMotion(f32.data(), frame.size(), assigned, kicp_bridge::from_params(last.data()), kicp_bridge::from_params(rel.data()), h[4]));
            print_pose("pose_widened", copied.ComputeRobotMotion(widened, assigned, kicp_bridge::from_params(last.data()), kicp_bridge::from_params(rel.data()), h[4]));
            // the pipeline object too (every member by value in the reference, KinematicICP.hpp:100-108)
            kinematic_icp::pipeline::Config cfg;
            kinematic_icp::pipeline::KinematicICP icp(cfg);
            icp.VoxelMap().AddPoints(map_pts);
            kinematic_icp::pipeline::KinematicICP icp_copy(icp);
            icp.SetPose(Sophus::SE3d());
            kinematic_icp::pipeline::KinematicICP icp_moved(std::move(icp));
            printf("pipeline_copy %zu %zu\n", icp_copy.LocalMap().size(), icp_moved.LocalMap().size());
            // max_num_threads <= 0 reads back as the hardware's thread count, as in the reference's constructor (Registration.cpp:141-142)
            const int hw = static_cast<int>(std::thread::hardware_concurrency());
            printf("threads_field %d %d %d\n", kinematic_icp::KinematicRegistration(10, 1e-3, 0, true, 0.0).max_num_threads_ == (hw > 0 ? hw : 1) ? 1 : 0,
                   kinematic_icp::KinematicRegistration(10, 1e-3, -3, true, 0.0).max_num_threads_ == (hw > 0 ? hw : 1) ? 1 : 0,
                   kinematic_icp::KinematicRegistration(10, 1e-3, 4, true, 0.0).max_num_threads_);
        } else if (mode == "pipeline") {
            const auto h = read_doubles(f, 4);  // n_frames, voxel, max_range, deskew
            kinematic_icp::pipeline::Config cfg;
            cfg.voxel_size = h[1], cfg.max_range = h[2], cfg.deskew = h[3] != 0.0;
            kinematic_icp::pipeline::KinematicICP icp(cfg);
            const auto ext = read_doubles(f, 7);
            for (int k = 0; k < static_cast<int>(h[0]); ++k) {
                const auto n = read_doubles(f, 1);
                const auto frame = to_points(read_doubles(f, static_cast<size_t>(n[0]) * 3));
                const auto stamps = read_doubles(f, static_cast<size_t>(n[0]));
                const auto delta = read_doubles(f, 7);
                const auto [deskewed, source] = icp.RegisterFrame(frame, stamps, kicp_bridge::from_params(ext.data()),
                                                                  kicp_bridge::from_params(delta.data()));
                print_pose("pose", icp.pose());
                printf("sizes %zu %zu %zu\n", deskewed.size(), source.size(), icp.LocalMap().size());
            }
            icp.SetPose(Sophus::SE3d());
            printf("after_setpose %zu %d\n", icp.LocalMap().size(), icp.VoxelMap().Empty() ? 1 : 0);
        } else if (mode == "pipeline_raw") {  // same input file, fed as PointCloud2-style records (x y z f32, pad, t f64; 24 B)
            const auto h = read_doubles(f, 4);
            kinematic_icp::pipeline::Config cfg;
            cfg.voxel_size = h[1], cfg.max_range = h[2], cfg.deskew = h[3] != 0.0;
            kinematic_icp::pipeline::KinematicICP icp(cfg);
            const auto ext = read_doubles(f, 7);
            const kicp_cloud_layout layout{24, 0, 4, 8, KICP_FIELD_FLOAT64, 16};
            for (int k = 0; k < static_cast<int>(h[0]); ++k) {
                const auto n = read_doubles(f, 1);
                const size_t np = static_cast<size_t>(n[0]);
                const auto xyz = read_doubles(f, np * 3), stamps = read_doubles(f, np);
                const auto delta = read_doubles(f, 7);
                std::vector<unsigned char> msg(np * 24);
                for (size_t i = 0; i < np; ++i) {
                    const float p[3] = {static_cast<float>(xyz[3 * i]), static_cast<float>(xyz[3 * i + 1]), static_cast<float>(xyz[3 * i + 2])};
                    std::memcpy(&msg[24 * i], p, 12), std::memcpy(&msg[24 * i + 16], &stamps[i], 8);
                }
                const auto [has_stamps, lo, hi] = icp.IngestCloud(msg.data(), np, layout);
                const auto [deskewed, source] = icp.RegisterIngestedFrame(kicp_bridge::from_params(ext.data()), kicp_bridge::from_params(delta.data()));
                print_pose("pose", icp.pose());
                printf("sizes %zu %zu %zu\n", deskewed.size(), source.size(), icp.LocalMap().size());
                if (k == 0) printf("stamps %d %.17g %.17g\n", has_stamps ? 1 : 0, lo, hi);
            }
        } else if (mode == "pipeline_steps") {  // RegisterFrame's backend calls one by one with a clock around each (diagnostics)
            const auto h = read_doubles(f, 4);
            const auto ext = read_doubles(f, 7);
            kicp_pre *pre = nullptr;
            kicp_map *map = nullptr;
            kicp_reg *reg = nullptr;
            kicp_reg_config rc{10, 1e-3, 1, 1, 0.0};
            kicp_bridge::check(kicp_pre_create(0, &pre), "pre");
            kicp_bridge::check(kicp_map_create(h[1], h[2], 20, &map), "map");
            kicp_bridge::check(kicp_reg_create(&rc, 0, &reg), "reg");
            double last[7] = {0, 0, 0, 1, 0, 0, 0};
            const double ident[7] = {0, 0, 0, 1, 0, 0, 0};
            for (int k = 0; k < static_cast<int>(h[0]); ++k) {
                const auto n = read_doubles(f, 1);
                const size_t np = static_cast<size_t>(n[0]);
                const auto xyz = read_doubles(f, np * 3), stamps = read_doubles(f, np);
                const auto delta = read_doubles(f, 7);
                double t[8];
                auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
                size_t n_frame = 0, n_down = 0, n_src = 0;
                t[0] = now();
                kicp_bridge::check(kicp_pre_preprocess(pre, xyz.data(), np, stamps.data(), np, ident, ext.data(), h[2], 0.0, h[3] != 0.0, 0, &n_frame), "pp");
                t[1] = now();
                kicp_bridge::check(kicp_pre_voxel_downsample(pre, 0, h[1] * 0.5, 1, &n_down), "d1");
                kicp_bridge::check(kicp_pre_voxel_downsample(pre, 1, h[1] * 1.5, 2, &n_src), "d2");
                t[2] = now();
                double pose[7];
                kicp_stats st;
                kicp_bridge::check(kicp_register_device(reg, map, kicp_pre_device_ptr(pre, 2, nullptr), n_src, last, delta.data(), 0.67 * h[1], pose, &st), "reg");
                t[3] = now();
                std::vector<double> a(3 * n_frame), b(3 * n_src);
                t[4] = now();
                kicp_bridge::check(kicp_pre_download(pre, 0, a.data(), n_frame, nullptr), "dl0");
                kicp_bridge::check(kicp_pre_download(pre, 2, b.data(), n_src, nullptr), "dl2");
                t[5] = now();
                kicp_bridge::check(kicp_map_update_pose_device(map, 0, kicp_pre_device_ptr(pre, 1, nullptr), n_down, pose), "upd");
                t[6] = now();
                { std::vector<double>().swap(a); std::vector<double>().swap(b); }  // free the buffers the copies landed in
                t[7] = now();
                std::memcpy(last, pose, sizeof last);
                printf("frame %d ms: preprocess+upload %.3f downsample %.3f register %.3f alloc %.3f download %.3f map_update %.3f free %.3f | total %.3f\n", k,
                       t[1] - t[0], t[2] - t[1], t[3] - t[2], t[4] - t[3], t[5] - t[4], t[6] - t[5], t[7] - t[6], t[7] - t[0]);
            }
            kicp_reg_destroy(reg), kicp_map_destroy(map), kicp_pre_destroy(pre);
        } else if (mode == "pipeline_timed") {  // same input file; RegisterFrame alone inside the clock, no map download
            const auto h = read_doubles(f, 4);
            kinematic_icp::pipeline::Config cfg;
            cfg.voxel_size = h[1], cfg.max_range = h[2], cfg.deskew = h[3] != 0.0;
            kinematic_icp::pipeline::KinematicICP icp(cfg);
            const auto ext = read_doubles(f, 7);
            std::vector<FrameLog> log;
            const size_t n_total = static_cast<size_t>(h[0]);
            auto drive_t0 = std::chrono::steady_clock::now();
            for (int k = 0; k < static_cast<int>(h[0]); ++k) {
                const auto n = read_doubles(f, 1);
                const auto frame = to_points(read_doubles(f, static_cast<size_t>(n[0]) * 3));
                const auto stamps = read_doubles(f, static_cast<size_t>(n[0]));
                const auto delta = read_doubles(f, 7);
                const auto t0 = std::chrono::steady_clock::now();
                double ms = 0.0;
                size_t n_deskewed = 0, n_source = 0;
                {
                    const auto [deskewed, source] = icp.RegisterFrame(frame, stamps, kicp_bridge::from_params(ext.data()),
                                                                      kicp_bridge::from_params(delta.data()));
                    ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
                    n_deskewed = deskewed.size(), n_source = source.size();
                }  // the caller drops the returned clouds here
                const double ms_with_free = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
                // (nothing is printed - and nothing asked of the map - between frames: the map update of frame k is still running when
                //  RegisterFrame returns and is collected by frame k + 1's registration, inside ITS clock)
                log.push_back({ms, ms_with_free, n_deskewed, n_source, icp.pose()});
                if (log.size() == n_total / 2) drive_t0 = std::chrono::steady_clock::now();  // (the steady half: tables, buffers and threads are in place)
            }
            const double drive_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - drive_t0).count();
            const size_t map_points = icp.LocalMap().size();  // (collects the last frame's map update)
            const unsigned long long on_device = kicp_map_device_updates(icp.VoxelMap().handle());
            for (size_t k = 0; k < log.size(); ++k) {
                printf("frame %zu ms %.4f (%.4f incl. freeing the results) in %zu source %zu map_on_device %d\n", k, log[k].ms, log[k].ms_with_free, log[k].n_deskewed,
                       log[k].n_source, on_device == log.size() ? 1 : 0);
                print_pose("pose", log[k].pose);
            }
            printf("drive %zu frames %.4f ms (the second half of the drive, wall clock around the loop: what a deferred map update cannot hide in)\n",
                   log.size() - n_total / 2, drive_ms);
            printf("map %zu\n", map_points);
        } else if (mode == "pipeline_timed_raw" || mode == "pipeline_timed_raw_ahead") {  // the same frames as 16-byte PointCloud2 records (x y z t, FLOAT32): IngestCloud + RegisterIngestedFrame in the clock
            // (_ahead: a bag replay that holds message k + 1 while it registers message k - AnnounceNextCloud: the next message's upload hides
            //  behind the current frame's pre-steps)
            const bool ahead = mode == "pipeline_timed_raw_ahead";
            const auto h = read_doubles(f, 4);
            kinematic_icp::pipeline::Config cfg;
            cfg.voxel_size = h[1], cfg.max_range = h[2], cfg.deskew = h[3] != 0.0;
            kinematic_icp::pipeline::KinematicICP icp(cfg);
            const auto ext = read_doubles(f, 7);
            const kicp_cloud_layout layout{16, 0, 4, 8, KICP_FIELD_FLOAT32, 12};
            const int n_frames = static_cast<int>(h[0]);
            std::vector<std::vector<float>> msgs(static_cast<size_t>(n_frames));  // (the messages as they arrive from the driver / the bag: outside the clock)
            std::vector<std::vector<double>> deltas(static_cast<size_t>(n_frames));
            for (int k = 0; k < n_frames; ++k) {
                const auto n = read_doubles(f, 1);
                const size_t np = static_cast<size_t>(n[0]);
                const auto xyz = read_doubles(f, np * 3), stamps = read_doubles(f, np);
                deltas[k] = read_doubles(f, 7);
                auto &msg = msgs[k];
                msg.resize(np * 4);
                for (size_t i = 0; i < np; ++i)
                    msg[4 * i] = static_cast<float>(xyz[3 * i]), msg[4 * i + 1] = static_cast<float>(xyz[3 * i + 1]), msg[4 * i + 2] = static_cast<float>(xyz[3 * i + 2]),
                              msg[4 * i + 3] = static_cast<float>(stamps[i]);
            }
            std::vector<FrameLog> log;
            const size_t n_total = static_cast<size_t>(n_frames);
            auto drive_t0 = std::chrono::steady_clock::now();
            for (int k = 0; k < n_frames; ++k) {
                const auto &msg = msgs[k];
                const auto &delta = deltas[k];
                const size_t np = msg.size() / 4;
                const auto t0 = std::chrono::steady_clock::now();
                double ms = 0.0;
                size_t n_deskewed = 0, n_source = 0;
                {
                    (void)icp.IngestCloud(msg.data(), np, layout);
                    if (ahead && k + 1 < n_frames) icp.AnnounceNextCloud(msgs[k + 1].data(), msgs[k + 1].size() / 4, layout);
                    const auto [deskewed, source] = icp.RegisterIngestedFrame(kicp_bridge::from_params(ext.data()), kicp_bridge::from_params(delta.data()));
                    ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
                    n_deskewed = deskewed.size(), n_source = source.size();
                }
                const double ms_with_free = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
                // (nothing is printed - and nothing asked of the map - between frames: the map update of frame k is still running when
                //  RegisterFrame returns and is collected by frame k + 1's registration, inside ITS clock)
                log.push_back({ms, ms_with_free, n_deskewed, n_source, icp.pose()});
                if (log.size() == n_total / 2) drive_t0 = std::chrono::steady_clock::now();  // (the steady half: tables, buffers and threads are in place)
            }
            const double drive_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - drive_t0).count();
            const size_t map_points = icp.LocalMap().size();  // (collects the last frame's map update)
            const unsigned long long on_device = kicp_map_device_updates(icp.VoxelMap().handle());
            for (size_t k = 0; k < log.size(); ++k) {
                printf("frame %zu ms %.4f (%.4f incl. freeing the results) in %zu source %zu map_on_device %d\n", k, log[k].ms, log[k].ms_with_free, log[k].n_deskewed,
                       log[k].n_source, on_device == log.size() ? 1 : 0);
                print_pose("pose", log[k].pose);
            }
            printf("drive %zu frames %.4f ms (the second half of the drive, wall clock around the loop: what a deferred map update cannot hide in)\n",
                   log.size() - n_total / 2, drive_ms);
            printf("map %zu\n", map_points);
        }
    } catch (const std::exception &e) {
        fprintf(stderr, "exception: %s\n", e.what());
        return 3;
    }
    fclose(f);
    return 0;
}
