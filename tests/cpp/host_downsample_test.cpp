// The drop-in HOST pre-steps (kinematic_icp_amd/cpp/kiss_icp/core/{VoxelUtils,Preprocessing}.hpp, the KICP_HOST_PRESTEPS
// path of KinematicICP::RegisterFrame) as a filter: reads fp64 data from files, writes the resulting cloud to stdout as raw
// doubles.  tests/test_table_order.py compares the bytes with the oracle's / the reference build's.
//   host_downsample_test downsample <points.bin> <voxel_size>
//   host_downsample_test preprocess <points.bin> <stamps.bin> <pose7.bin> <max_range> <min_range> <deskew 0|1>
//   host_downsample_test threshold <errors7.bin> <map_discretization_error> <max_range>   -> tau before / after every update
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <kiss_icp/core/Preprocessing.hpp>
#include <kiss_icp/core/VoxelUtils.hpp>
#include <kinematic_icp/correspondence_threshold/CorrespondenceThreshold.hpp>
#include <vector>

#include "kicp_bridge.hpp"

static std::vector<double> read_all(const char *path) {
    std::vector<double> v;
    FILE *f = std::fopen(path, "rb");
    if (!f) std::exit(2);
    double x;
    while (std::fread(&x, sizeof x, 1, f) == 1) v.push_back(x);
    std::fclose(f);
    return v;
}
static std::vector<Eigen::Vector3d> points_of(const std::vector<double> &v) {
    std::vector<Eigen::Vector3d> p;
    for (size_t i = 0; i + 2 < v.size(); i += 3) p.emplace_back(v[i], v[i + 1], v[i + 2]);
    return p;
}
int main(int argc, char **argv) {
    if (argc < 4) return 2;
    std::vector<Eigen::Vector3d> out;
    if (!std::strcmp(argv[1], "downsample")) {
        out = kiss_icp::VoxelDownsample(points_of(read_all(argv[2])), std::atof(argv[3]));
    } else if (!std::strcmp(argv[1], "preprocess") && argc >= 8) {
        const auto pose = read_all(argv[4]);
        const kiss_icp::Preprocessor pre(std::atof(argv[5]), std::atof(argv[6]), std::atoi(argv[7]) != 0, 1);
        out = pre.Preprocess(points_of(read_all(argv[2])), read_all(argv[3]), kicp_bridge::from_params(pose.data()));
    } else if (!std::strcmp(argv[1], "threshold") && argc >= 5) {
        const auto errs = read_all(argv[2]);
        kinematic_icp::CorrespondenceThreshold thr(std::atof(argv[3]), std::atof(argv[4]), true, 1.0);
        double tau = thr.ComputeThreshold();
        std::fwrite(&tau, sizeof tau, 1, stdout);
        for (size_t i = 0; i + 6 < errs.size(); i += 7) {
            thr.UpdateOdometryError(kicp_bridge::from_params(&errs[i]));
            tau = thr.ComputeThreshold();
            std::fwrite(&tau, sizeof tau, 1, stdout);
        }
        return 0;
    } else {
        return 2;
    }
    for (const auto &q : out) {
        const double v[3] = {q.x(), q.y(), q.z()};
        std::fwrite(v, sizeof(double), 3, stdout);
    }
    return 0;
}
