// kicp_bridge.hpp's Sophus::SE3d <-> [qx qy qz qw tx ty tz] conversion, compiled against whatever Eigen / Sophus is on the
// include path (cpp/compat here; the real libraries where they exist): parameter ORDER (Eigen::Quaterniond's constructor takes
// w first, the C-ABI has it fourth), round trip, and that a pose built this way acts on a point like the C-ABI's convention
// says (rotation about z by 90 degrees maps x to y).
#include <cmath>
#include <cstdio>

#include "kicp_bridge.hpp"

int main() {
    const double s = std::sqrt(0.5);
    const double p[7] = {0.0, 0.0, s, s, 1.0, 2.0, 3.0};  // 90 degrees about z, then translate
    const Sophus::SE3d T = kicp_bridge::from_params(p);
    const Eigen::Vector3d q = T * Eigen::Vector3d(1.0, 0.0, 0.0);
    if (std::fabs(q.x() - 1.0) > 1e-12 || std::fabs(q.y() - 3.0) > 1e-12 || std::fabs(q.z() - 3.0) > 1e-12) return std::printf("act\n"), 1;
    double r[7];
    kicp_bridge::to_params(T, r);
    for (int i = 0; i < 7; ++i)
        if (std::fabs(r[i] - p[i]) > 1e-15) return std::printf("round trip %d\n", i), 1;
    const Sophus::SE3d I = T.inverse() * T;
    kicp_bridge::to_params(I, r);
    if (std::fabs(r[3] - 1.0) > 1e-12 || std::fabs(r[4]) > 1e-12 || std::fabs(r[6]) > 1e-12) return std::printf("inverse\n"), 1;
    const std::vector<Eigen::Vector3d> v{Eigen::Vector3d(1, 2, 3), Eigen::Vector3d(4, 5, 6)};
    const double *xyz = kicp_bridge::xyz(v);
    if (xyz[0] != 1 || xyz[3] != 4 || xyz[5] != 6) return std::printf("layout\n"), 1;
    std::printf("OK\n");
    return 0;
}
