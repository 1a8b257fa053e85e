// kicp_bridge.hpp's Sophus::SE3d <-> [qx qy qz qw tx ty tz] conversion, compiled against whatever Eigen / Sophus is on the
// include path (cpp/compat here; the real libraries where they exist): parameter ORDER (Eigen::Quaterniond's constructor takes
// w first, the C-ABI has it fourth), round trip, and that a pose built this way acts on a point like the C-ABI's convention
// says (rotation about z by 90 degrees maps x to y).
#include <cmath>
#include <cstdio>

#include "kicp_bridge.hpp"

// `bridge_test math <in.bin>`: records of 23 doubles [a(7) b(7) xi(6) p(3)] -> per record 36 doubles
// [a*p (3), params(a*b) (7), params(a^-1) (7), params(exp(xi)) (7), log(exp(xi)) (6), log(a) (6)], all through the Sophus types
// on the include path and kicp_bridge's conversions (tests/test_host.py compares them with scipy).
static int math_mode(const char *path) {
    FILE *f = std::fopen(path, "rb");
    if (!f) return 2;
    double in[23];
    while (std::fread(in, sizeof(double), 23, f) == 23) {
        const Sophus::SE3d a = kicp_bridge::from_params(in), b = kicp_bridge::from_params(in + 7);
        Eigen::Matrix<double, 6, 1> xi;
        for (int i = 0; i < 6; ++i) xi[i] = in[14 + i];
        double out[36];
        const Eigen::Vector3d q = a * Eigen::Vector3d(in[20], in[21], in[22]);
        out[0] = q.x(), out[1] = q.y(), out[2] = q.z();
        kicp_bridge::to_params(a * b, out + 3);
        kicp_bridge::to_params(a.inverse(), out + 10);
        const Sophus::SE3d e = Sophus::SE3d::exp(xi);
        kicp_bridge::to_params(e, out + 17);
        const Eigen::Matrix<double, 6, 1> le = e.log(), la = a.log();
        for (int i = 0; i < 6; ++i) out[24 + i] = le[i], out[30 + i] = la[i];
        std::fwrite(out, sizeof(double), 36, stdout);
    }
    std::fclose(f);
    return 0;
}

int main(int argc, char **argv) {
    if (argc >= 3 && std::string(argv[1]) == "math") return math_mode(argv[2]);
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
