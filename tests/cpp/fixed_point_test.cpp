// to_fixed / basis_of of kicp_kernels.hpp on the HOST (the functions are __host__ __device__: the device runs the same operations):
//   * the four signed 21-bit limbs add up to rint(x 2^40) - checked against round 4's route (two 64-bit conversions and 128-bit integer
//     arithmetic) over magnitudes from 2^-60 to the range limit 2^43, ties at the 2^-41 grid, signs, zeros;
//   * the range flag for |x| >= 2^43, infinities and NaN;
//   * basis_of: R UnitX, R UnitY of a pose and the fixed-point term of |R UnitX|^2 (2^40 exactly for a unit quaternion).
// Built by tests/test_closed_form.py with `hipcc --cuda-host-only`.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>

#include "kicp_kernels.hpp"

using namespace kicp;

static __int128 reference_fixed(double x) {  // round 4: ip 2^40 + rint((x - ip) 2^40), ip = rint(x)
    const long long ip = std::llrint(x);
    const long long fp = std::llrint((x - static_cast<double>(ip)) * kFixScale);
    return (static_cast<__int128>(ip) << 40) + fp;
}
static int check(double x, long &bad) {
    int limb[4] = {7, 7, 7, 7}, range_error = 0;
    to_fixed(x, limb, range_error);
    const bool in_range = std::fabs(x) < kFixLimit;  // (false for NaN)
    if (!in_range) {
        if (!range_error || limb[0] || limb[1] || limb[2] || limb[3]) ++bad, std::printf("range: %a -> flag %d limbs %d %d %d %d\n", x, range_error, limb[0], limb[1], limb[2], limb[3]);
        return 0;
    }
    const __int128 got = static_cast<__int128>(limb[0]) + (static_cast<__int128>(limb[1]) << 21) + (static_cast<__int128>(limb[2]) << 42) + (static_cast<__int128>(limb[3]) << 63);
    bool ok = !range_error && got == reference_fixed(x);
    for (int k = 0; k < 4; ++k) ok = ok && std::abs(limb[k]) < (1 << 21) && (limb[k] == 0 || (limb[k] < 0) == (x < 0));
    if (!ok) ++bad, std::printf("value: %a -> limbs %d %d %d %d (flag %d)\n", x, limb[0], limb[1], limb[2], limb[3], range_error);
    return 1;
}
int main() {
    long bad = 0, n = 0;
    std::mt19937_64 rng(12345);
    std::uniform_real_distribution<double> mant(1.0, 2.0);
    for (int e = -60; e <= 43; ++e)
        for (int k = 0; k < 2000; ++k) {
            const double x = std::ldexp(mant(rng), e - 1);
            n += check(x, bad), n += check(-x, bad);
        }
    for (long long k = -2000; k <= 2000; ++k) {  // ties and their neighbours on the 2^-41 grid, small and large
        for (double base : {0.0, 1.0, 1048576.0, 4398046511104.0, -8796093022207.0}) {
            const double x = base + std::ldexp(static_cast<double>(k), -41);
            n += check(x, bad), n += check(std::nextafter(x, 1e300), bad), n += check(std::nextafter(x, -1e300), bad);
        }
    }
    const double edge[] = {0.0, -0.0, kFixLimit, -kFixLimit, std::nextafter(kFixLimit, 0.0), -std::nextafter(kFixLimit, 0.0), 1e300, -1e300, INFINITY, -INFINITY, NAN,
                           std::ldexp(1.0, -41), std::ldexp(1.0, -42), std::ldexp(3.0, -42), 5e-324, -5e-324};
    for (double x : edge) n += check(x, bad);
    // basis_of
    std::normal_distribution<double> g(0.0, 1.0);
    for (int k = 0; k < 1000; ++k) {
        double q[4] = {g(rng), g(rng), g(rng), g(rng)};
        const double nq = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
        const Pose T{q[0] / nq, q[1] / nq, q[2] / nq, q[3] / nq, 10.0 * g(rng), 10.0 * g(rng), g(rng)};
        const PassBasis B = basis_of(T);
        const Rt m = pose_to_rt(T);
        const double err = std::fabs(B.c0x - m.r[0]) + std::fabs(B.c0y - m.r[3]) + std::fabs(B.c0z - m.r[6]) + std::fabs(B.c1x - m.r[1]) + std::fabs(B.c1y - m.r[4]) + std::fabs(B.c1z - m.r[7]);
        const bool one = B.jtj00[0] == 0 && B.jtj00[1] == (1 << 19) && B.jtj00[2] == 0 && B.jtj00[3] == 0;  // 2^40 = 2^19 2^21
        if (!(err < 1e-14) || !one) ++bad, std::printf("basis: err %g limbs %d %d %d %d\n", err, B.jtj00[0], B.jtj00[1], B.jtj00[2], B.jtj00[3]);
    }
    std::printf("%ld values checked, %ld bad\n", n, bad);
    return bad ? 1 : 0;
}
