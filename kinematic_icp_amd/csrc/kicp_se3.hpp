// kicp_se3.hpp -- the fp64 rigid-body algebra of the registration loop, usable from host code and from
// gfx950 device code (the device-side solve/update kernel runs it on one lane).
//
// Mirrors what the reference gets from Sophus::SE3d / Eigen at these call sites:
//   registration/Registration.cpp:156      last_robot_pose * relative_wheel_odometry
//   registration/Registration.cpp:74,88    T * point
//   registration/Registration.cpp:119-125  Matrix2d normalise + regularise + inverse()
//   registration/Registration.cpp:159-167  motion_model -> SE3d::exp
//   registration/Registration.cpp:181-182  current_estimate * delta_motion
// Pose storage: unit quaternion (x,y,z,w) + translation, i.e. Sophus' own 7 parameters.
#pragma once
#include <cfloat>
#include <cmath>

#include "kicp_common.hpp"

namespace kicp {

struct Pose {
    double qx, qy, qz, qw, tx, ty, tz;
};
struct Rt {  // rotation matrix (row major) + translation: what the per-point code consumes
    double r[9];
    double t[3];
};

KICP_HD void quat_rotate(const Pose &T, double px, double py, double pz, double &ox, double &oy, double &oz) {
    // p + w*(2 v x p) + v x (2 v x p), the form Sophus' SO3::operator*(point) evaluates
    double ux = T.qy * pz - T.qz * py, uy = T.qz * px - T.qx * pz, uz = T.qx * py - T.qy * px;
    ux += ux, uy += uy, uz += uz;
    ox = px + T.qw * ux + (T.qy * uz - T.qz * uy);
    oy = py + T.qw * uy + (T.qz * ux - T.qx * uz);
    oz = pz + T.qw * uz + (T.qx * uy - T.qy * ux);
}

KICP_HD Pose pose_mul(const Pose &a, const Pose &b) {
    // Hamilton product, re-normalised (Sophus constructs the product SO3 from the raw quaternion)
    double x = a.qw * b.qx + a.qx * b.qw + a.qy * b.qz - a.qz * b.qy;
    double y = a.qw * b.qy + a.qy * b.qw + a.qz * b.qx - a.qx * b.qz;
    double z = a.qw * b.qz + a.qz * b.qw + a.qx * b.qy - a.qy * b.qx;
    double w = a.qw * b.qw - a.qx * b.qx - a.qy * b.qy - a.qz * b.qz;
    const double n = sqrt(x * x + y * y + z * z + w * w);
    Pose r;
    r.qx = x / n, r.qy = y / n, r.qz = z / n, r.qw = w / n;
    double rx, ry, rz;
    quat_rotate(a, b.tx, b.ty, b.tz, rx, ry, rz);
    r.tx = a.tx + rx, r.ty = a.ty + ry, r.tz = a.tz + rz;
    return r;
}

KICP_HD Rt pose_to_rt(const Pose &T) {
    const double tx = 2 * T.qx, ty = 2 * T.qy, tz = 2 * T.qz;
    const double twx = tx * T.qw, twy = ty * T.qw, twz = tz * T.qw;
    const double txx = tx * T.qx, txy = ty * T.qx, txz = tz * T.qx;
    const double tyy = ty * T.qy, tyz = tz * T.qy, tzz = tz * T.qz;
    Rt m;
    m.r[0] = 1 - (tyy + tzz), m.r[1] = txy - twz, m.r[2] = txz + twy;
    m.r[3] = txy + twz, m.r[4] = 1 - (txx + tzz), m.r[5] = tyz - twx;
    m.r[6] = txz - twy, m.r[7] = tyz + twx, m.r[8] = 1 - (txx + tyy);
    m.t[0] = T.tx, m.t[1] = T.ty, m.t[2] = T.tz;
    return m;
}

// SE3 exponential of a twist (v, w) = (vx, vy, vz, wx, wy, wz).
KICP_HD Pose pose_exp(const double xi[6]) {
    const double wx = xi[3], wy = xi[4], wz = xi[5];
    const double th2 = wx * wx + wy * wy + wz * wz;
    double theta, imag, real;
    if (th2 < 1e-20) {  // Sophus: theta_sq < eps^2, eps = 1e-10
        theta = 0.0;
        const double th4 = th2 * th2;
        imag = 0.5 - (1.0 / 48.0) * th2 + (1.0 / 3840.0) * th4;
        real = 1.0 - (1.0 / 8.0) * th2 + (1.0 / 384.0) * th4;
    } else {
        theta = sqrt(th2);
        const double h = 0.5 * theta;
        imag = sin(h) / theta;
        real = cos(h);
    }
    Pose T;
    T.qx = imag * wx, T.qy = imag * wy, T.qz = imag * wz, T.qw = real;
    // V = I + a*W + b*W^2 (W = hat(w)); V = R when theta is below eps
    double V[9];
    if (theta < 1e-10) {
        const Rt m = pose_to_rt(T);
        for (int i = 0; i < 9; ++i) V[i] = m.r[i];
    } else {
        const double tsq = theta * theta;
        const double a = (1.0 - cos(theta)) / tsq;
        const double b = (theta - sin(theta)) / (tsq * theta);
        // W = [[0,-wz,wy],[wz,0,-wx],[-wy,wx,0]], W^2 = w w^T - |w|^2 I (evaluated as the explicit product)
        const double W[9] = {0, -wz, wy, wz, 0, -wx, -wy, wx, 0};
        double W2[9];
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) W2[3 * i + j] = W[3 * i] * W[j] + W[3 * i + 1] * W[3 + j] + W[3 * i + 2] * W[6 + j];
        // Eigen evaluates `I + a * Omega + b * Omega_sq` coefficient-wise, left to right
        for (int i = 0; i < 9; ++i) V[i] = (((i % 4 == 0) ? 1.0 : 0.0) + a * W[i]) + b * W2[i];
    }
    T.tx = V[0] * xi[0] + V[1] * xi[1] + V[2] * xi[2];
    T.ty = V[3] * xi[0] + V[4] * xi[1] + V[5] * xi[2];
    T.tz = V[6] * xi[0] + V[7] * xi[1] + V[8] * xi[2];
    return T;
}

KICP_HD Pose pose_inverse(const Pose &a) {
    // Sophus: SE3(so3().inverse(), so3().inverse() * (translation() * -1)); SO3::inverse() hands the conjugate to the
    // normalising quaternion constructor
    const double n = sqrt(a.qx * a.qx + a.qy * a.qy + a.qz * a.qz + a.qw * a.qw);
    Pose r;
    r.qx = -a.qx / n, r.qy = -a.qy / n, r.qz = -a.qz / n, r.qw = a.qw / n;
    double x, y, z;
    quat_rotate(r, a.tx * -1.0, a.ty * -1.0, a.tz * -1.0, x, y, z);
    r.tx = x, r.ty = y, r.tz = z;
    return r;
}

// SE3 logarithm -> twist (v, w), Sophus' formulas (used once per frame for the deskewing velocity; kiss-icp v1.2.0
// core/Preprocessing.cpp): V^-1 = I - 1/2 W + c W^2 as a matrix (coefficient-wise, left to right),
// then V^-1 t
KICP_HD void pose_log(const Pose &T, double xi[6]) {
    const double sn = T.qx * T.qx + T.qy * T.qy + T.qz * T.qz, w = T.qw;
    double k, theta;
    if (sn < 1e-20) {
        k = 2.0 / w - (2.0 / 3.0) * sn / (w * (w * w));
        theta = 2.0 * sn / w;
    } else {
        const double n = sqrt(sn);
        const double a = (w < 0.0) ? atan2(-n, -w) : atan2(n, w);
        k = 2.0 * a / n;
        theta = k * n;
    }
    const double ox = k * T.qx, oy = k * T.qy, oz = k * T.qz;
    double c;
    if (fabs(theta) < 1e-10) {
        c = 1.0 / 12.0;
    } else {
        const double h = 0.5 * theta;
        c = (1.0 - theta * cos(h) / (2.0 * sin(h))) / (theta * theta);
    }
    const double W[9] = {0, -oz, oy, oz, 0, -ox, -oy, ox, 0};
    double Vi[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            const double w2 = W[3 * i] * W[j] + W[3 * i + 1] * W[3 + j] + W[3 * i + 2] * W[6 + j];
            Vi[3 * i + j] = (((i == j) ? 1.0 : 0.0) - 0.5 * W[3 * i + j]) + c * w2;
        }
    xi[0] = Vi[0] * T.tx + Vi[1] * T.ty + Vi[2] * T.tz;
    xi[1] = Vi[3] * T.tx + Vi[4] * T.ty + Vi[5] * T.tz;
    xi[2] = Vi[6] * T.tx + Vi[7] * T.ty + Vi[8] * T.tz;
    xi[3] = ox, xi[4] = oy, xi[5] = oz;
}

// motion_model(integrated_controls) -- Registration.cpp:159-167.  NB theta == 0.0 exactly gives dx(0) = 0
// (epsilon = DBL_MIN only avoids 0/0): the reference's own behaviour, kept.
KICP_HD Pose motion_model(double displacement, double theta) {
    double xi[6] = {0, 0, 0, 0, 0, 0};
    xi[0] = displacement * sin(theta) / (theta + DBL_MIN);
    xi[1] = displacement * (1.0 - cos(theta)) / (theta + DBL_MIN);
    xi[5] = theta;
    return pose_exp(xi);
}

// The tail of ComputePerturbation -- Registration.cpp:119-125.  s = raw sums {JTJ00, JTJ01, JTJ11, JTr0, JTr1}.
KICP_HD void solve_perturbation(const double s[5], double n, double beta, double &dx0, double &dx1) {
    const double a = s[0] / n + beta, b = s[1] / n, c = s[1] / n, d = s[2] / n + 0.0;
    const double g0 = s[3] / n, g1 = s[4] / n;
    const double invdet = 1.0 / (a * d - c * b);
    dx0 = -((d * invdet) * g0 + (-b * invdet) * g1);
    dx1 = -((-c * invdet) * g0 + (a * invdet) * g1);
}

}  // namespace kicp
