// compat/sophus/se3.hpp -- minimal stand-in used ONLY when the real Sophus is not installed.  Implements the
// subset of Sophus::SE3d the reference's callers of this path use (construction, *, inverse, point action,
// translation, exp/log, the 7 raw parameters in Sophus' order qx qy qz qw tx ty tz), following Sophus' formulas
// (SURVEY.md App. B.2).
#pragma once
#include <Eigen/Core>
#include <array>
#include <cmath>
namespace Sophus {
class SE3d {
public:
    using Tangent = std::array<double, 6>;  // (upsilon, omega)
    SE3d() : q_{0, 0, 0, 1}, t_() {}
    SE3d(const std::array<double, 4> &q_xyzw, const Eigen::Vector3d &t) : q_(q_xyzw), t_(t) { normalize(); }
    static SE3d fromParams(const double p[7]) { return SE3d({p[0], p[1], p[2], p[3]}, Eigen::Vector3d(p[4], p[5], p[6])); }
    void toParams(double p[7]) const {
        for (int i = 0; i < 4; ++i) p[i] = q_[i];
        p[4] = t_[0], p[5] = t_[1], p[6] = t_[2];
    }
    const Eigen::Vector3d &translation() const { return t_; }
    Eigen::Vector3d &translation() { return t_; }
    const std::array<double, 4> &unit_quaternion_xyzw() const { return q_; }

    SE3d operator*(const SE3d &b) const {
        const auto &a = q_;
        const auto &c = b.q_;
        SE3d r;
        r.q_ = {a[3] * c[0] + a[0] * c[3] + a[1] * c[2] - a[2] * c[1], a[3] * c[1] + a[1] * c[3] + a[2] * c[0] - a[0] * c[2],
                a[3] * c[2] + a[2] * c[3] + a[0] * c[1] - a[1] * c[0], a[3] * c[3] - a[0] * c[0] - a[1] * c[1] - a[2] * c[2]};
        r.normalize();
        r.t_ = t_ + rotate(b.t_);
        return r;
    }
    Eigen::Vector3d operator*(const Eigen::Vector3d &p) const { return rotate(p) + t_; }
    SE3d inverse() const {
        SE3d r;
        r.q_ = {-q_[0], -q_[1], -q_[2], q_[3]};
        r.normalize();  // Sophus: SO3::inverse() passes the conjugate through the normalising constructor
        r.t_ = r.rotate(t_ * -1.0);
        return r;
    }
    Eigen::Vector3d rotate(const Eigen::Vector3d &p) const {
        const Eigen::Vector3d qv(q_[0], q_[1], q_[2]);
        Eigen::Vector3d uv = qv.cross(p);
        uv = uv + uv;
        return p + q_[3] * uv + qv.cross(uv);
    }
    // rotation angle as Sophus' SO3::logAndTheta().theta
    double so3_theta(Eigen::Vector3d *omega = nullptr) const {
        const double sn = q_[0] * q_[0] + q_[1] * q_[1] + q_[2] * q_[2], w = q_[3];
        double k, theta;
        if (sn < 1e-20) {
            k = 2.0 / w - (2.0 / 3.0) * sn / (w * w * w);
            theta = 2.0 * sn / w;
        } else {
            const double n = std::sqrt(sn);
            const double a = (w < 0.0) ? std::atan2(-n, -w) : std::atan2(n, w);
            k = 2.0 * a / n;
            theta = k * n;
        }
        if (omega) *omega = Eigen::Vector3d(k * q_[0], k * q_[1], k * q_[2]);
        return theta;
    }
    static SE3d exp(const Tangent &a) {
        const Eigen::Vector3d ups(a[0], a[1], a[2]), om(a[3], a[4], a[5]);
        const double th2 = om.squaredNorm();
        double theta, im, re;
        if (th2 < 1e-20) {
            theta = 0.0;
            im = 0.5 - th2 / 48.0 + th2 * th2 / 3840.0, re = 1.0 - th2 / 8.0 + th2 * th2 / 384.0;
        } else {
            theta = std::sqrt(th2);
            im = std::sin(0.5 * theta) / theta, re = std::cos(0.5 * theta);
        }
        SE3d r;
        r.q_ = {im * om[0], im * om[1], im * om[2], re};
        const Eigen::Vector3d wu = om.cross(ups), wwu = om.cross(wu);
        if (theta < 1e-10) {
            r.t_ = r.rotate(ups);
        } else {
            r.t_ = ups + ((1.0 - std::cos(theta)) / th2) * wu + ((theta - std::sin(theta)) / (th2 * theta)) * wwu;
        }
        return r;
    }
    Tangent log() const {
        Eigen::Vector3d om;
        const double theta = so3_theta(&om);
        const Eigen::Vector3d wt = om.cross(t_), wwt = om.cross(wt);
        Eigen::Vector3d u;
        if (std::abs(theta) < 1e-10) {
            u = t_ - 0.5 * wt + (1.0 / 12.0) * wwt;
        } else {
            const double h = 0.5 * theta;
            u = t_ - 0.5 * wt + ((1.0 - theta * std::cos(h) / (2.0 * std::sin(h))) / (theta * theta)) * wwt;
        }
        return {u[0], u[1], u[2], om[0], om[1], om[2]};
    }

private:
    void normalize() {
        const double n = std::sqrt(q_[0] * q_[0] + q_[1] * q_[1] + q_[2] * q_[2] + q_[3] * q_[3]);
        for (double &c : q_) c /= n;
    }
    std::array<double, 4> q_;
    Eigen::Vector3d t_;
};
}  // namespace Sophus
#define KICP_COMPAT_SOPHUS 1
