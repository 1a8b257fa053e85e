// compat/sophus/se3.hpp -- stand-in used ONLY where the real Sophus is not installed: the parts of Sophus::SE3d the drop-in
// headers and the reference's callers of this path use, with Sophus' own formulas and interface.
#pragma once
#include <Eigen/Core>
#include <cmath>

#include "so3.hpp"

namespace Sophus {
template <typename Scalar>
class SE3 {
public:
    using Point = Eigen::Matrix<Scalar, 3, 1>;
    using Tangent = Eigen::Matrix<Scalar, 6, 1>;
    using TranslationType = Point;

    SE3() : so3_(), translation_(Point::Zero()) {}
    SE3(const SO3<Scalar> &so3, const Point &translation) : so3_(so3), translation_(translation) {}
    SE3(const Eigen::Quaternion<Scalar> &quaternion, const Point &translation) : so3_(quaternion), translation_(translation) {}

    const SO3<Scalar> &so3() const { return so3_; }
    SO3<Scalar> &so3() { return so3_; }
    const Point &translation() const { return translation_; }
    Point &translation() { return translation_; }
    const Eigen::Quaternion<Scalar> &unit_quaternion() const { return so3_.unit_quaternion(); }

    SE3 inverse() const {
        const SO3<Scalar> invR = so3_.inverse();
        return SE3(invR, invR * (translation_ * Scalar(-1)));
    }
    SE3 operator*(const SE3 &other) const { return SE3(so3_ * other.so3_, translation_ + so3_ * other.translation_); }
    Point operator*(const Point &p) const { return so3_ * p + translation_; }

    static SE3 exp(const Tangent &a) {
        const Point omega(a(3), a(4), a(5));
        Scalar theta;
        const SO3<Scalar> so3 = SO3<Scalar>::expAndTheta(omega, &theta);
        const Eigen::Matrix<Scalar, 3, 3> Omega = SO3<Scalar>::hat(omega);
        const Eigen::Matrix<Scalar, 3, 3> Omega_sq = Omega * Omega;
        Eigen::Matrix<Scalar, 3, 3> V;
        if (theta < Constants<Scalar>::epsilon()) {
            V = so3.matrix();
        } else {
            const Scalar theta_sq = theta * theta;
            const Scalar c1 = (Scalar(1) - std::cos(theta)) / (theta_sq), c2 = (theta - std::sin(theta)) / (theta_sq * theta);
            const Eigen::Matrix<Scalar, 3, 3> I = Eigen::Matrix<Scalar, 3, 3>::Identity();
            for (int j = 0; j < 3; ++j)
                for (int i = 0; i < 3; ++i) V(i, j) = (I(i, j) + c1 * Omega(i, j)) + c2 * Omega_sq(i, j);
        }
        return SE3(so3, V * Point(a(0), a(1), a(2)));
    }
    Tangent log() const {
        Tangent upsilon_omega;
        const auto omega_and_theta = so3_.logAndTheta();
        const Scalar theta = omega_and_theta.theta;
        const Point omega = omega_and_theta.tangent;
        const Eigen::Matrix<Scalar, 3, 3> Omega = SO3<Scalar>::hat(omega);
        const Eigen::Matrix<Scalar, 3, 3> Omega_sq = Omega * Omega;
        const Eigen::Matrix<Scalar, 3, 3> I = Eigen::Matrix<Scalar, 3, 3>::Identity();
        Scalar c;
        if (std::abs(theta) < Constants<Scalar>::epsilon()) {
            c = Scalar(1. / 12.);
        } else {
            const Scalar half_theta = Scalar(0.5) * theta;
            c = (Scalar(1) - theta * std::cos(half_theta) / (Scalar(2) * std::sin(half_theta))) / (theta * theta);
        }
        Eigen::Matrix<Scalar, 3, 3> V_inv;
        for (int j = 0; j < 3; ++j)
            for (int i = 0; i < 3; ++i) V_inv(i, j) = (I(i, j) - Scalar(0.5) * Omega(i, j)) + c * Omega_sq(i, j);
        const Point upsilon = V_inv * translation_;
        for (int i = 0; i < 3; ++i) upsilon_omega(i) = upsilon(i), upsilon_omega(3 + i) = omega(i);
        return upsilon_omega;
    }

private:
    SO3<Scalar> so3_;
    Point translation_;
};
using SE3d = SE3<double>;
}  // namespace Sophus
