// compat/sophus/so3.hpp -- stand-in used ONLY where the real Sophus is not installed: the parts of Sophus::SO3d the drop-in
// headers and the reference's callers of this path use, with Sophus' own formulas and interface.
#pragma once
#include <Eigen/Core>
#include <Eigen/Geometry>
#include <cmath>

namespace Sophus {
template <typename Scalar>
struct Constants {
    static Scalar epsilon() { return Scalar(1e-10); }
};

template <typename Scalar>
class SO3 {
public:
    using Point = Eigen::Matrix<Scalar, 3, 1>;
    using Tangent = Eigen::Matrix<Scalar, 3, 1>;
    using Transformation = Eigen::Matrix<Scalar, 3, 3>;
    struct TangentAndTheta {
        Tangent tangent;
        Scalar theta;
    };

    SO3() : q_(Scalar(1), Scalar(0), Scalar(0), Scalar(0)) {}
    explicit SO3(const Eigen::Quaternion<Scalar> &quat) : q_(quat) { q_.normalize(); }
    // (test plumbing: the raw parameters, no normalisation)
    static SO3 fromParams(Scalar x, Scalar y, Scalar z, Scalar w) {
        SO3 r;
        r.q_ = Eigen::Quaternion<Scalar>(w, x, y, z);
        return r;
    }

    const Eigen::Quaternion<Scalar> &unit_quaternion() const { return q_; }
    Transformation matrix() const { return q_.toRotationMatrix(); }
    SO3 inverse() const { return SO3(q_.conjugate()); }

    SO3 operator*(const SO3 &other) const {
        const Eigen::Quaternion<Scalar> &a = q_, &b = other.q_;
        return SO3(Eigen::Quaternion<Scalar>(a.w() * b.w() - a.x() * b.x() - a.y() * b.y() - a.z() * b.z(),
                                             a.w() * b.x() + a.x() * b.w() + a.y() * b.z() - a.z() * b.y(),
                                             a.w() * b.y() + a.y() * b.w() + a.z() * b.x() - a.x() * b.z(),
                                             a.w() * b.z() + a.z() * b.w() + a.x() * b.y() - a.y() * b.x()));
    }
    Point operator*(const Point &p) const {
        const Point v = q_.vec();
        Point uv = v.cross(p);
        uv += uv;
        return p + q_.w() * uv + v.cross(uv);
    }

    static Transformation hat(const Tangent &omega) {
        Transformation Omega;
        Omega(0, 0) = Scalar(0), Omega(0, 1) = -omega(2), Omega(0, 2) = omega(1);
        Omega(1, 0) = omega(2), Omega(1, 1) = Scalar(0), Omega(1, 2) = -omega(0);
        Omega(2, 0) = -omega(1), Omega(2, 1) = omega(0), Omega(2, 2) = Scalar(0);
        return Omega;
    }
    static SO3 expAndTheta(const Tangent &omega, Scalar *theta) {
        const Scalar theta_sq = omega.squaredNorm();
        Scalar imag_factor, real_factor;
        if (theta_sq < Constants<Scalar>::epsilon() * Constants<Scalar>::epsilon()) {
            *theta = Scalar(0);
            const Scalar theta_po4 = theta_sq * theta_sq;
            imag_factor = Scalar(0.5) - Scalar(1.0 / 48.0) * theta_sq + Scalar(1.0 / 3840.0) * theta_po4;
            real_factor = Scalar(1) - Scalar(1.0 / 8.0) * theta_sq + Scalar(1.0 / 384.0) * theta_po4;
        } else {
            *theta = std::sqrt(theta_sq);
            const Scalar half_theta = Scalar(0.5) * (*theta);
            imag_factor = std::sin(half_theta) / (*theta);
            real_factor = std::cos(half_theta);
        }
        SO3 q;  // set directly: exp does not pass through the normalising constructor
        q.q_ = Eigen::Quaternion<Scalar>(real_factor, imag_factor * omega.x(), imag_factor * omega.y(), imag_factor * omega.z());
        return q;
    }
    static SO3 exp(const Tangent &omega) {
        Scalar theta;
        return expAndTheta(omega, &theta);
    }
    TangentAndTheta logAndTheta() const {
        TangentAndTheta J;
        const Scalar squared_n = q_.vec().squaredNorm();
        const Scalar w = q_.w();
        Scalar two_atan_nbyw_by_n;
        if (squared_n < Constants<Scalar>::epsilon() * Constants<Scalar>::epsilon()) {
            const Scalar squared_w = w * w;
            two_atan_nbyw_by_n = Scalar(2) / w - Scalar(2.0 / 3.0) * (squared_n) / (w * squared_w);
            J.theta = Scalar(2) * squared_n / w;
        } else {
            const Scalar n = std::sqrt(squared_n);
            const Scalar atan_nbyw = (w < Scalar(0)) ? Scalar(std::atan2(-n, -w)) : Scalar(std::atan2(n, w));
            two_atan_nbyw_by_n = Scalar(2) * atan_nbyw / n;
            J.theta = two_atan_nbyw_by_n * n;
        }
        J.tangent = two_atan_nbyw_by_n * q_.vec();
        return J;
    }
    Tangent log() const { return logAndTheta().tangent; }

private:
    Eigen::Quaternion<Scalar> q_;
};
using SO3d = SO3<double>;
}  // namespace Sophus
