// kicp_bridge.hpp -- glue between the reference's C++ value types and the C-ABI of include/kicp.h.
// With the real Eigen/Sophus the same code compiles: only the two param-conversion helpers differ.
#pragma once
#include <Eigen/Core>
#include <Eigen/Geometry>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <sophus/se3.hpp>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "kicp.h"

namespace kicp_bridge {
// Sophus::SE3d <-> the C-ABI's 7 parameters [qx qy qz qw tx ty tz] (Sophus' own parameter order; Eigen::Quaterniond's
// constructor takes w first).  The same code for the real Sophus and for cpp/compat.  from_params goes through Sophus'
// normalising quaternion constructor, like any SE3d a caller builds from a quaternion.
inline void to_params(const Sophus::SE3d &T, double p[7]) {
    const auto &q = T.unit_quaternion();
    p[0] = q.x(), p[1] = q.y(), p[2] = q.z(), p[3] = q.w();
    p[4] = T.translation().x(), p[5] = T.translation().y(), p[6] = T.translation().z();
}
inline Sophus::SE3d from_params(const double p[7]) {
    return Sophus::SE3d(Eigen::Quaterniond(p[3], p[0], p[1], p[2]), Eigen::Vector3d(p[4], p[5], p[6]));
}
// what tbb::this_task_arena::max_concurrency() answers in the reference's constructors (Registration.cpp:141-142): the hardware
// threads of this machine, at least one
inline int hardware_threads() {
    const unsigned n = std::thread::hardware_concurrency();
    return n > 0u ? static_cast<int>(n) : 1;
}
inline const double *xyz(const std::vector<Eigen::Vector3d> &v) {
    static_assert(sizeof(Eigen::Vector3d) == 3 * sizeof(double), "Eigen::Vector3d must be 3 packed doubles");
    return v.empty() ? nullptr : v.front().data();
}
// Which GPU the drop-in classes use: the reference API has no notion of a device, so the choice travels out of band -
// KICP_DEVICE in the environment (default 0), read once per process.
inline int default_device() {
    static const int device = [] {
        const char *e = std::getenv("KICP_DEVICE");
        return e && *e ? std::atoi(e) : 0;
    }();
    return device;
}
// The reference's core throws nothing; a backend failure must not return garbage silently.
inline int check(int rc, const char *what) {
    if (rc < 0) throw std::runtime_error(std::string(what) + ": " + kicp_last_error());
    return rc;  // (> 0: a warning code of include/kicp.h; the result still follows the reference's convention)
}
// a warning the reference has no channel for: once per process on stderr
inline void warn_once(const char *message) {
    static bool said = false;
    if (!said) std::fprintf(stderr, "[kicp] warning: %s\n", message);
    said = true;
}
// KICP_TRACE=1: host-side sections of the drop-in headers report their wall time on stderr, next to the library's own
// per-call lines (debugging aid; a getenv once per process otherwise)
struct Trace {
    static bool enabled() {
        static const bool on = [] {
            const char *e = std::getenv("KICP_TRACE");
            return e && *e && *e != '0';
        }();
        return on;
    }
    const char *name;
    std::chrono::steady_clock::time_point t0;
    explicit Trace(const char *n) : name(n) {
        if (enabled()) t0 = std::chrono::steady_clock::now();
    }
    void lap(const char *next) {
        if (!enabled()) return;
        const auto t1 = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[kicp host] %-30s %9.3f ms\n", name, std::chrono::duration<double, std::milli>(t1 - t0).count());
        name = next, t0 = t1;
    }
    ~Trace() { lap(""); }
};
}  // namespace kicp_bridge
