// kicp_oracle.cpp -- CPU ORACLE (test infrastructure, NOT product code).
//
// A dependency-free, line-faithful restatement of the reference's per-scan registration
// hot path.  It exists only so that tests/, __graft_entry__.smoke() and bench.py's
// `cpu_baseline` leg can check / time the HIP path against the reference arithmetic.
// Nothing under kinematic_icp_amd/ may include, link or call this file.
//
// PINNING: the reference ships no tests, golden vectors or fixtures, so the pin is the
// reference itself: `make -C oracle ref` compiles the reference's own Registration.cpp / CorrespondenceThreshold.cpp /
// KinematicICP.cpp, unmodified, against stand-in headers for the dependencies this image lacks (oracle/ref_shim/) into
// oracle/_ref/libkicp_ref.so, and tests/test_ref.py requires this file to equal that build BIT FOR BIT (registrations,
// thresholds, pre-steps, whole RegisterFrame sequences); its outputs are frozen in tests/golden/ref_outputs.npz.  What
// stays recalled rather than verified: the third-party parts both builds share by construction (kiss-icp v1.2.0
// VoxelHashMap / VoxelUtils / Preprocessing, Sophus SE3/SO3, tsl::robin_map's iteration order), restated from their
// published algorithms, pinned by analytic known-answer tests, scipy, and an independent
// numpy restatement (tests/ref_numpy.py).
//
// Reference map (all under /root/reference/cpp/kinematic_icp/):
//   registration/Registration.cpp:42-46    LinearSystem, Correspondences, epsilon
//   registration/Registration.cpp:48-60    ComputeOdometryRegularization   -> compute_odometry_regularization
//   registration/Registration.cpp:62-81    DataAssociation                  -> data_association
//   registration/Registration.cpp:83-126   ComputePerturbation              -> compute_perturbation
//   registration/Registration.cpp:151-190  ComputeRobotMotion               -> compute_robot_motion
//   correspondence_threshold/CorrespondenceThreshold.cpp:29-64              -> Threshold
//   pipeline/KinematicICP.cpp:31-85        transform_points, Voxelize, RegisterFrame (host glue)
//   kiss-icp v1.2.0 core/VoxelHashMap.cpp, core/VoxelUtils.{hpp,cpp}        -> VoxelMap, point_to_voxel,
//                                                                              voxel_downsample
//   kiss-icp v1.2.0 core/Preprocessing.cpp                                  -> preprocess
//   Sophus so3.hpp / se3.hpp                                                -> so3_*, se3_*
#include <algorithm>
#include <array>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <functional>
#include <limits>
#include <numeric>
#include <utility>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

namespace {

// ----------------------------------------------------------------------------------------
// Minimal linear algebra (stands in for Eigen::Vector3d / Matrix2d)
// ----------------------------------------------------------------------------------------
struct V3 {
    double x, y, z;
};
inline V3 operator+(const V3 &a, const V3 &b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(const V3 &a, const V3 &b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator*(double s, const V3 &a) { return {s * a.x, s * a.y, s * a.z}; }
inline double dot(const V3 &a, const V3 &b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3 cross(const V3 &a, const V3 &b) {
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
inline double squared_norm(const V3 &a) { return dot(a, a); }
inline double norm(const V3 &a) { return std::sqrt(squared_norm(a)); }

struct M3 {
    double m[3][3];
};
inline M3 m3_identity() { return {{{1, 0, 0}, {0, 1, 0}, {0, 0, 1}}}; }
inline M3 m3_mul(const M3 &a, const M3 &b) {
    M3 r{};
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) r.m[i][j] = a.m[i][0] * b.m[0][j] + a.m[i][1] * b.m[1][j] + a.m[i][2] * b.m[2][j];
    return r;
}
inline M3 m3_add(const M3 &a, const M3 &b) {
    M3 r{};
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) r.m[i][j] = a.m[i][j] + b.m[i][j];
    return r;
}
inline M3 m3_scale(double s, const M3 &a) {
    M3 r{};
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) r.m[i][j] = s * a.m[i][j];
    return r;
}
inline V3 m3_apply(const M3 &a, const V3 &v) {
    return {a.m[0][0] * v.x + a.m[0][1] * v.y + a.m[0][2] * v.z, a.m[1][0] * v.x + a.m[1][1] * v.y + a.m[1][2] * v.z,
            a.m[2][0] * v.x + a.m[2][1] * v.y + a.m[2][2] * v.z};
}
// SO3::hat (Sophus so3.hpp)
inline M3 hat(const V3 &w) { return {{{0, -w.z, w.y}, {w.z, 0, -w.x}, {-w.y, w.x, 0}}}; }

// ----------------------------------------------------------------------------------------
// Sophus restatement.  SE3d = unit quaternion (x,y,z,w) + translation.
// ----------------------------------------------------------------------------------------
constexpr double kSophusEps = 1e-10;  // Sophus::Constants<double>::epsilon()

struct Quat {
    double x, y, z, w;
};
struct SE3 {
    Quat q{0, 0, 0, 1};
    V3 t{0, 0, 0};
};

// Sophus SO3(quaternion) constructor: stores the quaternion and normalises it.
inline Quat so3_from_quat(Quat q) {
    const double n = std::sqrt(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
    return {q.x / n, q.y / n, q.z / n, q.w / n};
}
// Sophus SO3Base::operator*(SO3): explicit Hamilton product, result passes through the constructor.
inline Quat so3_mul(const Quat &a, const Quat &b) {
    return so3_from_quat({a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y, a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z,
                          a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x, a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z});
}
// Sophus SO3Base::operator*(point): uv = 2 (q.vec x p); p + w uv + q.vec x uv
inline V3 so3_act(const Quat &q, const V3 &p) {
    const V3 qv{q.x, q.y, q.z};
    V3 uv = cross(qv, p);
    uv = uv + uv;
    return p + q.w * uv + cross(qv, uv);
}
// Sophus SO3Base::inverse(): SO3(unit_quaternion().conjugate()) - through the normalising constructor
inline Quat so3_inverse(const Quat &q) { return so3_from_quat({-q.x, -q.y, -q.z, q.w}); }
// Sophus SO3Base::matrix() == Eigen::Quaternion::toRotationMatrix()
inline M3 so3_matrix(const Quat &q) {
    const double tx = 2 * q.x, ty = 2 * q.y, tz = 2 * q.z;
    const double twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
    const double txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
    const double tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
    return {{{1 - (tyy + tzz), txy - twz, txz + twy}, {txy + twz, 1 - (txx + tzz), tyz - twx}, {txz - twy, tyz + twx, 1 - (txx + tyy)}}};
}
// Sophus SO3::expAndTheta
inline Quat so3_exp_and_theta(const V3 &omega, double *theta) {
    const double theta_sq = squared_norm(omega);
    double imag_factor, real_factor;
    if (theta_sq < kSophusEps * kSophusEps) {
        *theta = 0.0;
        const double theta_po4 = theta_sq * theta_sq;
        imag_factor = 0.5 - (1.0 / 48.0) * theta_sq + (1.0 / 3840.0) * theta_po4;
        real_factor = 1.0 - (1.0 / 8.0) * theta_sq + (1.0 / 384.0) * theta_po4;
    } else {
        *theta = std::sqrt(theta_sq);
        const double half_theta = 0.5 * (*theta);
        imag_factor = std::sin(half_theta) / (*theta);
        real_factor = std::cos(half_theta);
    }
    return {imag_factor * omega.x, imag_factor * omega.y, imag_factor * omega.z, real_factor};
}
// Sophus SO3Base::logAndTheta
inline void so3_log_and_theta(const Quat &q, V3 *tangent, double *theta) {
    const double squared_n = q.x * q.x + q.y * q.y + q.z * q.z;
    const double w = q.w;
    double two_atan_nbyw_by_n;
    if (squared_n < kSophusEps * kSophusEps) {
        const double squared_w = w * w;
        two_atan_nbyw_by_n = 2.0 / w - (2.0 / 3.0) * squared_n / (w * squared_w);
        *theta = 2.0 * squared_n / w;
    } else {
        const double n = std::sqrt(squared_n);
        const double atan_nbyw = (w < 0.0) ? std::atan2(-n, -w) : std::atan2(n, w);
        two_atan_nbyw_by_n = 2.0 * atan_nbyw / n;
        *theta = two_atan_nbyw_by_n * n;
    }
    *tangent = {two_atan_nbyw_by_n * q.x, two_atan_nbyw_by_n * q.y, two_atan_nbyw_by_n * q.z};
}
// Sophus SE3Base::operator*(SE3), operator*(point), inverse()
inline SE3 se3_mul(const SE3 &a, const SE3 &b) { return {so3_mul(a.q, b.q), a.t + so3_act(a.q, b.t)}; }
inline V3 se3_act(const SE3 &T, const V3 &p) { return so3_act(T.q, p) + T.t; }
inline SE3 se3_inverse(const SE3 &T) {
    const Quat qi = so3_inverse(T.q);
    return {qi, so3_act(qi, -1.0 * T.t)};
}
// Sophus SE3::exp; tangent order (upsilon, omega)
inline SE3 se3_exp(const double a[6]) {
    const V3 upsilon{a[0], a[1], a[2]}, omega{a[3], a[4], a[5]};
    double theta;
    const Quat so3 = so3_exp_and_theta(omega, &theta);
    const M3 Omega = hat(omega);
    const M3 Omega_sq = m3_mul(Omega, Omega);
    M3 V;
    if (theta < kSophusEps) {
        V = so3_matrix(so3);
    } else {
        const double theta_sq = theta * theta;
        // Eigen evaluates `I + a * Omega + b * Omega_sq` coefficient-wise, left to right
        V = m3_add(m3_add(m3_identity(), m3_scale((1.0 - std::cos(theta)) / theta_sq, Omega)),
                   m3_scale((theta - std::sin(theta)) / (theta_sq * theta), Omega_sq));
    }
    return {so3, m3_apply(V, upsilon)};
}
// Sophus SE3Base::log
inline void se3_log(const SE3 &T, double out[6]) {
    V3 omega;
    double theta;
    so3_log_and_theta(T.q, &omega, &theta);
    const M3 Omega = hat(omega);
    const M3 Omega_sq = m3_mul(Omega, Omega);
    M3 V_inv;
    if (std::abs(theta) < kSophusEps) {
        V_inv = m3_add(m3_add(m3_identity(), m3_scale(-0.5, Omega)), m3_scale(1.0 / 12.0, Omega_sq));
    } else {
        const double half_theta = 0.5 * theta;
        V_inv = m3_add(m3_add(m3_identity(), m3_scale(-0.5, Omega)),
                       m3_scale((1.0 - theta * std::cos(half_theta) / (2.0 * std::sin(half_theta))) / (theta * theta), Omega_sq));
    }
    const V3 u = m3_apply(V_inv, T.t);
    out[0] = u.x, out[1] = u.y, out[2] = u.z, out[3] = omega.x, out[4] = omega.y, out[5] = omega.z;
}
inline SE3 se3_from_qt(const double p[7]) { return {{p[0], p[1], p[2], p[3]}, {p[4], p[5], p[6]}}; }
inline void se3_to_qt(const SE3 &T, double p[7]) {
    p[0] = T.q.x, p[1] = T.q.y, p[2] = T.q.z, p[3] = T.q.w, p[4] = T.t.x, p[5] = T.t.y, p[6] = T.t.z;
}

// ----------------------------------------------------------------------------------------
// kiss-icp v1.2.0 core/VoxelUtils.hpp : Voxel, PointToVoxel, std::hash<Voxel>  (App. A.1)
// ----------------------------------------------------------------------------------------
struct Voxel {
    int32_t x, y, z;
    bool operator==(const Voxel &o) const { return x == o.x && y == o.y && z == o.z; }
};
// (Eigen::Vector3i addition is plain int addition.  After a pass without correspondences the pose is NaN, PointToVoxel(NaN) is
//  INT_MIN on x86, and the reference's `voxel + shift` overflows - which the compiled reference wraps.  `make -C oracle asan`
//  found it; the wrap is spelled out here so that the restatement has defined behaviour with the same result.)
inline int32_t wrap_add(int32_t a, int32_t b) { return static_cast<int32_t>(static_cast<uint32_t>(a) + static_cast<uint32_t>(b)); }
inline Voxel operator+(const Voxel &a, const Voxel &b) { return {wrap_add(a.x, b.x), wrap_add(a.y, b.y), wrap_add(a.z, b.z)}; }
inline Voxel point_to_voxel(const V3 &p, double voxel_size) {
    return {static_cast<int>(std::floor(p.x / voxel_size)), static_cast<int>(std::floor(p.y / voxel_size)),
            static_cast<int>(std::floor(p.z / voxel_size))};
}
inline size_t voxel_hash(const Voxel &v) {
    const uint32_t a = static_cast<uint32_t>(v.x), b = static_cast<uint32_t>(v.y), c = static_cast<uint32_t>(v.z);
    return static_cast<size_t>((a * 73856093u) ^ (b * 19349669u) ^ (c * 83492791u));
}

// Open-addressing stand-in for tsl::robin_map<Voxel, T> (power-of-two table, linear probing,
// backward-shift erase).  Only find / insert / erase / ordered iteration are used; the iteration
// order of the real robin_map is NOT reproduced (it never influences a query result - App. A.1).
template <typename T>
class VoxelTable {
public:
    struct Slot {
        Voxel key;
        int32_t used = 0;
        T value;
    };
    VoxelTable() { slots_.resize(16); }
    void clear() {
        slots_.assign(16, Slot{});
        size_ = 0;
    }
    size_t size() const { return size_; }
    bool empty() const { return size_ == 0; }
    const Slot *find(const Voxel &k) const {
        const size_t mask = slots_.size() - 1;
        for (size_t i = voxel_hash(k) & mask;; i = (i + 1) & mask) {
            const Slot &s = slots_[i];
            if (!s.used) return nullptr;
            if (s.key == k) return &s;
        }
    }
    Slot *find(const Voxel &k) { return const_cast<Slot *>(static_cast<const VoxelTable *>(this)->find(k)); }
    // precondition: key absent
    Slot *insert(const Voxel &k, T &&value) {
        if ((size_ + 1) * 2 > slots_.size()) grow();
        const size_t mask = slots_.size() - 1;
        size_t i = voxel_hash(k) & mask;
        while (slots_[i].used) i = (i + 1) & mask;
        slots_[i].key = k, slots_[i].used = 1, slots_[i].value = std::move(value);
        ++size_;
        return &slots_[i];
    }
    // erase slot index i (backward-shift); returns true if a later element moved into i
    bool erase_at(size_t i) {
        const size_t mask = slots_.size() - 1;
        slots_[i].used = 0;
        slots_[i].value = T{};
        --size_;
        bool moved_into_i = false;
        size_t hole = i;
        for (size_t j = (i + 1) & mask; slots_[j].used; j = (j + 1) & mask) {
            const size_t home = voxel_hash(slots_[j].key) & mask;
            // can slot j move to the hole?  yes iff home is cyclically outside (hole, j]
            const bool between = (hole <= j) ? (home > hole && home <= j) : (home > hole || home <= j);
            if (!between) {
                slots_[hole] = std::move(slots_[j]);
                slots_[j].used = 0;
                slots_[j].value = T{};
                if (hole == i) moved_into_i = true;
                hole = j;
            }
        }
        return moved_into_i;
    }
    std::vector<Slot> &slots() { return slots_; }
    const std::vector<Slot> &slots() const { return slots_; }

private:
    void grow() {
        std::vector<Slot> old;
        old.swap(slots_);
        slots_.resize(old.size() * 2);
        size_ = 0;
        for (auto &s : old)
            if (s.used) insert(s.key, std::move(s.value));
    }
    std::vector<Slot> slots_;
    size_t size_ = 0;
};

// ----------------------------------------------------------------------------------------
// kiss-icp v1.2.0 core/VoxelHashMap.{hpp,cpp}  (App. A.2 - A.6)
// ----------------------------------------------------------------------------------------
const std::array<Voxel, 27> kVoxelShifts{{{0, 0, 0},   {1, 0, 0},   {-1, 0, 0},  {0, 1, 0},   {0, -1, 0},  {0, 0, 1},   {0, 0, -1},
                                          {1, 1, 0},   {1, -1, 0},  {-1, 1, 0},  {-1, -1, 0}, {1, 0, 1},   {1, 0, -1},  {-1, 0, 1},
                                          {-1, 0, -1}, {0, 1, 1},   {0, 1, -1},  {0, -1, 1},  {0, -1, -1}, {1, 1, 1},   {1, 1, -1},
                                          {1, -1, 1},  {1, -1, -1}, {-1, 1, 1},  {-1, 1, -1}, {-1, -1, 1}, {-1, -1, -1}}};

struct QueryCounters {
    uint64_t probes = 0;         // map_.find calls
    uint64_t occupied = 0;       // finds that hit a bucket
    uint64_t points_scanned = 0; // bucket points visited by min_element
};

struct VoxelMap {
    double voxel_size_;
    double max_distance_;
    unsigned int max_points_per_voxel_;
    VoxelTable<std::vector<V3>> map_;

    VoxelMap(double vs, double md, unsigned int mp) : voxel_size_(vs), max_distance_(md), max_points_per_voxel_(mp) {}
    void Clear() { map_.clear(); }
    bool Empty() const { return map_.empty(); }

    // VoxelHashMap::AddPoints (App. A.4)
    void AddPoints(const V3 *points, size_t n) {
        const double map_resolution = std::sqrt(voxel_size_ * voxel_size_ / max_points_per_voxel_);
        for (size_t i = 0; i < n; ++i) {
            const V3 &point = points[i];
            const Voxel voxel = point_to_voxel(point, voxel_size_);
            auto *search = map_.find(voxel);
            if (search != nullptr) {
                auto &voxel_points = search->value;
                if (voxel_points.size() == max_points_per_voxel_ ||
                    std::any_of(voxel_points.cbegin(), voxel_points.cend(),
                                [&](const V3 &voxel_point) { return norm(voxel_point - point) < map_resolution; })) {
                    continue;
                }
                voxel_points.emplace_back(point);
            } else {
                std::vector<V3> voxel_points;
                voxel_points.reserve(max_points_per_voxel_);
                voxel_points.emplace_back(point);
                map_.insert(voxel, std::move(voxel_points));
            }
        }
    }
    // VoxelHashMap::RemovePointsFarFromLocation (App. A.6)
    void RemovePointsFarFromLocation(const V3 &origin) {
        const double max_distance2 = max_distance_ * max_distance_;
        auto &slots = map_.slots();
        for (size_t i = 0; i < slots.size();) {
            if (slots[i].used && squared_norm(slots[i].value.front() - origin) >= max_distance2) {
                if (map_.erase_at(i)) continue;  // a shifted element now sits at i: re-examine it
            }
            ++i;
        }
        // Backward-shift can wrap an element from the table's start to its end region only when the
        // probe chain crosses the wrap point; such an element was already examined (it lived at a
        // smaller index), so every surviving voxel has been tested exactly once or twice - the
        // predicate is idempotent, hence the result equals the reference's single sweep.
    }
    // VoxelHashMap::Update(points, origin) / Update(points, pose) (App. A.5)
    void Update(const V3 *points, size_t n, const V3 &origin) {
        AddPoints(points, n);
        RemovePointsFarFromLocation(origin);
    }
    void Update(const V3 *points, size_t n, const SE3 &pose) {
        std::vector<V3> points_transformed(n);
        std::transform(points, points + n, points_transformed.begin(), [&](const V3 &p) { return se3_act(pose, p); });
        Update(points_transformed.data(), n, pose.t);
    }
    // VoxelHashMap::Pointcloud
    std::vector<V3> Pointcloud() const {
        std::vector<V3> points;
        points.reserve(map_.size() * static_cast<size_t>(max_points_per_voxel_));
        for (const auto &s : map_.slots())
            if (s.used) points.insert(points.end(), s.value.cbegin(), s.value.cend());
        points.shrink_to_fit();
        return points;
    }
    size_t NumPoints() const {
        size_t n = 0;
        for (const auto &s : map_.slots())
            if (s.used) n += s.value.size();
        return n;
    }
    // VoxelHashMap::GetClosestNeighbor (App. A.3)
    std::pair<V3, double> GetClosestNeighbor(const V3 &query, QueryCounters *c = nullptr) const {
        const Voxel voxel = point_to_voxel(query, voxel_size_);
        V3 closest_neighbor{0, 0, 0};
        double closest_distance = std::numeric_limits<double>::max();
        for (const auto &voxel_shift : kVoxelShifts) {
            const Voxel query_voxel = voxel + voxel_shift;
            const auto *search = map_.find(query_voxel);
            if (c) ++c->probes;
            if (search != nullptr) {
                const auto &points = search->value;
                const V3 &neighbor = *std::min_element(points.cbegin(), points.cend(), [&](const V3 &lhs, const V3 &rhs) {
                    return norm(lhs - query) < norm(rhs - query);
                });
                const double distance = norm(neighbor - query);
                if (distance < closest_distance) {
                    closest_neighbor = neighbor;
                    closest_distance = distance;
                }
                if (c) ++c->occupied, c->points_scanned += points.size();
            }
        }
        return {closest_neighbor, closest_distance};
    }
};

// kiss-icp v1.2.0 core/VoxelUtils.cpp : VoxelDownsample (App. A.7): the first point of every voxel, emitted in the
// iteration order of the reference's `tsl::robin_map<Voxel, Vector3d> grid` after `grid.reserve(frame.size())`
// [RECALLED: robin-map 1.x semantics, see oracle/ref_shim/tsl/robin_map.h, which implements the general container;
// tests/test_ref.py checks this specialised restatement against it]:
//   * reserve(n) gives B = pow2ceil(ceil(n / 0.5)) buckets, so the table never re-hashes while <= n voxels go in;
//   * a voxel seen for the first time walks from its ideal bucket h = hash & (B-1) past every resident that is at
//     least as far from ITS ideal bucket, takes the first bucket whose resident is closer to home (or empty), and the
//     evicted resident walks on under the same rule - so an evicted element also passes the residents that share its
//     ideal bucket (the order inside such a group is a function of the whole insertion history, not of first-seen time);
//   * the output is the table read in ascending bucket order.
// `first_index_out` (optional) receives the input index of every emitted point.
std::vector<V3> voxel_downsample(const V3 *frame, size_t n, double voxel_size, std::vector<uint32_t> *first_index_out = nullptr) {
    size_t buckets = 0;
    const size_t want = static_cast<size_t>(std::ceil(static_cast<float>(n) / 0.5f));
    if (want > 0) {
        buckets = 1;
        while (buckets < want) buckets <<= 1;
    }
    constexpr uint32_t kFree = 0xFFFFFFFFu;
    std::vector<uint32_t> point_at(buckets, kFree);  // input index of the point stored in a bucket
    std::vector<int32_t> dist_at(buckets, -1);       // its distance from the ideal bucket (-1 = free)
    const size_t mask = buckets - 1;
    for (size_t i = 0; i < n; ++i) {
        const Voxel voxel = point_to_voxel(frame[i], voxel_size);
        size_t b = voxel_hash(voxel) & mask;
        int32_t dist = 0;
        bool present = false;
        while (dist <= dist_at[b]) {  // look-up: stop at the first bucket whose resident is closer to home than we are
            if (point_to_voxel(frame[point_at[b]], voxel_size) == voxel) {
                present = true;
                break;
            }
            b = (b + 1) & mask, ++dist;
        }
        if (present) continue;  // not the first point of its voxel
        uint32_t carry = static_cast<uint32_t>(i);
        while (dist_at[b] >= 0) {  // evict the closer-to-home resident and carry it on
            if (dist > dist_at[b]) std::swap(carry, point_at[b]), std::swap(dist, dist_at[b]);
            b = (b + 1) & mask, ++dist;
        }
        point_at[b] = carry, dist_at[b] = dist;
    }
    std::vector<V3> out;
    if (first_index_out) first_index_out->clear();
    for (size_t b = 0; b < buckets; ++b)
        if (dist_at[b] >= 0) {
            out.emplace_back(frame[point_at[b]]);
            if (first_index_out) first_index_out->push_back(point_at[b]);
        }
    return out;
}

// kiss-icp v1.2.0 core/Preprocessing.cpp : Preprocessor::Preprocess (App. A.8)
std::vector<V3> preprocess(const V3 *frame, size_t n, const double *timestamps, size_t n_ts, const SE3 &relative_motion,
                           double max_range, double min_range, bool deskew) {
    std::vector<V3> deskewed(frame, frame + n);
    if (deskew && n_ts != 0) {
        double omega[6];
        se3_log(relative_motion, omega);
        const SE3 motion_inverse = se3_inverse(relative_motion);
        for (size_t i = 0; i < n; ++i) {
            double xi[6];
            for (int k = 0; k < 6; ++k) xi[k] = timestamps[i] * omega[k];
            const SE3 pose = se3_mul(motion_inverse, se3_exp(xi));
            deskewed[i] = se3_act(pose, frame[i]);
        }
    }
    std::vector<V3> out;
    out.reserve(n);
    for (const auto &p : deskewed) {
        const double r = norm(p);
        if (r < max_range && r > min_range) out.emplace_back(p);
    }
    return out;
}

// ----------------------------------------------------------------------------------------
// registration/Registration.cpp
// ----------------------------------------------------------------------------------------
constexpr double epsilon = std::numeric_limits<double>::min();  // Registration.cpp:46
using Correspondences = std::vector<std::pair<V3, V3>>;         // Registration.cpp:43 (serial order)
struct LinearSystem {                                           // Registration.cpp:42
    double JTJ[2][2] = {{0, 0}, {0, 0}};
    double JTr[2] = {0, 0};
};

// Registration.cpp:48-60.  The reference sums with std::transform_reduce (no execution policy); libstdc++ evaluates that
// over random-access iterators four elements at a time - init + ((u0 + u1) + (u2 + u3)) - so calling the same
// algorithm here reproduces the reference's (libstdc++) summation order bit for bit (checked against oracle/_ref).
double compute_odometry_regularization(const Correspondences &associations, const SE3 &odometry_initial_guess,
                                       double *sum_sq_out = nullptr) {
    const double sum_of_squared_residuals =
        std::transform_reduce(associations.cbegin(), associations.cend(), 0.0, std::plus<double>(), [&](const auto &association) {
            const auto &[source, target] = association;
            return squared_norm(se3_act(odometry_initial_guess, source) - target);
        });
    const double N = static_cast<double>(associations.size());
    const double mean_squared_residual = sum_of_squared_residuals / N;
    const double beta = 1.0 / (mean_squared_residual + epsilon);
    if (sum_sq_out) *sum_sq_out = sum_of_squared_residuals;
    return beta;
}

// Registration.cpp:62-81.  num_threads==1 -> the reference's serial, deterministic order.
Correspondences data_association(const V3 *points, size_t n, const VoxelMap &voxel_map, const SE3 &T,
                                 double max_correspondance_distance, int num_threads, QueryCounters *counters) {
    Correspondences correspondences;
    correspondences.reserve(n);
#ifdef _OPENMP
    if (num_threads > 1) {
        std::vector<Correspondences> local(num_threads);
        std::vector<QueryCounters> lc(num_threads);
#pragma omp parallel num_threads(num_threads)
        {
            const int tid = omp_get_thread_num();
            auto &mine = local[tid];
            mine.reserve(n / num_threads + 16);
#pragma omp for schedule(static)
            for (long i = 0; i < static_cast<long>(n); ++i) {
                const auto [closest_neighbor, distance] =
                    voxel_map.GetClosestNeighbor(se3_act(T, points[i]), counters ? &lc[tid] : nullptr);
                if (distance < max_correspondance_distance) mine.emplace_back(points[i], closest_neighbor);
            }
        }
        for (int t = 0; t < num_threads; ++t) {
            correspondences.insert(correspondences.end(), local[t].begin(), local[t].end());
            if (counters) {
                counters->probes += lc[t].probes, counters->occupied += lc[t].occupied;
                counters->points_scanned += lc[t].points_scanned;
            }
        }
        return correspondences;
    }
#endif
    (void)num_threads;
    for (size_t i = 0; i < n; ++i) {
        const auto [closest_neighbor, distance] = voxel_map.GetClosestNeighbor(se3_act(T, points[i]), counters);
        if (distance < max_correspondance_distance) correspondences.emplace_back(points[i], closest_neighbor);
    }
    return correspondences;
}

// Registration.cpp:86-93 + :108-113 : one correspondence's (J^T J, J^T r)
inline LinearSystem linear_system_of(const std::pair<V3, V3> &correspondence, const SE3 &current_estimate) {
    const auto &[source, target] = correspondence;
    const V3 residual = se3_act(current_estimate, source) - target;
    const V3 J0 = so3_act(current_estimate.q, V3{1.0, 0.0, 0.0});
    const V3 J1 = so3_act(current_estimate.q, V3{-source.y, source.x, 0.0});
    LinearSystem a;
    a.JTJ[0][0] = dot(J0, J0), a.JTJ[0][1] = dot(J0, J1);
    a.JTJ[1][0] = dot(J1, J0), a.JTJ[1][1] = dot(J1, J1);
    a.JTr[0] = dot(J0, residual), a.JTr[1] = dot(J1, residual);
    return a;
}
// Registration.cpp:95-99
inline LinearSystem sum_linear_systems(LinearSystem a, const LinearSystem &b) {
    a.JTJ[0][0] += b.JTJ[0][0], a.JTJ[0][1] += b.JTJ[0][1], a.JTJ[1][0] += b.JTJ[1][0], a.JTJ[1][1] += b.JTJ[1][1];
    a.JTr[0] += b.JTr[0], a.JTr[1] += b.JTr[1];
    return a;
}
// Registration.cpp:102-118 : the (un-normalised) reduction.  One thread = one tbb::blocked_range covering everything,
// folded by std::transform_reduce (libstdc++: four elements at a time, see compute_odometry_regularization); several
// threads = equal contiguous chunks folded the same way from the identity and joined in chunk order.
LinearSystem reduce_linear_system(const Correspondences &correspondences, const SE3 &current_estimate, int num_threads) {
    const auto fold = [&](size_t lo, size_t hi) {
        return std::transform_reduce(correspondences.cbegin() + static_cast<std::ptrdiff_t>(lo), correspondences.cbegin() + static_cast<std::ptrdiff_t>(hi),
                                     LinearSystem{}, sum_linear_systems,
                                     [&](const auto &correspondence) { return linear_system_of(correspondence, current_estimate); });
    };
    const size_t n = correspondences.size();
#ifdef _OPENMP
    if (num_threads > 1 && n >= 2 * static_cast<size_t>(num_threads)) {
        std::vector<LinearSystem> part(num_threads);
#pragma omp parallel for num_threads(num_threads) schedule(static)
        for (int c = 0; c < num_threads; ++c) part[c] = fold(n * c / num_threads, n * (c + 1) / num_threads);
        LinearSystem sys = part[0];
        for (int c = 1; c < num_threads; ++c) sys = sum_linear_systems(sys, part[c]);
        return sys;
    }
#endif
    (void)num_threads;
    return fold(0, n);
}
// Registration.cpp:119-125 : normalise, regularise, closed-form 2x2 solve (Eigen Matrix2d::inverse())
inline void solve_perturbation(LinearSystem sys, double num_correspondences, double beta, double dx[2]) {
    double (&JTJ)[2][2] = sys.JTJ;
    double (&JTr)[2] = sys.JTr;
    for (auto &row : JTJ)
        for (double &v : row) v /= num_correspondences;
    JTr[0] /= num_correspondences, JTr[1] /= num_correspondences;
    JTJ[0][0] += beta;  // Omega = diag(beta, 0)
    JTJ[1][1] += 0.0;
    // Eigen compute_inverse_size2_helper: invdet = 1/det; inv = [[d,-b],[-c,a]] * invdet
    const double invdet = 1.0 / (JTJ[0][0] * JTJ[1][1] - JTJ[1][0] * JTJ[0][1]);
    const double inv00 = JTJ[1][1] * invdet, inv01 = -JTJ[0][1] * invdet, inv10 = -JTJ[1][0] * invdet, inv11 = JTJ[0][0] * invdet;
    dx[0] = -(inv00 * JTr[0] + inv01 * JTr[1]);
    dx[1] = -(inv10 * JTr[0] + inv11 * JTr[1]);
}
// Registration.cpp:83-126
inline void compute_perturbation(const Correspondences &correspondences, const SE3 &current_estimate, double beta, int num_threads,
                                 double dx[2], LinearSystem *raw = nullptr) {
    const LinearSystem sys = reduce_linear_system(correspondences, current_estimate, num_threads);
    if (raw) *raw = sys;
    solve_perturbation(sys, static_cast<double>(correspondences.size()), beta, dx);
}
// Registration.cpp:159-167
inline SE3 motion_model(const double integrated_controls[2]) {
    double dx[6] = {0, 0, 0, 0, 0, 0};
    const double displacement = integrated_controls[0];
    const double theta = integrated_controls[1];
    dx[0] = displacement * std::sin(theta) / (theta + epsilon);
    dx[1] = displacement * (1.0 - std::cos(theta)) / (theta + epsilon);
    dx[5] = theta;
    return se3_exp(dx);
}

constexpr int kMaxPasses = 64;
}  // namespace

extern "C" {
struct okicp_stats {
    int32_t iterations;    // ComputePerturbation calls executed
    int32_t associations;  // DataAssociation calls executed (incl. the discarded trailing one)
    int32_t converged;     // 1 if the loop left through `break`
    int32_t empty_map;     // 1 if the early-out at Registration.cpp:157 fired
    double beta;
    double n_corr[kMaxPasses];
    double sums[kMaxPasses][6];  // raw JTJ00, JTJ01, JTJ11, JTr0, JTr1 and (pass 0 only) sum ||r||^2
    double dx[kMaxPasses][2];
    uint64_t probes[kMaxPasses], occupied[kMaxPasses], points_scanned[kMaxPasses];
    double seconds;  // wall time of the call
};
}

namespace {
// Registration.cpp:151-190
SE3 compute_robot_motion(const V3 *frame, size_t n, const VoxelMap &voxel_map, const SE3 &last_robot_pose,
                         const SE3 &relative_wheel_odometry, double max_correspondence_distance, int max_num_iterations,
                         double convergence_criterion, int num_threads, bool use_adaptive_odometry_regularization,
                         double fixed_regularization, okicp_stats *st, bool count) {
    SE3 current_estimate = se3_mul(last_robot_pose, relative_wheel_odometry);
    if (voxel_map.Empty()) {
        if (st) st->empty_map = 1;
        return current_estimate;
    }
    int pass = 0;
    auto associate = [&]() {
        QueryCounters qc;
        auto c = data_association(frame, n, voxel_map, current_estimate, max_correspondence_distance, num_threads,
                                  (st && count) ? &qc : nullptr);
        if (st && pass < kMaxPasses) {
            st->n_corr[pass] = static_cast<double>(c.size());
            st->probes[pass] = qc.probes, st->occupied[pass] = qc.occupied, st->points_scanned[pass] = qc.points_scanned;
            st->associations = pass + 1;
        }
        return c;
    };
    auto correspondences = associate();
    double ssq = 0.0;
    const double regularization_term = use_adaptive_odometry_regularization
                                           ? compute_odometry_regularization(correspondences, current_estimate, &ssq)
                                           : fixed_regularization;
    if (st) st->beta = regularization_term, st->sums[0][5] = ssq;
    for (int j = 0; j < max_num_iterations; ++j) {
        double dx[2];
        LinearSystem raw;
        compute_perturbation(correspondences, current_estimate, regularization_term, num_threads, dx, &raw);
        if (st && pass < kMaxPasses) {
            st->sums[pass][0] = raw.JTJ[0][0], st->sums[pass][1] = raw.JTJ[0][1], st->sums[pass][2] = raw.JTJ[1][1];
            st->sums[pass][3] = raw.JTr[0], st->sums[pass][4] = raw.JTr[1];
            st->dx[pass][0] = dx[0], st->dx[pass][1] = dx[1];
            st->iterations = pass + 1;
        }
        const SE3 delta_motion = motion_model(dx);
        current_estimate = se3_mul(current_estimate, delta_motion);
        ++pass;
        if (std::sqrt(dx[0] * dx[0] + dx[1] * dx[1]) < convergence_criterion) {
            if (st) st->converged = 1;
            break;
        }
        correspondences = associate();
    }
    return current_estimate;
}
}  // namespace

// ----------------------------------------------------------------------------------------
// C entry points (ctypes / bench)
// ----------------------------------------------------------------------------------------
extern "C" {
void *okicp_map_create(double voxel_size, double max_distance, unsigned int max_points_per_voxel) {
    return new VoxelMap(voxel_size, max_distance, max_points_per_voxel);
}
void okicp_map_destroy(void *m) { delete static_cast<VoxelMap *>(m); }
void okicp_map_clear(void *m) { static_cast<VoxelMap *>(m)->Clear(); }
int okicp_map_empty(void *m) { return static_cast<VoxelMap *>(m)->Empty() ? 1 : 0; }
void okicp_map_add_points(void *m, const double *xyz, size_t n) {
    static_cast<VoxelMap *>(m)->AddPoints(reinterpret_cast<const V3 *>(xyz), n);
}
void okicp_map_remove_far(void *m, const double origin[3]) {
    static_cast<VoxelMap *>(m)->RemovePointsFarFromLocation({origin[0], origin[1], origin[2]});
}
void okicp_map_update_origin(void *m, const double *xyz, size_t n, const double origin[3]) {
    static_cast<VoxelMap *>(m)->Update(reinterpret_cast<const V3 *>(xyz), n, V3{origin[0], origin[1], origin[2]});
}
void okicp_map_update_pose(void *m, const double *xyz, size_t n, const double pose_qt[7]) {
    static_cast<VoxelMap *>(m)->Update(reinterpret_cast<const V3 *>(xyz), n, se3_from_qt(pose_qt));
}
size_t okicp_map_num_voxels(void *m) { return static_cast<VoxelMap *>(m)->map_.size(); }
size_t okicp_map_num_points(void *m) { return static_cast<VoxelMap *>(m)->NumPoints(); }
size_t okicp_map_pointcloud(void *m, double *out_xyz, size_t cap_points) {
    const auto pts = static_cast<VoxelMap *>(m)->Pointcloud();
    const size_t k = std::min(cap_points, pts.size());
    if (k) std::memcpy(out_xyz, pts.data(), k * sizeof(V3));
    return pts.size();
}
void okicp_map_closest(void *m, const double *queries, size_t n, double *out_nn, double *out_dist) {
    const auto *map = static_cast<VoxelMap *>(m);
    for (size_t i = 0; i < n; ++i) {
        const auto [nn, d] = map->GetClosestNeighbor({queries[3 * i], queries[3 * i + 1], queries[3 * i + 2]});
        out_nn[3 * i] = nn.x, out_nn[3 * i + 1] = nn.y, out_nn[3 * i + 2] = nn.z;
        out_dist[i] = d;
    }
}

// One fused association + accumulation pass at a fixed pose: the per-pass quantity the HIP kernel is
// checked against.  out_sums = {JTJ00, JTJ01, JTJ11, JTr0, JTr1, sum||r||^2, N_corr};
// out_counters = {probes, occupied, points_scanned}.
void okicp_pass(void *m, const double *frame_xyz, size_t n, const double pose_qt[7], double tau, int num_threads,
                double out_sums[7], uint64_t out_counters[3]) {
    const auto *map = static_cast<VoxelMap *>(m);
    const SE3 T = se3_from_qt(pose_qt);
    QueryCounters qc;
    const auto corr = data_association(reinterpret_cast<const V3 *>(frame_xyz), n, *map, T, tau, num_threads, &qc);
    const LinearSystem sys = reduce_linear_system(corr, T, num_threads);
    double ssq = 0.0;
    compute_odometry_regularization(corr, T, &ssq);
    out_sums[0] = sys.JTJ[0][0], out_sums[1] = sys.JTJ[0][1], out_sums[2] = sys.JTJ[1][1];
    out_sums[3] = sys.JTr[0], out_sums[4] = sys.JTr[1], out_sums[5] = ssq, out_sums[6] = static_cast<double>(corr.size());
    if (out_counters) out_counters[0] = qc.probes, out_counters[1] = qc.occupied, out_counters[2] = qc.points_scanned;
}

// DataAssociation per query (Registration.cpp:73-77, serial): accepted[i] = 1 and nn / dist = GetClosestNeighbor(T * frame[i]) when
// `distance < max_correspondance_distance`, else accepted[i] = 0 (nn / dist still what GetClosestNeighbor returned).  The checker of
// kicp_pass_correspondences (tests/test_gpu_correspondences.py).
void okicp_associate(void *m, const double *frame_xyz, size_t n, const double pose_qt[7], double tau, int32_t *accepted, double *out_nn,
                     double *out_dist) {
    const auto *map = static_cast<VoxelMap *>(m);
    const SE3 T = se3_from_qt(pose_qt);
    const V3 *points = reinterpret_cast<const V3 *>(frame_xyz);
    for (size_t i = 0; i < n; ++i) {
        const auto [closest_neighbor, distance] = map->GetClosestNeighbor(se3_act(T, points[i]), nullptr);
        accepted[i] = distance < tau ? 1 : 0;
        out_nn[3 * i] = closest_neighbor.x, out_nn[3 * i + 1] = closest_neighbor.y, out_nn[3 * i + 2] = closest_neighbor.z;
        out_dist[i] = distance;
    }
}

// KinematicRegistration::ComputeRobotMotion.  count_work!=0 also fills the probe/scan counters
// (slower; leave 0 when timing).  Returns 0, or 1 if the result contains NaN (zero correspondences).
int okicp_register(void *m, const double *frame_xyz, size_t n, const double last_pose_qt[7], const double rel_odom_qt[7],
                   double tau, int max_num_iterations, double convergence_criterion, int num_threads,
                   int use_adaptive_odometry_regularization, double fixed_regularization, int count_work, double out_pose_qt[7],
                   okicp_stats *stats) {
    if (stats) std::memset(stats, 0, sizeof(*stats));
#ifdef _OPENMP
    if (num_threads <= 0) num_threads = omp_get_max_threads();  // reference: <=0 -> all cores (Registration.cpp:140-141)
#else
    num_threads = 1;
#endif
    const auto t0 = std::chrono::steady_clock::now();
    const SE3 T = compute_robot_motion(reinterpret_cast<const V3 *>(frame_xyz), n, *static_cast<VoxelMap *>(m),
                                       se3_from_qt(last_pose_qt), se3_from_qt(rel_odom_qt), tau, max_num_iterations,
                                       convergence_criterion, num_threads, use_adaptive_odometry_regularization != 0,
                                       fixed_regularization, stats, count_work != 0);
    const auto t1 = std::chrono::steady_clock::now();
    if (stats) stats->seconds = std::chrono::duration<double>(t1 - t0).count();
    se3_to_qt(T, out_pose_qt);
    for (int i = 0; i < 7; ++i)
        if (std::isnan(out_pose_qt[i])) return 1;
    return 0;
}
int okicp_max_threads() {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

// --- Lie-group helpers, exported so tests can pin them against scipy ---
void okicp_se3_exp(const double xi[6], double out_qt[7]) { se3_to_qt(se3_exp(xi), out_qt); }
void okicp_se3_log(const double qt[7], double out_xi[6]) { se3_log(se3_from_qt(qt), out_xi); }
void okicp_se3_mul(const double a[7], const double b[7], double out[7]) { se3_to_qt(se3_mul(se3_from_qt(a), se3_from_qt(b)), out); }
void okicp_se3_inverse(const double a[7], double out[7]) { se3_to_qt(se3_inverse(se3_from_qt(a)), out); }
void okicp_se3_act(const double a[7], const double *xyz, size_t n, double *out_xyz) {
    const SE3 T = se3_from_qt(a);
    for (size_t i = 0; i < n; ++i) {
        const V3 r = se3_act(T, {xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]});
        out_xyz[3 * i] = r.x, out_xyz[3 * i + 1] = r.y, out_xyz[3 * i + 2] = r.z;
    }
}
void okicp_motion_model(const double controls[2], double out_qt[7]) { se3_to_qt(motion_model(controls), out_qt); }
void okicp_solve(const double sums[5], double n_corr, double beta, double out_dx[2]) {
    LinearSystem s;
    s.JTJ[0][0] = sums[0], s.JTJ[0][1] = s.JTJ[1][0] = sums[1], s.JTJ[1][1] = sums[2], s.JTr[0] = sums[3], s.JTr[1] = sums[4];
    solve_perturbation(s, n_corr, beta, out_dx);
}

// --- correspondence_threshold/CorrespondenceThreshold.{hpp,cpp} ---
struct okicp_threshold {
    double map_discretization_error_, max_range_;
    int use_adaptive_threshold_;
    double fixed_threshold_, odom_sse_, num_samples_;
};
void okicp_threshold_init(okicp_threshold *t, double map_discretization_error, double max_range, int use_adaptive_threshold,
                          double fixed_threshold) {  // CorrespondenceThreshold.cpp:37-47
    *t = {map_discretization_error, max_range, use_adaptive_threshold, fixed_threshold, 0.0, 1e-8};
}
double okicp_threshold_compute(const okicp_threshold *t) {  // CorrespondenceThreshold.cpp:49-56
    if (!t->use_adaptive_threshold_) return t->fixed_threshold_;
    const double sigma_odom = std::sqrt(t->odom_sse_ / t->num_samples_);
    const double sigma_map = t->map_discretization_error_;
    return 3.0 * (sigma_map + sigma_odom);
}
void okicp_threshold_update(okicp_threshold *t, const double odometry_error_qt[7]) {  // CorrespondenceThreshold.cpp:29-34,58-64
    if (!t->use_adaptive_threshold_) return;
    const SE3 pose = se3_from_qt(odometry_error_qt);
    V3 tangent;
    double theta;
    so3_log_and_theta(pose.q, &tangent, &theta);
    const double delta_rot = 2.0 * t->max_range_ * std::sin(theta / 2.0);
    const double delta_trans = norm(pose.t);
    const double e = delta_trans + delta_rot;
    t->odom_sse_ += e * e;
    t->num_samples_ += 1.0;
}
void okicp_threshold_reset(okicp_threshold *t) { t->odom_sse_ = 0.0, t->num_samples_ = 1e-8; }  // CorrespondenceThreshold.hpp:40-43

// --- pre-steps: VoxelDownsample, Preprocess ---
size_t okicp_voxel_downsample(const double *xyz, size_t n, double voxel_size, double *out_xyz) {
    const auto out = voxel_downsample(reinterpret_cast<const V3 *>(xyz), n, voxel_size);
    if (!out.empty()) std::memcpy(out_xyz, out.data(), out.size() * sizeof(V3));
    return out.size();
}
size_t okicp_preprocess(const double *xyz, size_t n, const double *timestamps, size_t n_ts, const double relative_motion_qt[7],
                        double max_range, double min_range, int deskew, double *out_xyz) {
    const auto out = preprocess(reinterpret_cast<const V3 *>(xyz), n, timestamps, n_ts, se3_from_qt(relative_motion_qt), max_range,
                                min_range, deskew != 0);
    if (!out.empty()) std::memcpy(out_xyz, out.data(), out.size() * sizeof(V3));
    return out.size();
}

// --- wire-format ingest ---
// ros/src/kinematic_icp_ros/utils/RosUtils.cpp:30-39 : PointCloud2ToEigen(msg, T) walks a float iterator over field
// "x" (y and z are the next two floats of the record as far as the iterator is concerned: iter[1], iter[2]) and stores
// T * Vector3d.  The byte offsets of y and z are passed in explicitly here; callers with the usual packed x,y,z get the
// iterator's behaviour.  stamp_type uses sensor_msgs PointField codes (6 UINT32, 7 FLOAT32, 8 FLOAT64; 0 = no field).
// Returns -1 for an unsupported stamp type (TimeStampHandler.cpp:103 throws).
int okicp_ingest(const unsigned char *data, size_t n, unsigned point_step, unsigned off_x, unsigned off_y, unsigned off_z, int stamp_type,
                 unsigned off_t, const double *sensor_pose_qt, double *out_xyz, double *out_stamps, double out_minmax[2]) {
    if (stamp_type != 0 && stamp_type != 6 && stamp_type != 7 && stamp_type != 8) return -1;
    const SE3 T = sensor_pose_qt ? se3_from_qt(sensor_pose_qt) : SE3{};
    for (size_t i = 0; i < n; ++i) {
        const unsigned char *rec = data + i * point_step;
        float x, y, z;
        std::memcpy(&x, rec + off_x, 4), std::memcpy(&y, rec + off_y, 4), std::memcpy(&z, rec + off_z, 4);
        const V3 q = se3_act(T, V3{static_cast<double>(x), static_cast<double>(y), static_cast<double>(z)});
        out_xyz[3 * i] = q.x, out_xyz[3 * i + 1] = q.y, out_xyz[3 * i + 2] = q.z;
    }
    out_minmax[0] = out_minmax[1] = 0.0;
    if (stamp_type == 0 || n == 0) return 0;
    // TimeStampHandler.cpp:57-83 : ExtractTimestampsFromMsg
    auto number_of_digits_integer_part = [](double stamp) {
        const uint64_t number_of_seconds = static_cast<uint64_t>(std::round(stamp));
        return number_of_seconds > 0 ? std::floor(std::log10(static_cast<double>(number_of_seconds)) + 1) : 1.0;
    };
    for (size_t i = 0; i < n; ++i) {
        const unsigned char *rec = data + i * point_step + off_t;
        double stampd;
        if (stamp_type == 6) {
            uint32_t v;
            std::memcpy(&v, rec, 4);
            stampd = static_cast<double>(v);
        } else if (stamp_type == 7) {
            float v;
            std::memcpy(&v, rec, 4);
            stampd = static_cast<double>(v);
        } else {
            std::memcpy(&stampd, rec, 8);
        }
        if (number_of_digits_integer_part(stampd) > 10) stampd *= 1e-9;  // nanoseconds -> seconds
        out_stamps[i] = stampd;
    }
    // TimeStampHandler.cpp:106,121-128 : minmax_element + normalisation
    const auto mm = std::minmax_element(out_stamps, out_stamps + n);
    const double lo = *mm.first, hi = *mm.second;
    out_minmax[0] = lo, out_minmax[1] = hi;
    for (size_t i = 0; i < n; ++i) out_stamps[i] = (out_stamps[i] - lo) / (hi - lo);
    return 1;
}
}  // extern "C"
