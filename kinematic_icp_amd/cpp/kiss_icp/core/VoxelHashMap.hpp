// kiss_icp/core/VoxelHashMap.hpp -- drop-in for kiss-icp v1.2.0's header of the same path, backed by the
// MI355X library (host map + HBM mirror, either side may hold the newest state).  Same struct name, constructor, methods and public
// configuration fields (reference call sites: registration/Registration.cpp:63,74,157,
// pipeline/KinematicICP.hpp:79,88,92,94-95, pipeline/KinematicICP.cpp:79).
// Copyable and movable like the reference's struct (a copy is a deep copy of the newest state, wherever it lives).
// The public `map_` member (a tsl::robin_map<Voxel, std::vector<Eigen::Vector3d>> in the reference, reachable through
// KinematicICP::VoxelMap(), pipeline/KinematicICP.hpp:94-95) is a READ-ONLY view here: size / empty / iteration / find / at /
// count over (voxel, points) pairs, materialised from the backend on first use after a change (the container itself lives
// behind the C-ABI as a flat table + bucket pools).  Writing through it is not supported and does not
// compile - clear / erase / insert / emplace / operator[] ... end in a static_assert that names the way out: the map is
// changed through AddPoints / Update / RemovePointsFarFromLocation / Clear, as every caller in the reference does.
#pragma once
#include <Eigen/Core>
#include <cmath>
#include <cstdint>
#include <limits>
#include <sophus/se3.hpp>
#include <stdexcept>
#include <tuple>
#include <type_traits>
#include <unordered_map>
#include <utility>
#include <vector>

#include "kicp_bridge.hpp"

namespace kiss_icp {
using Voxel = Eigen::Vector3i;  // kiss-icp v1.2.0 core/VoxelUtils.hpp
struct VoxelHashMap {
    // read-only stand-in for the reference's `tsl::robin_map<Voxel, std::vector<Eigen::Vector3d>> map_`
    class MapView {
    public:
        using key_type = Voxel;
        using mapped_type = std::vector<Eigen::Vector3d>;
        using value_type = std::pair<Voxel, mapped_type>;
        using const_iterator = std::vector<value_type>::const_iterator;
        using iterator = const_iterator;
        size_t size() const { return snapshot().size(); }
        bool empty() const { return owner_->Empty(); }
        const_iterator begin() const { return snapshot().begin(); }
        const_iterator end() const { return snapshot().end(); }
        const_iterator cbegin() const { return begin(); }
        const_iterator cend() const { return end(); }
        const_iterator find(const Voxel &v) const {
            const auto &items = snapshot();
            const auto it = index_.find(Key{v.x(), v.y(), v.z()});
            return it == index_.end() ? items.end() : items.begin() + static_cast<std::ptrdiff_t>(it->second);
        }
        size_t count(const Voxel &v) const { return find(v) == end() ? 0u : 1u; }
        bool contains(const Voxel &v) const { return count(v) != 0u; }
        const mapped_type &at(const Voxel &v) const {
            const auto it = find(v);
            if (it == end()) throw std::out_of_range("VoxelHashMap::map_.at: no such voxel");
            return it->second;
        }
        // Every mutator of the reference's container is a COMPILE-TIME error with the way out in its text (the reference itself
        // never writes `map_` from outside the struct: its writers are AddPoints / Update / RemovePointsFarFromLocation / Clear).
        template <class...>
        struct mutation : std::false_type {};
#define KICP_MAP_VIEW_READ_ONLY(name)                                                                                              \
    template <class... A>                                                                                                          \
    void name(A &&...) {                                                                                                           \
        static_assert(mutation<A...>::value,                                                                                       \
                      "kiss_icp::VoxelHashMap::map_ is a read-only view in this backend (the container lives behind the C-ABI): "  \
                      "change the map through AddPoints / Update / RemovePointsFarFromLocation / Clear");                          \
    }
        KICP_MAP_VIEW_READ_ONLY(clear)
        KICP_MAP_VIEW_READ_ONLY(erase)
        KICP_MAP_VIEW_READ_ONLY(insert)
        KICP_MAP_VIEW_READ_ONLY(insert_or_assign)
        KICP_MAP_VIEW_READ_ONLY(emplace)
        KICP_MAP_VIEW_READ_ONLY(emplace_hint)
        KICP_MAP_VIEW_READ_ONLY(try_emplace)
        KICP_MAP_VIEW_READ_ONLY(reserve)
        KICP_MAP_VIEW_READ_ONLY(rehash)
        KICP_MAP_VIEW_READ_ONLY(swap)
        KICP_MAP_VIEW_READ_ONLY(operator[])
#undef KICP_MAP_VIEW_READ_ONLY

    private:
        friend struct VoxelHashMap;
        explicit MapView(const VoxelHashMap *owner) : owner_(owner) {}
        // the index is keyed on the voxel itself (the reference's std::hash<Voxel>, kiss-icp v1.2.0 core/VoxelUtils.hpp, and
        // component-wise equality): two voxels can share a hash, never an entry
        struct Key {
            int x, y, z;
            bool operator==(const Key &o) const { return x == o.x && y == o.y && z == o.z; }
        };
        struct KeyHash {
            size_t operator()(const Key &k) const {
                return static_cast<size_t>((static_cast<uint32_t>(k.x) * 73856093u) ^ (static_cast<uint32_t>(k.y) * 19349669u) ^ (static_cast<uint32_t>(k.z) * 83492791u));
            }
        };
        // Pointcloud() lists the voxels one after another; a voxel's points are recognised by PointToVoxel (VoxelUtils.hpp)
        const std::vector<value_type> &snapshot() const {
            if (seen_ == owner_->version_) return items_;
            items_.clear(), index_.clear();
            const std::vector<Eigen::Vector3d> points = owner_->Pointcloud();
            const double vs = owner_->voxel_size_;
            for (const auto &p : points) {
                const Voxel v(static_cast<int>(std::floor(p.x() / vs)), static_cast<int>(std::floor(p.y() / vs)), static_cast<int>(std::floor(p.z() / vs)));
                if (items_.empty() || items_.back().first.x() != v.x() || items_.back().first.y() != v.y() || items_.back().first.z() != v.z()) {
                    index_[Key{v.x(), v.y(), v.z()}] = items_.size();
                    items_.emplace_back(v, mapped_type{});
                }
                items_.back().second.push_back(p);
            }
            seen_ = owner_->version_;
            return items_;
        }
        const VoxelHashMap *owner_;
        mutable std::vector<value_type> items_;
        mutable std::unordered_map<Key, size_t, KeyHash> index_;
        mutable uint64_t seen_ = ~uint64_t(0);
    };

    explicit VoxelHashMap(double voxel_size, double max_distance, unsigned int max_points_per_voxel)
        : voxel_size_(voxel_size), max_distance_(max_distance), max_points_per_voxel_(max_points_per_voxel), map_(this) {
        kicp_bridge::check(kicp_map_create(voxel_size, max_distance, max_points_per_voxel, &handle_), "VoxelHashMap");
        kicp_map_set_device(handle_, device_);  // bulk AddPoints / Update calls from host vectors insert on the GPU
    }
    ~VoxelHashMap() { kicp_map_destroy(handle_); }
    VoxelHashMap(const VoxelHashMap &o)
        : voxel_size_(o.voxel_size_), max_distance_(o.max_distance_), max_points_per_voxel_(o.max_points_per_voxel_), map_(this), device_(o.device_) {
        kicp_bridge::check(kicp_map_clone(o.handle_, &handle_), "VoxelHashMap(const VoxelHashMap&)");
    }
    VoxelHashMap(VoxelHashMap &&o) noexcept
        : voxel_size_(o.voxel_size_), max_distance_(o.max_distance_), max_points_per_voxel_(o.max_points_per_voxel_), map_(this), device_(o.device_),
          handle_(o.handle_) {
        o.handle_ = nullptr;
    }
    VoxelHashMap &operator=(VoxelHashMap o) noexcept {  // copy / move and swap
        std::swap(voxel_size_, o.voxel_size_), std::swap(max_distance_, o.max_distance_), std::swap(max_points_per_voxel_, o.max_points_per_voxel_);
        std::swap(device_, o.device_), std::swap(handle_, o.handle_);
        ++version_;
        return *this;
    }

    inline void Clear() { kicp_map_clear(handle_), ++version_; }
    inline bool Empty() const { return kicp_map_empty(handle_) != 0; }
    void Update(const std::vector<Eigen::Vector3d> &points, const Eigen::Vector3d &origin) {
        ++version_;
        kicp_bridge::check(kicp_map_update_origin(handle_, kicp_bridge::xyz(points), points.size(), origin.data()), "VoxelHashMap::Update");
    }
    void Update(const std::vector<Eigen::Vector3d> &points, const Sophus::SE3d &pose) {
        double p[7];
        kicp_bridge::to_params(pose, p);
        ++version_;
        kicp_bridge::check(kicp_map_update_pose(handle_, kicp_bridge::xyz(points), points.size(), p), "VoxelHashMap::Update");
    }
    // Update(points, pose) with the points already in HBM on device_ (e.g. a kicp_pre buffer); backend extension
    void UpdateDevice(const double *d_points_xyz, size_t n, const Sophus::SE3d &pose) {
        double p[7];
        kicp_bridge::to_params(pose, p);
        ++version_;
        kicp_bridge::check(kicp_map_update_pose_device(handle_, device_, d_points_xyz, n, p), "VoxelHashMap::Update");
    }
    // the same in two halves: Begin queues the update and returns without waiting, Finish collects it (kicp.h); backend extension
    void UpdateDeviceBegin(const double *d_points_xyz, size_t n, const Sophus::SE3d &pose) {
        double p[7];
        kicp_bridge::to_params(pose, p);
        ++version_;
        kicp_bridge::check(kicp_map_update_pose_device_begin(handle_, device_, d_points_xyz, n, p), "VoxelHashMap::Update");
    }
    void UpdateFinish() { kicp_bridge::check(kicp_map_update_finish(handle_), "VoxelHashMap::Update"); }
    void AddPoints(const std::vector<Eigen::Vector3d> &points) {
        ++version_;
        kicp_bridge::check(kicp_map_add_points(handle_, kicp_bridge::xyz(points), points.size()), "VoxelHashMap::AddPoints");
    }
    void RemovePointsFarFromLocation(const Eigen::Vector3d &origin) { kicp_map_remove_far(handle_, origin.data()), ++version_; }
    std::vector<Eigen::Vector3d> Pointcloud() const {
        std::vector<Eigen::Vector3d> points(kicp_map_num_points(handle_));
        if (!points.empty()) kicp_map_pointcloud(handle_, points.front().data(), points.size());
        return points;
    }
    // One query -> (closest point, distance); (0, DBL_MAX) when the 27 voxels hold nothing.  Runs the device search.
    std::tuple<Eigen::Vector3d, double> GetClosestNeighbor(const Eigen::Vector3d &query) const {
        Eigen::Vector3d nn;
        double d = std::numeric_limits<double>::max();
        kicp_bridge::check(kicp_map_closest(handle_, device_, query.data(), 1, nn.data(), &d), "VoxelHashMap::GetClosestNeighbor");
        return std::make_tuple(nn, d);
    }

    double voxel_size_;
    double max_distance_;
    unsigned int max_points_per_voxel_;
    MapView map_;  // read-only view (see the header comment)

    // backend access (not part of the reference API)
    kicp_map *handle() const { return handle_; }
    int device_ = kicp_bridge::default_device();  // KICP_DEVICE

private:
    kicp_map *handle_ = nullptr;
    uint64_t version_ = 0;  // bumped by every call that may change the map: map_'s snapshot is rebuilt on its next use
};
}  // namespace kiss_icp
