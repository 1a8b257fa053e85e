// The drop-in kiss_icp::VoxelHashMap's read-only `map_` view (the reference's public tsl::robin_map member,
// pipeline/KinematicICP.hpp:94-95 makes it reachable): iteration, size, find / at / count, refresh after every mutating call.
// Host-only (small insertions stay on the host map): runs without a GPU.
#include <cmath>
#include <cstdio>
#include <kiss_icp/core/VoxelHashMap.hpp>
#include <random>

#define EXPECT(c)                                        \
    do {                                                 \
        if (!(c)) return std::printf("FAILED: %s (line %d)\n", #c, __LINE__), 1; \
    } while (0)

int main() {
    std::mt19937 rng(5);
    std::uniform_real_distribution<double> u(-6.0, 6.0);
    std::vector<Eigen::Vector3d> pts;
    for (int i = 0; i < 900; ++i) pts.emplace_back(u(rng), u(rng), 0.2 * u(rng));
    kiss_icp::VoxelHashMap map(1.0, 100.0, 5);
    EXPECT(map.map_.empty() && map.map_.size() == 0 && map.map_.begin() == map.map_.end());
    map.AddPoints(pts);
    const auto &m = map.map_;
    EXPECT(!m.empty() && m.size() == kicp_map_num_voxels(map.handle()));
    size_t total = 0;
    for (const auto &[voxel, points] : m) {
        EXPECT(!points.empty() && points.size() <= 5);
        for (const auto &p : points)
            EXPECT(static_cast<int>(std::floor(p.x())) == voxel.x() && static_cast<int>(std::floor(p.y())) == voxel.y() && static_cast<int>(std::floor(p.z())) == voxel.z());
        total += points.size();
        EXPECT(m.find(voxel) != m.end() && m.count(voxel) == 1 && m.contains(voxel) && &m.at(voxel) == &m.find(voxel)->second);
    }
    EXPECT(total == map.Pointcloud().size());
    EXPECT(m.find(kiss_icp::Voxel(1000, 0, 0)) == m.end() && m.count(kiss_icp::Voxel(1000, 0, 0)) == 0);
    // the first point offered to a voxel is the first of its bucket (insertion order is kept)
    const kiss_icp::Voxel v0(static_cast<int>(std::floor(pts[0].x())), static_cast<int>(std::floor(pts[0].y())), static_cast<int>(std::floor(pts[0].z())));
    EXPECT(m.at(v0).front().x() == pts[0].x() && m.at(v0).front().y() == pts[0].y());
    // every mutating call refreshes the view
    const size_t before = m.size();
    map.AddPoints({Eigen::Vector3d(50.5, 50.5, 0.5)});
    EXPECT(m.size() == before + 1 && m.contains(kiss_icp::Voxel(50, 50, 0)));
    map.RemovePointsFarFromLocation(Eigen::Vector3d(0.0, 0.0, 0.0));  // max_distance 100: nothing goes
    EXPECT(m.size() == before + 1);
    kiss_icp::VoxelHashMap copy(map);  // a copy has its own view of its own (deep-copied) state
    map.Clear();
    EXPECT(m.empty() && m.size() == 0 && copy.map_.size() == before + 1);
    map = copy;
    EXPECT(map.map_.size() == before + 1);
    std::printf("OK\n");
    return 0;
}
