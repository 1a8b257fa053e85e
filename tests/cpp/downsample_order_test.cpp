// CPU check of kinematic_icp_amd/csrc/kicp_table_order.hpp (the integer core of the table-order VoxelDownsample kernels):
// linear-probing claim in ARBITRARY order + per-cluster replay  ==  sequential tsl::robin_map-style insertion in input
// order, bucket for bucket.  The sequential table below is written independently (robin-hood insertion as published:
// swap with the first resident that is strictly closer to home), the claim mimics the concurrent k_downsample_claim by
// processing the points in a shuffled order.  Prints "ok <cases>" or the first mismatch.
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "../../kinematic_icp_amd/csrc/kicp_table_order.hpp"

using kicp::kFreeBucket;
namespace {
struct Vox {
    int32_t x, y, z;
    bool operator==(const Vox &o) const { return x == o.x && y == o.y && z == o.z; }
};
unsigned long long pack(const Vox &v) {
    const int lim = 1 << 20;
    return (static_cast<unsigned long long>(static_cast<uint32_t>(v.z + lim) & 0x1FFFFFu) << 42) |
           (static_cast<unsigned long long>(static_cast<uint32_t>(v.y + lim) & 0x1FFFFFu) << 21) |
           static_cast<unsigned long long>(static_cast<uint32_t>(v.x + lim) & 0x1FFFFFu);
}
// the sequential container: returns, per bucket, the input index of the stored point (kFreeBucket = empty)
std::vector<uint32_t> sequential_robin_hood(const std::vector<Vox> &pts, size_t buckets, int *max_probe = nullptr) {
    std::vector<uint32_t> who(buckets, kFreeBucket);
    std::vector<int> dist(buckets, -1);
    const size_t mask = buckets - 1;
    for (uint32_t i = 0; i < pts.size(); ++i) {
        size_t b = kicp::reference_voxel_hash(pts[i].x, pts[i].y, pts[i].z) & mask;
        int d = 0;
        bool present = false;
        while (d <= dist[b]) {
            if (pts[who[b]] == pts[i]) {
                present = true;
                break;
            }
            b = (b + 1) & mask, ++d;
        }
        if (present) continue;
        uint32_t carry = i;
        while (dist[b] >= 0) {
            if (d > dist[b]) std::swap(carry, who[b]), std::swap(d, dist[b]);
            b = (b + 1) & mask, ++d;
        }
        who[b] = carry, dist[b] = d;
        if (max_probe) *max_probe = std::max(*max_probe, d);  // (the longest probe any insertion ended with: what robin_map's growth rule looks at)
    }
    return who;
}
// what the kernels do: claim (any order) -> replay per cluster
std::vector<uint32_t> claimed_and_replayed(const std::vector<Vox> &pts, size_t buckets, std::mt19937 &rng, uint32_t *max_probe = nullptr) {
    const uint32_t mask = static_cast<uint32_t>(buckets - 1);
    std::vector<unsigned long long> keys(buckets, ~0ull);
    std::vector<uint32_t> min_index(buckets, 0xFFFFFFFFu), order(buckets, kFreeBucket), home_at(buckets, 0xDEADBEEFu);
    std::vector<uint32_t> perm(pts.size());
    for (uint32_t i = 0; i < perm.size(); ++i) perm[i] = i;
    std::shuffle(perm.begin(), perm.end(), rng);
    for (uint32_t i : perm) {
        const unsigned long long key = pack(pts[i]);
        uint32_t slot = kicp::reference_voxel_hash(pts[i].x, pts[i].y, pts[i].z) & mask;
        while (keys[slot] != ~0ull && keys[slot] != key) slot = (slot + 1) & mask;
        keys[slot] = key;
        min_index[slot] = std::min(min_index[slot], i);
    }
    for (uint32_t s = 0; s <= mask; ++s) {
        if (keys[s] == ~0ull || keys[(s - 1u) & mask] != ~0ull) continue;
        uint32_t len = 1;
        while (keys[(s + len) & mask] != ~0ull) ++len;
        const uint32_t probe = kicp::replay_cluster(keys.data(), min_index.data(), order.data(), home_at.data(), mask, s, len);
        if (max_probe) *max_probe = std::max(*max_probe, probe);
        // the kernels' LDS form of the same replay (replay_window): the cluster copied into a window that starts `lead` buckets before
        // its head must come out in the same order with the same probe figure
        const uint32_t lead = s % 7u;
        std::vector<unsigned long long> wkey(lead + len, ~0ull);
        std::vector<uint32_t> wmin(lead + len, 0xFFFFFFFFu), word(lead + len, kFreeBucket), whome(lead + len, 0xDEADBEEFu);
        for (uint32_t j = 0; j < len; ++j) wkey[lead + j] = keys[(s + j) & mask], wmin[lead + j] = min_index[(s + j) & mask];
        const uint32_t wprobe = kicp::replay_window(wkey.data(), wmin.data(), word.data(), whome.data(), lead, lead + len, s, mask);
        for (uint32_t j = 0; j < len; ++j)
            if (word[lead + j] != order[(s + j) & mask] || wprobe != probe) {
                std::printf("WINDOW replay differs: cluster at %u len %u position %u: %u vs %u (probe %u vs %u)\n", s, len, j, word[lead + j], order[(s + j) & mask], wprobe, probe);
                std::exit(1);
            }
    }
    return order;
}
int check(const std::vector<Vox> &pts, std::mt19937 &rng, const char *what) {
    const size_t buckets = kicp::reference_bucket_count(pts.size());
    if (buckets == 0) return 0;
    int seq_probe = 0;
    uint32_t replay_probe = 0;
    const auto a = sequential_robin_hood(pts, buckets, &seq_probe), b = claimed_and_replayed(pts, buckets, rng, &replay_probe);
    // the replay reports the longest probe it walked: never below the container's own figure (it is what kicp_pre_last_max_probe hands out)
    if (static_cast<int>(replay_probe) < seq_probe) {
        std::printf("PROBE %s: n %zu sequential %d replayed %u\n", what, pts.size(), seq_probe, replay_probe);
        return 1;
    }
    for (size_t s = 0; s < buckets; ++s)
        if (a[s] != b[s]) {
            std::printf("MISMATCH %s: n %zu buckets %zu bucket %zu sequential %u replayed %u\n", what, pts.size(), buckets, s, a[s], b[s]);
            return 1;
        }
    return 0;
}
}  // namespace

int main() {
    std::mt19937 rng(12345);
    int cases = 0;
    // bucket counts follow robin_map::reserve
    const size_t expect[][2] = {{0, 0}, {1, 2}, {2, 4}, {3, 8}, {4, 8}, {5, 16}, {1000, 2048}, {1024, 2048}, {1025, 4096}, {131072, 262144}};
    for (auto &e : expect)
        if (kicp::reference_bucket_count(e[0]) != e[1]) return std::printf("bucket count of %zu: %zu\n", e[0], kicp::reference_bucket_count(e[0])), 1;
    // 1. random clouds: all distinct (load 0.5 at n = power of two), heavy duplication, tiny tables
    for (int n : {1, 2, 3, 4, 5, 7, 8, 16, 33, 64, 100, 256, 1000, 4096, 20000})
        for (int span : {1, 2, 4, 16, 200}) {
            std::vector<Vox> pts(n);
            std::uniform_int_distribution<int> u(-span, span);
            for (auto &p : pts) p = {u(rng), u(rng), u(rng) / 4};
            if (check(pts, rng, "random")) return 1;
            ++cases;
        }
    // 2. adversarial: many distinct voxels with the SAME ideal bucket, and with ideal buckets at the end of the table (the cluster
    //    wraps around to bucket 0), mixed with background points and duplicates
    for (int n : {64, 256, 1024})
        for (uint32_t target : {0u, 1u, 7u, 0xFFFFFFFFu, 0xFFFFFFFEu}) {
            const uint32_t mask = static_cast<uint32_t>(kicp::reference_bucket_count(n) - 1);
            std::vector<Vox> pts;
            std::uniform_int_distribution<int> u(-300, 300);
            while (static_cast<int>(pts.size()) < n / 3) {  // same home
                const Vox v{u(rng), u(rng), u(rng)};
                if ((kicp::reference_voxel_hash(v.x, v.y, v.z) & mask) == (target & mask)) pts.push_back(v);
            }
            while (static_cast<int>(pts.size()) < n / 2) {  // homes just before the target: they push the group along
                const Vox v{u(rng), u(rng), u(rng)};
                const uint32_t h = kicp::reference_voxel_hash(v.x, v.y, v.z) & mask;
                if (((target - h) & mask) <= 6) pts.push_back(v);
            }
            while (static_cast<int>(pts.size()) < n) pts.push_back(rng() % 3 ? Vox{u(rng), u(rng), u(rng)} : pts[rng() % pts.size()]);
            std::shuffle(pts.begin(), pts.end(), rng);
            if (check(pts, rng, "adversarial")) return 1;
            ++cases;
        }
    // 3. structured grids (what a planar scan looks like after voxelisation)
    for (int w : {10, 40, 90}) {
        std::vector<Vox> pts;
        for (int rep = 0; rep < 2; ++rep)
            for (int x = -w; x < w; ++x)
                for (int y = -w; y < w; ++y) pts.push_back({x, y, (x * y) % 3 == 0 ? 1 : 0});
        std::shuffle(pts.begin(), pts.end(), rng);
        if (check(pts, rng, "grid")) return 1;
        ++cases;
    }
    std::printf("ok %d\n", cases);
    return 0;
}
