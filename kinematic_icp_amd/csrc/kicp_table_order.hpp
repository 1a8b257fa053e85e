// kicp_table_order.hpp -- the pieces of the table-order VoxelDownsample (kicp_pre.hpp) that are plain integer code: the
// reference's voxel hash and bucket count, and the per-cluster replay of its robin-hood insertions.  No HIP in here, so
// tests/cpp/downsample_order_test.cpp compiles this very file with g++ and checks it on the CPU against a sequential
// robin-hood table (the kernels that call it are checked on the GPU against the oracle and the reference build).
#pragma once
#include <cmath>
#include <cstddef>
#include <cstdint>
#ifndef KICP_HD
#define KICP_HD inline
#endif

namespace kicp {
constexpr uint32_t kFreeBucket = 0xFFFFFFFFu;
// std::hash<kiss_icp::Voxel> (kiss-icp v1.2.0 core/VoxelUtils.hpp): uint32 wrap-around products, XOR
KICP_HD uint32_t reference_voxel_hash(int32_t x, int32_t y, int32_t z) {
    return (static_cast<uint32_t>(x) * 73856093u) ^ (static_cast<uint32_t>(y) * 19349669u) ^ (static_cast<uint32_t>(z) * 83492791u);
}
KICP_HD uint32_t reference_hash_of_packed(unsigned long long key) {
    const int lim = 1 << 20;
    return reference_voxel_hash(static_cast<int32_t>(key & 0x1FFFFFu) - lim, static_cast<int32_t>((key >> 21) & 0x1FFFFFu) - lim,
                                static_cast<int32_t>((key >> 42) & 0x1FFFFFu) - lim);
}
// tsl::robin_map::reserve(n) = rehash(ceil(float(n) / 0.5f)), rounded up to a power of two (0 stays 0)
inline size_t reference_bucket_count(size_t n) {
    const size_t want = static_cast<size_t>(std::ceil(static_cast<float>(n) / 0.5f));
    size_t buckets = 0;
    if (want > 0) {
        buckets = 1;
        while (buckets < want) buckets <<= 1;
    }
    return buckets;
}

// the same for a count that fits 32 bits, host + device (the chained pre-steps size the table on the device, from a count that
// never visits the host in between: kicp_pre.hpp)
KICP_HD uint32_t reference_bucket_count_u32(uint32_t n) {
    const float want = ceilf(static_cast<float>(n) / 0.5f);
    uint32_t buckets = 0u;
    if (want > 0.f) {
        buckets = 1u;
        while (static_cast<float>(buckets) < want && buckets < 0x80000000u) buckets <<= 1;
    }
    return buckets;
}

// One cluster (slots head .. head+len-1, cyclic) of the claimed table -> the reference's arrangement of the same keys.
// keys / min_index are read-only here; order / home_at are written inside the cluster only.  order[] must be kFreeBucket
// on entry.  Host + device.
// Returns the largest displacement from its ideal bucket any key ended up with (or travelled through): tsl::robin_map gives up
// robin-hood probing and GROWS the table when an insertion's probe length exceeds its limit (128 in robin-map 0.6.x once the load
// factor is >= 0.15, 8192 in 1.x) - a re-hash this replay does not model, so beyond the limit of the container the reference was
// built against the output order may differ from it.  The caller reports the value (kicp_pre_last_max_probe).
KICP_HD uint32_t replay_cluster(const unsigned long long *keys, const uint32_t *min_index, uint32_t *order, uint32_t *home_at, uint32_t mask,
                                uint32_t head, uint32_t len) {
    if (len == 1u) {
        order[head] = min_index[head];
        return 0u;
    }
    uint32_t max_probe = 0u;
    uint32_t last = 0u;  // input index of the previous insertion (+1), so "greater than last" selects the next one
    for (uint32_t t = 0; t < len; ++t) {
        uint32_t best = kFreeBucket, best_slot = head;
        for (uint32_t j = 0; j < len; ++j) {  // the cluster's key with the lowest input index not inserted yet
            const uint32_t s = (head + j) & mask, v = min_index[s];
            if (v >= last && v < best) best = v, best_slot = s;
        }
        last = best + 1u;
        // tsl::robin_map::insert: walk from the ideal bucket; the traveller takes the place of the first resident that is
        // strictly closer to its own ideal bucket, which travels on the same way, until a free bucket
        uint32_t carry = best, carry_home = reference_hash_of_packed(keys[best_slot]) & mask;
        uint32_t pos = carry_home;
        for (;;) {
            const uint32_t resident = order[pos];
            const uint32_t dist = (pos - carry_home) & mask;
            max_probe = dist > max_probe ? dist : max_probe;
            if (resident == kFreeBucket) {
                order[pos] = carry, home_at[pos] = carry_home;
                break;
            }
            const uint32_t resident_home = home_at[pos];
            if (((pos - carry_home) & mask) > ((pos - resident_home) & mask)) {
                order[pos] = carry, home_at[pos] = carry_home;
                carry = resident, carry_home = resident_home;
            }
            pos = (pos + 1u) & mask;
        }
    }
    return max_probe;
}

// The same replay on a WINDOW of the table held in arrays of its own (the kernels' LDS copy of a 256-bucket tile and the buckets behind
// it: kicp_pre.hpp replay_tile): position j of the window is bucket (window_bucket0 + j) & mask; the cluster occupies positions
// first .. end-1 and its head is bucket (window_bucket0 + first) & mask.  key / min_index: the window's copies (read); ord: kFreeBucket
// on entry inside the cluster, the reference's arrangement on return; home: scratch.  Distances need no wrap here - a key's ideal
// bucket lies inside its cluster, so every position involved is a plain offset.  Returns replay_cluster's figure.
KICP_HD uint32_t replay_window(const unsigned long long *key, const uint32_t *min_index, uint32_t *ord, uint32_t *home, uint32_t first, uint32_t end,
                               uint32_t head_bucket, uint32_t mask) {
    uint32_t max_probe = 0u, last = 0u;
    for (uint32_t t = first; t < end; ++t) {
        uint32_t best = kFreeBucket, best_at = first;
        for (uint32_t j = first; j < end; ++j) {  // the cluster's key with the lowest input index not inserted yet
            const uint32_t v = min_index[j];
            if (v >= last && v < best) best = v, best_at = j;
        }
        last = best + 1u;
        uint32_t carry = best, carry_home = first + ((reference_hash_of_packed(key[best_at]) - head_bucket) & mask);
        for (uint32_t pos = carry_home;; ++pos) {
            const uint32_t resident = ord[pos], dist = pos - carry_home;
            max_probe = dist > max_probe ? dist : max_probe;
            if (resident == kFreeBucket) {
                ord[pos] = carry, home[pos] = carry_home;
                break;
            }
            const uint32_t resident_home = home[pos];
            if (pos - carry_home > pos - resident_home) {
                ord[pos] = carry, home[pos] = carry_home;
                carry = resident, carry_home = resident_home;
            }
        }
    }
    return max_probe;
}

}  // namespace kicp
