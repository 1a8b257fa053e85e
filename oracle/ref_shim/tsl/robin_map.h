// tsl/robin_map.h -- SHIM (test infrastructure, see oracle/ref_shim/README.md).
//
// Stand-in for Tessil's tsl::robin_map (the container behind kiss_icp::VoxelHashMap::map_ and VoxelDownsample in
// kiss-icp v1.2.0), written from the published algorithm of robin-map 1.x so that not only look-ups but also the
// ITERATION ORDER (which decides the output order of VoxelDownsample / Pointcloud()) follows the
// real container:
//   * open addressing, power-of-two bucket count (growth factor 2, bucket = hash & mask), default max load factor 0.5:
//     the table doubles when size() >= bucket_count * 0.5 BEFORE an insertion (0 -> 2 -> 4 -> ...);
//   * robin-hood insertion: walk from the ideal bucket; the newcomer takes the place of the first resident that is
//     closer to its own ideal bucket (strictly smaller distance), which then moves on the same way;
//   * rehash = insert every element of the old array, in array order, into the new one;
//   * erase = clear + backward shift until an empty bucket or a resident at distance 0; erase(it) returns the
//     iterator to the same bucket if something shifted into it, else to the next occupied bucket;
//   * reserve(n) = rehash(ceil(n / 0.5)) rounded up to a power of two; clear() keeps the bucket array;
//   * iteration = ascending bucket index.
// Not reproduced: the "distance > 8192 -> grow on next insert" escape (never reached by these workloads), shrinking
// (min load factor is 0 by default), allocator / exception details.  RECALLED, not verifiable offline.
#pragma once
#include <cmath>
#include <cstddef>
#include <functional>
#include <utility>
#include <vector>

namespace tsl {
template <class Key, class T, class Hash = std::hash<Key>, class KeyEqual = std::equal_to<Key>>
class robin_map {
    struct Entry {
        int dist = -1;  // distance from the ideal bucket; -1 = empty
        std::pair<Key, T> kv;
    };

public:
    using value_type = std::pair<Key, T>;
    using size_type = std::size_t;

    template <bool Const>
    class iter {
        friend class robin_map;
        using Map = typename std::conditional<Const, const robin_map, robin_map>::type;
        Map *m_ = nullptr;
        size_type i_ = 0;
        iter(Map *m, size_type i) : m_(m), i_(i) {}

    public:
        iter() = default;
        template <bool C2, typename = typename std::enable_if<Const && !C2>::type>
        iter(const iter<C2> &o) : m_(o.m_), i_(o.i_) {}
        const value_type &operator*() const { return m_->buckets_[i_].kv; }
        const value_type *operator->() const { return &m_->buckets_[i_].kv; }
        const Key &key() const { return m_->buckets_[i_].kv.first; }
        template <bool C = Const, typename = typename std::enable_if<!C>::type>
        T &value() const {
            return m_->buckets_[i_].kv.second;
        }
        template <bool C = Const, typename = typename std::enable_if<C>::type>
        const T &value() const {
            return m_->buckets_[i_].kv.second;
        }
        iter &operator++() {
            ++i_;
            while (i_ < m_->buckets_.size() && m_->buckets_[i_].dist < 0) ++i_;
            return *this;
        }
        bool operator==(const iter &o) const { return i_ == o.i_; }
        bool operator!=(const iter &o) const { return i_ != o.i_; }
        template <bool>
        friend class iter;
    };
    using iterator = iter<false>;
    using const_iterator = iter<true>;

    robin_map() = default;

    size_type size() const { return size_; }
    bool empty() const { return size_ == 0; }
    size_type bucket_count() const { return buckets_.size(); }
    float max_load_factor() const { return 0.5f; }

    iterator begin() { return iterator(this, first_from(0)); }
    iterator end() { return iterator(this, buckets_.size()); }
    const_iterator begin() const { return const_iterator(this, first_from(0)); }
    const_iterator end() const { return const_iterator(this, buckets_.size()); }
    const_iterator cbegin() const { return begin(); }
    const_iterator cend() const { return end(); }

    void clear() {
        for (auto &b : buckets_) b = Entry{};
        size_ = 0;
    }
    void reserve(size_type count) { rehash(static_cast<size_type>(std::ceil(static_cast<float>(count) / max_load_factor()))); }
    void rehash(size_type count) {
        const size_type need = static_cast<size_type>(std::ceil(static_cast<float>(size_) / max_load_factor()));
        rehash_impl(count > need ? count : need);
    }

    iterator find(const Key &k) { return iterator(this, find_index(k)); }
    const_iterator find(const Key &k) const { return const_iterator(this, find_index(k)); }
    bool contains(const Key &k) const { return find_index(k) != buckets_.size(); }
    size_type count(const Key &k) const { return contains(k) ? 1 : 0; }

    std::pair<iterator, bool> insert(const value_type &v) { return insert_impl(value_type(v)); }
    std::pair<iterator, bool> insert(value_type &&v) { return insert_impl(std::move(v)); }
    template <class... Args>
    std::pair<iterator, bool> emplace(Args &&...args) {
        return insert_impl(value_type(std::forward<Args>(args)...));
    }
    T &operator[](const Key &k) { return insert_impl(value_type(k, T())).first.value(); }

    iterator erase(iterator pos) {
        size_type prev = pos.i_;
        buckets_[prev] = Entry{};
        --size_;
        size_type i = next(prev);
        while (buckets_[i].dist > 0) {  // backward shift
            buckets_[prev].kv = std::move(buckets_[i].kv);
            buckets_[prev].dist = buckets_[i].dist - 1;
            buckets_[i] = Entry{};
            prev = i, i = next(i);
        }
        if (buckets_[pos.i_].dist < 0) ++pos;  // nothing moved into the erased bucket: go on to the next occupied one
        return pos;
    }
    size_type erase(const Key &k) {
        const size_type i = find_index(k);
        if (i == buckets_.size()) return 0;
        erase(iterator(this, i));
        return 1;
    }

private:
    size_type mask() const { return buckets_.size() - 1; }
    size_type next(size_type i) const { return (i + 1) & mask(); }
    size_type first_from(size_type i) const {
        while (i < buckets_.size() && buckets_[i].dist < 0) ++i;
        return i;
    }
    size_type load_threshold() const { return static_cast<size_type>(static_cast<float>(buckets_.size()) * max_load_factor()); }
    size_type find_index(const Key &k) const {
        if (buckets_.empty()) return 0;  // == end()
        size_type i = Hash()(k) & mask();
        int dist = 0;
        while (dist <= buckets_[i].dist) {
            if (KeyEqual()(buckets_[i].kv.first, k)) return i;
            i = next(i), ++dist;
        }
        return buckets_.size();
    }
    // place `v` (arriving at bucket i with distance dist, i's resident being richer) robin-hood style
    void displace_insert(size_type i, int dist, value_type &v) {
        for (;;) {
            if (buckets_[i].dist < 0) {
                buckets_[i].kv = std::move(v), buckets_[i].dist = dist;
                return;
            }
            if (dist > buckets_[i].dist) {
                std::swap(v, buckets_[i].kv);
                std::swap(dist, buckets_[i].dist);
            }
            i = next(i), ++dist;
        }
    }
    void rehash_impl(size_type count) {
        size_type cap = 0;
        if (count > 0) {
            cap = 1;
            while (cap < count) cap <<= 1;
        }
        std::vector<Entry> old;
        old.swap(buckets_);
        buckets_.assign(cap, Entry{});
        for (auto &b : old) {
            if (b.dist < 0) continue;
            size_type i = Hash()(b.kv.first) & mask();
            int dist = 0;
            while (dist <= buckets_[i].dist) i = next(i), ++dist;
            displace_insert(i, dist, b.kv);
        }
    }
    std::pair<iterator, bool> insert_impl(value_type &&v) {
        const std::size_t hash = Hash()(v.first);
        size_type i = 0;
        int dist = 0;
        if (!buckets_.empty()) {
            i = hash & mask();
            while (dist <= buckets_[i].dist) {
                if (KeyEqual()(buckets_[i].kv.first, v.first)) return {iterator(this, i), false};
                i = next(i), ++dist;
            }
        }
        if (size_ >= load_threshold()) {  // rehash_on_extreme_load: grow by the policy's factor (0 -> 2 -> 4 ...)
            rehash_impl(buckets_.empty() ? 2 : buckets_.size() * 2);
            i = hash & mask(), dist = 0;
            while (dist <= buckets_[i].dist) i = next(i), ++dist;
        }
        displace_insert(i, dist, v);  // the newcomer itself always lands in bucket i
        ++size_;
        return {iterator(this, i), true};
    }

    std::vector<Entry> buckets_;
    size_type size_ = 0;
};
}  // namespace tsl
