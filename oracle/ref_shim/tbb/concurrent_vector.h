// tbb/concurrent_vector.h -- SHIM (test infrastructure, see oracle/ref_shim/README.md).
// Growth-only vector whose emplace_back may be called from several threads at once: a slot index is claimed with an
// atomic counter inside the reserved storage (the reference reserves points.size() before filling,
// registration/Registration.cpp:68).  Appending beyond the reservation re-allocates and is only safe while no other
// thread is appending (it never happens on the reference's path).
// With 1 thread the element order is the input order (the reference's deterministic default); with more threads it is
// scheduling dependent, exactly as with oneTBB.
#pragma once
#include <atomic>
#include <cstddef>
#include <memory>
#include <mutex>
#include <new>
#include <utility>
namespace tbb {
template <typename T>
class concurrent_vector {
public:
    using value_type = T;
    using const_iterator = const T *;
    using iterator = T *;
    concurrent_vector() = default;
    concurrent_vector(concurrent_vector &&o) noexcept { steal(o); }
    concurrent_vector &operator=(concurrent_vector &&o) noexcept {
        if (this != &o) release(), steal(o);
        return *this;
    }
    concurrent_vector(const concurrent_vector &) = delete;
    concurrent_vector &operator=(const concurrent_vector &) = delete;
    ~concurrent_vector() { release(); }

    void reserve(std::size_t n) {
        if (n > cap_) regrow(n);
    }
    template <class... Args>
    iterator emplace_back(Args &&...args) {
        std::size_t i = size_.fetch_add(1, std::memory_order_relaxed);
        if (i >= cap_) {  // beyond the reservation (never on the reference's path; single-threaded use only)
            std::lock_guard<std::mutex> g(grow_);
            if (i >= cap_) regrow(cap_ ? 2 * cap_ : 16);
        }
        return new (data_ + i) T(std::forward<Args>(args)...);
    }
    std::size_t size() const { return size_.load(std::memory_order_acquire); }
    bool empty() const { return size() == 0; }
    const_iterator begin() const { return data_; }
    const_iterator end() const { return data_ + size(); }
    const_iterator cbegin() const { return data_; }
    const_iterator cend() const { return data_ + size(); }
    const T &operator[](std::size_t i) const { return data_[i]; }

private:
    void regrow(std::size_t n) {
        T *fresh = static_cast<T *>(::operator new(n * sizeof(T)));
        const std::size_t have = size_.load() < cap_ ? size_.load() : cap_;
        for (std::size_t i = 0; i < have; ++i) new (fresh + i) T(std::move(data_[i])), data_[i].~T();
        ::operator delete(data_);
        data_ = fresh, cap_ = n;
    }
    void release() {
        const std::size_t n = size_.load() < cap_ ? size_.load() : cap_;
        for (std::size_t i = 0; i < n; ++i) data_[i].~T();
        ::operator delete(data_);
        data_ = nullptr, cap_ = 0, size_.store(0);
    }
    void steal(concurrent_vector &o) {
        data_ = o.data_, cap_ = o.cap_, size_.store(o.size_.load());
        o.data_ = nullptr, o.cap_ = 0, o.size_.store(0);
    }
    T *data_ = nullptr;
    std::size_t cap_ = 0;
    std::atomic<std::size_t> size_{0};
    std::mutex grow_;
};
}  // namespace tbb
