// tbb/blocked_range.h -- SHIM (test infrastructure, see oracle/ref_shim/README.md): the whole range, never split.
#pragma once
#include <cstddef>
namespace tbb {
template <typename Value>
class blocked_range {
public:
    using const_iterator = Value;
    blocked_range(Value begin, Value end, std::size_t /*grainsize*/ = 1) : begin_(begin), end_(end) {}
    Value begin() const { return begin_; }
    Value end() const { return end_; }
    bool empty() const { return !(begin_ < end_); }

private:
    Value begin_, end_;
};
}  // namespace tbb
