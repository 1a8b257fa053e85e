// tbb/global_control.h -- SHIM (test infrastructure, see oracle/ref_shim/README.md): max_allowed_parallelism is
// recorded in tbb::shim::cap() (see shim_threads.h); the other parameters are accepted and ignored.
#pragma once
#include <cstddef>

#include "shim_threads.h"
namespace tbb {
class global_control {
public:
    enum parameter { max_allowed_parallelism, thread_stack_size, terminate_on_exception };
    global_control(parameter p, std::size_t value) : param_(p), value_(value) {
        if (p == max_allowed_parallelism) {
            const int v = value > 0 ? static_cast<int>(value) : 1, cur = shim::cap().load();
            if (cur == 0 || v < cur) shim::cap().store(v);  // the smallest live cap applies (oneTBB semantics)
        }
    }
    static std::size_t active_value(parameter) { return static_cast<std::size_t>(shim::threads()); }

private:
    parameter param_;
    std::size_t value_;
};
}  // namespace tbb
