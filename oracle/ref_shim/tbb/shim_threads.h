// tbb/shim_threads.h -- SHIM (test infrastructure, see oracle/ref_shim/README.md): how many threads the stand-in
// parallel_for / parallel_reduce use.  The reference caps oneTBB process-wide through a function-local static
// tbb::global_control (registration/Registration.cpp:147-148: the first-constructed KinematicRegistration wins); the shim
// records the smallest cap ever requested, and oracle/ref_capi.cpp may override it per call
// (tbb::shim::override_threads) so that one process can time the reference's code at 1 and at all cores.  1 thread = the
// reference's default (max_num_threads = 1): strictly serial, input order, deterministic.
#pragma once
#include <atomic>
#ifdef _OPENMP
#include <omp.h>
#endif
namespace tbb {
namespace shim {
inline std::atomic<int> &cap() {
    static std::atomic<int> v{0};  // 0 = no global_control seen
    return v;
}
inline std::atomic<int> &override_threads() {
    static std::atomic<int> v{0};  // 0 = follow global_control
    return v;
}
inline int hardware_threads() {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
inline int threads() {
    const int o = override_threads().load(std::memory_order_relaxed);
    if (o > 0) return o;
    const int c = cap().load(std::memory_order_relaxed);
    return c > 0 ? c : hardware_threads();
}
}  // namespace shim
}  // namespace tbb
