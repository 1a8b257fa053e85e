// tbb/parallel_reduce.h -- SHIM (test infrastructure, see oracle/ref_shim/README.md): functional form.
// 1 thread: the body folds the whole range onto the identity (serial, input order).  More threads: equal contiguous
// chunks are folded from the identity in parallel and joined in chunk order (deterministic for a given thread count,
// unlike oneTBB's scheduling-dependent tree).
#pragma once
#include <vector>

#include "shim_threads.h"
namespace tbb {
template <typename Range, typename Value, typename RealBody, typename Reduction>
Value parallel_reduce(const Range &range, const Value &identity, const RealBody &real_body, const Reduction &reduction) {
    const int nt = shim::threads();
    const auto first = range.begin();
    const long long n = static_cast<long long>(range.end() - first);
    if (nt <= 1 || n < 2 * nt) return real_body(range, identity);
    std::vector<Value> part(static_cast<size_t>(nt), identity);
#ifdef _OPENMP
#pragma omp parallel for num_threads(nt) schedule(static)
#endif
    for (int c = 0; c < nt; ++c) part[c] = real_body(Range(first + (n * c) / nt, first + (n * (c + 1)) / nt), identity);
    Value total = part[0];
    for (int c = 1; c < nt; ++c) total = reduction(total, part[c]);
    return total;
}
}  // namespace tbb
