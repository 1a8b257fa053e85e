// tbb/parallel_for.h -- SHIM (test infrastructure, see oracle/ref_shim/README.md).
// 1 thread (the reference's default max_num_threads = 1): ONE call of the body over the whole range, i.e. strictly
// serial in input order.  More threads: the range is cut into equal contiguous chunks, one OpenMP thread each.
#pragma once
#include "shim_threads.h"
namespace tbb {
template <typename Range, typename Body>
void parallel_for(const Range &range, const Body &body) {
    const int nt = shim::threads();
    const auto first = range.begin();
    const long long n = static_cast<long long>(range.end() - first);
    if (nt <= 1 || n < 2 * nt) {
        body(range);
        return;
    }
#ifdef _OPENMP
#pragma omp parallel for num_threads(nt) schedule(static)
#endif
    for (int c = 0; c < nt; ++c) body(Range(first + (n * c) / nt, first + (n * (c + 1)) / nt));
}
}  // namespace tbb
