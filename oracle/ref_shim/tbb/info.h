// tbb/info.h -- SHIM (test infrastructure, see oracle/ref_shim/README.md).
#pragma once
#include "shim_threads.h"
namespace tbb {
namespace info {
inline int default_concurrency() { return shim::hardware_threads(); }
}  // namespace info
}  // namespace tbb
