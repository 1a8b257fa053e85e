// tbb/task_arena.h -- SHIM (test infrastructure, see oracle/ref_shim/README.md).
#pragma once
#include "shim_threads.h"
namespace tbb {
namespace this_task_arena {
inline int max_concurrency() { return shim::hardware_threads(); }
}  // namespace this_task_arena
}  // namespace tbb
