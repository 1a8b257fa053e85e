// Micro-benchmark behind the small-scan path's hand-off (round 3): what does it cost to get the exact sums of N workgroups to
// the host?  (A) every workgroup stores its row (16 self-validating 8-byte words = two 64-byte lines) straight into host-mapped
// memory; (B) rows to device memory (write-through), one ticket atomic per workgroup, the last arriver folds <= 32 rows per
// group and stores ONE row per group to the host (the generic pass kernel's tree); (C) 64-bit atomic adds into 8 replicas in
// device memory + ticket, the last arriver folds the replicas and stores one row.  (0) baseline: only workgroup 0 stores a row.
// Wall clock from the launch call to the last row seen by the host, median of many; the launch cost is common to all.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x)                                                                                   \
    do {                                                                                        \
        hipError_t e_ = (x);                                                                    \
        if (e_ != hipSuccess) {                                                                 \
            std::printf("HIP error %s at line %d: %s\n", #x, __LINE__, hipGetErrorString(e_)); \
            std::exit(1);                                                                       \
        }                                                                                       \
    } while (0)
using Clock = std::chrono::steady_clock;
constexpr int kWords = 16;

struct P {
    unsigned long long *host_rows;  // [N][16]
    unsigned long long *dev_rows;   // [N][16]
    unsigned long long *replicas;   // [8][16]
    unsigned int *tickets;          // [groups * 32]
    unsigned long long tag;
    int mode, spin;
};

__global__ __launch_bounds__(256) void k(const P p) {
    // a little dependent work so that the workgroups do not all arrive in the same cycle
    unsigned long long v = blockIdx.x + 1;
    for (int i = 0; i < p.spin; ++i) v = v * 6364136223846793005ull + 1442695040888963407ull;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (wave != 0) return;
    const unsigned long long word = ((v & 0xFFFFull) << 16) | p.tag;
    if (p.mode == 0) {
        if (blockIdx.x == 0 && lane < kWords) __hip_atomic_store(p.host_rows + lane, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    } else if (p.mode == 1) {
        if (lane < kWords) __hip_atomic_store(p.host_rows + static_cast<size_t>(blockIdx.x) * kWords + lane, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    } else if (p.mode == 2) {
        const unsigned g = blockIdx.x / 32, gsize = min(32u, gridDim.x - g * 32);
        if (lane < kWords) __hip_atomic_store(p.dev_rows + static_cast<size_t>(blockIdx.x) * kWords + lane, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned t = 0;
        if (lane == 0) t = __hip_atomic_fetch_add(p.tickets + g * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        t = __shfl(t, 0, 64);
        if (t != gsize - 1) return;
        if (lane == 0) __hip_atomic_store(p.tickets + g * 32, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned long long sum;
        bool ok;
        do {
            sum = 0, ok = true;
            if (lane < kWords)
                for (unsigned j = 0; j < gsize; ++j) {
                    const unsigned long long w = __hip_atomic_load(p.dev_rows + (static_cast<size_t>(g) * 32 + j) * kWords + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    ok = ok && (w & 0xFFFFull) == p.tag;
                    sum += w >> 16;
                }
        } while (!__all(ok));
        if (lane < kWords) __hip_atomic_store(p.host_rows + static_cast<size_t>(g) * kWords + lane, (sum << 16) | p.tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    } else {
        if (lane < kWords) __hip_atomic_fetch_add(p.replicas + (blockIdx.x % 8) * kWords + lane, v & 0xFFFFull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        unsigned t = 0;
        if (lane == 0) t = __hip_atomic_fetch_add(p.tickets, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        t = __shfl(t, 0, 64);
        if (t != gridDim.x - 1) return;
        if (lane == 0) __hip_atomic_store(p.tickets, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned long long sum = 0;
        if (lane < kWords)
            for (int r = 0; r < 8; ++r) sum += __hip_atomic_exchange(p.replicas + r * kWords + lane, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (lane < kWords) __hip_atomic_store(p.host_rows + lane, (sum << 16) | p.tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

int main() {
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    const int maxN = 1024;
    unsigned long long *rows, *d_rows_host;
    CK(hipHostMalloc(reinterpret_cast<void **>(&rows), maxN * kWords * 8, hipHostMallocMapped | hipHostMallocCoherent));
    std::memset(rows, 0, maxN * kWords * 8);
    CK(hipHostGetDevicePointer(reinterpret_cast<void **>(&d_rows_host), rows, 0));
    P p{};
    p.host_rows = d_rows_host;
    CK(hipMalloc(&p.dev_rows, maxN * kWords * 8));
    CK(hipMalloc(&p.replicas, 8 * kWords * 8));
    CK(hipMalloc(&p.tickets, 64 * 32 * 4));
    CK(hipMemset(p.dev_rows, 0, maxN * kWords * 8));
    CK(hipMemset(p.replicas, 0, 8 * kWords * 8));
    CK(hipMemset(p.tickets, 0, 64 * 32 * 4));
    unsigned long long tag = 0;
    const char *names[4] = {"(0) one row from workgroup 0        ", "(A) every workgroup's row to the host", "(B) tree: groups of 32, rows of groups", "(C) atomics into 8 replicas + ticket "};
    for (int spin : {0, 2000}) {
        std::printf("dependent work per workgroup before the hand-off: %d multiply-adds\n", spin);
        for (int N : {1, 2, 4, 8, 16, 32, 64, 128, 272, 512, 1024}) {
            for (int mode = 0; mode < 4; ++mode) {
                p.mode = mode, p.spin = spin;
                const int rows_expected = mode == 1 ? N : (mode == 2 ? (N + 31) / 32 : 1);
                std::vector<double> us;
                for (int r = 0; r < 400; ++r) {
                    tag = tag % 65535 + 1;
                    p.tag = tag;
                    const auto t0 = Clock::now();
                    hipLaunchKernelGGL(k, dim3(N), dim3(256), 0, st, p);
                    for (int g = 0; g < rows_expected; ++g)
                        for (int i = 0; i < kWords; ++i)
                            while ((__atomic_load_n(rows + static_cast<size_t>(g) * kWords + i, __ATOMIC_RELAXED) & 0xFFFFull) != tag) {
                            }
                    us.push_back(std::chrono::duration<double, std::micro>(Clock::now() - t0).count());
                    CK(hipStreamSynchronize(st));
                }
                std::sort(us.begin() + 50, us.end());
                std::printf("  N %4d %s: median %.2f us (p10 %.2f, p90 %.2f)\n", N, names[mode], us[50 + 175], us[50 + 35], us[50 + 315]);
            }
        }
    }
    return 0;
}
