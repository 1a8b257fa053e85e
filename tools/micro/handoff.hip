// Microbenchmarks behind two design decisions of round 2 (built and run on the GPU box by tools/micro/run.sh):
//   A. what a host -> GPU -> host round trip costs (i) as one kernel launch whose last lane writes host-mapped memory
//      (today's per-iteration cost) and (ii) as a command word polled by an already resident kernel, with 1 workgroup and
//      with a full grid (workgroup 0 polls the host word and republishes it in device memory; every workgroup acknowledges
//      with one ticket atomic; the last arriver writes the answer to host-mapped memory);
//   B. how fast 3 MB of scan points travel from caller memory to HBM: CPU copy into pinned memory (1..8 threads), DMA from
//      pinned memory (whole / in 4 pieces), and a kernel reading host-mapped pinned memory directly (zero copy).
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#define CK(x)                                                                                    \
    do {                                                                                         \
        hipError_t e_ = (x);                                                                     \
        if (e_ != hipSuccess) {                                                                  \
            std::printf("HIP error %s at line %d: %s\n", #x, __LINE__, hipGetErrorString(e_));  \
            std::exit(1);                                                                        \
        }                                                                                        \
    } while (0)

using Clock = std::chrono::steady_clock;
static double us_since(Clock::time_point t0) { return std::chrono::duration<double, std::micro>(Clock::now() - t0).count(); }

struct Mailbox {  // host-mapped, one cache line per word
    unsigned long long cmd;
    unsigned long long pad0[15];
    unsigned long long ans;
    unsigned long long pad1[15];
};

__global__ void k_oneshot(Mailbox *mb, unsigned long long v, unsigned int *ticket) {
    // every workgroup takes a ticket; the last one answers (the shape of today's reduction hand-off)
    __shared__ unsigned int t;
    if (threadIdx.x == 0) t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (t == gridDim.x - 1 && threadIdx.x == 0) {
        __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&mb->ans, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// resident kernel: serves commands 1..n_cmds, exits on the last one or when the budget of polls runs out
__global__ void k_resident(Mailbox *mb, unsigned long long *dev_cmd, unsigned int *ticket, unsigned long long n_cmds, unsigned long long max_polls) {
    unsigned long long seen = 0, polls = 0;
    for (;;) {
        unsigned long long c = seen;
        if (blockIdx.x == 0) {
            if (threadIdx.x == 0) {
                while ((c = __hip_atomic_load(&mb->cmd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) == seen && ++polls < max_polls) {
                }
                __hip_atomic_store(dev_cmd, c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // republish for the other workgroups
            }
        } else if (threadIdx.x == 0) {
            while ((c = __hip_atomic_load(dev_cmd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == seen && ++polls < 10 * max_polls) __builtin_amdgcn_s_sleep(1);
        }
        __shared__ unsigned long long s_c;
        if (threadIdx.x == 0) s_c = c;
        __syncthreads();
        c = s_c;
        if (c == seen) return;  // poll budget exhausted
        seen = c;
        __shared__ unsigned int t;
        if (threadIdx.x == 0) t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        if (t == gridDim.x - 1 && threadIdx.x == 0) {
            __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&mb->ans, c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        __syncthreads();
        if (c >= n_cmds) return;
    }
}

__global__ void k_copy24(const double *__restrict__ src, double *__restrict__ dst, unsigned n) {
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[3 * i] = src[3 * i], dst[3 * i + 1] = src[3 * i + 1], dst[3 * i + 2] = src[3 * i + 2];
}
__global__ void k_copy16(const double2 *__restrict__ src, double2 *__restrict__ dst, unsigned n16) {
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n16) dst[i] = src[i];
}

static void wait_ans(Mailbox *mb, unsigned long long v) {
    while (__atomic_load_n(&mb->ans, __ATOMIC_ACQUIRE) != v) {
    }
}

int main() {
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    Mailbox *mb, *d_mb;
    CK(hipHostMalloc(reinterpret_cast<void **>(&mb), sizeof(Mailbox), hipHostMallocMapped | hipHostMallocCoherent));
    std::memset(mb, 0, sizeof(Mailbox));
    CK(hipHostGetDevicePointer(reinterpret_cast<void **>(&d_mb), mb, 0));
    unsigned int *d_ticket;
    unsigned long long *d_cmd;
    CK(hipMalloc(&d_ticket, 256));
    CK(hipMalloc(&d_cmd, 256));
    CK(hipMemset(d_ticket, 0, 256));
    CK(hipMemset(d_cmd, 0, 256));
    const int reps = 2000;
    // ---- A(i): launch per command --------------------------------------------------------------------------------------
    for (int grid : {1, 256, 1024, 2048}) {
        for (int r = 0; r < 200; ++r) {
            hipLaunchKernelGGL(k_oneshot, dim3(grid), dim3(128), 0, st, d_mb, 1000000ull + r, d_ticket);
            wait_ans(mb, 1000000ull + r);
        }
        double launch_us = 0;
        const auto t0 = Clock::now();
        for (int r = 0; r < reps; ++r) {
            const auto t1 = Clock::now();
            hipLaunchKernelGGL(k_oneshot, dim3(grid), dim3(128), 0, st, d_mb, 2000000ull + r, d_ticket);
            launch_us += us_since(t1);
            wait_ans(mb, 2000000ull + r);
        }
        std::printf("A(i)  launch per command, grid %4d x128: round trip %.2f us (of which the launch call itself %.2f us)\n", grid, us_since(t0) / reps,
                    launch_us / reps);
    }
    // ---- A(ii): resident kernel ------------------------------------------------------------------------------------------
    for (int grid : {1, 256, 1024, 1536}) {
        __atomic_store_n(&mb->cmd, 0ull, __ATOMIC_RELEASE);
        __atomic_store_n(&mb->ans, 0ull, __ATOMIC_RELEASE);
        CK(hipMemsetAsync(d_cmd, 0, 8, st));
        CK(hipMemsetAsync(d_ticket, 0, 4, st));
        CK(hipStreamSynchronize(st));
        hipLaunchKernelGGL(k_resident, dim3(grid), dim3(128), 0, st, d_mb, d_cmd, d_ticket, static_cast<unsigned long long>(reps + 200), 3000000ull);
        for (int r = 1; r <= 200; ++r) {
            __atomic_store_n(&mb->cmd, static_cast<unsigned long long>(r), __ATOMIC_RELEASE);
            wait_ans(mb, r);
        }
        const auto t0 = Clock::now();
        for (int r = 201; r <= reps + 200; ++r) {
            __atomic_store_n(&mb->cmd, static_cast<unsigned long long>(r), __ATOMIC_RELEASE);
            wait_ans(mb, r);
        }
        const double rt = us_since(t0) / reps;
        CK(hipStreamSynchronize(st));
        std::printf("A(ii) resident kernel,   grid %4d x128: round trip %.2f us per command\n", grid, rt);
    }
    // ---- B: 3 MB of points from caller memory to HBM ------------------------------------------------------------------------
    const size_t n = 131072, bytes = n * 24;
    std::vector<double> src(n * 3, 1.5);
    unsigned char *pin;
    CK(hipHostMalloc(reinterpret_cast<void **>(&pin), bytes, hipHostMallocMapped));
    double *d_dst, *d_pin;
    CK(hipMalloc(&d_dst, bytes));
    CK(hipHostGetDevicePointer(reinterpret_cast<void **>(&d_pin), pin, 0));
    for (int threads : {1, 2, 4, 8}) {
        const int it = 200;
        const auto t0 = Clock::now();
        for (int r = 0; r < it; ++r) {
            if (threads == 1) {
                std::memcpy(pin, src.data(), bytes);
            } else {
                std::vector<std::thread> th;
                for (int k = 0; k < threads; ++k)
                    th.emplace_back([&, k] { std::memcpy(pin + bytes * k / threads, reinterpret_cast<unsigned char *>(src.data()) + bytes * k / threads, bytes / threads); });
                for (auto &t : th) t.join();
            }
        }
        const double us = us_since(t0) / it;
        std::printf("B  CPU copy of %.2f MB into pinned memory, %d thread(s)%s: %.1f us (%.1f GB/s)\n", bytes / 1e6, threads,
                    threads > 1 ? " (threads created per copy)" : "", us, bytes / us / 1e3);
    }
    for (int pieces : {1, 4}) {
        const int it = 300;
        for (int r = 0; r < 20; ++r) CK(hipMemcpyAsync(d_dst, pin, bytes, hipMemcpyHostToDevice, st));
        CK(hipStreamSynchronize(st));
        const auto t0 = Clock::now();
        for (int r = 0; r < it; ++r) {
            for (int k = 0; k < pieces; ++k)
                CK(hipMemcpyAsync(reinterpret_cast<unsigned char *>(d_dst) + bytes * k / pieces, pin + bytes * k / pieces, bytes / pieces, hipMemcpyHostToDevice, st));
            CK(hipStreamSynchronize(st));
        }
        const double us = us_since(t0) / it;
        std::printf("B  DMA pinned -> HBM, %d piece(s) + stream sync: %.1f us (%.1f GB/s)\n", pieces, us, bytes / us / 1e3);
    }
    {
        const int it = 300;
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0));
        CK(hipEventCreate(&e1));
        for (int variant = 0; variant < 2; ++variant) {
            float ms_total = 0;
            const auto t0 = Clock::now();
            for (int r = 0; r < it; ++r) {
                CK(hipEventRecord(e0, st));
                if (variant == 0)
                    hipLaunchKernelGGL(k_copy24, dim3((n + 255) / 256), dim3(256), 0, st, d_pin, d_dst, static_cast<unsigned>(n));
                else
                    hipLaunchKernelGGL(k_copy16, dim3((bytes / 16 + 255) / 256), dim3(256), 0, st, reinterpret_cast<const double2 *>(d_pin),
                                       reinterpret_cast<double2 *>(d_dst), static_cast<unsigned>(bytes / 16));
                CK(hipEventRecord(e1, st));
                CK(hipEventSynchronize(e1));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                ms_total += ms;
            }
            std::printf("B  kernel reading host-mapped pinned memory (%s): %.1f us by events, %.1f us wall (%.1f GB/s)\n",
                        variant == 0 ? "24 B per lane, 3 x 8-B loads" : "16 B per lane", ms_total * 1e3 / it, us_since(t0) / it, bytes / (ms_total * 1e3 / it) / 1e3);
        }
    }
    return 0;
}
