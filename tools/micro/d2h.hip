// Microbenchmark behind the frame download of round 6: how fast do 3 MB of preprocessed points (the frame RegisterFrame returns,
// pipeline/KinematicICP.cpp:84) get from HBM into a caller's pageable std::vector?
//   A. the DMA engine: hipMemcpyAsync device -> pinned, whole and in 3 / 6 pieces
//   B. a kernel that PUSHES the bytes into host-mapped pinned memory (16 B per lane), by workgroup count
//   C. the CPU's copy pinned -> pageable (a fresh vector, a touched vector; 1, 2, 3 threads)
// Build + run on the GPU box: hipcc --offload-arch=gfx950 -O3 -o tools/micro/d2h tools/micro/d2h.hip -lpthread && tools/micro/d2h
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
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
static double us_since(Clock::time_point t0) { return std::chrono::duration<double, std::micro>(Clock::now() - t0).count(); }
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void k_push(const u32x4 *__restrict__ src, u32x4 *__restrict__ dst, size_t n16) {
    for (size_t i = static_cast<size_t>(blockIdx.x) * 256 + threadIdx.x; i < n16; i += static_cast<size_t>(gridDim.x) * 256) dst[i] = src[i];
}

static double median(std::vector<double> v) {
    std::sort(v.begin(), v.end());
    return v[v.size() / 2];
}

int main() {
    const size_t bytes = 131072ull * 24;
    unsigned char *d = nullptr, *h = nullptr, *hdev = nullptr;
    CK(hipMalloc(&d, bytes));
    CK(hipMemset(d, 7, bytes));
    CK(hipHostMalloc(reinterpret_cast<void **>(&h), bytes, hipHostMallocDefault));
    CK(hipHostGetDevicePointer(reinterpret_cast<void **>(&hdev), h, 0));
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    for (int pieces : {1, 3, 6, 12}) {
        std::vector<double> t;
        for (int rep = 0; rep < 30; ++rep) {
            const auto t0 = Clock::now();
            const size_t piece = (bytes / pieces + 4095) / 4096 * 4096;
            for (size_t off = 0; off < bytes; off += piece) CK(hipMemcpyAsync(h + off, d + off, std::min(piece, bytes - off), hipMemcpyDeviceToHost, s));
            CK(hipStreamSynchronize(s));
            t.push_back(us_since(t0));
        }
        std::printf("A. DMA device -> pinned, %2d piece(s): median %.1f us (%.1f GB/s)\n", pieces, median(t), bytes / median(t) * 1e-3);
    }
    for (int wgs : {16, 32, 64, 128, 256, 512}) {
        std::vector<double> t;
        for (int rep = 0; rep < 30; ++rep) {
            const auto t0 = Clock::now();
            hipLaunchKernelGGL(k_push, dim3(wgs), dim3(256), 0, s, reinterpret_cast<const u32x4 *>(d), reinterpret_cast<u32x4 *>(hdev), bytes / 16);
            CK(hipStreamSynchronize(s));
            t.push_back(us_since(t0));
        }
        std::printf("B. push kernel -> host-mapped pinned, %3d workgroups: median %.1f us (%.1f GB/s)\n", wgs, median(t), bytes / median(t) * 1e-3);
    }
    for (int threads : {1, 2, 3, 4}) {
        for (int fresh = 0; fresh < 2; ++fresh) {
            std::vector<double> t;
            std::vector<unsigned char> keep(bytes, 1);
            for (int rep = 0; rep < 20; ++rep) {
                std::vector<unsigned char> fresh_vec;
                if (fresh) fresh_vec.assign(bytes, 0);  // (what the drop-in does: a value-initialised std::vector per frame)
                unsigned char *dst = fresh ? fresh_vec.data() : keep.data();
                const auto t0 = Clock::now();
                std::vector<std::thread> th;
                const size_t share = bytes / threads;
                for (int k = 1; k < threads; ++k) th.emplace_back([=] { std::memcpy(dst + k * share, h + k * share, k == threads - 1 ? bytes - k * share : share); });
                std::memcpy(dst, h, threads == 1 ? bytes : share);
                for (auto &x : th) x.join();
                t.push_back(us_since(t0));
            }
            std::printf("C. CPU copy pinned -> %s vector, %d thread(s) (incl. thread start): median %.1f us (%.1f GB/s)\n", fresh ? "just-allocated" : "long-lived", threads,
                        median(t), bytes / median(t) * 1e-3);
        }
    }
    {
        std::vector<double> t;
        for (int rep = 0; rep < 20; ++rep) {
            const auto t0 = Clock::now();
            std::vector<unsigned char> v(bytes);
            t.push_back(us_since(t0));
            if (v[rep] != 0) std::printf("?");
        }
        std::printf("D. std::vector<unsigned char>(3 MB) value-initialised: median %.1f us\n", median(t));
    }
    return 0;
}
