// Pointer-chase latency microbenchmark (one lane, dependent loads) over a random cyclic permutation.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <numeric>
#include <algorithm>
#include <random>
__global__ void chase(const unsigned *next, unsigned start, int steps, unsigned *out, long long *cycles) {
    unsigned i = start;
    long long t0 = clock64();
    for (int s = 0; s < steps; ++s) i = next[i];
    long long t1 = clock64();
    *out = i;
    *cycles = t1 - t0;
}
// many waves each chasing independently (loaded machine)
__global__ void chase_many(const unsigned *next, int steps, unsigned stride, unsigned n, unsigned *out, int group) {
    unsigned i = ((blockIdx.x * blockDim.x + threadIdx.x) / group) * stride % n;
    for (int s = 0; s < steps; ++s) i = next[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = i;
}
int main() {
    for (size_t mb : {1, 8, 32, 64, 256, 1024}) {
        const size_t elem_stride = 32;  // one element per 128-byte line
        const size_t lines = mb * 1024 * 1024 / 128;
        std::vector<unsigned> perm(lines);
        std::iota(perm.begin(), perm.end(), 0u);
        std::mt19937 g(1);
        std::shuffle(perm.begin(), perm.end(), g);
        std::vector<unsigned> next(lines * elem_stride, 0);
        for (size_t k = 0; k < lines; ++k) next[perm[k] * elem_stride] = perm[(k + 1) % lines] * elem_stride;
        unsigned *d_next, *d_out;
        long long *d_cyc;
        hipMalloc(&d_next, next.size() * 4);
        hipMalloc(&d_out, 4 * 1024 * 1024);
        hipMalloc(&d_cyc, 8);
        hipMemcpy(d_next, next.data(), next.size() * 4, hipMemcpyHostToDevice);
        const int steps = 20000;
        hipEvent_t e0, e1;
        hipEventCreate(&e0), hipEventCreate(&e1);
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(chase, dim3(1), dim3(1), 0, 0, d_next, perm[0] * elem_stride, steps, d_out, d_cyc);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
        }
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        long long cyc;
        hipMemcpy(&cyc, d_cyc, 8, hipMemcpyDeviceToHost);
        printf("%5zu MB: single lane %.0f ns/load (%.0f clk) | 2048 waves, ns per dependent step with lanes sharing a chain in groups of", mb, ms * 1e6 / steps, (double)cyc / steps);
        for (int group : {1, 8, 64, 128 * 4}) {
            const int steps2 = 200;
            float ms2 = 0;
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                hipLaunchKernelGGL(chase_many, dim3(1024), dim3(128), 0, 0, d_next, steps2, 97u * (unsigned)elem_stride, (unsigned)(lines * elem_stride), d_out, group);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                hipEventElapsedTime(&ms2, e0, e1);
            }
            printf("  %d: %.0f", group, ms2 * 1e6 / steps2);
        }
        printf("\n");
        hipFree(d_next), hipFree(d_out), hipFree(d_cyc);
    }
    return 0;
}
