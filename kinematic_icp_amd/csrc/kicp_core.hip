// kicp_core.hip -- shared host-side plumbing of libkicp_amd.so: error state, tracing switch, the staging policy for caller
// memory, version / device queries and the raw device-memory helpers of include/kicp.h.
#include <dlfcn.h>

#include <sched.h>
#include <sys/syscall.h>
#include <unistd.h>

#include <cctype>

#include "kicp_internal.hpp"

namespace kicp {
namespace host {

// (plain function on purpose: hipcc gave two namespace-scope initialiser lambdas of this shape the same closure symbol and
// ran the first one's body for both)
bool env_flag(const char *name) {
    const char *e = std::getenv(name);
    return e && *e && *e != '0';
}
const bool g_trace = env_flag("KICP_TRACE");
thread_local std::chrono::steady_clock::time_point g_trace_t0;

// roctx ranges (SURVEY.md section 5 "tracing"): libroctx64 is bound with dlopen the first time a range is opened, so that nothing
// links against the profiler's library; a process without it simply gets no ranges
namespace {
struct Roctx {
    int (*push)(const char *) = nullptr;
    int (*pop)() = nullptr;
    Roctx() {
        // (rocprofv3 listens to the rocprofiler-sdk flavour of the library; the roctracer one is what older tools hook)
        for (const char *nm : {"librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so", "/opt/rocm/lib/librocprofiler-sdk-roctx.so.1", "libroctx64.so.4", "libroctx64.so"}) {
            if (void *h = dlopen(nm, RTLD_NOW | RTLD_GLOBAL)) {
                push = reinterpret_cast<int (*)(const char *)>(dlsym(h, "roctxRangePushA"));
                pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
                if (push && pop) return;
            }
        }
        push = nullptr, pop = nullptr;
    }
};
Roctx &roctx() {
    static Roctx r;
    return r;
}
}  // namespace
const bool g_roctx = env_flag("KICP_ROCTX");
void roctx_push(const char *name) {
    if (roctx().push) roctx().push(name);
}
void roctx_pop() {
    if (roctx().pop) roctx().pop();
}

std::string &last_error() {
    thread_local std::string g_error;
    return g_error;
}
int fail(int code, const std::string &msg) {
    last_error() = msg;
    return code;
}

// Transfers from / to caller memory never hand the caller's pointer to the HIP runtime.  The runtime pins a pageable
// buffer for the DMA and remembers pinned ranges by address; with buffers that live at new or recycled addresses every
// frame (a new message, a new std::vector) a 4 MB scan took 17-27 ms to upload instead of 0.16 ms in most processes we
// measured (always in multiples of ~9 ms, and whether a process was hit depended on its allocation pattern only).  So
// both directions go through a pinned staging buffer owned by the handle: CPU copy in 1 MB pieces (~0.03 ms each), each
// followed by its asynchronous DMA - about 0.2 ms for that scan, every time.  KICP_DIRECT_UPLOAD=1 restores the direct
// DMA for callers that pass pinned (hipHostMalloc / hipHostRegister) memory.  An event recorded behind the last copy of an
// upload guards the buffer: the next transfer through it waits for that event first, so an entry point that returns early
// (error, max_num_iterations <= 0) cannot have its copy overtaken by the next call's CPU writes.
// ---- NUMA locality of the GPU (kicp_internal.hpp) ---------------------------------------------------------------------------------
namespace {
struct GpuLocality {
    int node = -1;
    cpu_set_t cpus;
    bool have_cpus = false;
    std::string cpulist;
};
const bool g_numa = [] { const char *e = std::getenv("KICP_NUMA"); return !(e && *e == '0'); }();
std::string read_line(const std::string &path) {
    std::string out;
    if (FILE *f = std::fopen(path.c_str(), "r")) {
        char buf[4096];
        if (std::fgets(buf, sizeof buf, f)) out = buf;
        std::fclose(f);
    }
    while (!out.empty() && (out.back() == '\n' || out.back() == ' ')) out.pop_back();
    return out;
}
const GpuLocality &gpu_locality(int device) {
    static std::mutex mutex;
    static GpuLocality known[64];
    static bool looked[64] = {};
    static const GpuLocality none{};
    if (device < 0 || device >= 64 || !g_numa) return none;
    std::lock_guard<std::mutex> lock(mutex);
    GpuLocality &g = known[device];
    if (looked[device]) return g;
    looked[device] = true;
    CPU_ZERO(&g.cpus);
    char bdf[64] = {};
    if (hipDeviceGetPCIBusId(bdf, sizeof bdf, device) != hipSuccess) {
        (void)hipGetLastError();
        return g;
    }
    for (char *c = bdf; *c; ++c) *c = static_cast<char>(std::tolower(*c));
    const std::string base = std::string("/sys/bus/pci/devices/") + bdf + "/";
    const std::string node = read_line(base + "numa_node");
    if (!node.empty()) g.node = std::atoi(node.c_str());
    // "0-63,128-191"
    const std::string list = read_line(base + "local_cpulist");
    g.cpulist = list;
    for (size_t i = 0; i < list.size();) {
        char *end = nullptr;
        const long lo = std::strtol(list.c_str() + i, &end, 10);
        long hi = lo;
        if (end == list.c_str() + i) break;
        i = static_cast<size_t>(end - list.c_str());
        if (i < list.size() && list[i] == '-') {
            hi = std::strtol(list.c_str() + i + 1, &end, 10);
            i = static_cast<size_t>(end - list.c_str());
        }
        for (long c = lo; c <= hi && c < CPU_SETSIZE; ++c) CPU_SET(static_cast<int>(c), &g.cpus), g.have_cpus = true;
        if (i < list.size() && list[i] == ',') ++i;
    }
    return g;
}
}  // namespace
hipError_t pinned_alloc(void **ptr, size_t bytes, unsigned int flags) {
    int device = 0;
    if (hipGetDevice(&device) != hipSuccess) (void)hipGetLastError(), device = -1;
    const int node = gpu_locality(device).node;
    bool policy_set = false;
    int old_mode = 0;
    unsigned long old_mask[16] = {};
    if (node >= 0 && node < 1024 && syscall(SYS_get_mempolicy, &old_mode, old_mask, 8 * sizeof old_mask + 1, nullptr, 0ul) == 0) {
        // MPOL_PREFERRED (1): pages come from `node` while it has any, from anywhere else after that; the thread's own policy (numactl
        // --membind and the like) is put back behind the allocation
        unsigned long mask[16] = {};
        mask[node / (8 * sizeof(unsigned long))] = 1ul << (node % (8 * sizeof(unsigned long)));
        policy_set = syscall(SYS_set_mempolicy, 1, mask, 8 * sizeof mask + 1) == 0;
    }
    const hipError_t e = hipHostMalloc(ptr, bytes, flags);
    if (policy_set) (void)syscall(SYS_set_mempolicy, old_mode, old_mode == 0 ? nullptr : old_mask, old_mode == 0 ? 0ul : 8 * sizeof old_mask + 1);
    return e;
}
int device_locality(int device, int *node, char *cpulist, size_t cap) {
    const GpuLocality &g = gpu_locality(device);
    if (node) *node = g.node;
    if (cpulist && cap) std::snprintf(cpulist, cap, "%s", g.cpulist.c_str());
    return KICP_OK;
}
void bind_thread_near_gpu(int device) {
    const GpuLocality &g = gpu_locality(device);
    if (!g.have_cpus) return;
    cpu_set_t allowed, want;
    if (sched_getaffinity(0, sizeof allowed, &allowed) != 0) return;
    CPU_AND(&want, &allowed, &g.cpus);
    if (CPU_COUNT(&want) == 0 || CPU_EQUAL(&want, &allowed)) return;
    (void)sched_setaffinity(0, sizeof want, &want);
}

const bool g_direct_upload = env_flag("KICP_DIRECT_UPLOAD");
static int stage_wait(HostStage &hs) {
    if (hs.pending) {
        HIP_TRY(hipEventSynchronize(hs.done));
        hs.pending = false;
    }
    return KICP_OK;
}
int stage_reserve(HostStage &hs, size_t bytes, hipStream_t stream) {
    if (bytes <= hs.cap) return KICP_OK;
    HIP_TRY(hipStreamSynchronize(stream));
    hs.release();
    const size_t want = bytes + bytes / 2 + (1u << 20);
    HIP_TRY(pinned_alloc(reinterpret_cast<void **>(&hs.p), want, hipHostMallocDefault));
    hs.cap = want;
    hs.dev = nullptr;
    if (hipHostGetDevicePointer(reinterpret_cast<void **>(&hs.dev), hs.p, 0) != hipSuccess) hs.dev = nullptr, (void)hipGetLastError();
    return KICP_OK;
}
// a transfer that fills the staging buffer itself (kicp_register's pipelined frame upload): wait for the previous user, make room;
// stage_end records the event that guards the buffer against the next one
int stage_begin(HostStage &hs, size_t bytes, hipStream_t stream) {
    if (int rc = stage_wait(hs)) return rc;
    return stage_reserve(hs, bytes, stream);
}
int stage_end(HostStage &hs, hipStream_t stream) {
    if (!hs.done) HIP_TRY(hipEventCreateWithFlags(&hs.done, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(hs.done, stream));
    hs.pending = true;
    return KICP_OK;
}
constexpr size_t kStagePiece = 1u << 20;
// Larger transfers are PULLED by the GPU: the calling thread copies the caller's memory into the pinned staging buffer in 384 KB
// pieces and launches k_pull_bytes behind each, which reads the piece straight out of host memory (16 bytes per lane) while the
// CPU copies the next one.  A kernel launch costs the host ~3 us where a hipMemcpyAsync costs ~10, and the copy engine's start-up
// per transfer is gone (round 4: a 2.1 MB PointCloud2 in ~75 us instead of ~105; kicp_register_f32 uses the widening twin of
// this kernel).  KICP_PULL_UPLOAD=0 restores the DMA engine.
static __global__ __launch_bounds__(256) void k_pull_bytes(const unsigned char *__restrict__ staged, unsigned char *__restrict__ dst, size_t bytes) {
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    const size_t i = (static_cast<size_t>(blockIdx.x) * 256u + threadIdx.x) * 16u;
    if (i + 16u <= bytes) {
        *reinterpret_cast<u32x4 *>(dst + i) = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(staged + i));
    } else {
        for (size_t k = i; k < bytes; ++k) dst[k] = staged[k];
    }
}
constexpr size_t kPullPiece = 384u << 10, kPullMinBytes = 256u << 10;
const bool g_pull_upload = [] {
    const char *e = std::getenv("KICP_PULL_UPLOAD");
    return !(e && *e == '0');
}();
// `offset`: where in the staging buffer this transfer may start (several may be in flight within one call; reserve first)
int staged_upload(HostStage &hs, size_t offset, void *dst, const void *src, size_t bytes, hipStream_t stream) {
    if (bytes == 0) return KICP_OK;
    if (g_direct_upload) {
        HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, stream));
        return KICP_OK;
    }
    if (offset == 0) {
        if (int rc = stage_wait(hs)) return rc;  // a copy of the previous call may still be reading the buffer
        if (int rc = stage_reserve(hs, bytes, stream)) return rc;
    }
    if (offset + bytes > hs.cap) return fail(KICP_ERR_ARG, "staging buffer too small for a follow-up transfer");
    // (16-byte loads and stores: both ends and the place in the staging buffer must be 16-byte aligned - device allocations and
    //  the frame / cloud buffers of this library are)
    const bool pull = g_pull_upload && hs.dev && bytes >= kPullMinBytes && offset % 16 == 0 && reinterpret_cast<uintptr_t>(dst) % 16 == 0;
    const size_t piece = pull ? kPullPiece : kStagePiece;
    for (size_t off = 0; off < bytes; off += piece) {
        const size_t len = std::min(piece, bytes - off);
        std::memcpy(hs.p + offset + off, static_cast<const unsigned char *>(src) + off, len);
        if (pull)
            hipLaunchKernelGGL(k_pull_bytes, dim3(static_cast<uint32_t>((len + 4095) / 4096)), dim3(256), 0, stream, hs.dev + offset + off,
                               static_cast<unsigned char *>(dst) + off, len);
        else
            HIP_TRY(hipMemcpyAsync(static_cast<unsigned char *>(dst) + off, hs.p + offset + off, len, hipMemcpyHostToDevice, stream));
    }
    if (pull) HIP_TRY(hipGetLastError());
    if (!hs.done) HIP_TRY(hipEventCreateWithFlags(&hs.done, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(hs.done, stream));
    hs.pending = true;
    return KICP_OK;
}
// device -> caller memory; returns with the data in place
int staged_download(HostStage &hs, void *dst, const void *src, size_t bytes, hipStream_t stream) {
    if (bytes == 0) return KICP_OK;
    if (g_direct_upload) {
        HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        return KICP_OK;
    }
    if (int rc = stage_wait(hs)) return rc;
    if (int rc = stage_reserve(hs, bytes, stream)) return rc;
    HIP_TRY(hipMemcpyAsync(hs.p, src, bytes, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    std::memcpy(dst, hs.p, bytes);
    return KICP_OK;
}

}  // namespace host
}  // namespace kicp

using namespace kicp;
using namespace kicp::host;

// every lane walks its own chain of dependent loads through a random cyclic permutation of 128-byte lines (diagnostic: the
// latency model of bench.py prices the pass kernel's chain of dependent accesses with what THIS box measures)
static __global__ void k_chase(const uint32_t *__restrict__ next, uint32_t lines, int steps, uint32_t *__restrict__ sink) {
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t i = static_cast<uint32_t>((static_cast<unsigned long long>(gid) * 2654435761ull) % lines) * 32u;
    for (int s = 0; s < steps; ++s) i = next[i];
    sink[gid] = i;
}

extern "C" {

int kicp_probe_dependent_load(int device, size_t working_set_bytes, int workgroups, int block, int steps, double *out_ns_per_step) {
    if (!out_ns_per_step || workgroups <= 0 || block <= 0 || block > 1024 || steps <= 0 || working_set_bytes < 256) return fail(KICP_ERR_ARG, "bad argument");
    if (int rc = set_device(device)) return rc;
    const size_t lines = std::min<size_t>(working_set_bytes / 128, 0x7FFFFFFu);
    std::vector<uint32_t> perm(lines), next(lines * 32, 0u);
    for (size_t k = 0; k < lines; ++k) perm[k] = static_cast<uint32_t>(k);
    unsigned long long x = 0x9E3779B97F4A7C15ull;  // splitmix64: the same permutation on every box
    for (size_t k = lines - 1; k > 0; --k) {
        x += 0x9E3779B97F4A7C15ull;
        unsigned long long z = x;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull, z = (z ^ (z >> 27)) * 0x94D049BB133111EBull, z ^= z >> 31;
        std::swap(perm[k], perm[z % (k + 1)]);
    }
    for (size_t k = 0; k < lines; ++k) next[static_cast<size_t>(perm[k]) * 32] = perm[(k + 1) % lines] * 32u;
    uint32_t *d_next = nullptr, *d_sink = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    const size_t lanes = static_cast<size_t>(workgroups) * block;
    hipError_t e = hipMalloc(&d_next, next.size() * 4);
    if (e == hipSuccess) e = hipMalloc(&d_sink, lanes * 4);
    if (e == hipSuccess) e = hipMemcpy(d_next, next.data(), next.size() * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipEventCreate(&e0);
    if (e == hipSuccess) e = hipEventCreate(&e1);
    float best = 0.f;
    for (int rep = 0; rep < 4 && e == hipSuccess; ++rep) {  // (the first launch warms the caches as far as the working set lets it)
        hipEventRecord(e0, nullptr);
        hipLaunchKernelGGL(k_chase, dim3(workgroups), dim3(block), 0, nullptr, d_next, static_cast<uint32_t>(lines), steps, d_sink);
        hipEventRecord(e1, nullptr);
        e = hipEventSynchronize(e1);
        float ms = 0.f;
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
        if (rep > 0 && (best == 0.f || ms < best)) best = ms;
    }
    if (e0) hipEventDestroy(e0);
    if (e1) hipEventDestroy(e1);
    hipFree(d_next), hipFree(d_sink);
    if (e != hipSuccess) return fail(KICP_ERR_HIP, std::string("kicp_probe_dependent_load: ") + hipGetErrorString(e));
    *out_ns_per_step = static_cast<double>(best) * 1.0e6 / steps;
    return KICP_OK;
}


const char *kicp_last_error(void) { return last_error().c_str(); }
int kicp_version(void) { return KICP_VERSION; }
int kicp_device_locality(int device, int *out_numa_node, char *out_cpulist, size_t cap) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n) return fail(KICP_ERR_ARG, "no such device");
    return device_locality(device, out_numa_node, out_cpulist, cap);
}
int kicp_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return -1;
    return n;
}

// ---- device helpers -------------------------------------------------------------------------------------------------
int kicp_device_malloc(int device, size_t bytes, void **out_dptr) {
    if (!out_dptr) return fail(KICP_ERR_ARG, "null argument");
    if (int rc = set_device(device)) return rc;
    HIP_TRY(hipMalloc(out_dptr, bytes ? bytes : 1));
    return KICP_OK;
}
int kicp_device_free(int device, void *dptr) {
    if (int rc = set_device(device)) return rc;
    HIP_TRY(hipFree(dptr));
    return KICP_OK;
}
int kicp_device_upload(int device, void *dst_dptr, const void *src_host, size_t bytes) {
    if (int rc = set_device(device)) return rc;
    HIP_TRY(hipMemcpy(dst_dptr, src_host, bytes, hipMemcpyHostToDevice));
    return KICP_OK;
}
int kicp_device_synchronize(int device) {
    if (int rc = set_device(device)) return rc;
    HIP_TRY(hipDeviceSynchronize());
    return KICP_OK;
}

}  // extern "C"
