// kicp_aql.hpp -- direct AQL dispatch of the pass kernel on a user-mode HSA queue owned by the registration handle.
//
// Why: one ICP iteration is ONE small kernel whose result the host waits for, so the launch path is on the critical path of
// every iteration.  hipLaunchKernelGGL costs ~3 us of host time per call on this runtime (argument marshalling, stream
// bookkeeping, packet, doorbell: tools/micro/handoff.hip measures 2.97 us); writing the 64-byte AQL packet and ringing the
// doorbell ourselves costs a few hundred nanoseconds.  Nothing else changes: same code object (the gfx950 image of
// kicp_reg_launch.hip, embedded in the library by kicp_hsaco.S), same kernel, same arguments, results still handed to the host
// through tagged rows in host-mapped memory.
//
// What this file does (host code, ROCr / HSA runtime API - the layer HIP itself sits on):
//   * finds the HSA agent of the handle's HIP device (PCI bus id), creates one single-producer queue on it;
//   * loads the embedded code object into an executable and looks kernels up by their mangled names (<name>.kd);
//   * dispatch(): copies the explicit argument block into a kernarg slot (ring in host-coherent pinned memory), fills the
//     code-object-v5 implicit arguments the device library reads (block counts, group sizes, remainders, grid dims),
//     writes the packet (barrier bit, system-scope acquire + release like HIP's own packets), publishes the header with a
//     release store and rings the doorbell.
// The queue is independent of the handle's HIP stream: the caller dispatches here only when nothing is pending on that
// stream and nothing must be ordered behind the kernel on it (kicp_reg_launch.hip: launch_pass).  Kernels that need scratch
// memory are refused (the pass kernels need none); any failure at set-up leaves `ready` false and the handle keeps
// launching through HIP.  KICP_AQL=0 in the environment disables this path.
#pragma once
#include <hip/hip_runtime.h>
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cxxabi.h>
#include <immintrin.h>

#include <map>
#include <string>
#include <vector>

extern "C" const unsigned char kicp_hsaco_start[];
extern "C" const unsigned char kicp_hsaco_end[];

namespace kicp {
namespace host {

struct AqlKernel {
    uint64_t object = 0;  // address of the kernel descriptor
    uint32_t kernarg_size = 0, group_size = 0, private_size = 0;
    bool usable = false;
};

class AqlDispatcher {
public:
    bool ready = false;
    std::string why;                  // why it is not ready
    volatile int queue_error = 0;     // set by the queue's error callback

    int init(int hip_device) {
        if (ready) return 0;
        if (const char *e = std::getenv("KICP_AQL"))
            if (std::atoi(e) == 0) return off("disabled by KICP_AQL=0");
        if (kicp_hsaco_end - kicp_hsaco_start < 64) return off("no embedded code object");
        if (hsa_init() != HSA_STATUS_SUCCESS) return off("hsa_init failed");
        inited_ = true;
        char bus[64] = {};
        if (hipDeviceGetPCIBusId(bus, sizeof bus, hip_device) != hipSuccess) return off("hipDeviceGetPCIBusId failed");
        unsigned dom = 0, b = 0, d = 0, f = 0;
        if (std::sscanf(bus, "%x:%x:%x.%x", &dom, &b, &d, &f) != 4) return off(std::string("cannot parse PCI bus id ") + bus);
        want_bdf_ = (b << 8) | (d << 3) | f, want_domain_ = dom;
        hsa_iterate_agents(&AqlDispatcher::pick_agent, this);
        if (!found_) return off(std::string("no HSA agent at ") + bus);
        if (hsa_queue_create(agent_, 256, HSA_QUEUE_TYPE_SINGLE, &AqlDispatcher::on_queue_error, this, UINT32_MAX, UINT32_MAX, &queue_) != HSA_STATUS_SUCCESS)
            return off("hsa_queue_create failed");
        if (hsa_code_object_reader_create_from_memory(kicp_hsaco_start, static_cast<size_t>(kicp_hsaco_end - kicp_hsaco_start), &reader_) != HSA_STATUS_SUCCESS)
            return off("code object reader failed");
        have_reader_ = true;
        if (hsa_executable_create_alt(HSA_PROFILE_FULL, HSA_DEFAULT_FLOAT_ROUNDING_MODE_DEFAULT, nullptr, &exe_) != HSA_STATUS_SUCCESS)
            return off("hsa_executable_create_alt failed");
        have_exe_ = true;
        if (hsa_executable_load_agent_code_object(exe_, agent_, reader_, nullptr, nullptr) != HSA_STATUS_SUCCESS) return off("loading the code object failed");
        if (hsa_executable_freeze(exe_, nullptr) != HSA_STATUS_SUCCESS) return off("hsa_executable_freeze failed");
        // Kernarg ring.  Default: in the GPU's own memory, written by the CPU through the PCIe BAR - the scalar loads of a kernel's
        // first waves then hit HBM instead of crossing PCIe to host memory (measured: 2.2 us less per dispatch; the whole GPU test
        // suite passes with it).  KICP_KERNARG=devhdp also pokes the host-data-path flush register before the doorbell (measured
        // 0.3 us slower, never needed); KICP_KERNARG=host, or a platform without a CPU-writable BAR: host-coherent pinned memory.
        const char *place = std::getenv("KICP_KERNARG");
        if (!place || std::strcmp(place, "host") != 0) {
            if (!kernarg_in_hbm(place && std::strcmp(place, "devhdp") == 0)) why = "device-memory kernarg ring unavailable (" + why + "): using host memory";
        }
        if (!kernarg_) {
            if (hipHostMalloc(reinterpret_cast<void **>(&kernarg_), kSlots * kSlotBytes, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) {
                (void)hipGetLastError();
                return off("kernarg allocation failed");
            }
            void *dev = nullptr;
            if (hipHostGetDevicePointer(&dev, kernarg_, 0) != hipSuccess || dev != kernarg_) {
                (void)hipGetLastError();
                return off("kernarg memory is not identity mapped");
            }
        }
        if (hsa_signal_create(1, 0, nullptr, &done_) != HSA_STATUS_SUCCESS) return off("hsa_signal_create failed");
        have_signal_ = true;
        ready = true;
        return 0;
    }

    // Kernel descriptor + segment sizes of the kernel whose DEMANGLED name starts with `prefix`, e.g.
    // "void kicp::k_pass_gather32<256, 1, 4, false>(" - resolved by enumerating the code object's kernel symbols
    // (hsa_executable_iterate_symbols) and demangling them, so a change in the compiler's mangling scheme cannot silently
    // turn every look-up into a miss; tests/test_host.py checks the same names against build/kicp_reg.hsaco.  Cached.
    const AqlKernel &kernel(const std::string &prefix) {
        auto it = kernels_.find(prefix);
        if (it != kernels_.end()) return it->second;
        if (symbols_.empty() && ready) hsa_executable_iterate_symbols(exe_, &AqlDispatcher::collect_symbol, this);
        AqlKernel k;
        for (const auto &entry : symbols_) {
            if (entry.first.compare(0, prefix.size(), prefix) != 0) continue;
            const hsa_executable_symbol_t sym = entry.second;
            if (hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_OBJECT, &k.object) == HSA_STATUS_SUCCESS &&
                hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_KERNARG_SEGMENT_SIZE, &k.kernarg_size) == HSA_STATUS_SUCCESS &&
                hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_GROUP_SEGMENT_SIZE, &k.group_size) == HSA_STATUS_SUCCESS &&
                hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_PRIVATE_SEGMENT_SIZE, &k.private_size) == HSA_STATUS_SUCCESS)
                k.usable = k.object != 0 && k.private_size == 0 && k.kernarg_size <= kSlotBytes;
            break;
        }
        return kernels_.emplace(prefix, k).first->second;
    }
    size_t symbol_count() const { return symbols_.size(); }
    const char *kernarg_place() const { return kernarg_in_hbm_ ? (hdp_flush_ ? "device memory + HDP flush" : "device memory") : "host memory"; }

    // one 1-D dispatch: `grid` workgroups of `block` work-items; `args` = the kernel's explicit argument block
    // acquire / release: HSA_FENCE_SCOPE_{NONE, AGENT, SYSTEM} of the packet's fences
    bool dispatch(const AqlKernel &k, uint32_t grid, uint32_t block_size, const void *args, size_t args_bytes, int acquire = HSA_FENCE_SCOPE_SYSTEM,
                  int release = HSA_FENCE_SCOPE_SYSTEM) {
        const size_t implicit = (args_bytes + 7) & ~size_t(7);  // code object v5: the implicit arguments follow, 8-byte aligned
        // (a kernel that reads no hidden argument has none: its kernarg segment ends with the explicit block)
        const bool has_implicit = implicit + 80 <= k.kernarg_size;
        if (!ready || !k.usable || queue_error || args_bytes > k.kernarg_size) return false;
        unsigned char *ka = kernarg_ + (slot_++ % kSlots) * kSlotBytes;
        // The whole kernarg segment is built in a local block (zeroed: every hidden argument this file does not set - hostcall /
        // printf buffer, heap, dynamic LDS size, queue pointer - reads as 0 instead of a stale byte of an earlier dispatch) and
        // copied to the slot in one go.
        // hidden_block_count_{x,y,z} u32 @0, hidden_group_size_{x,y,z} u16 @12, hidden_remainder_{x,y,z} u16 @18,
        // hidden_global_offset_{x,y,z} u64 @40, hidden_grid_dims u16 @64 (llvm AMDGPU usage, code object v5)
        alignas(64) unsigned char block[kSlotBytes];
        const size_t total = (static_cast<size_t>(k.kernarg_size) + 63) & ~size_t(63);
        std::memset(block, 0, total);
        std::memcpy(block, args, args_bytes);
        unsigned char *ia = block + implicit;
        const uint32_t counts[3] = {grid, 1u, 1u};
        const uint16_t sizes[3] = {static_cast<uint16_t>(block_size), 1, 1}, dims = 1;
        if (has_implicit) std::memcpy(ia, counts, 12), std::memcpy(ia + 12, sizes, 6), std::memcpy(ia + 64, &dims, 2);
        std::memcpy(ka, block, total);
        if (kernarg_in_hbm_) {
            // The ring lives in HBM and was written through the PCIe BAR (write-combining): drain the CPU's WC buffers, then
            // make the device's host data path hand the bytes on to memory.  Both the flush register and the doorbell below
            // are posted writes to the same device, so PCIe keeps them behind the argument bytes.
            _mm_sfence();
            if (hdp_flush_) *hdp_flush_ = 1u;
        }

        const uint64_t index = hsa_queue_add_write_index_relaxed(queue_, 1);
        while (index - hsa_queue_load_read_index_scacquire(queue_) >= queue_->size) {
        }
        auto *pkt = static_cast<hsa_kernel_dispatch_packet_t *>(queue_->base_address) + (index & (queue_->size - 1));
        pkt->workgroup_size_x = static_cast<uint16_t>(block_size), pkt->workgroup_size_y = 1, pkt->workgroup_size_z = 1;
        pkt->reserved0 = 0;
        pkt->grid_size_x = grid * block_size, pkt->grid_size_y = 1, pkt->grid_size_z = 1;
        pkt->private_segment_size = 0, pkt->group_segment_size = k.group_size;
        pkt->kernel_object = k.object;
        pkt->kernarg_address = ka;
        pkt->reserved2 = 0;
        pkt->completion_signal.handle = 0;
        const uint16_t header = static_cast<uint16_t>((HSA_PACKET_TYPE_KERNEL_DISPATCH << HSA_PACKET_HEADER_TYPE) | (1u << HSA_PACKET_HEADER_BARRIER) |
                                                      (acquire << HSA_PACKET_HEADER_SCACQUIRE_FENCE_SCOPE) |
                                                      (release << HSA_PACKET_HEADER_SCRELEASE_FENCE_SCOPE));
        const uint16_t setup = 1u << HSA_KERNEL_DISPATCH_PACKET_SETUP_DIMENSIONS;
        __atomic_store_n(reinterpret_cast<uint32_t *>(pkt), static_cast<uint32_t>(header) | (static_cast<uint32_t>(setup) << 16), __ATOMIC_RELEASE);
        hsa_signal_store_screlease(queue_->doorbell_signal, static_cast<hsa_signal_value_t>(index));
        ++dispatched_;
        return true;
    }
    // Wait until every kernel dispatched so far has FINISHED (a barrier-AND packet with a completion signal behind them; the
    // queue is in order).  Needed only when other work must be ordered behind those kernels - before the handle launches
    // through its HIP stream again, and before its buffers are freed.  Returns false on time-out / queue error.
    bool drain(double timeout_s) {
        if (!ready || drained_ == dispatched_) return true;
        hsa_signal_store_relaxed(done_, 1);
        const uint64_t index = hsa_queue_add_write_index_relaxed(queue_, 1);
        while (index - hsa_queue_load_read_index_scacquire(queue_) >= queue_->size) {
        }
        auto *pkt = static_cast<hsa_barrier_and_packet_t *>(queue_->base_address) + (index & (queue_->size - 1));
        std::memset(reinterpret_cast<unsigned char *>(pkt) + 4, 0, sizeof(*pkt) - 4);
        pkt->completion_signal = done_;
        const uint16_t header = static_cast<uint16_t>((HSA_PACKET_TYPE_BARRIER_AND << HSA_PACKET_HEADER_TYPE) | (1u << HSA_PACKET_HEADER_BARRIER) |
                                                      (HSA_FENCE_SCOPE_SYSTEM << HSA_PACKET_HEADER_SCACQUIRE_FENCE_SCOPE) |
                                                      (HSA_FENCE_SCOPE_SYSTEM << HSA_PACKET_HEADER_SCRELEASE_FENCE_SCOPE));
        __atomic_store_n(reinterpret_cast<uint32_t *>(pkt), static_cast<uint32_t>(header), __ATOMIC_RELEASE);
        hsa_signal_store_screlease(queue_->doorbell_signal, static_cast<hsa_signal_value_t>(index));
        const auto t0 = std::chrono::steady_clock::now();
        while (hsa_signal_wait_scacquire(done_, HSA_SIGNAL_CONDITION_LT, 1, 1000000, HSA_WAIT_STATE_ACTIVE) >= 1) {
            if (queue_error || std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s) return false;
        }
        drained_ = dispatched_;
        return true;
    }
    bool busy() const { return ready && drained_ != dispatched_; }

    // After a queue error nothing more can be dispatched here: forget the kernels in flight (the queue is dead) so that the
    // handle can go on through its HIP stream.
    void disable() {
        ready = false;
        drained_ = dispatched_;
    }

    void release() {
        if (kernarg_ && kernarg_in_hbm_) (void)hsa_amd_memory_pool_free(kernarg_);
        else if (kernarg_) (void)hipHostFree(kernarg_);
        kernarg_in_hbm_ = false, hdp_flush_ = nullptr;
        symbols_.clear();
        if (have_signal_) hsa_signal_destroy(done_);
        if (queue_) hsa_queue_destroy(queue_);
        if (have_exe_) hsa_executable_destroy(exe_);
        if (have_reader_) hsa_code_object_reader_destroy(reader_);
        if (inited_) hsa_shut_down();
        kernarg_ = nullptr, queue_ = nullptr, have_exe_ = have_reader_ = have_signal_ = inited_ = ready = false;
        kernels_.clear();
    }

private:
    static constexpr size_t kSlots = 64, kSlotBytes = 1024;
    int off(const std::string &reason) {
        why = reason;
        ready = false;
        return 1;
    }
    static hsa_status_t pick_agent(hsa_agent_t a, void *self_) {
        auto *self = static_cast<AqlDispatcher *>(self_);
        hsa_device_type_t type;
        if (hsa_agent_get_info(a, HSA_AGENT_INFO_DEVICE, &type) != HSA_STATUS_SUCCESS || type != HSA_DEVICE_TYPE_GPU) return HSA_STATUS_SUCCESS;
        uint32_t bdf = 0, domain = 0;
        hsa_agent_get_info(a, static_cast<hsa_agent_info_t>(HSA_AMD_AGENT_INFO_BDFID), &bdf);
        hsa_agent_get_info(a, static_cast<hsa_agent_info_t>(HSA_AMD_AGENT_INFO_DOMAIN), &domain);
        if ((bdf & 0xFFFFu) == self->want_bdf_ && domain == self->want_domain_) {
            self->agent_ = a, self->found_ = true;
            return HSA_STATUS_INFO_BREAK;
        }
        return HSA_STATUS_SUCCESS;
    }
    static hsa_status_t collect_symbol(hsa_executable_t, hsa_executable_symbol_t sym, void *self_) {
        auto *self = static_cast<AqlDispatcher *>(self_);
        hsa_symbol_kind_t kind;
        uint32_t len = 0;
        if (hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_TYPE, &kind) != HSA_STATUS_SUCCESS || kind != HSA_SYMBOL_KIND_KERNEL) return HSA_STATUS_SUCCESS;
        if (hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_NAME_LENGTH, &len) != HSA_STATUS_SUCCESS || len == 0) return HSA_STATUS_SUCCESS;
        std::string name(len, '\0');
        if (hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_NAME, &name[0]) != HSA_STATUS_SUCCESS) return HSA_STATUS_SUCCESS;
        while (!name.empty() && name.back() == '\0') name.pop_back();
        if (name.size() > 3 && name.compare(name.size() - 3, 3, ".kd") == 0) name.resize(name.size() - 3);
        int status = 0;
        char *dem = abi::__cxa_demangle(name.c_str(), nullptr, nullptr, &status);
        if (status == 0 && dem) self->symbols_.emplace_back(dem, sym);
        std::free(dem);
        return HSA_STATUS_SUCCESS;
    }
    // the GPU's own memory pools / the first CPU agent, for the device-memory kernarg ring
    struct PoolPick {
        hsa_amd_memory_pool_t fine{}, coarse{};
        bool have_fine = false, have_coarse = false;
    };
    static hsa_status_t pick_pool(hsa_amd_memory_pool_t pool, void *out_) {
        auto *out = static_cast<PoolPick *>(out_);
        hsa_amd_segment_t seg;
        uint32_t flags = 0;
        bool alloc = false;
        if (hsa_amd_memory_pool_get_info(pool, HSA_AMD_MEMORY_POOL_INFO_SEGMENT, &seg) != HSA_STATUS_SUCCESS || seg != HSA_AMD_SEGMENT_GLOBAL) return HSA_STATUS_SUCCESS;
        hsa_amd_memory_pool_get_info(pool, HSA_AMD_MEMORY_POOL_INFO_GLOBAL_FLAGS, &flags);
        hsa_amd_memory_pool_get_info(pool, HSA_AMD_MEMORY_POOL_INFO_RUNTIME_ALLOC_ALLOWED, &alloc);
        if (!alloc) return HSA_STATUS_SUCCESS;
        if ((flags & HSA_AMD_MEMORY_POOL_GLOBAL_FLAG_FINE_GRAINED) && !out->have_fine) out->fine = pool, out->have_fine = true;
        if ((flags & HSA_AMD_MEMORY_POOL_GLOBAL_FLAG_COARSE_GRAINED) && !out->have_coarse) out->coarse = pool, out->have_coarse = true;
        return HSA_STATUS_SUCCESS;
    }
    static hsa_status_t pick_cpu(hsa_agent_t a, void *out_) {
        hsa_device_type_t type;
        if (hsa_agent_get_info(a, HSA_AGENT_INFO_DEVICE, &type) == HSA_STATUS_SUCCESS && type == HSA_DEVICE_TYPE_CPU) {
            *static_cast<hsa_agent_t *>(out_) = a;
            return HSA_STATUS_INFO_BREAK;
        }
        return HSA_STATUS_SUCCESS;
    }
    // `bytes` of the GPU's own memory that the CPU may write through the PCIe BAR (fine-grained pool when there is one: GPU reads
    // are then not served from a stale L2 line); verified by a write / read-back of the first and last word.  nullptr (and
    // `why`) when the platform does not allow it.  Free with free_bar().
public:
    void *alloc_bar(size_t bytes) {
        if (!inited_ || !found_) return why = "HSA not initialised", nullptr;
        PoolPick pick;
        hsa_amd_agent_iterate_memory_pools(agent_, &AqlDispatcher::pick_pool, &pick);
        if (!pick.have_fine && !pick.have_coarse) return why = "no allocatable global pool on the GPU agent", nullptr;
        hsa_agent_t cpu{};
        if (hsa_iterate_agents(&AqlDispatcher::pick_cpu, &cpu) != HSA_STATUS_INFO_BREAK) return why = "no CPU agent", nullptr;
        void *ptr = nullptr;
        if (hsa_amd_memory_pool_allocate(pick.have_fine ? pick.fine : pick.coarse, bytes, 0, &ptr) != HSA_STATUS_SUCCESS || !ptr)
            return why = "hsa_amd_memory_pool_allocate failed", nullptr;
        const hsa_agent_t both[2] = {cpu, agent_};
        if (hsa_amd_agents_allow_access(2, both, nullptr, ptr) != HSA_STATUS_SUCCESS) {
            hsa_amd_memory_pool_free(ptr);
            return why = "the CPU cannot map the GPU's memory (no large BAR?)", nullptr;
        }
        volatile uint64_t *probe = static_cast<volatile uint64_t *>(ptr);
        probe[0] = 0x4B49435041524753ull, probe[bytes / 8 - 1] = 0x1234567890ABCDEFull;
        _mm_sfence();
        if (probe[0] != 0x4B49435041524753ull || probe[bytes / 8 - 1] != 0x1234567890ABCDEFull) {
            hsa_amd_memory_pool_free(ptr);
            return why = "BAR write / read-back mismatch", nullptr;
        }
        probe[0] = 0ull, probe[bytes / 8 - 1] = 0ull;
        _mm_sfence();
        return ptr;
    }
    void free_bar(void *ptr) {
        if (ptr) (void)hsa_amd_memory_pool_free(ptr);
    }

private:
    bool kernarg_in_hbm(bool with_hdp_flush) {
        void *ptr = alloc_bar(kSlots * kSlotBytes);
        if (!ptr) return false;
        if (with_hdp_flush) {
            hsa_amd_hdp_flush_t hdp{};
            if (hsa_agent_get_info(agent_, static_cast<hsa_agent_info_t>(HSA_AMD_AGENT_INFO_HDP_FLUSH), &hdp) == HSA_STATUS_SUCCESS && hdp.HDP_MEM_FLUSH_CNTL)
                hdp_flush_ = hdp.HDP_MEM_FLUSH_CNTL;
        }
        kernarg_ = static_cast<unsigned char *>(ptr), kernarg_in_hbm_ = true;
        return true;
    }
    static void on_queue_error(hsa_status_t status, hsa_queue_t *, void *self_) {
        static_cast<AqlDispatcher *>(self_)->queue_error = static_cast<int>(status) ? static_cast<int>(status) : -1;
    }

    hsa_agent_t agent_{};
    hsa_queue_t *queue_ = nullptr;
    hsa_code_object_reader_t reader_{};
    hsa_executable_t exe_{};
    unsigned char *kernarg_ = nullptr;
    hsa_signal_t done_{};
    uint64_t slot_ = 0, dispatched_ = 0, drained_ = 0;
    uint32_t want_bdf_ = 0, want_domain_ = 0;
    bool found_ = false, inited_ = false, have_reader_ = false, have_exe_ = false, have_signal_ = false;
    std::map<std::string, AqlKernel> kernels_;
    std::vector<std::pair<std::string, hsa_executable_symbol_t>> symbols_;  // demangled kernel names of the code object
    bool kernarg_in_hbm_ = false;
    volatile uint32_t *hdp_flush_ = nullptr;
};

}  // namespace host
}  // namespace kicp
