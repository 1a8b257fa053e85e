// kicp_aql.hpp -- direct AQL dispatch of the pass kernel on a user-mode HSA queue owned by the registration handle.
//
// Why: one ICP iteration is ONE small kernel whose result the host waits for, so the launch path is on the critical path of
// every iteration.  hipLaunchKernelGGL costs ~3 us of host time per call on this runtime (argument marshalling, stream
// bookkeeping, packet, doorbell: tools/micro/handoff.hip measures 2.97 us); writing the 64-byte AQL packet and ringing the
// doorbell ourselves costs a few hundred nanoseconds.  Nothing else changes: same code object (the gfx950 image of
// kicp_reg.hip, embedded in the library by kicp_hsaco.S), same kernel, same arguments, results still handed to the host
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
// stream and nothing must be ordered behind the kernel on it (kicp_reg.hip: can_use_aql()).  Kernels that need scratch
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
#include <map>
#include <string>

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
        // kernarg ring: host-coherent pinned memory the GPU reads the arguments from
        if (hipHostMalloc(reinterpret_cast<void **>(&kernarg_), kSlots * kSlotBytes, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) {
            (void)hipGetLastError();
            return off("kernarg allocation failed");
        }
        void *dev = nullptr;
        if (hipHostGetDevicePointer(&dev, kernarg_, 0) != hipSuccess || dev != kernarg_) {
            (void)hipGetLastError();
            return off("kernarg memory is not identity mapped");
        }
        if (hsa_signal_create(1, 0, nullptr, &done_) != HSA_STATUS_SUCCESS) return off("hsa_signal_create failed");
        have_signal_ = true;
        ready = true;
        return 0;
    }

    // kernel descriptor + segment sizes of `name` (mangled, without the .kd suffix); cached
    const AqlKernel &kernel(const std::string &name) {
        auto it = kernels_.find(name);
        if (it != kernels_.end()) return it->second;
        AqlKernel k;
        hsa_executable_symbol_t sym;
        const std::string kd = name + ".kd";
        if (hsa_executable_get_symbol_by_name(exe_, kd.c_str(), &agent_, &sym) == HSA_STATUS_SUCCESS &&
            hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_OBJECT, &k.object) == HSA_STATUS_SUCCESS &&
            hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_KERNARG_SEGMENT_SIZE, &k.kernarg_size) == HSA_STATUS_SUCCESS &&
            hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_GROUP_SEGMENT_SIZE, &k.group_size) == HSA_STATUS_SUCCESS &&
            hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_PRIVATE_SEGMENT_SIZE, &k.private_size) == HSA_STATUS_SUCCESS)
            k.usable = k.object != 0 && k.private_size == 0 && k.kernarg_size <= kSlotBytes;
        return kernels_.emplace(name, k).first->second;
    }

    // one 1-D dispatch: `grid` workgroups of `block` work-items; `args` = the kernel's explicit argument block
    // acquire / release: HSA_FENCE_SCOPE_{NONE, AGENT, SYSTEM} of the packet's fences
    bool dispatch(const AqlKernel &k, uint32_t grid, uint32_t block, const void *args, size_t args_bytes, int acquire = HSA_FENCE_SCOPE_SYSTEM,
                  int release = HSA_FENCE_SCOPE_SYSTEM) {
        const size_t implicit = (args_bytes + 7) & ~size_t(7);  // code object v5: the implicit arguments follow, 8-byte aligned
        if (!ready || !k.usable || queue_error || implicit + 80 > k.kernarg_size) return false;
        unsigned char *ka = kernarg_ + (slot_++ % kSlots) * kSlotBytes;
        std::memcpy(ka, args, args_bytes);
        // hidden_block_count_{x,y,z} u32 @0, hidden_group_size_{x,y,z} u16 @12, hidden_remainder_{x,y,z} u16 @18,
        // hidden_global_offset_{x,y,z} u64 @40, hidden_grid_dims u16 @64 (llvm AMDGPU usage, code object v5)
        unsigned char *ia = ka + implicit;
        std::memset(ia, 0, 80);
        const uint32_t counts[3] = {grid, 1u, 1u};
        const uint16_t sizes[3] = {static_cast<uint16_t>(block), 1, 1}, dims = 1;
        std::memcpy(ia, counts, 12), std::memcpy(ia + 12, sizes, 6), std::memcpy(ia + 64, &dims, 2);

        const uint64_t index = hsa_queue_add_write_index_relaxed(queue_, 1);
        while (index - hsa_queue_load_read_index_scacquire(queue_) >= queue_->size) {
        }
        auto *pkt = static_cast<hsa_kernel_dispatch_packet_t *>(queue_->base_address) + (index & (queue_->size - 1));
        pkt->workgroup_size_x = static_cast<uint16_t>(block), pkt->workgroup_size_y = 1, pkt->workgroup_size_z = 1;
        pkt->reserved0 = 0;
        pkt->grid_size_x = grid * block, pkt->grid_size_y = 1, pkt->grid_size_z = 1;
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

    void release() {
        if (kernarg_) (void)hipHostFree(kernarg_);
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
};

}  // namespace host
}  // namespace kicp
