// kicp_reg_comm.hip -- the multi-GPU exchanges behind include/kicp.h (see kicp_reg_internal.hpp)
#include "kicp_reg_internal.hpp"

using namespace kicp;
using namespace kicp::host;

namespace kicp {
namespace host {
CommApi g_comm;
}  // namespace host
}  // namespace kicp

extern "C" {

// ---- multi-GPU ------------------------------------------------------------------------------------------------------
int kicp_comm_unique_id(char id[KICP_COMM_ID_BYTES]) {
    static_assert(KICP_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "id size");
    static_assert(KICP_REDUCE_WORDS == kReduceWords, "payload size");
    if (!id) return fail(KICP_ERR_ARG, "null argument");
    std::string err;
    if (!g_comm.load(err)) return fail(KICP_ERR_COMM, err);
    ncclUniqueId uid;
    const ncclResult_t rc = g_comm.GetUniqueId(&uid);
    if (rc != ncclSuccess) return fail(KICP_ERR_COMM, std::string("ncclGetUniqueId: ") + g_comm.GetErrorString(rc));
    std::memcpy(id, uid.internal, KICP_COMM_ID_BYTES);
    return KICP_OK;
}
static void destroy_lane_comms(kicp_reg *reg) {  // the lanes' sub-communicators go before the communicator they were split off
    for (kicp_reg *lane : reg->batch_lanes)
        if (lane->stream) hipStreamSynchronize(lane->stream), lane->comm = nullptr;
    for (ncclComm_t &c : reg->lane_comms) {
        if (c) g_comm.CommDestroy(c);
        c = nullptr;
    }
    reg->lane_comms_failed = false;
}
int kicp_reg_comm_init(kicp_reg *reg, int nranks, int rank, const char id[KICP_COMM_ID_BYTES]) {
    if (!reg || !id || nranks < 1 || rank < 0 || rank >= nranks) return fail(KICP_ERR_ARG, "bad communicator arguments");
    std::string err;
    if (!g_comm.load(err)) return fail(KICP_ERR_COMM, err);
    if (int rc = set_device(reg->device)) return rc;
    if (reg->comm) destroy_lane_comms(reg), g_comm.CommDestroy(reg->comm), reg->comm = nullptr;
    ncclUniqueId uid;
    std::memcpy(uid.internal, id, KICP_COMM_ID_BYTES);
    const ncclResult_t rc = g_comm.CommInitRank(&reg->comm, nranks, uid, rank);
    if (rc != ncclSuccess) {
        reg->comm = nullptr;
        return fail(KICP_ERR_COMM, std::string("ncclCommInitRank: ") + g_comm.GetErrorString(rc));
    }
    reg->nranks = nranks, reg->rank = rank;
    return KICP_OK;
}
int kicp_reg_comm_destroy(kicp_reg *reg) {
    if (!reg) return fail(KICP_ERR_ARG, "null argument");
    if (reg->comm) {
        hipSetDevice(reg->device);
        hipStreamSynchronize(reg->stream);
        destroy_lane_comms(reg);
        g_comm.CommDestroy(reg->comm);
        reg->comm = nullptr;
    }
    reg->nranks = 1, reg->rank = 0;
    return KICP_OK;
}
// Shared segment layout: one header slot (magic word written LAST by rank 0, then the rank count) followed by the
// [2 buffers][nranks] hand-off slots of single calls and, behind them, one such area per lane of a sharded batch call.
constexpr unsigned long long kShmMagic = 0x4B49435053484D31ull;  // "KICPSHM1"
// Rank 0 writes this over the magic word before it unlinks a segment (its own at destroy, a left-over of that name at init): a
// rank that opened the OLD segment in the window before the unlink sees it, lets go and opens the name again.  A rank
// that had already passed the check exchanges on the dead segment and runs into the exchange's time-out - an error, never a hang
// or a mixed-up sum; callers that re-initialise should put a barrier between kicp_reg_shm_destroy and kicp_reg_shm_init.
constexpr unsigned long long kShmRetired = 0x4B49435044454144ull;  // "KICPDEAD"
static void shm_retire_by_name(const std::string &nm) {
    const int fd = shm_open(nm.c_str(), O_RDWR, 0600);
    if (fd < 0) return;
    struct stat st {};
    if (fstat(fd, &st) == 0 && static_cast<size_t>(st.st_size) >= sizeof(kicp_reg::ShmSlot)) {
        void *ptr = mmap(nullptr, sizeof(kicp_reg::ShmSlot), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        if (ptr != MAP_FAILED) {
            __atomic_store_n(&static_cast<kicp_reg::ShmSlot *>(ptr)->seq, kShmRetired, __ATOMIC_RELEASE);
            munmap(ptr, sizeof(kicp_reg::ShmSlot));
        }
    }
    close(fd);
}
int kicp_reg_shm_destroy(kicp_reg *reg) {
    if (!reg) return fail(KICP_ERR_ARG, "null argument");
    if (reg->shm) {
        hipSetDevice(reg->device);
        hipStreamSynchronize(reg->stream);
        if (reg->d_shm) (void)hipHostUnregister(reg->shm_base);
        if (reg->rank == 0) __atomic_store_n(&static_cast<kicp_reg::ShmSlot *>(reg->shm_base)->seq, kShmRetired, __ATOMIC_RELEASE);
        munmap(reg->shm_base, reg->shm_bytes);
        if (reg->rank == 0) shm_unlink(reg->shm_name.c_str());
        reg->shm = nullptr, reg->d_shm = nullptr, reg->shm_base = nullptr, reg->shm_bytes = 0;
    }
    reg->nranks = 1, reg->rank = 0;
    return KICP_OK;
}
int kicp_reg_shm_init(kicp_reg *reg, int nranks, int rank, const char *name) {
    if (!reg || !name || nranks < 1 || rank < 0 || rank >= nranks) return fail(KICP_ERR_ARG, "bad shared-segment arguments");
    if (reg->comm) return fail(KICP_ERR_ARG, "an RCCL communicator is already attached");
    kicp_reg_shm_destroy(reg);
    if (int rc = set_device(reg->device)) return rc;
    const size_t bytes = (1 + 2 * static_cast<size_t>(nranks) * (1 + kicp_reg::kShmLanes)) * sizeof(kicp_reg::ShmSlot);  // header, single-call area, the lanes' areas
    const std::string nm = std::string(name[0] == '/' ? "" : "/") + name;
    void *ptr = MAP_FAILED;
    if (rank == 0) {
        // a segment of this name left behind by a crashed run must not be adopted: mark it dead, remove it, then create exclusively
        shm_retire_by_name(nm);
        shm_unlink(nm.c_str());
        const int fd = shm_open(nm.c_str(), O_CREAT | O_EXCL | O_RDWR, 0600);
        if (fd < 0) return fail(KICP_ERR_COMM, "shm_open(" + nm + ", O_CREAT | O_EXCL) failed");
        if (ftruncate(fd, static_cast<off_t>(bytes)) != 0) {
            close(fd);
            shm_unlink(nm.c_str());
            return fail(KICP_ERR_COMM, "ftruncate on the shared segment failed");
        }
        ptr = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        close(fd);
        if (ptr == MAP_FAILED) return fail(KICP_ERR_COMM, "mmap of the shared segment failed");
        std::memset(ptr, 0, bytes);
        auto *hdr = static_cast<kicp_reg::ShmSlot *>(ptr);
        hdr->words[0] = nranks;
        __atomic_store_n(&hdr->seq, kShmMagic, __ATOMIC_RELEASE);  // published last: the other ranks wait for it
    } else {
        // wait (bounded) until rank 0 has created, sized, zeroed and published the segment
        const Deadline deadline;
        for (;;) {
            const int fd = shm_open(nm.c_str(), O_RDWR, 0600);
            if (fd >= 0) {
                struct stat st {};
                if (fstat(fd, &st) == 0 && static_cast<size_t>(st.st_size) == bytes) {
                    ptr = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
                    close(fd);
                    if (ptr == MAP_FAILED) return fail(KICP_ERR_COMM, "mmap of the shared segment failed");
                    auto *hdr = static_cast<kicp_reg::ShmSlot *>(ptr);
                    unsigned long long word;
                    while ((word = __atomic_load_n(&hdr->seq, __ATOMIC_ACQUIRE)) != kShmMagic && word != kShmRetired) {
                        if (deadline.passed()) {
                            munmap(ptr, bytes);
                            return fail(KICP_ERR_COMM, "timed out waiting for rank 0 to publish the shared segment");
                        }
                        usleep(50);
                    }
                    if (word == kShmRetired) {  // the previous incarnation's segment: rank 0 is about to replace it
                        munmap(ptr, bytes);
                        ptr = MAP_FAILED;
                        if (deadline.passed()) return fail(KICP_ERR_COMM, "timed out waiting for rank 0 to replace the retired shared segment " + nm);
                        usleep(200);
                        continue;
                    }
                    if (hdr->words[0] != nranks) {
                        munmap(ptr, bytes);
                        return fail(KICP_ERR_COMM, "the shared segment was created for a different number of ranks");
                    }
                    break;
                }
                close(fd);
            }
            if (deadline.passed()) return fail(KICP_ERR_COMM, "timed out waiting for rank 0 to create shared segment " + nm);
            usleep(200);
        }
    }
    // The device view is only needed when the GPU itself writes the slot ("group_rows" = 0); by default the rank's host adds
    // its GPU's tagged rows and stores the totals, so a failed registration is not fatal.
    hipError_t e = hipHostRegister(ptr, bytes, hipHostRegisterMapped | hipHostRegisterPortable);
    void *dptr = nullptr;
    if (e == hipSuccess) e = hipHostGetDevicePointer(&dptr, ptr, 0);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        dptr = nullptr;
    }
    reg->shm_base = ptr;
    reg->shm = static_cast<kicp_reg::ShmSlot *>(ptr) + 1;
    reg->d_shm = dptr ? static_cast<kicp_reg::ShmSlot *>(dptr) + 1 : nullptr;
    reg->shm_bytes = bytes, reg->shm_step = 0, reg->shm_name = nm, reg->nranks = nranks, reg->rank = rank;
    for (auto &st : reg->shm_lane_step) st = 0;
    reg->shm_poisoned = false;
    return KICP_OK;
}
// ---- one-shot exchange over peer mappings -----------------------------------------------------
// Each rank owns a mailbox in its own HBM: [2 parities][nranks][kP2pWords] tagged words, fine-grained so that a peer's stores
// (over xGMI) become visible to a kernel that is polling it.  Export -> the caller gathers every rank's handle (any
// transport: torch.distributed, MPI, a file) -> connect opens the peers' mailboxes -> every pass kernel's last workgroup
// writes this rank's totals into all mailboxes and collects its own (mode 5, kicp_kernels.hpp::p2p_exchange).
int kicp_reg_p2p_destroy(kicp_reg *reg) {
    if (!reg) return fail(KICP_ERR_ARG, "null argument");
    if (reg->p2p_box || reg->d_p2p_table) {
        hipSetDevice(reg->device);
        hipStreamSynchronize(reg->stream);
        for (void *&m : reg->p2p_mapped)
            if (m) (void)hipIpcCloseMemHandle(m), m = nullptr;
        if (reg->d_p2p_table) hipFree(reg->d_p2p_table);
        if (reg->p2p_box) hipFree(reg->p2p_box);
        reg->d_p2p_table = nullptr, reg->p2p_box = nullptr;
        (void)hipGetLastError();
    }
    reg->nranks = 1, reg->rank = 0, reg->p2p_step = 0, reg->p2p_poisoned = false;
    return KICP_OK;
}
int kicp_reg_p2p_export(kicp_reg *reg, int nranks, int rank, char handle[KICP_P2P_HANDLE_BYTES]) {
    static_assert(KICP_P2P_HANDLE_BYTES == sizeof(hipIpcMemHandle_t), "handle size");
    static_assert(KICP_P2P_MAX_RANKS == kP2pMaxRanks, "rank limit");
    if (!reg || !handle || nranks < 1 || nranks > kP2pMaxRanks || rank < 0 || rank >= nranks) return fail(KICP_ERR_ARG, "bad peer-mailbox arguments");
    if (reg->comm || reg->shm || reg->allreduce_fn) return fail(KICP_ERR_ARG, "another exchange is already attached");
    kicp_reg_p2p_destroy(reg);
    if (int rc = set_device(reg->device)) return rc;
    const size_t bytes = p2p_box_words(nranks) * sizeof(unsigned long long);  // totals area (mode 5) + group-row area (mode 6)
    // fine-grained: stores arriving from a peer GPU must be visible to a wave that is polling (no stale L2 line)
    hipError_t e = hipExtMallocWithFlags(reinterpret_cast<void **>(&reg->p2p_box), bytes, hipDeviceMallocFinegrained);
    if (e != hipSuccess) {
        reg->p2p_box = nullptr;
        return fail(KICP_ERR_HIP, std::string("hipExtMallocWithFlags(fine-grained mailbox): ") + hipGetErrorString(e));
    }
    HIP_TRY(hipMemset(reg->p2p_box, 0, bytes));  // tag 0 never matches
    hipIpcMemHandle_t h;
    e = hipIpcGetMemHandle(&h, reg->p2p_box);
    if (e != hipSuccess) {
        hipFree(reg->p2p_box), reg->p2p_box = nullptr;
        return fail(KICP_ERR_COMM, std::string("hipIpcGetMemHandle: ") + hipGetErrorString(e));
    }
    std::memcpy(handle, &h, sizeof h);
    reg->nranks = nranks, reg->rank = rank;
    return KICP_OK;
}
int kicp_reg_p2p_connect(kicp_reg *reg, const char *handles) {
    if (!reg || !handles) return fail(KICP_ERR_ARG, "null argument");
    if (!reg->p2p_box) return fail(KICP_ERR_ARG, "kicp_reg_p2p_export first");
    if (reg->d_p2p_table) return fail(KICP_ERR_ARG, "already connected: kicp_reg_p2p_destroy / _export first");
    if (int rc = set_device(reg->device)) return rc;
    unsigned long long *table[kP2pMaxRanks] = {};
    for (int k = 0; k < reg->nranks; ++k) {
        if (k == reg->rank) {
            table[k] = reg->p2p_box;
            continue;
        }
        hipIpcMemHandle_t h;
        std::memcpy(&h, handles + static_cast<size_t>(k) * KICP_P2P_HANDLE_BYTES, sizeof h);
        void *ptr = nullptr;
        const hipError_t e = hipIpcOpenMemHandle(&ptr, h, hipIpcMemLazyEnablePeerAccess);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            return fail(KICP_ERR_COMM, "hipIpcOpenMemHandle(rank " + std::to_string(k) + "): " + hipGetErrorString(e));
        }
        reg->p2p_mapped[k] = ptr, table[k] = static_cast<unsigned long long *>(ptr);
    }
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&reg->d_p2p_table), sizeof table));
    HIP_TRY(hipMemcpy(reg->d_p2p_table, table, sizeof table, hipMemcpyHostToDevice));
    reg->p2p_step = 0;
    return KICP_OK;
}
int kicp_reg_set_allreduce(kicp_reg *reg, kicp_allreduce_fn fn, void *user) {
    if (!reg) return fail(KICP_ERR_ARG, "null argument");
    reg->allreduce_fn = fn, reg->allreduce_user = user;
    return KICP_OK;
}

}  // extern "C"
