// kicp_prestep.hip -- the pipeline's pre-steps behind include/kicp.h (kicp_pre_*): wire-format ingest, deskew + crop +
// transform, voxel downsample (kernels: kicp_pre.hpp).
#include <hip/hip_ext.h>  // hipExtLaunchKernelGGL: a launch with its own stop event

#include "kicp_internal.hpp"
#include "kicp_pre.hpp"

using namespace kicp;
using namespace kicp::host;

struct kicp_pre {
    int device = 0;
    hipStream_t stream = nullptr;
    double *buf[KICP_PRE_BUFFERS] = {};
    size_t buf_cap[KICP_PRE_BUFFERS] = {}, buf_n[KICP_PRE_BUFFERS] = {};
    double *d_in = nullptr, *d_ts = nullptr, *d_staged = nullptr;
    uint32_t *d_flags = nullptr, *d_block_counts = nullptr, *d_misc = nullptr;  // misc: [0] total, [1] error, [2] largest robin-hood displacement of the last downsample
    uint32_t last_max_probe = 0;
    // longest probe the reference's tsl::robin_map tolerates before it grows its table instead (kicp_pre_set_probe_limit)
    uint32_t probe_limit = [] {
        const char *e = std::getenv("KICP_ROBIN_PROBE_LIMIT");
        return e && *e ? static_cast<uint32_t>(std::strtoul(e, nullptr, 10)) : 128u;
    }();
    unsigned char *d_table = nullptr;  // downsampling table: keys | min_index | order | home_at, 20 B per bucket (kicp_pre.hpp)
    unsigned char *d_table2 = nullptr; // the fused chain's table of the second level (the first level's is still being emptied when it fills)
    uint32_t *d_counts1 = nullptr, *d_counts2 = nullptr;  // the fused chain's occupied-bucket counts per tile of either table
    size_t cap_n = 0, table_slots = 0;
    bool table_clean = false;  // every byte of d_table / d_table2 is 0xFF (what a downsample needs to find; its gather step leaves it so)
    // the fused chain (k_frame_*): the survivor count its table size is guessed from (0: none yet), the record its last workgroup
    // writes for the host, and how often the guess was wrong (the unfused steps then run from buffer 0 on)
    uint32_t spec_tiles_b = 0;  // 256-bucket tiles of the previous frame's second-level table (sizes that level's launches)
    uint32_t spec_n0 = 0, spec_n_in = 0;  // (... and the input count it belonged to: the guess scales with the frame)
    unsigned long long *h_rec = nullptr;  // pinned, host-coherent: [0..4] the chain's tagged words, [8..10] / [12..14] the ingest records (this call's / the look-ahead's),
                                          // [16..23] the pushed frame's piece flags (k_push_frame)
    unsigned long long *d_push_tickets = nullptr;  // [kPushPieces] device counters of k_push_frame, never reset
    unsigned long long push_drawn = 0;
    uint32_t push_seq = 0;
    // the registration source (buffer 2, ~100 KB) also lands in host memory as the fused chain's last launch writes it: the pipeline
    // returns it too (KinematicICP.cpp:84), and a copy + stream synchronisation for it cost the frame ~40 us
    unsigned char *h_src = nullptr, *h_src_dev = nullptr;
    size_t h_src_cap = 0;
    bool src_on_host = false;  // h_src holds buffer 2's current contents - once word 5 of h_rec carries src_seq (k_frame_src_host)
    uint32_t src_seq = 0;
    unsigned char *copy_host_dev = nullptr;  // the landing area as the device sees it (nullptr: not mapped - the DMA engine moves the frame)
    unsigned long long *d_ticket = nullptr;  // [2] the ingest kernels' tickets (never reset), one per record
    unsigned long long ticket_drawn[2] = {0ull, 0ull};
    uint32_t chain_seq = 0;
    unsigned long long ingest_seq = 0;
    unsigned long long spec_misses = 0, fused_frames = 0;
    bool fused = [] {
        const char *e = std::getenv("KICP_PRE_FUSED");
        return !(e && *e == '0');
    }();
    // wire-format ingest: the raw message bytes (only where the device cannot read the staging buffer), the stamps' extrema, what
    // d_in / d_ts currently hold
    unsigned char *d_raw = nullptr;
    size_t raw_cap = 0;
    double ts_lo = 0.0, ts_hi = 0.0;  // d_ts holds the stamps in seconds; consumers normalise with these (PreprocessParams::ts_normalise)
    bool ts_raw = false;
    mutable HostStage stage;  // pinned staging for transfers from / to caller memory
    hipEvent_t chain_ready = nullptr;  // chained pre-steps: buffer 0 is complete (its background download may start)
    // background download of one buffer (kicp_pre_download_begin / _finish): its own stream, pinned landing area and event
    hipStream_t copy_stream = nullptr;
    hipEvent_t copy_done = nullptr;
    // the transfer goes in pieces, an event behind each: the helper thread moves piece i into the caller's memory while piece i + 1 is on
    // its way (round 5; one 3 MB copy behind one 3 MB transfer made the frame wait for the CPU's copy, ~60 us)
    static constexpr int kCopyPieces = 3;
    hipEvent_t copy_piece_done[kCopyPieces] = {};
    size_t copy_piece_bytes = 0;  // bytes per piece of the transfer in flight (0: one piece, only copy_done)
    unsigned char *copy_host = nullptr;
    size_t copy_cap = 0, copy_n = 0;
    size_t copy_points = 0;  // points the helper thread moves (set before the job is posted, never written while it runs; copy_n is what _finish REPORTS)
    int copy_buffer = -1;
    // kicp_pre_download_begin_into: a helper thread of the handle moves the landed bytes into the caller's memory while the
    // calling thread goes on with the pipeline (waits for the DMA's event, then one memcpy); _finish only joins it
    std::thread copy_thread;
    std::mutex copy_mutex;
    std::condition_variable copy_cv;
    int copy_state = 0;  // 0 idle, 1 job posted, 2 job done, -1 shut down
    double *copy_dst = nullptr;
    size_t copy_dst_points = 0;
    hipError_t copy_error = hipSuccess;
    unsigned long long *d_block_minmax = nullptr;
    size_t ingested_n = 0;
    bool ingested = false, ingested_stamps = false;
    // LOOK-AHEAD ingest (kicp_pre_ingest_ahead, round 5): the NEXT message is uploaded and decoded into a second slot (d_in2 / d_ts2) on
    // a stream of its own by the kicp_pre_frame_ingested call of the CURRENT one - while that call's kernels run and its thread would
    // only wait -, and the kicp_pre_ingest call for the same message then just swaps the slots.
    double *d_in2 = nullptr, *d_ts2 = nullptr;
    hipStream_t ahead_stream = nullptr;
    struct Ahead {
        const void *data = nullptr;
        size_t n = 0;
        kicp_cloud_layout layout{};
        bool has_pose = false;
        Pose pose{};
        int state = 0;  // 0 none | 1 announced | 2 in the second slot
        double lo = 0.0, hi = 0.0;
        // what the first and the last 64 bytes of the message were when it was uploaded: a caller that reuses one receive buffer and hands
        // over ANOTHER message of the same size at the same address (against the contract) is caught instead of served the stale cloud
        unsigned char edge[128] = {};
        size_t edge_bytes = 0;
    } ahead;
    unsigned long long ahead_hits = 0;  // kicp_pre_ingest calls that found their message decoded ahead (kicp_pre_ahead_hits)
    JobThread ahead_thread;             // queues the look-ahead upload beside the calling thread's own kernels
    hipStream_t ingest_stream2 = nullptr;  // this call's message: its pieces' decodes alternate between the handle's stream and this one
    bool ahead_job_out = false;         // ... and is only waited for where its result (or a buffer it uses) is needed: ahead_join
    HostStage stage_ahead;              // its own pinned staging buffer (the calling thread goes on using `stage` meanwhile)
    // the chained pre-steps hand the WHOLE download of buffer 0 to the helper thread (its dozen API calls cost the calling thread ~40 us):
    bool copy_job_begins = false;       // the posted job starts with download_queue(0, copy_job_n, after chain_ready)
    bool copy_job_push = false;         // the frame arrives as k_push_frame's pieces, announced in h_rec[16..]: the job only follows them
    uint32_t copy_push_piece = 0, copy_push_seq = 0;
    size_t copy_job_n = 0;
};
namespace {
int ahead_join(kicp_pre *p);
int pre_ensure(kicp_pre *p, size_t n) {
    if (n <= p->cap_n) return KICP_OK;
    const size_t cap = n + n / 4 + 1024;
    if (int rc = ahead_join(p)) return rc;
    if (p->ahead_stream) HIP_TRY(hipStreamSynchronize(p->ahead_stream));
    p->ahead.state = 0;  // (a cloud waiting in the second slot goes with it: its kicp_pre_ingest call uploads it again)
    hipFree(p->d_in), hipFree(p->d_ts), hipFree(p->d_in2), hipFree(p->d_ts2), hipFree(p->d_staged), hipFree(p->d_flags), hipFree(p->d_block_counts), hipFree(p->d_table);
    hipFree(p->d_table2), hipFree(p->d_counts1), hipFree(p->d_counts2), hipFree(p->d_block_minmax);
    p->d_in = p->d_ts = p->d_in2 = p->d_ts2 = p->d_staged = nullptr, p->d_flags = p->d_block_counts = nullptr, p->d_table = nullptr, p->cap_n = 0;
    p->d_table2 = nullptr, p->d_counts1 = p->d_counts2 = nullptr, p->d_block_minmax = nullptr;
    const size_t slots = reference_bucket_count(cap);  // >= the reference's bucket count for every frame of <= cap points
    HIP_TRY(hipMalloc(&p->d_in, cap * 24));
    HIP_TRY(hipMalloc(&p->d_ts, cap * 8));
    HIP_TRY(hipMalloc(&p->d_in2, cap * 24));
    HIP_TRY(hipMalloc(&p->d_ts2, cap * 8));
    HIP_TRY(hipMalloc(&p->d_staged, cap * 24));
    HIP_TRY(hipMalloc(&p->d_flags, cap * 4));
    HIP_TRY(hipMalloc(&p->d_block_counts, (std::max(cap, slots) / 256 + 2) * 4));
    HIP_TRY(hipMalloc(&p->d_table, slots * 20));
    HIP_TRY(hipMalloc(&p->d_table2, slots * 20));
    HIP_TRY(hipMalloc(&p->d_counts1, (slots / 256 + 2) * 4));
    HIP_TRY(hipMalloc(&p->d_counts2, (slots / 256 + 2) * 4));
    HIP_TRY(hipMalloc(&p->d_block_minmax, (cap / 256 + 2) * 16));
    p->cap_n = cap, p->table_slots = slots, p->table_clean = false;
    return KICP_OK;
}
int pre_ensure_buf(kicp_pre *p, int b, size_t n) {
    if (b == 2) p->src_on_host = false;  // (every path that refills a buffer comes through here first)
    if (n <= p->buf_cap[b]) return KICP_OK;
    if (p->buf[b]) HIP_TRY(hipFree(p->buf[b]));
    p->buf[b] = nullptr;
    const size_t cap = n + n / 4 + 1024;
    HIP_TRY(hipMalloc(&p->buf[b], cap * 24));
    p->buf_cap[b] = cap;
    return KICP_OK;
}
// the survivor count (misc[0]) and the range flag (misc[1]) of the kernels queued so far -> buf_n[dst]
int pre_finish(kicp_pre *p, int dst, size_t *out_n) {
    uint32_t misc[3] = {0, 0, 0};
    HIP_TRY(hipMemcpyAsync(misc, p->d_misc, sizeof misc, hipMemcpyDeviceToHost, p->stream));
    HIP_TRY(hipStreamSynchronize(p->stream));
    p->last_max_probe = misc[2];
    if (misc[1]) {  // report once: the flag must not poison the calls that follow on this handle
        HIP_TRY(hipMemsetAsync(p->d_misc + 1, 0, 4, p->stream));
        return fail(KICP_ERR_CAPACITY, "a voxel coordinate left the +-2^20 range of the downsampling table");
    }
    p->buf_n[dst] = misc[0];
    if (out_n) *out_n = misc[0];
    return KICP_OK;
}
// flags + block counts are in place: scan, compact staged -> buffer dst, return the survivor count
int pre_compact(kicp_pre *p, const double *staged, size_t n, int dst, size_t *out_n) {
    const uint32_t grid = static_cast<uint32_t>((n + 255) / 256);
    if (int rc = pre_ensure_buf(p, dst, n)) return rc;
    const int raw = grid <= kFusedScanBlocks ? 1 : 0;  // (frame-sized grids: every workgroup adds up the counts before it itself)
    if (!raw) hipLaunchKernelGGL(k_scan_blocks, dim3(1), dim3(1024), 0, p->stream, p->d_block_counts, grid, p->d_misc);
    hipLaunchKernelGGL(k_compact, dim3(grid), dim3(256), 0, p->stream, staged, static_cast<const uint32_t *>(p->d_flags), static_cast<const uint32_t *>(p->d_block_counts), raw,
                       p->d_misc, static_cast<uint32_t>(n), p->buf[dst]);
    HIP_TRY(hipGetLastError());
    return pre_finish(p, dst, out_n);
}
// k_preprocess over what d_in / d_ts hold, then compaction into buffer dst
int pre_run_preprocess(kicp_pre *p, size_t n, bool do_deskew, const double relative_motion_qt[7], const double lidar_to_base_qt[7],
                       double max_range, double min_range, int dst_buffer, size_t *out_n) {
    if (n == 0) {
        p->buf_n[dst_buffer] = 0;
        if (out_n) *out_n = 0;
        return KICP_OK;
    }
    PreprocessParams pp{};
    pp.in = p->d_in, pp.timestamps = p->d_ts, pp.n = static_cast<uint32_t>(n), pp.deskew = do_deskew ? 1 : 0;
    const Pose rel = pose_from(relative_motion_qt);
    pose_log(rel, pp.omega);
    pp.motion_inverse = pose_inverse(rel), pp.lidar_to_base = pose_from(lidar_to_base_qt);
    pp.max_range = max_range, pp.min_range = min_range;
    pp.ts_normalise = p->ts_raw ? 1 : 0, pp.ts_lo = p->ts_lo, pp.ts_hi = p->ts_hi;
    pp.flags = p->d_flags, pp.staged = p->d_staged, pp.block_counts = p->d_block_counts;
    hipLaunchKernelGGL(k_preprocess, dim3(static_cast<uint32_t>((n + 255) / 256)), dim3(256), 0, p->stream, pp);
    return pre_compact(p, p->d_staged, n, dst_buffer, out_n);
}
}  // namespace
extern "C" {
int kicp_pre_create(int device, kicp_pre **out) {
    if (!out) return fail(KICP_ERR_ARG, "null argument");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(KICP_ERR_HIP, "no HIP device visible: this library has no CPU fallback");
    if (device < 0 || device >= ndev) return fail(KICP_ERR_ARG, "device index out of range");
    if (int rc = set_device(device)) return rc;
    kicp_pre *p = new kicp_pre;
    p->device = device;
    hipError_t e = hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipMalloc(&p->d_misc, 64);  // [0] total [1] error [2] longest probe [3] ticket [4..6] the chained pre-steps' three counts [7], [8]: kicp_pre.hpp
    if (e == hipSuccess) e = hipMemset(p->d_misc, 0, 64);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&p->chain_ready, hipEventDisableTiming);
    if (e == hipSuccess) e = pinned_alloc(reinterpret_cast<void **>(&p->h_rec), 32 * sizeof(unsigned long long), hipHostMallocMapped | hipHostMallocCoherent);
    if (e == hipSuccess) std::memset(p->h_rec, 0, 32 * sizeof(unsigned long long));
    if (e == hipSuccess) e = hipMalloc(&p->d_ticket, 16);
    if (e == hipSuccess) e = hipMemset(p->d_ticket, 0, 16);
    if (e == hipSuccess) e = hipMalloc(&p->d_push_tickets, kPushPieces * 8);
    if (e == hipSuccess) e = hipMemset(p->d_push_tickets, 0, kPushPieces * 8);
    if (e != hipSuccess) {
        kicp_pre_destroy(p);
        return fail(KICP_ERR_HIP, std::string("kicp_pre_create: ") + hipGetErrorString(e));
    }
    *out = p;
    return KICP_OK;
}
void kicp_pre_destroy(kicp_pre *p) {
    if (!p) return;
    hipSetDevice(p->device);
    if (p->stream) hipStreamSynchronize(p->stream);
    for (double *b : p->buf) hipFree(b);
    (void)ahead_join(p);
    p->ahead_thread.stop();
    p->stage_ahead.release();
    if (p->ahead_stream) hipStreamSynchronize(p->ahead_stream), hipStreamDestroy(p->ahead_stream);
    if (p->ingest_stream2) hipStreamSynchronize(p->ingest_stream2), hipStreamDestroy(p->ingest_stream2);
    if (p->h_rec) hipHostFree(p->h_rec);
    if (p->h_src) hipHostFree(p->h_src);
    hipFree(p->d_in), hipFree(p->d_ts), hipFree(p->d_in2), hipFree(p->d_ts2), hipFree(p->d_staged), hipFree(p->d_flags), hipFree(p->d_block_counts), hipFree(p->d_table);
    hipFree(p->d_table2), hipFree(p->d_counts1), hipFree(p->d_counts2);
    hipFree(p->d_misc), hipFree(p->d_raw), hipFree(p->d_block_minmax), hipFree(p->d_ticket), hipFree(p->d_push_tickets);
    p->stage.release();
    if (p->copy_thread.joinable()) {  // the helper thread finishes the job it has, then leaves
        {
            std::unique_lock<std::mutex> lock(p->copy_mutex);
            p->copy_cv.wait(lock, [p] { return p->copy_state != 1; });
            p->copy_state = -1;
        }
        p->copy_cv.notify_all();
        p->copy_thread.join();
    }
    if (p->copy_stream) hipStreamSynchronize(p->copy_stream), hipStreamDestroy(p->copy_stream);
    if (p->copy_done) hipEventDestroy(p->copy_done);
    for (hipEvent_t e : p->copy_piece_done)
        if (e) hipEventDestroy(e);
    if (p->chain_ready) hipEventDestroy(p->chain_ready);
    if (p->copy_host) hipHostFree(p->copy_host);
    if (p->stream) hipStreamDestroy(p->stream);
    delete p;
}
int kicp_pre_preprocess(kicp_pre *p, const double *frame_xyz, size_t n, const double *timestamps, size_t n_timestamps,
                        const double relative_motion_qt[7], const double lidar_to_base_qt[7], double max_range, double min_range,
                        int deskew, int dst_buffer, size_t *out_n) {
    KICP_TRACE_CALL();
    if (!p || (!frame_xyz && n) || !relative_motion_qt || !lidar_to_base_qt || dst_buffer < 0 || dst_buffer >= KICP_PRE_BUFFERS)
        return fail(KICP_ERR_ARG, "bad argument");
    const bool do_deskew = deskew && n_timestamps != 0;  // Preprocessing.cpp: `if (deskew_ && !timestamps.empty())`
    if (do_deskew && (!timestamps || n_timestamps < n)) return fail(KICP_ERR_ARG, "one timestamp per point is required for deskewing");
    if (int rc = set_device(p->device)) return rc;
    if (n > 0x7FFFFFF0ull / 3) return fail(KICP_ERR_CAPACITY, "frame too large");
    if (n) {
        if (int rc = pre_ensure(p, n)) return rc;
        p->ingested = false, p->ts_raw = false;  // d_in / d_ts are overwritten (the caller's stamps are normalised already)
        if (int rc = stage_reserve(p->stage, n * 32, p->stream)) return rc;  // one buffer for both arrays
        if (int rc = staged_upload(p->stage, 0, p->d_in, frame_xyz, n * 24, p->stream)) return rc;
        if (do_deskew)
            if (int rc = staged_upload(p->stage, n * 24, p->d_ts, timestamps, n * 8, p->stream)) return rc;
    }
    return pre_run_preprocess(p, n, do_deskew, relative_motion_qt, lidar_to_base_qt, max_range, min_range, dst_buffer, out_n);
}
namespace {
int ingest_validate(const kicp_pre *p, const void *data, size_t n_points, const kicp_cloud_layout *layout) {
    if (!p || !layout || (!data && n_points)) return fail(KICP_ERR_ARG, "bad argument");
    const kicp_cloud_layout &L = *layout;
    const int st = L.stamp_datatype;
    if (st != 0 && st != KICP_FIELD_UINT32 && st != KICP_FIELD_FLOAT32 && st != KICP_FIELD_FLOAT64)
        return fail(KICP_ERR_ARG, "timestamp field type not supported");  // TimeStampHandler.cpp:103
    const uint32_t stamp_bytes = st == KICP_FIELD_FLOAT64 ? 8u : 4u;
    if (L.point_step == 0 || L.offset_x + 4ull > L.point_step || L.offset_y + 4ull > L.point_step || L.offset_z + 4ull > L.point_step ||
        (st != 0 && L.offset_stamp + static_cast<unsigned long long>(stamp_bytes) > L.point_step))
        return fail(KICP_ERR_ARG, "field offsets do not fit inside point_step");
    if (n_points > 0x7FFFFFF0ull / 3) return fail(KICP_ERR_CAPACITY, "cloud too large");
    return KICP_OK;
}
// Poll a tagged word of the handle's host record until it reads `want` - the kernels' way of saying "done" without a copy and a
// stream synchronisation behind them (~15 us of API time per frame).  Bounded: after 2 ms the stream is synchronised instead
// (whatever is wrong then surfaces as a HIP error, or the word is there after all).
int wait_word(const volatile unsigned long long *word, unsigned long long want, unsigned long long mask, hipStream_t stream) {
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spins = 0;; ++spins) {
        if ((*word & mask) == want) return KICP_OK;
        if ((spins & 255u) == 255u && std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() > 2.0) break;
        __builtin_ia32_pause();
    }
    HIP_TRY(hipStreamSynchronize(stream));
    if ((*word & mask) == want) return KICP_OK;
    return fail(KICP_ERR_HIP, "the pre-step kernels finished without handing their result over");
}
// Upload + decode of one message (n_points > 0, <= p->cap_n) on `stream` into (out_xyz, out_ts: stamps in SECONDS) and wait for it:
// the CPU copies the message into the pinned staging buffer piece by piece and launches k_ingest behind each piece, which decodes
// the records straight out of host memory (16 bytes per lane for the usual x y z t layout) while the CPU copies the next piece;
// the cloud's last workgroup folds the stamps' extrema and says so in the host record `slot` (0: this call's, 1: the look-ahead's).
// Round 5 pulled the bytes into HBM (k_pull_bytes per piece), decoded them with one more launch, normalised the stamps with another
// and copied the extrema back: four stream operations and 6 MB of traffic more per frame.
constexpr size_t kIngestPiece = 512u << 10;
int ingest_run(kicp_pre *p, const void *data, size_t n_points, const kicp_cloud_layout &L, const Pose *sensor_pose, hipStream_t stream, double *out_xyz,
               double *out_ts, HostStage &stage, int slot, double *out_lo, double *out_hi) {
    const int st = L.stamp_datatype;
    const uint32_t stamp_bytes = st == KICP_FIELD_FLOAT64 ? 8u : 4u;
    const size_t bytes = n_points * static_cast<size_t>(L.point_step);
    if (int rc = stage_begin(stage, bytes, stream)) return rc;
    if (slot == 0) trace_lap("staging buffer free");
    const bool direct = stage.dev != nullptr;  // the device reads the staging buffer itself
    if (!direct && bytes > p->raw_cap) {
        HIP_TRY(hipDeviceSynchronize());  // (rare: the buffer grows; nothing may still be reading the old one)
        hipFree(p->d_raw);
        p->d_raw = nullptr, p->raw_cap = 0;
        HIP_TRY(hipMalloc(&p->d_raw, bytes + bytes / 4 + 4096));
        p->raw_cap = bytes + bytes / 4 + 4096;
    }
    IngestParams ip{};
    ip.point_step = L.point_step;
    ip.off_x = L.offset_x, ip.off_y = L.offset_y, ip.off_z = L.offset_z, ip.off_t = L.offset_stamp, ip.stamp_type = st;
    ip.transform = sensor_pose ? 1 : 0;
    // (the staging buffer and d_raw are 256-byte aligned; a piece starts at a multiple of 256 records)
    ip.aligned = (L.point_step % 4 == 0 && L.offset_x % 4 == 0 && L.offset_y % 4 == 0 && L.offset_z % 4 == 0 &&
                  (st == 0 || (L.offset_stamp % stamp_bytes == 0 && L.point_step % stamp_bytes == 0))) ? 1 : 0;
    if (sensor_pose) ip.T = *sensor_pose;
    ip.out_xyz = out_xyz, ip.out_stamps = out_ts, ip.block_minmax = p->d_block_minmax;
    ip.total_blocks = static_cast<uint32_t>((n_points + 255) / 256);
    ip.ticket = p->d_ticket + slot;
    // a look-ahead message: few workgroups going round its tiles, so that at most ~200 KB of it are on request at any time (k_ingest;
    // 24 / 48 / 64 workgroups and 0.5 / 1 / 4 MB pieces measured within the boxes' noise of each other, 8 and 512 clearly worse)
    static const uint32_t ahead_wgs = [] { const char *e = std::getenv("KICP_PRE_AHEAD_WGS"); return e && *e ? static_cast<uint32_t>(std::max(1, std::atoi(e))) : 48u; }();
    static const size_t ahead_piece = [] { const char *e = std::getenv("KICP_PRE_AHEAD_PIECE_KB"); return (e && *e ? static_cast<size_t>(std::max(64, std::atoi(e))) : 512u) << 10; }();
    const uint32_t wgs_cap = slot == 1 ? ahead_wgs : 0xFFFFFFFFu;
    unsigned long long *rec = p->h_rec + 8 + 4 * slot;
    ip.host_rec = rec, ip.seq = ++p->ingest_seq;
    // pieces, so that the GPU decodes piece k while the CPU copies piece k + 1 (a look-ahead message too: as ONE launch behind the
    // whole 2 MB copy it was not there when the next frame asked for it)
    const size_t piece_records = std::max<size_t>(256, (slot == 1 ? ahead_piece : kIngestPiece) / L.point_step / 256 * 256);
    // What the call waits for is the link: a kernel's loads pull ~30 GB/s out of host memory (2 MB: ~65 us from the first launch to the
    // record), whoever copies and however the launches are arranged.  Measured and not kept: a helper thread sharing the copy into the
    // staging buffer (all four launches queued by 47 us instead of 66 - the record arrives at 87 us either way), the copy engine moving
    // the pieces into HBM first (106 us).
    // this call's message: consecutive pieces' decodes go to two streams in turn - on one stream each launch waits for the one before
    // it to drain; side by side they keep the link busy (2 % of the frame)
    static const bool two_streams = [] { const char *e = std::getenv("KICP_PRE_INGEST_STREAMS"); return !(e && *e == '1'); }();
    hipStream_t lanes[2] = {stream, stream};
    if (slot == 0 && two_streams) {
        if (!p->ingest_stream2) HIP_TRY(hipStreamCreateWithFlags(&p->ingest_stream2, hipStreamNonBlocking));
        lanes[1] = p->ingest_stream2;
    }
    unsigned piece_index = 0;
    for (size_t first = 0; first < n_points; first += piece_records)  // (one ticket per workgroup of every launch)
        p->ticket_drawn[slot] += std::min<uint32_t>(wgs_cap, static_cast<uint32_t>((std::min(piece_records, n_points - first) + 255) / 256));
    ip.ticket_done = p->ticket_drawn[slot];
    for (size_t first = 0; first < n_points; first += piece_records) {
        const size_t count = std::min(piece_records, n_points - first), off = first * L.point_step, len = count * L.point_step;
        std::memcpy(stage.p + off, static_cast<const unsigned char *>(data) + off, len);
        if (direct) {
            ip.raw = stage.dev + off;
        } else {
            HIP_TRY(hipMemcpyAsync(p->d_raw + off, stage.p + off, len, hipMemcpyHostToDevice, lanes[piece_index & 1u]));
            ip.raw = p->d_raw + off;
        }
        ip.first = static_cast<uint32_t>(first), ip.n = static_cast<uint32_t>(count);
        hipLaunchKernelGGL(k_ingest, dim3(std::min<uint32_t>(wgs_cap, static_cast<uint32_t>((count + 255) / 256))), dim3(256), 0, lanes[piece_index & 1u], ip);
        ++piece_index;
        if (slot == 0) trace_lap("piece copied, its decode launched");
    }
    HIP_TRY(hipGetLastError());
    if (int rc = wait_word(rec + 2, ip.seq, ~0ull, stream)) return rc;  // (`data` and the staging buffer are free again behind this)
    if (slot == 0) trace_lap("the decoded cloud's record at the host");
    *out_lo = *out_hi = 0.0;
    if (st != 0) *out_lo = ordered_value(rec[0]), *out_hi = ordered_value(rec[1]);
    return KICP_OK;
}
bool same_layout(const kicp_cloud_layout &a, const kicp_cloud_layout &b) {
    return a.point_step == b.point_step && a.offset_x == b.offset_x && a.offset_y == b.offset_y && a.offset_z == b.offset_z && a.stamp_datatype == b.stamp_datatype &&
           (a.stamp_datatype == 0 || a.offset_stamp == b.offset_stamp);
}
// the announced next message goes into the second slot now (the job the chained pre-steps post to the look-ahead thread)
int ahead_run(kicp_pre *p) {
    kicp_pre::Ahead &a = p->ahead;
    if (a.state != 1 || a.n == 0 || a.n > p->cap_n) return KICP_OK;  // (a cloud that does not fit the buffers is left to its kicp_pre_ingest call)
    if (!p->ahead_stream) {
        // lowest priority: background work - and the runtime keeps a pool of hardware queues per priority, so that this stream does
        // not share a queue with the frame's way back (same pool: the look-ahead's kernel waited for the push kernel's 60 us)
        int least = 0, greatest = 0;
        static const bool low = [] { const char *e = std::getenv("KICP_PRE_AHEAD_PRIORITY"); return !(e && *e == '0'); }();
        if (!(low && hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && least != greatest &&
              hipStreamCreateWithPriority(&p->ahead_stream, hipStreamNonBlocking, least) == hipSuccess)) {
            (void)hipGetLastError();  // (no priorities on this platform: an ordinary stream)
            p->ahead_stream = nullptr;
            HIP_TRY(hipStreamCreateWithFlags(&p->ahead_stream, hipStreamNonBlocking));
        }
    }
    if (int rc = ingest_run(p, a.data, a.n, a.layout, a.has_pose ? &a.pose : nullptr, p->ahead_stream, p->d_in2, p->d_ts2, p->stage_ahead, 1, &a.lo, &a.hi)) return rc;
    const size_t bytes = a.n * static_cast<size_t>(a.layout.point_step);
    a.edge_bytes = std::min<size_t>(64, bytes);
    std::memcpy(a.edge, a.data, a.edge_bytes);
    std::memcpy(a.edge + 64, static_cast<const unsigned char *>(a.data) + bytes - a.edge_bytes, a.edge_bytes);
    a.state = 2;
    return KICP_OK;
}
// the look-ahead upload, if one is out, must be over before its result is looked at or anything it uses changes hands
int ahead_join(kicp_pre *p) {
    if (!p->ahead_job_out) return KICP_OK;
    p->ahead_job_out = false;
    return p->ahead_thread.wait();
}
}  // namespace
int kicp_pre_ingest(kicp_pre *p, const void *data, size_t n_points, const kicp_cloud_layout *layout, const double sensor_pose_qt[7],
                    double *out_min_stamp, double *out_max_stamp) {
    KICP_TRACE_CALL();
    if (int rc = ingest_validate(p, data, n_points, layout)) return rc;
    const kicp_cloud_layout &L = *layout;
    const int st = L.stamp_datatype;
    if (int rc = set_device(p->device)) return rc;
    if (out_min_stamp) *out_min_stamp = 0.0;
    if (out_max_stamp) *out_max_stamp = 0.0;
    if (int rc = ahead_join(p)) return rc;  // (the announced message's upload has had the registration and the map update to finish behind)
    kicp_pre::Ahead &a = p->ahead;
    const bool pose_matches = sensor_pose_qt ? (a.has_pose && std::memcmp(&a.pose, sensor_pose_qt, 7 * sizeof(double)) == 0) : !a.has_pose;
    bool same_bytes = false;
    if (a.state == 2 && a.data == data && a.n == n_points && same_layout(a.layout, L) && pose_matches) {
        const size_t bytes = n_points * static_cast<size_t>(L.point_step);
        same_bytes = std::memcmp(a.edge, data, a.edge_bytes) == 0 &&
                     std::memcmp(a.edge + 64, static_cast<const unsigned char *>(data) + bytes - a.edge_bytes, a.edge_bytes) == 0;
    }
    if (same_bytes) {
        // the message was uploaded and decoded ahead (kicp_pre_ingest_ahead + the previous frame's chained pre-steps): take the slot
        std::swap(p->d_in, p->d_in2), std::swap(p->d_ts, p->d_ts2);
        a.state = 0, ++p->ahead_hits;
        p->ingested = true, p->ingested_n = n_points, p->ingested_stamps = st != 0 && n_points != 0;
        p->ts_raw = p->ingested_stamps, p->ts_lo = a.lo, p->ts_hi = a.hi;
        if (out_min_stamp) *out_min_stamp = a.lo;
        if (out_max_stamp) *out_max_stamp = a.hi;
        return KICP_OK;
    }
    a.state = 0;  // (another message than the one announced: the announcement is void)
    p->ingested = true, p->ingested_n = n_points, p->ingested_stamps = st != 0 && n_points != 0;
    p->ts_raw = false, p->ts_lo = p->ts_hi = 0.0;
    if (n_points == 0) return KICP_OK;
    if (int rc = pre_ensure(p, n_points)) return rc;
    Pose T{};
    if (sensor_pose_qt) T = pose_from(sensor_pose_qt);
    double lo = 0.0, hi = 0.0;
    if (int rc = ingest_run(p, data, n_points, L, sensor_pose_qt ? &T : nullptr, p->stream, p->d_in, p->d_ts, p->stage, 0, &lo, &hi)) {
        p->ingested = false;
        return rc;
    }
    p->ts_raw = st != 0, p->ts_lo = lo, p->ts_hi = hi;
    if (out_min_stamp) *out_min_stamp = lo;
    if (out_max_stamp) *out_max_stamp = hi;
    return KICP_OK;
}
int kicp_pre_ingest_ahead(kicp_pre *p, const void *data, size_t n_points, const kicp_cloud_layout *layout, const double sensor_pose_qt[7]) {
    if (int rc = ingest_validate(p, data, n_points, layout)) return rc;
    if (int rc = ahead_join(p)) return rc;
    kicp_pre::Ahead &a = p->ahead;
    a.data = data, a.n = n_points, a.layout = *layout, a.has_pose = sensor_pose_qt != nullptr, a.state = n_points ? 1 : 0;
    if (sensor_pose_qt) a.pose = pose_from(sensor_pose_qt);
    return KICP_OK;
}
int kicp_pre_preprocess_ingested(kicp_pre *p, const double relative_motion_qt[7], const double lidar_to_base_qt[7], double max_range,
                                 double min_range, int deskew, int dst_buffer, size_t *out_n) {
    KICP_TRACE_CALL();
    if (!p || !relative_motion_qt || !lidar_to_base_qt || dst_buffer < 0 || dst_buffer >= KICP_PRE_BUFFERS)
        return fail(KICP_ERR_ARG, "bad argument");
    if (!p->ingested) return fail(KICP_ERR_ARG, "no ingested cloud: call kicp_pre_ingest first");
    if (int rc = set_device(p->device)) return rc;
    return pre_run_preprocess(p, p->ingested_n, deskew && p->ingested_stamps, relative_motion_qt, lidar_to_base_qt, max_range, min_range,
                              dst_buffer, out_n);
}
int kicp_pre_ingested(const kicp_pre *p, double *out_xyz, double *out_stamps, size_t cap_points, size_t *out_n, int *out_has_stamps) {
    if (!p) return fail(KICP_ERR_ARG, "bad argument");
    if (!p->ingested) return fail(KICP_ERR_ARG, "no ingested cloud: call kicp_pre_ingest first");
    if (int rc = set_device(p->device)) return rc;
    const size_t k = std::min(p->ingested_n, cap_points);
    if (k && out_xyz)
        if (int rc = staged_download(p->stage, out_xyz, p->d_in, k * 24, p->stream)) return rc;
    if (k && out_stamps && p->ingested_stamps) {
        if (int rc = staged_download(p->stage, out_stamps, p->d_ts, k * 8, p->stream)) return rc;
        // TimeStampHandler.cpp:121-128 - the two fp64 operations the device applies where it consumes a stamp (kicp_pre.hpp: ts_normalise)
        if (p->ts_raw)
            for (size_t i = 0; i < k; ++i) out_stamps[i] = (out_stamps[i] - p->ts_lo) / (p->ts_hi - p->ts_lo);
    }
    if (out_n) *out_n = p->ingested_n;
    if (out_has_stamps) *out_has_stamps = p->ingested_stamps ? 1 : 0;
    return KICP_OK;
}
int kicp_pre_voxel_downsample(kicp_pre *p, int src, double voxel_size, int dst, size_t *out_n) {
    KICP_TRACE_CALL();
    if (!p || src < 0 || src >= KICP_PRE_BUFFERS || dst < 0 || dst >= KICP_PRE_BUFFERS || src == dst || !(voxel_size > 0.0))
        return fail(KICP_ERR_ARG, "bad argument");
    if (int rc = set_device(p->device)) return rc;
    const size_t n = p->buf_n[src];
    if (n == 0) {
        p->buf_n[dst] = 0;
        if (out_n) *out_n = 0;
        return KICP_OK;
    }
    if (int rc = pre_ensure(p, n)) return rc;
    if (int rc = pre_ensure_buf(p, dst, n)) return rc;
    // the reference's table: tsl::robin_map::reserve(frame.size()) buckets, ideal bucket = std::hash<Voxel> & mask (kicp_pre.hpp)
    const size_t slots = reference_bucket_count(n);
    if (slots > p->table_slots || slots > 0x80000000ull || n > slots / 2)  // n > 2^24: float rounding in reserve() lets the reference re-hash mid-way
        return fail(KICP_ERR_CAPACITY, "frame too large for the downsampling table");
    DownsampleParams dp{};
    dp.in = p->buf[src], dp.n = static_cast<uint32_t>(n), dp.voxel_size = voxel_size, dp.mask = static_cast<uint32_t>(slots - 1);
    dp.keys = reinterpret_cast<unsigned long long *>(p->d_table);
    dp.min_index = reinterpret_cast<uint32_t *>(p->d_table + slots * 8);
    dp.order = dp.min_index + slots, dp.home_at = dp.order + slots;
    dp.block_counts = p->d_block_counts, dp.error = p->d_misc + 1, dp.probe_max = p->d_misc + 2;
    // keys free, no winner yet, buckets of the replay free: all bytes 0xFF.  The gather step puts every bucket it has read back into
    // that state, so only the first call after an allocation (or after a failure) clears the table - one stream operation (~5 us) less
    // per call; the layout inside the buffer depends on `slots`, hence "every byte", not "these fields".
    if (!p->table_clean) HIP_TRY(hipMemsetAsync(p->d_table, 0xFF, p->table_slots * 20, p->stream));
    p->table_clean = false;  // (until this call is known to have run to its end)
    const uint32_t grid = static_cast<uint32_t>((n + 255) / 256), sgrid = static_cast<uint32_t>((slots + 255) / 256);
    hipLaunchKernelGGL(k_downsample_claim, dim3(grid), dim3(256), 0, p->stream, dp);
    hipLaunchKernelGGL(k_downsample_replay, dim3(sgrid), dim3(256), 0, p->stream, dp);
    const int raw = sgrid <= kFusedScanBlocks ? 1 : 0;
    if (!raw) hipLaunchKernelGGL(k_scan_blocks, dim3(1), dim3(1024), 0, p->stream, p->d_block_counts, sgrid, p->d_misc);
    hipLaunchKernelGGL(k_downsample_gather, dim3(sgrid), dim3(256), 0, p->stream, dp, static_cast<const uint32_t *>(p->d_block_counts), raw, p->d_misc, p->buf[dst]);
    HIP_TRY(hipGetLastError());
    const int rc = pre_finish(p, dst, out_n);
    p->table_clean = rc == KICP_OK;
    if (rc == KICP_OK && p->last_max_probe > p->probe_limit) {
        // The survivors are the reference's; their ORDER is only the reference's while its robin_map never grew mid-way, which it
        // does when a probe exceeds the container's limit - a re-hash the parallel replay does not model.  Say so instead of
        // handing out an order that may differ (the points are in place, the caller decides).
        last_error() = "VoxelDownsample: a robin-hood probe of " + std::to_string(p->last_max_probe) + " buckets exceeds the limit of " +
                       std::to_string(p->probe_limit) + " at which tsl::robin_map grows its table: the output ORDER may differ from the reference's";
        return KICP_WARN_TABLE_ORDER;
    }
    return rc;
}
static int download_begin_impl(kicp_pre *p, int buffer, size_t n, hipEvent_t after);
static int download_settle(kicp_pre *p);
static int download_queue(kicp_pre *p, int buffer, size_t n, hipEvent_t after);
static int download_reserve(kicp_pre *p, size_t bytes);
static void copy_worker(kicp_pre *p);
// ---- the whole pre-step chain of one frame behind ONE host synchronisation (KinematicICP.cpp:54-62) ------------------------
// What d_in / d_ts hold (n_in points: an uploaded frame or an ingested cloud) is preprocessed into buffer 0, buffer 0 is
// downsampled into buffer 1 at `voxel_a`, buffer 1 into buffer 2 at `voxel_b`.  The survivor count of each step stays on the
// device and is the next step's input count (every launch is sized for n_in, every kernel derives the reference's bucket count
// from the real count itself); the three counts come back together at the end.  Buffer 0 - the preprocessed frame the pipeline
// returns - starts travelling to `out_frame_xyz` (room for n_in points; may be nullptr) as soon as it is complete, on the
// download stream, while the downsamples run.
static int pre_frame_chain(kicp_pre *p, size_t n_in, bool do_deskew, const double relative_motion_qt[7], const double lidar_to_base_qt[7], double max_range,
                           double min_range, double voxel_a, double voxel_b, double *out_frame_xyz, size_t cap_points, size_t counts[3]) {
    counts[0] = counts[1] = counts[2] = 0;
    p->buf_n[0] = p->buf_n[1] = p->buf_n[2] = 0;
    p->src_on_host = false;
    if (n_in == 0) return KICP_OK;
    if (!(voxel_a > 0.0) || !(voxel_b > 0.0)) return fail(KICP_ERR_ARG, "bad voxel size");
    // Buffers 1 and 3 take turns: what the PREVIOUS frame left in buffer 1 - the points a map update that was begun with
    // kicp_map_update_pose_device_begin may still be reading - stays untouched, as buffer 3, until the frame after this one.
    std::swap(p->buf[1], p->buf[3]), std::swap(p->buf_cap[1], p->buf_cap[3]), std::swap(p->buf_n[1], p->buf_n[3]);
    p->buf_n[1] = 0;
    for (int b = 0; b < 3; ++b)
        if (int rc = pre_ensure_buf(p, b, n_in)) return rc;
    const size_t slots_up = reference_bucket_count(n_in);
    if (slots_up > p->table_slots || slots_up > 0x80000000ull || n_in > slots_up / 2) return fail(KICP_ERR_CAPACITY, "frame too large for the downsampling table");
    const uint32_t grid = static_cast<uint32_t>((n_in + 255) / 256), sgrid = static_cast<uint32_t>((slots_up + 255) / 256);
    uint32_t *cnt = p->d_misc + 4;
    PreprocessParams pp{};
    pp.in = p->d_in, pp.timestamps = p->d_ts, pp.n = static_cast<uint32_t>(n_in), pp.deskew = do_deskew ? 1 : 0;
    const Pose rel = pose_from(relative_motion_qt);
    pose_log(rel, pp.omega);
    pp.motion_inverse = pose_inverse(rel), pp.lidar_to_base = pose_from(lidar_to_base_qt);
    pp.max_range = max_range, pp.min_range = min_range;
    pp.ts_normalise = p->ts_raw ? 1 : 0, pp.ts_lo = p->ts_lo, pp.ts_hi = p->ts_hi;
    pp.flags = p->d_flags, pp.staged = p->d_staged, pp.block_counts = p->d_block_counts;
    // the next message, if one was announced, goes up NOW, from a thread of its own: its 2 MB copy into the staging buffer and its
    // launches run beside this thread's queueing of the frame's kernels, the GPU decodes it on a stream of its own
    bool ahead_out = false;
    if (int rc = ahead_join(p)) return rc;
    if (p->ahead.state == 1 && p->ahead.n != 0 && p->ahead.n <= p->cap_n) {
        p->ahead_thread.post(p->device, [p] { return ahead_run(p); });
        ahead_out = p->ahead_job_out = true;  // (collected by the kicp_pre_ingest call of that message - or whoever needs its buffers first: ahead_join)
    }
    // buffer 0 is complete behind the event: its download overlaps the downsamples (n_in points: an upper bound)
    // (two halves: the event goes into the stream right behind the launch that completes buffer 0; everything else - a dozen API calls,
    //  ~15 us of this thread - waits until the chain's remaining launches are queued, so the device never idles on it)
    auto mark_frame_ready = [&]() -> int {
        if (out_frame_xyz) HIP_TRY(hipEventRecord(p->chain_ready, p->stream));
        return KICP_OK;
    };
    auto start_download = [&]() -> int {
        if (!out_frame_xyz) return KICP_OK;
        if (int rc = download_settle(p)) return rc;  // (an earlier download nobody collected)
        if (!p->copy_thread.joinable()) p->copy_thread = std::thread(copy_worker, p);
        if (int rc = download_reserve(p, n_in * 24)) return rc;
        bool pushed = false;
        static const int push_wgs = [] { const char *e = std::getenv("KICP_PRE_PUSH_WGS"); return e && *e ? std::atoi(e) : 16; }();
        if (push_wgs <= 0) {  // (A/B: the DMA engine moves the frame, in pieces, queued here instead of by the helper thread)
            if (int rc = download_queue(p, 0, n_in, p->chain_ready)) return rc;
        } else if (p->copy_host_dev) {
            // k_push_frame on the download stream, behind the event: the frame crosses PCIe as a kernel's stores, piece by piece, each
            // piece announced in host memory; the helper thread needs no HIP call to follow it
            PushParams q{};
            q.src = reinterpret_cast<const unsigned char *>(p->buf[0]), q.dst = p->copy_host_dev, q.n_points = p->d_misc + 4;
            q.piece_bytes = static_cast<uint32_t>(((n_in * 24 + kPushPieces - 1) / kPushPieces + 4095) / 4096 * 4096);
            const uint32_t push_grid = static_cast<uint32_t>(push_wgs);
            p->push_drawn += push_grid;
            q.tickets = p->d_push_tickets, q.ticket_done = p->push_drawn, q.host_flags = p->h_rec + 16, q.seq = ++p->push_seq;
            if (q.seq == 0u) q.seq = ++p->push_seq;
            HIP_TRY(hipStreamWaitEvent(p->copy_stream, p->chain_ready, 0));
            hipLaunchKernelGGL(k_push_frame, dim3(push_grid), dim3(256), 0, p->copy_stream, q);
            HIP_TRY(hipGetLastError());
            p->copy_push_piece = q.piece_bytes, p->copy_push_seq = q.seq;
            pushed = true;
        }
        {   // the helper thread moves the pieces into the caller's memory as they land (without the push: it also queues the DMA transfer, in pieces)
            std::lock_guard<std::mutex> lock(p->copy_mutex);
            p->copy_dst = out_frame_xyz, p->copy_dst_points = cap_points, p->copy_job_begins = !pushed && push_wgs > 0, p->copy_job_push = pushed, p->copy_job_n = n_in;
            p->copy_buffer = 0, p->copy_n = n_in, p->copy_points = n_in, p->copy_state = 1;
        }
        p->copy_cv.notify_all();
        return KICP_OK;
    };
    // the tables' arrays laid out for the LARGEST table this handle can hold: every layout agrees on "all bytes 0xFF = clean"
    if (!p->table_clean) {
        HIP_TRY(hipMemsetAsync(p->d_table, 0xFF, p->table_slots * 20, p->stream));
        HIP_TRY(hipMemsetAsync(p->d_table2, 0xFF, p->table_slots * 20, p->stream));
    }
    p->table_clean = false;
    auto table_at = [p](unsigned char *base) {
        DsTable t{};
        t.keys = reinterpret_cast<unsigned long long *>(base);
        t.min_index = reinterpret_cast<uint32_t *>(base + p->table_slots * 8);
        t.order = t.min_index + p->table_slots, t.home_at = t.order + p->table_slots;
        return t;
    };
    const auto t_start = std::chrono::steady_clock::now();
    bool unfused_tail = true;   // the downsamples still have to run as launches of their own, from buffer 0 on
    uint32_t misc[8] = {};
    if (p->fused && grid <= kFusedScanBlocks) {
        // FIVE launches (kicp_pre.hpp "the pre-steps of one frame in FIVE launches"); table A's size is guessed from the last frame's
        // survivor count (the first frame: from the input count - right whenever the crop leaves more than half of the points)
        size_t guess = n_in;
        if (p->spec_n0 != 0 && p->spec_n_in == 0) guess = std::min<size_t>(n_in, p->spec_n0);
        if (p->spec_n0 != 0 && p->spec_n_in != 0) guess = std::min<size_t>(n_in, static_cast<size_t>(static_cast<double>(p->spec_n0) * static_cast<double>(n_in) / static_cast<double>(p->spec_n_in) + 0.5));
        if (guess == 0) guess = n_in;
        FrameParams f{};
        f.pre = pp;
        f.A = table_at(p->d_table), f.B = table_at(p->d_table2);
        f.voxel_a = voxel_a, f.voxel_b = voxel_b;
        f.spec_mask = static_cast<uint32_t>(reference_bucket_count(guess) - 1);
        f.tiles_pts = grid, f.tiles_spec = (f.spec_mask >> 8) + 1u;
        f.counts1 = p->d_counts1, f.counts2 = p->d_counts2, f.misc = p->d_misc;
        f.buf0 = p->buf[0], f.buf1 = p->buf[1], f.buf2 = p->buf[2];
        if (n_in * 24 > p->h_src_cap) {
            if (p->h_src) HIP_TRY(hipHostFree(p->h_src));
            p->h_src = p->h_src_dev = nullptr, p->h_src_cap = 0;
            const size_t want = n_in * 24 + n_in * 6 + 4096;
            HIP_TRY(pinned_alloc(reinterpret_cast<void **>(&p->h_src), want, hipHostMallocDefault));
            p->h_src_cap = want;
            if (hipHostGetDevicePointer(reinterpret_cast<void **>(&p->h_src_dev), p->h_src, 0) != hipSuccess) p->h_src_dev = nullptr, (void)hipGetLastError();
        }
        static const int src_to_host = [] { const char *e = std::getenv("KICP_PRE_SRC_HOST"); return e && *e ? std::atoi(e) : 1; }();
        f.host_buf2 = src_to_host ? reinterpret_cast<double *>(p->h_src_dev) : nullptr;
        f.host_rec = p->h_rec, f.seq = ++p->chain_seq;
        if (f.seq == 0u) f.seq = ++p->chain_seq;  // (0 is what the device words hold before the first frame)
        // table B's size is known on the device only: its two launches walk the tiles there are with the workgroups they get - as
        // many as the previous frame's second table had tiles (twice that, for a frame that keeps more), sgrid at most
        const uint32_t grid_b = p->spec_tiles_b ? std::min(sgrid, std::max(32u, 2u * p->spec_tiles_b)) : std::min(sgrid, 1024u);
        hipLaunchKernelGGL(k_frame_pre, dim3(grid), dim3(256), 0, p->stream, f);
        // buffer 0 is complete behind this launch: its way back starts beside the downsamples' remaining launches.  The event rides on the
        // launch's own completion signal (hipExtLaunchKernelGGL's stop event) where that is on - an event recorded behind the launch
        // is a packet of its own, and cost the device ~7 us between this launch and the next
        static const bool event_on_launch = [] { const char *e = std::getenv("KICP_PRE_EVENT_ON_LAUNCH"); return !(e && *e == '0'); }();
        if (event_on_launch && out_frame_xyz) {
            hipExtLaunchKernelGGL(k_frame_l1_replay, dim3(std::max(grid, f.tiles_spec)), dim3(256), 0, p->stream, nullptr, p->chain_ready, 0, f);
        } else {
            hipLaunchKernelGGL(k_frame_l1_replay, dim3(std::max(grid, f.tiles_spec)), dim3(256), 0, p->stream, f);
            if (int rc = mark_frame_ready()) return rc;
        }
        hipLaunchKernelGGL(k_frame_l1_gather, dim3(f.tiles_spec), dim3(256), 0, p->stream, f);
        hipLaunchKernelGGL(k_frame_l2_replay, dim3(grid_b), dim3(256), 0, p->stream, f);
        hipLaunchKernelGGL(k_frame_l2_gather, dim3(grid_b), dim3(256), 0, p->stream, f);
        if (f.host_buf2) hipLaunchKernelGGL(k_frame_src_host, dim3(4), dim3(256), 0, p->stream, f);  // (nobody waits for it before kicp_pre_download(2))
        // (starting the frame's way back one, two or three launches later instead - the push's 3 MB of PCIe writes make the launches
        //  beside it 2-3 x as long, and the event costs the device 6 us between two launches - was measured: 1-4 % slower each)
        HIP_TRY(hipGetLastError());
        trace_lap("the frame's launches queued");
        if (int rc = start_download()) return rc;
        trace_lap("its way back queued");
        const unsigned long long tag = static_cast<unsigned long long>(f.seq) << 32, hi = 0xFFFFFFFF00000000ull;
        for (int w = 4; w >= 0; --w)
            if (int rc = wait_word(p->h_rec + w, tag, hi, p->stream)) return rc;
        const volatile unsigned long long *rec = p->h_rec;
        trace_lap("the chain's record at the host");
        ++p->fused_frames;
        misc[4] = static_cast<uint32_t>(rec[0]), misc[5] = static_cast<uint32_t>(rec[1]), misc[6] = static_cast<uint32_t>(rec[2]);
        misc[2] = static_cast<uint32_t>(rec[3]), misc[1] = static_cast<uint32_t>(rec[4]) & 1u;
        unfused_tail = (static_cast<uint32_t>(rec[4]) & 2u) != 0u;
        p->spec_n0 = misc[4], p->spec_n_in = static_cast<uint32_t>(n_in);
        p->src_on_host = !unfused_tail && f.host_buf2 != nullptr;
        p->src_seq = f.seq;
        p->spec_tiles_b = misc[5] ? static_cast<uint32_t>(reference_bucket_count(misc[5]) + 255) / 256u : 1u;
        if (unfused_tail) {  // a guess was wrong: the tables hold claims made under the wrong size; buffer 0 and its count are in place
            ++p->spec_misses;
            HIP_TRY(hipMemsetAsync(p->d_table, 0xFF, p->table_slots * 20, p->stream));
            HIP_TRY(hipMemsetAsync(p->d_table2, 0xFF, p->table_slots * 20, p->stream));
        }
    } else {
        hipLaunchKernelGGL(k_preprocess, dim3(grid), dim3(256), 0, p->stream, pp);
        const int raw_c = grid <= kFusedScanBlocks ? 1 : 0;
        if (!raw_c) hipLaunchKernelGGL(k_scan_blocks, dim3(1), dim3(1024), 0, p->stream, p->d_block_counts, grid, cnt + 0);
        hipLaunchKernelGGL(k_compact, dim3(grid), dim3(256), 0, p->stream, static_cast<const double *>(p->d_staged), static_cast<const uint32_t *>(p->d_flags),
                           static_cast<const uint32_t *>(p->d_block_counts), raw_c, cnt + 0, static_cast<uint32_t>(n_in), p->buf[0]);
        if (int rc = mark_frame_ready()) return rc;
        if (int rc = start_download()) return rc;
    }
    if (unfused_tail) {
        // the downsamples as three launches each; every step's survivor count stays on the device as the next step's input count
        const int raw_g = sgrid <= kFusedScanBlocks ? 1 : 0;
        DownsampleParams dp{};
        const DsTable A = table_at(p->d_table);
        dp.keys = A.keys, dp.min_index = A.min_index, dp.order = A.order, dp.home_at = A.home_at;
        dp.block_counts = p->d_block_counts, dp.error = p->d_misc + 1, dp.probe_max = p->d_misc + 2;
        for (int stage = 0; stage < 2; ++stage) {
            dp.in = p->buf[stage], dp.voxel_size = stage == 0 ? voxel_a : voxel_b, dp.n_dev = cnt + stage;
            dp.probe_max_sticky = stage == 0 ? nullptr : dp.probe_max;
            hipLaunchKernelGGL(k_downsample_claim, dim3(grid), dim3(256), 0, p->stream, dp);
            hipLaunchKernelGGL(k_downsample_replay, dim3(sgrid), dim3(256), 0, p->stream, dp);
            if (!raw_g) hipLaunchKernelGGL(k_scan_blocks, dim3(1), dim3(1024), 0, p->stream, p->d_block_counts, sgrid, cnt + stage + 1);
            hipLaunchKernelGGL(k_downsample_gather, dim3(sgrid), dim3(256), 0, p->stream, dp, static_cast<const uint32_t *>(p->d_block_counts), raw_g, cnt + stage + 1,
                               p->buf[stage + 1]);
        }
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(misc, p->d_misc, sizeof misc, hipMemcpyDeviceToHost, p->stream));
        HIP_TRY(hipStreamSynchronize(p->stream));
    }
    if (g_trace) {
        std::fprintf(stderr, "[kicp]   chained pre-steps: %zu -> %u -> %u -> %u points; %s, results at the host %.3f ms after the first launch; look-ahead upload: %s\n", n_in, misc[4], misc[5], misc[6],
                     !p->fused || grid > kFusedScanBlocks ? "unfused" : (unfused_tail ? "fused, table size guessed wrong: unfused downsamples" : "five launches"),
                     std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count(),
                     ahead_out ? "on its own thread, collected by the message's kicp_pre_ingest" : "none");
    }
    p->last_max_probe = misc[2];
    if (misc[1]) {
        HIP_TRY(hipMemsetAsync(p->d_misc + 1, 0, 4, p->stream));
        return fail(KICP_ERR_CAPACITY, "a voxel coordinate left the +-2^20 range of the downsampling table");
    }
    p->table_clean = true;
    for (int b = 0; b < 3; ++b) counts[b] = p->buf_n[b] = misc[4 + b];
    if (out_frame_xyz) p->copy_n = counts[0];  // (what _finish reports; the helper thread moves the upper bound, copy_points, and never reads this)
    if (p->last_max_probe > p->probe_limit) {
        last_error() = "VoxelDownsample: a robin-hood probe of " + std::to_string(p->last_max_probe) + " buckets exceeds the limit of " +
                       std::to_string(p->probe_limit) + " at which tsl::robin_map grows its table: the output ORDER may differ from the reference's";
        return KICP_WARN_TABLE_ORDER;
    }
    return KICP_OK;
}
int kicp_pre_frame_ingested(kicp_pre *p, const double relative_motion_qt[7], const double lidar_to_base_qt[7], double max_range, double min_range, int deskew,
                            double voxel_a, double voxel_b, double *out_frame_xyz, size_t cap_points, size_t out_counts[3]) {
    KICP_TRACE_CALL();
    if (!p || !relative_motion_qt || !lidar_to_base_qt || !out_counts) return fail(KICP_ERR_ARG, "bad argument");
    if (!p->ingested) return fail(KICP_ERR_ARG, "no ingested cloud: call kicp_pre_ingest first");
    if (out_frame_xyz && cap_points < p->ingested_n) return fail(KICP_ERR_ARG, "the frame's landing area must hold every ingested point");
    if (int rc = set_device(p->device)) return rc;
    return pre_frame_chain(p, p->ingested_n, deskew && p->ingested_stamps, relative_motion_qt, lidar_to_base_qt, max_range, min_range, voxel_a, voxel_b,
                           out_frame_xyz, cap_points, out_counts);
}
int kicp_pre_frame(kicp_pre *p, const double *frame_xyz, size_t n, const double *timestamps, size_t n_timestamps, const double relative_motion_qt[7],
                   const double lidar_to_base_qt[7], double max_range, double min_range, int deskew, double voxel_a, double voxel_b, double *out_frame_xyz,
                   size_t cap_points, size_t out_counts[3]) {
    KICP_TRACE_CALL();
    if (!p || (!frame_xyz && n) || !relative_motion_qt || !lidar_to_base_qt || !out_counts) return fail(KICP_ERR_ARG, "bad argument");
    const bool do_deskew = deskew && n_timestamps != 0;
    if (do_deskew && (!timestamps || n_timestamps < n)) return fail(KICP_ERR_ARG, "one timestamp per point is required for deskewing");
    if (out_frame_xyz && cap_points < n) return fail(KICP_ERR_ARG, "the frame's landing area must hold every input point");
    if (n > 0x7FFFFFF0ull / 3) return fail(KICP_ERR_CAPACITY, "frame too large");
    if (int rc = set_device(p->device)) return rc;
    if (n) {
        if (int rc = pre_ensure(p, n)) return rc;
        p->ingested = false, p->ts_raw = false;
        if (int rc = stage_reserve(p->stage, n * 32, p->stream)) return rc;
        if (int rc = staged_upload(p->stage, 0, p->d_in, frame_xyz, n * 24, p->stream)) return rc;
        if (do_deskew)
            if (int rc = staged_upload(p->stage, n * 24, p->d_ts, timestamps, n * 8, p->stream)) return rc;
    }
    return pre_frame_chain(p, n, do_deskew, relative_motion_qt, lidar_to_base_qt, max_range, min_range, voxel_a, voxel_b, out_frame_xyz, cap_points, out_counts);
}
unsigned long long kicp_pre_ahead_hits(const kicp_pre *p) { return p ? p->ahead_hits : 0ull; }
int kicp_pre_set_option(kicp_pre *p, const char *name, double value) {
    if (!p || !name) return fail(KICP_ERR_ARG, "bad argument");
    const std::string n(name);
    if (n == "fused") p->fused = value != 0.0;
    else if (n == "guess") p->spec_n0 = value > 0.0 ? static_cast<uint32_t>(value) : 0u, p->spec_n_in = 0u;  // (taken as it is for the next frame)
    else return fail(KICP_ERR_ARG, "unknown pre-step option: " + n);
    return KICP_OK;
}
double kicp_pre_get_option(const kicp_pre *p, const char *name) {
    if (!p || !name) return -1.0;
    const std::string n(name);
    if (n == "fused") return p->fused ? 1.0 : 0.0;
    if (n == "guess") return static_cast<double>(p->spec_n0);
    if (n == "fused_frames") return static_cast<double>(p->fused_frames);
    if (n == "guess_misses") return static_cast<double>(p->spec_misses);
    return -1.0;
}
size_t kicp_pre_ingested_count(const kicp_pre *p) { return (p && p->ingested) ? p->ingested_n : 0; }
unsigned int kicp_pre_last_max_probe(const kicp_pre *p) { return p ? p->last_max_probe : 0u; }
int kicp_pre_set_probe_limit(kicp_pre *p, unsigned int limit) {
    if (!p || limit == 0u) return fail(KICP_ERR_ARG, "bad argument");
    p->probe_limit = limit;
    return KICP_OK;
}
int kicp_pre_upload(kicp_pre *p, int buffer, const double *xyz, size_t n) {
    KICP_TRACE_CALL();
    if (!p || buffer < 0 || buffer >= KICP_PRE_BUFFERS || (!xyz && n)) return fail(KICP_ERR_ARG, "bad argument");
    if (int rc = set_device(p->device)) return rc;
    if (int rc = pre_ensure_buf(p, buffer, n ? n : 1)) return rc;
    if (n)
        if (int rc = staged_upload(p->stage, 0, p->buf[buffer], xyz, n * 24, p->stream)) return rc;
    HIP_TRY(hipStreamSynchronize(p->stream));
    p->buf_n[buffer] = n;
    return KICP_OK;
}
int kicp_pre_download(const kicp_pre *p, int buffer, double *out_xyz, size_t cap_points, size_t *out_n) {
    KICP_TRACE_CALL();
    if (!p || buffer < 0 || buffer >= KICP_PRE_BUFFERS) return fail(KICP_ERR_ARG, "bad argument");
    if (int rc = set_device(p->device)) return rc;
    const size_t n = p->buf_n[buffer], k = std::min(n, cap_points);
    if (k && out_xyz) {
        if (buffer == 2 && p->src_on_host) {  // (the fused chain leaves a copy in host memory)
            if (int rc = wait_word(p->h_rec + 5, static_cast<unsigned long long>(p->src_seq) << 32, 0xFFFFFFFF00000000ull, p->stream)) return rc;
            std::memcpy(out_xyz, p->h_src, k * 24);
        } else if (int rc = staged_download(p->stage, out_xyz, p->buf[buffer], k * 24, p->stream)) return rc;
    }
    if (out_n) *out_n = n;
    return KICP_OK;
}
// Background download: the copy runs on its own stream while the caller goes on with the next steps (downsampling,
// registration, map update), and lands in pinned memory; _finish waits for it and hands the points over.
// `n`: points to copy; `after`: an event on the pre-step stream the copy has to wait for (nullptr: the buffer is final already)
static int download_begin_impl(kicp_pre *p, int buffer, size_t n, hipEvent_t after);
int kicp_pre_download_begin(kicp_pre *p, int buffer) {
    KICP_TRACE_CALL();
    if (!p || buffer < 0 || buffer >= KICP_PRE_BUFFERS) return fail(KICP_ERR_ARG, "bad argument");
    if (int rc = set_device(p->device)) return rc;
    return download_begin_impl(p, buffer, p->buf_n[buffer], nullptr);
}
static int download_settle(kicp_pre *p) {
    if (!p->copy_stream) HIP_TRY(hipStreamCreateWithFlags(&p->copy_stream, hipStreamNonBlocking));
    if (!p->copy_done) HIP_TRY(hipEventCreateWithFlags(&p->copy_done, hipEventDisableTiming));
    if (p->copy_buffer >= 0) {  // an earlier download nobody collected: let it (and the helper thread's copy) finish first
        std::unique_lock<std::mutex> lock(p->copy_mutex);
        if (p->copy_state == 1 || p->copy_state == 2) {
            p->copy_cv.wait(lock, [p] { return p->copy_state == 2; });
            p->copy_state = 0;
        }
        lock.unlock();
        HIP_TRY(hipStreamSynchronize(p->copy_stream));
    }
    return KICP_OK;
}
static int download_queue(kicp_pre *p, int buffer, size_t n, hipEvent_t after);
static int download_begin_impl(kicp_pre *p, int buffer, size_t n, hipEvent_t after) {
    if (int rc = download_settle(p)) return rc;
    if (int rc = download_queue(p, buffer, n, after)) return rc;
    p->copy_buffer = buffer, p->copy_n = n, p->copy_points = n;
    return KICP_OK;
}
// the transfer itself (calling thread, or the helper thread for the chained pre-steps)
static int download_reserve(kicp_pre *p, size_t bytes) {
    if (bytes > p->copy_cap) {
        if (p->copy_host) HIP_TRY(hipHostFree(p->copy_host));
        p->copy_host = nullptr, p->copy_host_dev = nullptr, p->copy_cap = 0;
        HIP_TRY(pinned_alloc(reinterpret_cast<void **>(&p->copy_host), bytes + bytes / 2 + (1u << 20), hipHostMallocDefault));
        p->copy_cap = bytes + bytes / 2 + (1u << 20);
        if (hipHostGetDevicePointer(reinterpret_cast<void **>(&p->copy_host_dev), p->copy_host, 0) != hipSuccess) p->copy_host_dev = nullptr, (void)hipGetLastError();
    }
    return KICP_OK;
}
static int download_queue(kicp_pre *p, int buffer, size_t n, hipEvent_t after) {
    const size_t bytes = n * 24;
    if (int rc = download_reserve(p, bytes)) return rc;
    // the buffer's contents are final (every call that fills a buffer returns only after its kernels have finished) - or will be
    // once `after` has happened on the pre-step stream
    if (after) HIP_TRY(hipStreamWaitEvent(p->copy_stream, after, 0));
    p->copy_piece_bytes = 0;
    if (bytes >= (1u << 20)) {  // frame-sized: in pieces, so that the helper thread's copy runs behind the transfer instead of after it
        const size_t piece = ((bytes + kicp_pre::kCopyPieces - 1) / kicp_pre::kCopyPieces + 4095) / 4096 * 4096;
        for (int i = 0; i < kicp_pre::kCopyPieces; ++i) {
            if (!p->copy_piece_done[i]) HIP_TRY(hipEventCreateWithFlags(&p->copy_piece_done[i], hipEventDisableTiming));
            const size_t off = std::min(bytes, piece * i), len = std::min(piece, bytes - off);
            if (len) HIP_TRY(hipMemcpyAsync(p->copy_host + off, reinterpret_cast<const unsigned char *>(p->buf[buffer]) + off, len, hipMemcpyDeviceToHost, p->copy_stream));
            HIP_TRY(hipEventRecord(p->copy_piece_done[i], p->copy_stream));
        }
        p->copy_piece_bytes = piece;
    } else if (bytes) {
        HIP_TRY(hipMemcpyAsync(p->copy_host, p->buf[buffer], bytes, hipMemcpyDeviceToHost, p->copy_stream));
    }
    HIP_TRY(hipEventRecord(p->copy_done, p->copy_stream));
    return KICP_OK;
}
// wait for an event of the download WITHOUT going to sleep on it: a blocking wait wakes the thread by interrupt, tens of microseconds
// after the transfer has landed - more than the copy it is waiting to start takes; the wait is bounded (a frame's transfer takes
// ~100 us), after 5 ms the thread does block
static hipError_t spin_on_event(hipEvent_t ev) {
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        const hipError_t q = hipEventQuery(ev);
        if (q != hipErrorNotReady) return q;
        if (std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() > 5.0) return hipEventSynchronize(ev);
    }
}
static void copy_worker(kicp_pre *p) {
    hipSetDevice(p->device);
    bind_thread_near_gpu(p->device);
    std::unique_lock<std::mutex> lock(p->copy_mutex);
    for (;;) {
        p->copy_cv.wait(lock, [p] { return p->copy_state == 1 || p->copy_state == -1; });
        if (p->copy_state == -1) return;
        lock.unlock();
        hipError_t e = hipSuccess;
        if (p->copy_job_begins) {  // (chained pre-steps: the transfer is queued here, behind the event that says buffer 0 is complete)
            p->copy_job_begins = false;
            if (download_queue(p, 0, p->copy_job_n, p->chain_ready) != KICP_OK) e = hipErrorUnknown;
        }
        const size_t want = std::min(p->copy_points, p->copy_dst_points) * 24;  // (copy_points, copy_dst*, copy_piece_bytes: written before the job was posted)
        if (e != hipSuccess) {
        } else if (p->copy_job_push) {
            // k_push_frame's pieces: a flag per piece in host memory, (seq << 32) | bytes - a short piece is the last one
            const volatile unsigned long long *flags = p->h_rec + 16;
            const unsigned long long tag = static_cast<unsigned long long>(p->copy_push_seq) << 32;
            const auto t0 = std::chrono::steady_clock::now();
            double landed_us[kPushPieces] = {}, copied_us[kPushPieces] = {};
            int pieces_seen = 0;
            for (int i = 0; i < kPushPieces && e == hipSuccess; ++i) {
                unsigned long long f = 0;
                for (unsigned spins = 0;; ++spins) {
                    f = flags[i];
                    if ((f & 0xFFFFFFFF00000000ull) == tag) break;
                    if ((spins & 1023u) == 1023u && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 2.0) {
                        e = hipStreamSynchronize(p->copy_stream);  // (something is badly wrong: surface it instead of spinning for ever)
                        if (e == hipSuccess && (flags[i] & 0xFFFFFFFF00000000ull) != tag) e = hipErrorUnknown;
                        f = flags[i];
                        break;
                    }
                    __builtin_ia32_pause();
                }
                if (e != hipSuccess) break;
                const size_t len = static_cast<size_t>(f & 0xFFFFFFFFull), off = static_cast<size_t>(p->copy_push_piece) * i;
                if (g_trace) landed_us[i] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
                if (len && p->copy_dst && off < want) std::memcpy(reinterpret_cast<unsigned char *>(p->copy_dst) + off, p->copy_host + off, std::min(len, want - off));
                if (g_trace) copied_us[i] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count(), pieces_seen = i + 1;
                if (len < p->copy_push_piece) break;
            }
            if (g_trace) {  // (since this thread took the job: when each piece's flag was seen, when its copy into the caller's memory was over)
                std::string line = "[kicp]     the frame's way back, us (landed/copied):";
                char buf[48];
                for (int i = 0; i < pieces_seen; ++i) std::snprintf(buf, sizeof buf, " %.0f/%.0f", landed_us[i], copied_us[i]), line += buf;
                std::fprintf(stderr, "%s\n", line.c_str());
            }
        } else if (p->copy_piece_bytes && p->copy_dst) {
            for (int i = 0; i < kicp_pre::kCopyPieces && e == hipSuccess; ++i) {
                e = spin_on_event(p->copy_piece_done[i]);
                const size_t off = std::min(want, p->copy_piece_bytes * i), len = std::min(p->copy_piece_bytes, want - off);
                if (e == hipSuccess && len) std::memcpy(reinterpret_cast<unsigned char *>(p->copy_dst) + off, p->copy_host + off, len);
            }
        } else {
            e = spin_on_event(p->copy_done);
            if (e == hipSuccess && want && p->copy_dst) std::memcpy(p->copy_dst, p->copy_host, want);
        }
        lock.lock();
        p->copy_error = e, p->copy_state = 2;
        p->copy_cv.notify_all();
    }
}
int kicp_pre_download_begin_into(kicp_pre *p, int buffer, double *out_xyz, size_t cap_points) {
    if (int rc = kicp_pre_download_begin(p, buffer)) return rc;
    if (!p->copy_thread.joinable()) p->copy_thread = std::thread(copy_worker, p);
    {
        std::lock_guard<std::mutex> lock(p->copy_mutex);
        p->copy_dst = out_xyz, p->copy_dst_points = cap_points, p->copy_job_begins = false, p->copy_job_push = false, p->copy_state = 1;
    }
    p->copy_cv.notify_all();
    return KICP_OK;
}
int kicp_pre_download_finish(kicp_pre *p, int buffer, double *out_xyz, size_t cap_points, size_t *out_n) {
    KICP_TRACE_CALL();
    if (!p || buffer < 0 || buffer >= KICP_PRE_BUFFERS) return fail(KICP_ERR_ARG, "bad argument");
    if (p->copy_buffer != buffer) return fail(KICP_ERR_ARG, "no download of this buffer in flight: kicp_pre_download_begin first");
    if (int rc = set_device(p->device)) return rc;
    bool delivered = false;
    {
        std::unique_lock<std::mutex> lock(p->copy_mutex);
        if (p->copy_state == 1 || p->copy_state == 2) {  // the helper thread owns this download: wait for it
            p->copy_cv.wait(lock, [p] { return p->copy_state == 2; });
            p->copy_state = 0;
            HIP_TRY(p->copy_error);
            delivered = p->copy_dst == out_xyz || out_xyz == nullptr;
        }
    }
    if (!delivered) {
        HIP_TRY(hipEventSynchronize(p->copy_done));
        const size_t k = std::min(p->copy_n, cap_points);
        if (k && out_xyz) std::memcpy(out_xyz, p->copy_host, k * 24);
    }
    if (out_n) *out_n = p->copy_n;
    p->copy_buffer = -1;
    return KICP_OK;
}
const double *kicp_pre_device_ptr(const kicp_pre *p, int buffer, size_t *out_n) {
    if (!p || buffer < 0 || buffer >= KICP_PRE_BUFFERS) return nullptr;
    if (out_n) *out_n = p->buf_n[buffer];
    return p->buf[buffer];
}

}  // extern "C"
