/* kicp.h -- C-ABI of the MI355X-native kinematic-ICP registration hot path (libkicp_amd.so).
 *
 * Drop-in boundary #2 of SURVEY.md section 8(b): everything the reference's
 *   kinematic_icp::KinematicRegistration      (cpp/kinematic_icp/registration/Registration.hpp:32-50)
 *   kiss_icp::VoxelHashMap (kiss-icp v1.2.0)  (used at registration/Registration.cpp:63,74,152,157,
 *                                              pipeline/KinematicICP.hpp:79,88,92 and KinematicICP.cpp:79)
 * need from a backend, as plain C: opaque handles, pointers and sizes, no C++/torch types - plus, further down, the
 * steps either side of that path in KinematicICP::RegisterFrame: kicp_pre_* for the wire-format
 * ingest, deskew + crop + downsample (pipeline/KinematicICP.cpp:31-62, ros/.../RosUtils.cpp:30-39,
 * TimeStampHandler.cpp:57-128) and kicp_map_update_pose_device / kicp_map_pointcloud for the map side (KinematicICP.cpp:79,
 * KinematicICP.hpp:92).
 *
 * Conventions
 *   points : contiguous AoS float64 xyz (the memory layout of std::vector<Eigen::Vector3d>, so
 *            `frame.data()->data()` can be passed without a copy).
 *   poses  : double[7] = [qx, qy, qz, qw, tx, ty, tz] (Sophus::SE3d parameter order).
 *   return : 0 = OK;  >0 = warning, result still follows the reference convention (e.g. NaN pose on
 *            zero correspondences, Registration.cpp:56-58,119-125);  <0 = backend error, outputs untouched.
 *   threads: one handle is used by one host thread at a time (the reference is called from a single ROS
 *            executor thread; SURVEY.md section 8b "Threading").
 * There is NO CPU fallback: if no gfx950 device / HIP runtime is usable, *_create fails with KICP_ERR_HIP.
 */
#ifndef KICP_H_
#define KICP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KICP_VERSION 100
#define KICP_MAX_LOG_PASSES 32

enum {
    KICP_OK = 0,
    KICP_WARN_NO_CORRESPONDENCES = 1, /* N_corr == 0 in some pass -> NaN pose, as the reference produces */
    KICP_WARN_TABLE_ORDER = 2,        /* kicp_pre_voxel_downsample: the survivors are the reference's, but a robin-hood probe exceeded the
                                         limit at which tsl::robin_map re-hashes mid-way: their ORDER may differ (kicp_pre_set_probe_limit) */
    KICP_ERR_HIP = -1,                /* HIP runtime / device failure (message in kicp_last_error) */
    KICP_ERR_ARG = -2,                /* bad argument */
    KICP_ERR_CAPACITY = -3,           /* a documented limit exceeded (max_points_per_voxel > 65 535, more voxels than the table's bucket
                                         index addresses - 2^24-2 at <= 255 points per voxel -, a per-point term of the normal
                                         equations >= 2^43: source points ~2 900 km from the base frame, a voxel coordinate beyond
                                         +-2^20 in kicp_pre_voxel_downsample) */
    KICP_ERR_COMM = -4                /* a multi-GPU exchange failed (RCCL, peer mailboxes, shared segment): a peer did not show up in time */
};

typedef struct kicp_map kicp_map; /* twin of kiss_icp::VoxelHashMap: host-authoritative voxel map + HBM mirror */
typedef struct kicp_reg kicp_reg; /* twin of kinematic_icp::KinematicRegistration + device workspace/stream */

/* Constructor arguments / public fields of KinematicRegistration (Registration.hpp:33-37,45-49). */
typedef struct kicp_reg_config {
    int32_t max_num_iterations;
    double convergence_criterion;
    int32_t max_num_threads; /* kept for API parity; the HIP path ignores it */
    int32_t use_adaptive_odometry_regularization;
    double fixed_regularization;
} kicp_reg_config;

/* Per-call diagnostics (SURVEY.md section 5 "Metrics"): what the reference computes internally but never exposes. */
typedef struct kicp_stats {
    int32_t iterations; /* ComputePerturbation evaluations executed (Registration.cpp:179-187) */
    int32_t converged;  /* 1 if ||dx|| < convergence_criterion ended the loop (Registration.cpp:184) */
    int32_t empty_map;  /* 1 if the early-out of Registration.cpp:157 fired */
    int32_t reserved;
    double beta;                            /* regularisation used (Registration.cpp:171-177) */
    double n_corr[KICP_MAX_LOG_PASSES];     /* correspondences per pass */
    double sums[KICP_MAX_LOG_PASSES][6];    /* raw JTJ00, JTJ01, JTJ11, JTr0, JTr1, sum||r||^2 per pass */
    double dx[KICP_MAX_LOG_PASSES][2];      /* solved (displacement, yaw) per pass */
    double gpu_ms;                          /* device time of the call, HIP events on the handle's stream ("timing" >= 1) */
    double pass_ms[KICP_MAX_LOG_PASSES];    /* device time of each fused pass kernel, HIP events around the launch ("timing" == 2) */
} kicp_stats;

const char *kicp_last_error(void); /* thread-local, valid until the next failing call on this thread */
int kicp_version(void);
int kicp_device_count(void); /* number of visible HIP devices, <0 on runtime failure */
/* Where the host should run the thread that calls this library for `device`: the NUMA node the GPU is attached to (-1: unknown) and
 * its CPUs as the kernel prints them ("0-63,128-191"; empty: unknown), from /sys/bus/pci/devices/<bdf>/{numa_node,local_cpulist}.
 * The library places its own helper threads and pinned buffers there (KICP_NUMA=0: not); the caller's thread is the caller's to
 * place - a frame is ~5 % faster from that node and ~10 % from inside one L3 domain of it. */
int kicp_device_locality(int device, int *out_numa_node, char *out_cpulist, size_t cap);

/* ---- kiss_icp::VoxelHashMap (kiss-icp v1.2.0 core/VoxelHashMap.hpp) ------------------ */
/* VoxelHashMap(voxel_size, max_distance, max_points_per_voxel)   -- pipeline/KinematicICP.hpp:79 */
int kicp_map_create(double voxel_size, double max_distance, unsigned int max_points_per_voxel, kicp_map **out);
void kicp_map_destroy(kicp_map *map);
/* VoxelHashMap(const VoxelHashMap &): the reference's map is copyable (it holds its tsl::robin_map by value); a deep copy
 * of the newest state (pulled back from the GPU first if the device copy is ahead), with its own HBM mirror on first use */
int kicp_map_clone(const kicp_map *map, kicp_map **out);
int kicp_map_clear(kicp_map *map);                                           /* Clear()  -- KinematicICP.hpp:88 */
int kicp_map_empty(const kicp_map *map);                                     /* Empty()  -- Registration.cpp:157 */
int kicp_map_add_points(kicp_map *map, const double *xyz, size_t n);         /* AddPoints(points) */
int kicp_map_remove_far(kicp_map *map, const double origin[3]);              /* RemovePointsFarFromLocation(origin) */
int kicp_map_update_origin(kicp_map *map, const double *xyz, size_t n, const double origin[3]); /* Update(points, origin) */
int kicp_map_update_pose(kicp_map *map, const double *xyz, size_t n, const double pose_qt[7]);  /* Update(points, pose) -- KinematicICP.cpp:79 */
/* Update(points, pose) with the points already in HBM (e.g. a kicp_pre buffer): transform, AddPoints and
 * RemovePointsFarFromLocation run on the device - every touched voxel's new points are judged in input order with the
 * reference's fp64 rule (one wave per voxel for frame-sized updates, one thread per voxel for bulk insertions), so the
 * accepted points and their order inside each voxel are exactly the sequential reference's.  Table growth / clean-up (a
 * device-side re-hash) and pool growth happen in HBM as well; the host copy is refreshed lazily when a host-side call
 * (AddPoints, kicp_map_check, ...) needs it.  A map that leaves +-2^20 voxels from its origin is updated by the host map
 * from then on (same result). */
int kicp_map_update_pose_device(kicp_map *map, int device, const double *d_points_xyz, size_t n, const double pose_qt[7]);
/* The same update in two halves (round 4): _begin queues the update's kernels and returns WITHOUT waiting for them, so that the caller
 * can do host-side work that does not touch the map meanwhile (the drop-in RegisterFrame collects the frame's two returned clouds);
 * kicp_map_update_finish waits and takes the result over (host fallback included).  d_points_xyz stays borrowed until then.  Only
 * frame-sized updates into a table with room for the worst case are actually deferred; every other case runs to completion inside
 * _begin.  Any other call on the map finishes a pending update first, so forgetting _finish costs nothing but the overlap. */
int kicp_map_update_pose_device_begin(kicp_map *map, int device, const double *d_points_xyz, size_t n, const double pose_qt[7]);
int kicp_map_update_finish(kicp_map *map);
int kicp_map_last_update_on_device(const kicp_map *map); /* 1 if the last kicp_map_update_pose_device ran on the GPU (collects a pending update first) */
/* Updates so far that ran on the GPU and have been collected; unlike every other call on the map this one does NOT wait for a pending
 * update (a diagnostic a timed loop can read without disturbing what it measures). */
unsigned long long kicp_map_device_updates(const kicp_map *map);
/* Preferred device for BULK host-side insertions (not part of the reference API): with device >= 0, kicp_map_add_points /
 * kicp_map_update_origin / kicp_map_update_pose calls of 4096 points or more stage their points into HBM and insert them
 * there (the same map as the sequential host insertion builds, an order of magnitude faster); -1 (default) = always on the
 * host.  Small calls stay on the host either way. */
int kicp_map_set_device(kicp_map *map, int device);
size_t kicp_map_num_points(const kicp_map *map);
size_t kicp_map_num_voxels(const kicp_map *map);
/* Pointcloud() -- KinematicICP.hpp:92.  Writes min(cap_points, total) points, returns total. */
size_t kicp_map_pointcloud(const kicp_map *map, double *out_xyz, size_t cap_points);
/* GetClosestNeighbor(query) for n queries -- Registration.cpp:74.  Host arrays in/out; runs on `device` the plain fp64 search
 * over the 27 neighbour voxels (kicp_kernels.hpp search_global: the reference's loop, and the exact fallback of the fused pass
 * kernels - NOT their mirror pre-selection: what the shipped pass kernels pick per query is what kicp_pass_correspondences returns).
 * No candidate -> nn = (0,0,0), dist = DBL_MAX, exactly like the reference. */
int kicp_map_closest(kicp_map *map, int device, const double *queries_xyz, size_t n, double *out_nn_xyz, double *out_dist);
/* Debug aid: verifies the table invariants the kernels rely on (neighbour masks, bucket records, halo entries, fp32
 * mirror, counters) on the host copy; returns the number of violations, 0 when consistent. */
size_t kicp_map_check(const kicp_map *map);
/* Make the HBM mirror on `device` current (a no-op when nothing changed since the last upload).
 * kicp_register* call it implicitly; exposed so map upload can be kept out of a timed region. */
int kicp_map_sync(kicp_map *map, int device);
/* What the last upload of the mirror moved: bytes sent host->device and whether it was a full re-send (1) or a delta
 * of the changed table slots / buckets (0).  A re-hash of the table or Clear() forces a full re-send. */
int kicp_map_last_upload(const kicp_map *map, size_t *bytes, int *was_full);

/* ---- kinematic_icp::KinematicRegistration (registration/Registration.hpp:32-50) ------------------------- */
int kicp_reg_create(const kicp_reg_config *config, int device, kicp_reg **out);
void kicp_reg_destroy(kicp_reg *reg);
/* KinematicRegistration(const KinematicRegistration &): the reference's struct is a plain copyable aggregate
 * (Registration.hpp:32-50).  A new handle on the same device with the same parameters and tuning options and workspaces of
 * its own (stream, queue, hand-off buffers).  Multi-GPU exchanges are per handle and are not carried over. */
int kicp_reg_clone(const kicp_reg *reg, kicp_reg **out);
int kicp_reg_get_config(const kicp_reg *reg, kicp_reg_config *out);
int kicp_reg_set_config(kicp_reg *reg, const kicp_reg_config *config); /* the reference's fields are public & mutable */
/* Backend tuning knobs (not part of the reference API).  Seventeen settable options; everything that lost its A/B over five rounds
 * (other workgroup sizes and register budgets, the plain fp64 gather, the device-side solve, single-record hand-offs, ...) was deleted
 * in round 6, and the pass kernels' ablation switches ("dbg") exist in libkicp_amd_dbg.so only (make -C kinematic_icp_amd/csrc dbg).
 * Which kernel runs:
 *   "lanes_per_query"  sub-lanes sharing one query of the generic pass kernel (1 | 2 | 4; 0 = by scan size, default: 4 up to 4 096 points,
 *                  2 up to 32 768, else 1)
 *   "latency_kernel" one lane per query: 1 (default) scans of at most 131 072 points - which leave the device two waves per SIMD
 *                  anyway - run the build that has TWO neighbour voxels in flight per round; 2: every such scan; 0: never (the
 *                  four-waves-per-SIMD build)
 *   "small"        1 (default): scans of at most 16 384 lanes (points x sub-lanes) take the small-scan path - the kernel stays resident for
 *                  the iterations of a call, polling for the next pose; 0: always the generic pass kernel
 *   "small_wave"   1 (default): scans of at most 4 096 points run ONE WAVE PER QUERY (k_pass_wave); 0: sub-lanes per query (k_pass_small)
 *   "small_resident" 1 (default): resident unless the previous call converged in one iteration; 2: always; 0: one launch per iteration.
 *                  A resident kernel holds the CUs it runs on until its host answers (at most "small_timeout_us"): set 0 where several
 *                  threads or processes register small scans on one device
 *   "small_timeout_us" how long a resident workgroup waits for the next command before it leaves on its own (default 20 000; the host
 *                  then launches afresh)
 *   "resident_generic" 1 (default): scans beyond the small-scan kernels and up to 131 072 points keep the generic kernel's latency build
 *                  resident for the later iterations of a call too; 0: one launch per iteration
 * Batches of independent scans (kicp_register_device_batch):
 *   "batch_queues"   (default 4, 0 .. 8) scans in flight at a time, each on a handle and HSA queue of its own; < 2: off
 *   "batch_resident" 1 (default): batches "batch_queues" does not take keep ONE resident kernel across the batch's scans; 0: scan by scan
 *   "batch_depth"    (default 3, 1 .. 4) scans that kernel has in flight
 *   "batch_threads"  (default 8, 0 .. 9) batches of scans that leave most of the device empty: up to this many resident kernels side by
 *                  side, a host thread of the library's pool each (capped by the CPUs this process may use: cpuset and cgroup quota)
 *   "batch_rotate"   1 (default): the workgroups of a resident kernel take turns at the parts of a scan; 0: fixed shares
 * Transfers and launches:
 *   "bar_frame"    1 (default): kicp_register writes host frames of up to 8 192 points straight into HBM through the PCIe BAR
 *   "fetch_upload" 1 (default): larger host frames are pulled out of the pinned staging buffer by a small kernel per piece; 0: DMA engine
 *   "aql"          1 (default): pass kernels are dispatched with hand-written AQL packets on the handle's own HSA queue; 0: HIP launches
 *   "wait"         0 (default): poll the tagged rows in host memory; 1: hipStreamSynchronize
 *   "timing"       1: kicp_stats.gpu_ms from HIP events; 2: also kicp_stats.pass_ms[]
 * Read only: "small_active" (path of the last call: 0 generic, 1 sub-lanes, 2 wave per query), "resident_passes", "batch_queue_passes",
 *   "batch_resident_passes", "batch_threads_active", "small_relaunches", "aql_active", "aql_kernarg", "comm_ranks".
 * Test hooks (exercise fall-backs that this hardware does not reach by itself): "small_cmd" 0 - workgroup 0 relays the resident kernels'
 *   commands (platforms without a CPU-writable BAR); "debug_tag" - jump next to the 16-bit pass tag's wrap-around; "debug_stall_us" -
 *   be late with one command; "debug_p2p_one_row" - send this rank's total as one mailbox row, as launches of more than 32 groups
 *   do; "small_trace" - in-kernel time stamps (tools/trace_small.py). */
int kicp_reg_set_option(kicp_reg *reg, const char *name, double value);
double kicp_reg_get_option(const kicp_reg *reg, const char *name);

/* ComputeRobotMotion(frame, voxel_map, last_robot_pose, relative_wheel_odometry, max_correspondence_distance)
 * -- Registration.hpp:39-43 / Registration.cpp:151-190.  `frame_xyz` is a HOST pointer (uploaded inside). */
int kicp_register(kicp_reg *reg, kicp_map *map, const double *frame_xyz, size_t n, const double last_pose_qt[7],
                  const double rel_odom_qt[7], double max_correspondence_distance, double out_pose_qt[7], kicp_stats *stats);
/* Same with the frame still float32 - the wire format of a PointCloud2, which the reference widens on the host with
 * static_cast<double> before it registers anything (ros/src/kinematic_icp_ros/utils/RosUtils.cpp:30-39): half the bytes cross
 * PCIe and the widening (exact) happens on the device, so the result is bit-equal to kicp_register on the widened frame. */
int kicp_register_f32(kicp_reg *reg, kicp_map *map, const float *frame_xyz_f32, size_t n, const double last_pose_qt[7],
                      const double rel_odom_qt[7], double max_correspondence_distance, double out_pose_qt[7], kicp_stats *stats);
/* Same, but the frame is already in HBM: `d_frame_xyz` is a DEVICE pointer on the handle's device
 * (e.g. the output of an on-device pre-step, or a torch tensor's data_ptr()). */
int kicp_register_device(kicp_reg *reg, kicp_map *map, const double *d_frame_xyz, size_t n, const double last_pose_qt[7],
                         const double rel_odom_qt[7], double max_correspondence_distance, double out_pose_qt[7],
                         kicp_stats *stats);
/* A queue of INDEPENDENT registrations against one map (several robots on one map, replayed scans), without a language
 * binding's per-call cost in between.  Every pose is bit-equal to what kicp_register_device returns for that scan alone; the
 * scans share nothing but the read-only map, which must not be updated during the call.  Because the scans do not depend on each
 * other, the call keeps SEVERAL OF THEM IN FLIGHT (one host thread - the caller's - drives them all, except in the first case):
 *   - batches of at least 32 scans that leave most of the device empty (all of at most 4 096 points, or all generic of at most 24 576):
 *     "batch_threads" (default 8) resident kernels side by side - as many as fit the device at once -, each with a contiguous part of
 *     the batch, "batch_depth" scans of it in flight, and a host thread of the library's pool;
 *   - otherwise, batches of at least two scans per queue that hold at least one scan for the generic pass kernel (more than 4 096 points by
 *     default): option "batch_queues" (default 4) scans at a time, each on a handle + HSA queue of its own (clones of `reg`,
 *     created on first use and kept until kicp_reg_destroy(reg); they follow reg's configuration and kernel-shape options at every
 *     call), every pass an ordinary launch (large scans: the four-waves-per-SIMD build; small scans in such a batch: their own
 *     kernels, one launch per pass); the thread goes round the scans in flight: rows complete -> solve -> next pass or next scan;
 *   - otherwise, batches of eight scans and more of one kind (all small, or all up to 131 072 points): ONE kernel resident across
 *     the batch's scans ("batch_resident"), option "batch_depth" (default 3, at most 4) scans in flight - the command that starts a
 *     pass names the scan it belongs to, and the command of pass k + depth goes out when the rows of pass k are in;
 *   - anything else, and "batch_queues" 0 with "batch_threads" 0 and "batch_depth" 1, or "batch_resident" 0: the scans strictly one after the other
 *     (every scan runs launch -> hand-off -> solve to completion before the next one starts) - what a caller needs whose next
 *     scan depends on the previous result, and what kicp_register_device gives one call at a time.
 * SHARDED batches (round 5): with the shared segment attached (kicp_reg_shm_init) EVERY rank calls with the same count and the same
 * poses and hands over ITS shard of every scan (frame k of rank r = points [n_k r / R, n_k (r + 1) / R) of scan k; an empty shard
 * is legal: n[k] == 0).  "batch_queues" >= 2 then keeps that many SHARDED scans in flight per rank: lane j of the call registers scans
 * j, j + lanes, j + 2 lanes, ... on every rank and exchanges its per-pass totals through an area of its own in the segment, so the
 * lanes' exchanges interleave freely while every lane's sequence of exchanges is the same on every rank.  Poses and iteration counts
 * are bit-equal to the single-GPU batch on the whole scans, on every rank.  A sharded batch that fails half-way leaves the ranks'
 * lane counters in doubt: later sharded batches return KICP_ERR_COMM until kicp_reg_shm_destroy / _init have been redone on every
 * rank.  The same with the RCCL communicator attached (round 6): lane j issues its all-reduces on a sub-communicator of its own
 * (ncclCommSplit on first use) and on its own stream.  (Callback / peer-mailbox exchanges: the scans one after the other.)
 * Poses: count x 7 doubles.  `out_iterations` (nullable): ICP iterations each scan ran.  Returns the first error (< 0) - scans
 * after the first one that failed are then unspecified (some of them may have completed), nothing of the call is still running
 * on the device - otherwise the largest warning code seen (KICP_OK if none). */
int kicp_register_device_batch(kicp_reg *reg, kicp_map *map, size_t count, const double *const *d_frames_xyz, const size_t *n,
                               const double *last_poses_qt, const double *rel_odoms_qt, double max_correspondence_distance,
                               double *out_poses_qt, int *out_iterations);
/* The same queue of INDEPENDENT registrations with `lanes` of them in flight: lane t drives handle regs[t] (its own stream,
 * AQL queue and hand-off buffers; options and config are taken from each handle, except that for the duration of the call
 * small scans are not kept resident and large scans use the four-waves-per-SIMD kernel, which leaves room for the other
 * lanes' workgroups) from a host thread of its own (a pool the library starts on first use and keeps, unpinned - they do not
 * inherit the caller's CPU affinity; the calling thread sleeps until the lanes are done; one concurrent call at a time uses the
 * pool), scans are
 * dealt to the lanes first come, first served.  This is a THROUGHPUT mode for workloads that have independent scans -
 * several robots localising in one map, replayed logs - and nothing the reference's sequential pipeline can use: a scan's
 * initial guess there is the previous scan's result.  Every pose is bit-equal to what kicp_register_device returns for that
 * scan (the lanes share nothing but the read-only map).  The map must not be updated during the call.  Returns like
 * kicp_register_device_batch; after an error the remaining scans are not started and their poses are unspecified. */
int kicp_register_device_concurrent(kicp_reg *const *regs, size_t lanes, kicp_map *map, size_t count, const double *const *d_frames_xyz,
                                    const size_t *n, const double *last_poses_qt, const double *rel_odoms_qt,
                                    double max_correspondence_distance, double *out_poses_qt, int *out_iterations);
/* One fused association+accumulation pass at a fixed pose (DataAssociation + the reduction of
 * ComputePerturbation + the sum of ComputeOdometryRegularization; Registration.cpp:62-81,102-118,51-55).
 * out_sums = {JTJ00, JTJ01, JTJ11, JTr0, JTr1, sum||r||^2, N_corr}, un-normalised.  Host frame pointer. */
int kicp_pass_sums(kicp_reg *reg, kicp_map *map, const double *frame_xyz, size_t n, const double pose_qt[7],
                   double max_correspondence_distance, double out_sums[7]);

/* The correspondences of that pass themselves - what DataAssociation appends (Registration.cpp:73-77) -, one entry per source point,
 * written by the SAME pass kernel build the handle registers a scan of this size with (its EXPORT instantiation: the search, the exact
 * resolution, the tie rule and the acceptance test are the shipped code, with the decision also stored per query):
 *   out_index[i]   index of the chosen map point in the device pool (bucket * max_points_per_voxel + position), -1: no correspondence
 *                  (nothing within max_correspondence_distance)
 *   out_d2[i]      its squared distance to pose * frame[i] (fp64, the reference's operation order); DBL_MAX without a correspondence
 *   out_nn_xyz     its coordinates (3 doubles per point; zeros without a correspondence)
 * Kernel-shape options ("small", "small_wave", "lanes_per_query", "latency_kernel") steer it as they steer kicp_register. */
int kicp_pass_correspondences(kicp_reg *reg, kicp_map *map, const double *frame_xyz, size_t n, const double pose_qt[7],
                              double max_correspondence_distance, int32_t *out_index, double *out_d2, double *out_nn_xyz);

/* Same pass, but returns the raw all-reduce payload: KICP_REDUCE_WORDS int64 words = 3 limbs per sum (value * 2^40 =
 * l0 + l1*2^40 + l2*2^80) + range flag + padding.  Summing the words of disjoint shards element-wise gives exactly the
 * words' value of the union: this is what the multi-GPU mode all-reduces. */
int kicp_pass_words(kicp_reg *reg, kicp_map *map, const double *frame_xyz, size_t n, const double pose_qt[7],
                    double max_correspondence_distance, long long out_words[24]);

/* ---- pre-steps of the pipeline on the GPU (pipeline/KinematicICP.cpp:54-62) -----------
 * A kicp_pre owns KICP_PRE_BUFFERS device point buffers.  Results stay in HBM (feed kicp_register_device with
 * kicp_pre_device_ptr) and are downloaded only when the host needs them (map update, return values). */
#define KICP_PRE_BUFFERS 4
typedef struct kicp_pre kicp_pre;
int kicp_pre_create(int device, kicp_pre **out);
void kicp_pre_destroy(kicp_pre *pre);
/* kiss_icp::Preprocessor::Preprocess(frame, timestamps, relative_motion) followed by transform_points(., lidar_to_base):
 * optional constant-velocity deskew to the scan end (when `deskew` and n_timestamps != 0), crop to
 * min_range < |p| < max_range in the sensor frame, then into the base frame.  Order preserved.  Host input. */
int kicp_pre_preprocess(kicp_pre *pre, const double *frame_xyz, size_t n, const double *timestamps, size_t n_timestamps,
                        const double relative_motion_qt[7], const double lidar_to_base_qt[7], double max_range, double min_range,
                        int deskew, int dst_buffer, size_t *out_n);
/* PointCloud2 wire-format ingest: the raw message bytes go to the GPU (point_step bytes per
 * point instead of 32 B of fp64 xyz + stamp) and are decoded there.  Replaces
 *   ros/src/kinematic_icp_ros/utils/RosUtils.cpp:30-39        PointCloud2ToEigen(msg, T): FLOAT32 x,y,z -> T * Vector3d
 *   ros/src/kinematic_icp_ros/utils/TimeStampHandler.cpp:57-104  per-point stamp (UINT32 / FLOAT32 / FLOAT64) -> seconds as
 *                                                             double; values with more than 10 integer digits are ns
 *   ros/src/kinematic_icp_ros/utils/TimeStampHandler.cpp:106,121-128  min / max and normalisation (t - min) / (max - min)
 * `data` = msg->data.data(), n_points = width * height, borrowed for the call.  sensor_pose_qt may be NULL (= the identity
 * LidarOdometryServer.cpp:203 passes).  out_min/max_stamp (seconds; 0 when the cloud has no stamp field or is empty)
 * are what ProcessTimestamps needs for its begin/end-stamp logic (TimeStampHandler.cpp:111-119).  Any other stamp
 * datatype -> KICP_ERR_ARG ("timestamp field type not supported", TimeStampHandler.cpp:103). */
#define KICP_FIELD_UINT32 6  /* sensor_msgs::msg::PointField datatype codes */
#define KICP_FIELD_FLOAT32 7
#define KICP_FIELD_FLOAT64 8
typedef struct kicp_cloud_layout {
    unsigned int point_step;                  /* msg->point_step */
    unsigned int offset_x, offset_y, offset_z; /* offsets of the FLOAT32 fields "x", "y", "z" */
    int stamp_datatype;                       /* 0 = no "t"/"timestamp"/"time"/"stamps" field, else its datatype code */
    unsigned int offset_stamp;
} kicp_cloud_layout;
int kicp_pre_ingest(kicp_pre *pre, const void *data, size_t n_points, const kicp_cloud_layout *layout, const double sensor_pose_qt[7],
                    double *out_min_stamp, double *out_max_stamp);
/* LOOK-AHEAD (round 5; a node that knows its next message - a bag replay, a queue of depth 2): announce the NEXT cloud.  Nothing is
 * copied yet; the next kicp_pre_frame[_ingested] call - the pre-steps of the CURRENT cloud - uploads and decodes the announced one into
 * a second slot on a stream of its own once its own kernels are queued, so the 2 MB of message k + 1 cross PCIe while the GPU works on
 * message k and the calling thread would only wait.  The kicp_pre_ingest call for the same message (same pointer, count, layout and
 * sensor pose) then finds it decoded and returns at once; any other message voids the announcement.  `data` must stay valid and
 * unchanged until that kicp_pre_ingest call.  Results are those of the plain kicp_pre_ingest, bit for bit. */
int kicp_pre_ingest_ahead(kicp_pre *pre, const void *data, size_t n_points, const kicp_cloud_layout *layout, const double sensor_pose_qt[7]);
unsigned long long kicp_pre_ahead_hits(const kicp_pre *pre); /* kicp_pre_ingest calls so far that found their message decoded ahead */
/* kicp_pre_preprocess on the ingested cloud (no host input; deskews only if the cloud carried stamps) */
int kicp_pre_preprocess_ingested(kicp_pre *pre, const double relative_motion_qt[7], const double lidar_to_base_qt[7], double max_range,
                                 double min_range, int deskew, int dst_buffer, size_t *out_n);
/* Download the decoded cloud (fp64 xyz) and its normalised stamps: what PointCloud2ToEigen / ProcessTimestamps return. */
int kicp_pre_ingested(const kicp_pre *pre, double *out_xyz, double *out_stamps, size_t cap_points, size_t *out_n, int *out_has_stamps);
/* kiss_icp::VoxelDownsample(buffer src, voxel_size) -> buffer dst: the first point (lowest index) of every voxel, in the
 * ITERATION ORDER of the reference's tsl::robin_map after reserve(frame.size()) - the order the second downsample and the map
 * update of the pipeline depend on (kiss-icp v1.2.0 core/VoxelUtils.cpp; call sites pipeline/KinematicICP.cpp:40,42).
 * KICP_ERR_CAPACITY when a voxel coordinate leaves +-2^20 (the handle stays usable). */
int kicp_pre_voxel_downsample(kicp_pre *pre, int src_buffer, double voxel_size, int dst_buffer, size_t *out_n);
/* Largest robin-hood displacement (in buckets) the last kicp_pre_voxel_downsample saw while replaying the reference's table
 * (0 when every probe stayed below 32).  tsl::robin_map grows its table when an insertion's probe exceeds the container's
 * limit (128 in robin-map 0.6.x once the load factor is >= 0.15, 8192 in 1.x) - a re-hash the replay does not model: a value at
 * or beyond the limit of the robin-map the reference was built against means the ORDER of that output may differ from the
 * reference's (the set of survivors never does). */
unsigned int kicp_pre_last_max_probe(const kicp_pre *pre);
/* The probe length beyond which the reference's container grows its table (default 128 = robin-map 0.6.x, what Ubuntu 22.04 ships
 * and USE_SYSTEM_TSL-ROBIN-MAP picks up; 8192 for robin-map 1.x; also KICP_ROBIN_PROBE_LIMIT in the environment).  A downsample whose
 * replay sees a longer probe returns KICP_WARN_TABLE_ORDER (> 0: the output is complete, its order is not vouched for). */
int kicp_pre_set_probe_limit(kicp_pre *pre, unsigned int limit);
/* The pre-steps of one frame as ONE call behind one host synchronisation (pipeline/KinematicICP.cpp:54-62): Preprocess +
 * transform_points into buffer 0, VoxelDownsample(buffer 0, voxel_a) into buffer 1, VoxelDownsample(buffer 1, voxel_b) into
 * buffer 2 - the results of kicp_pre_preprocess[_ingested] + two kicp_pre_voxel_downsample calls, element for element; each
 * step's survivor count stays on the device as the next step's input count.  out_counts = {points of buffer 0, 1, 2}.
 * out_frame_xyz (nullable; room for every INPUT point, cap_points >= n): buffer 0 starts travelling there in the background as
 * soon as it is complete - collect it with kicp_pre_download_finish(pre, 0, ...), whose out_n is out_counts[0]; the first
 * out_counts[0] points are the frame.  kicp_pre_frame: host input as kicp_pre_preprocess; kicp_pre_frame_ingested: the cloud
 * of the last kicp_pre_ingest (kicp_pre_ingested_count points).  May return KICP_WARN_TABLE_ORDER like the downsample.
 * Buffers 1 and 3 take turns between calls (round 6): the previous call's buffer 1 - device memory a map update begun with
 * kicp_map_update_pose_device_begin may still be reading - is buffer 3 afterwards, unchanged until the call after this one; a pointer
 * taken with kicp_pre_device_ptr(pre, 1, ..) stays valid that long. */
int kicp_pre_frame(kicp_pre *pre, const double *frame_xyz, size_t n, const double *timestamps, size_t n_timestamps,
                   const double relative_motion_qt[7], const double lidar_to_base_qt[7], double max_range, double min_range, int deskew,
                   double voxel_a, double voxel_b, double *out_frame_xyz, size_t cap_points, size_t out_counts[3]);
int kicp_pre_frame_ingested(kicp_pre *pre, const double relative_motion_qt[7], const double lidar_to_base_qt[7], double max_range,
                            double min_range, int deskew, double voxel_a, double voxel_b, double *out_frame_xyz, size_t cap_points,
                            size_t out_counts[3]);
size_t kicp_pre_ingested_count(const kicp_pre *pre);
/* Backend knobs of the chained pre-steps (not part of the reference API):
 *   "fused"  1 (default; KICP_PRE_FUSED=0 in the environment): kicp_pre_frame* run the frame's pre-steps as FIVE launches
 *            (kicp_pre.hpp: k_frame_*) and hand the three counts over through host memory the call polls; 0: one launch per
 *            step (preprocess, compact, 2 x {claim, replay, gather}) and a copy + stream synchronisation at the end.  Same results.
 *   "guess"  the survivor count from which the fused chain sizes its first table before it knows the real one (default: the
 *            previous frame's; the first frame: the input count).  A wrong power-of-two bracket costs the frame the unfused
 *            downsamples, never a different result.
 *   "fused_frames" / "guess_misses" (read only): frames the fused chain served / of those, frames whose guess was wrong. */
int kicp_pre_set_option(kicp_pre *pre, const char *name, double value);
double kicp_pre_get_option(const kicp_pre *pre, const char *name);
int kicp_pre_upload(kicp_pre *pre, int buffer, const double *xyz, size_t n);
int kicp_pre_download(const kicp_pre *pre, int buffer, double *out_xyz, size_t cap_points, size_t *out_n);
/* The same download in the background: _begin queues the copy of the buffer's current contents on a stream of its own
 * (pinned landing area) and returns at once; the caller goes on with the pipeline's next steps and collects the points
 * with _finish.  One download in flight per handle; the buffer must not be refilled in between.  (RegisterFrame returns the
 * whole preprocessed frame - pipeline/KinematicICP.cpp:84 - 3 MB that nothing on the device waits for.) */
int kicp_pre_download_begin(kicp_pre *pre, int buffer);
/* Same, and a helper thread of the handle also moves the landed points into `out_xyz` (room for cap_points points, which must
 * stay valid until _finish) while the caller goes on: _finish with the same pointer (or NULL) then only waits for it. */
int kicp_pre_download_begin_into(kicp_pre *pre, int buffer, double *out_xyz, size_t cap_points);
int kicp_pre_download_finish(kicp_pre *pre, int buffer, double *out_xyz, size_t cap_points, size_t *out_n);
const double *kicp_pre_device_ptr(const kicp_pre *pre, int buffer, size_t *out_n);

/* Diagnostics: the demangled-name prefixes under which the registration's direct-dispatch path looks its kernels up in the
 * embedded code object, newline separated (tests check each against build/kicp_reg.hsaco, so a change of the compiler's
 * mangling or of a template signature fails a CPU test instead of silently disabling the path).  Returns the bytes needed. */
size_t kicp_aql_kernel_names(char *out, size_t cap);

/* Diagnostic behind bench.py's latency model: `workgroups` x `block` lanes each walk their own chain of `steps` DEPENDENT loads
 * through a random cyclic permutation of the 128-byte lines of a `working_set_bytes` buffer (the pass kernel's situation: a
 * wave's step costs the slowest of its 64 lanes' accesses, with the whole launch's accesses in flight around it).  Returns the
 * time per dependent step (whole-launch duration / steps, best of three warm launches, HIP events). */
int kicp_probe_dependent_load(int device, size_t working_set_bytes, int workgroups, int block, int steps, double *out_ns_per_step);
/* Bytes of the map's device copy a query can touch: table entries (occupied + halo, 128 B each) and the occupied voxels' buckets
 * (fp64 pool + 16-bit mirror), without the head-room around them; 0 before the first upload. */
size_t kicp_map_device_bytes(const kicp_map *map);

/* ---- device memory helpers for callers without a HIP runtime binding of their own ------------------------ */
int kicp_device_malloc(int device, size_t bytes, void **out_dptr);
int kicp_device_free(int device, void *dptr);
int kicp_device_upload(int device, void *dst_dptr, const void *src_host, size_t bytes);
int kicp_device_synchronize(int device);

/* ---- multi-GPU: scan points sharded across ranks, map replicated, one tiny all-reduce per ICP iteration
 *      (SURVEY.md section 8e).  Every rank calls kicp_register* with ITS shard and gets the identical pose.
 *      The payload is KICP_REDUCE_WORDS int64 values: the seven sums (JTJ00, JTJ01, JTJ11, JTr0, JTr1, sum||r||^2,
 *      N_corr) are accumulated exactly as 3 x 40-bit fixed-point limbs each, so an integer sum-all-reduce makes the
 *      result independent of the number of ranks, bit for bit. ---- */
#define KICP_REDUCE_WORDS 24
#define KICP_COMM_ID_BYTES 128
int kicp_comm_unique_id(char id[KICP_COMM_ID_BYTES]); /* rank 0 creates, the caller broadcasts the bytes */
int kicp_reg_comm_init(kicp_reg *reg, int nranks, int rank, const char id[KICP_COMM_ID_BYTES]); /* RCCL comm on reg's device */
int kicp_reg_comm_destroy(kicp_reg *reg);
/* Single-node alternative without any device collective: all ranks map one POSIX shared-memory segment (`name`, created
 * by rank 0 - call it there first, e.g. before a barrier).  Each rank's 24 limb totals + a sequence word go into its own slot
 * of the segment - written by its host, which has just added its GPU's tagged group rows exactly as in the single-GPU
 * hand-off -; every rank's host polls all slots, adds the integers and solves.  The "all-reduce" thus costs no more than the single-GPU hand-off (a 192-byte RCCL all-reduce
 * costs tens of microseconds per ICP iteration, as long as the kernel itself).  Every rank must issue the same sequence
 * of kicp_register* calls. */
int kicp_reg_shm_init(kicp_reg *reg, int nranks, int rank, const char *name);
int kicp_reg_shm_destroy(kicp_reg *reg);
/* One-shot exchange over peer mappings: no collective library, no host shared segment.  Every
 * rank (one process per GPU) owns a mailbox in its HBM.  The pass kernel's first reduction level ends in groups of 32
 * workgroups; the last workgroup of every group writes the group's row of 24 exact limb sums - tagged - into ALL ranks'
 * mailboxes (the peers' through IPC mappings: stores over xGMI), and the last workgroup of group 0 adds the rows of all
 * ranks' groups as they arrive in its own mailbox (integers, fixed order: bit-identical on every rank) and hands the totals
 * to its host, which solves - no second reduction level, no collective.  (A launch of more than 32 groups - 262 144 lanes
 * per rank - reduces on two levels and sends one row.)  Usage: every rank calls kicp_reg_p2p_export, the caller all-gathers the handles (any
 * transport), every rank calls kicp_reg_p2p_connect with the nranks handles in rank order; a barrier between connect and
 * the first registration, and before kicp_reg_p2p_destroy, is the caller's.  nranks <= KICP_P2P_MAX_RANKS.
 * Recovery contract: the ranks stay in step only while every exchange completes on every rank.  When a registration fails
 * after it has started an exchange (a peer's slot did not arrive within 0.8 x KICP_WAIT_TIMEOUT_S - the in-kernel wait is
 * derived from the host's setting -, a device fault), this rank's state is POISONED: every later kicp_register* call on the
 * handle returns KICP_ERR_COMM until kicp_reg_p2p_destroy, kicp_reg_p2p_export and kicp_reg_p2p_connect have been redone
 * (on every rank: the step counters restart from zero). */
#define KICP_P2P_HANDLE_BYTES 64
#define KICP_P2P_MAX_RANKS 16
int kicp_reg_p2p_export(kicp_reg *reg, int nranks, int rank, char handle[KICP_P2P_HANDLE_BYTES]);
int kicp_reg_p2p_connect(kicp_reg *reg, const char *handles /* nranks * KICP_P2P_HANDLE_BYTES, rank order */);
int kicp_reg_p2p_destroy(kicp_reg *reg);
/* Alternative to the built-in RCCL communicator: the caller supplies the sum-all-reduce (e.g. torch.distributed).
 * Called once per ICP iteration with a device buffer of `count` int64 values to be sum-reduced IN PLACE, ordered
 * on `stream` (a hipStream_t).  Return 0 on success.  Pass NULL to remove. */
typedef int (*kicp_allreduce_fn)(void *user, long long *d_buf, int count, void *stream);
int kicp_reg_set_allreduce(kicp_reg *reg, kicp_allreduce_fn fn, void *user);

#ifdef __cplusplus
}
#endif
#endif /* KICP_H_ */
