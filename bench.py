#!/usr/bin/env python3
"""bench.py -- scans/sec of the ICP registration hot path (KinematicRegistration::ComputeRobotMotion) on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One "step" = one BATCH of ComputeRobotMotion registrations (all ICP iterations of each scan; the batch is ONE call of the C-ABI's
kicp_register_device_batch - a queue of INDEPENDENT scans against the fixed map, of which the library keeps several in flight at a
time, every result bit-identical to registering that scan alone; the rate with ONE scan in flight at a time is reported beside it,
top-level `value_one_scan_in_flight`, and the rate with one Python call per scan in `config`) on
synthetic data of BASELINE.json's headline config: cfg2 = 64-beam x 2048 = 131 072-point scan vs a ~1M-point voxel map
(voxel 1.0 m, 20 pts/voxel), default ICP parameters (max 10 iterations, 1e-3 stop, adaptive regularisation), tau =
first-frame adaptive value.  Inputs (scans, map mirror) are resident in HBM when the timed region starts; map
build/upload is outside it.  `value` = scans per second = K * scans_per_step / wall time of the K timed steps.

The headline scans carry the seeded initial-guess error of SURVEY.md section 8d, for which the reference stops after ONE
iteration.  A second workload ("multi_iteration" in `config`) times the same scans with an extra 0.05 m / 0.5 deg odometry error
that needs several iterations (Registration.cpp:179-187: solve, update, stop test, re-association), in scans/s and in
ms per ICP iteration.

N > 1: the scan's points are sharded contiguously across the N ranks, the map is replicated, and every ICP iteration
sums 24 int64 words per rank (the exact limb sums of the 2x2 normal equations), so every rank returns the bit-identical
pose.  --comm selects the exchange behind `value` (also printed as "comm"): "rccl" (default, the north star: ncclAllReduce
over xGMI; a batch call keeps several sharded scans in flight, each lane on a sub-communicator and stream of its own),
"shm" (every rank's host adds its GPU's rows and the ranks meet in a node-wide host shared segment: no device collective),
"p2p" (one-shot exchange over xGMI peer mappings - no collective library), "torch" (torch.distributed all-reduce callback).
Every exchange is measured in the same run (value_rccl, value_shm, value_p2p).  Total work is fixed -> "scaling": "strong";
"scaling_bound" says what sharding one scan can reach at best (Amdahl from the single-GPU floor), "value_replicas" what the same
GPUs deliver as independent replicas (--mode replicas makes that the headline: every rank registers whole scans, "scaling": "weak").

Prints ONE JSON line on rank 0 with the contract's keys plus
  "roofline"     the dominant kernel (fused association+accumulation pass).  `achieved` / `frac` = HBM bytes per launch MEASURED
                 in this run (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over the same registrations) / the kernel's live
                 HIP-event duration, against the 8 TB/s peak.  The kernel is not HBM bound: `frac_latency` (the chain of dependent
                 accesses a wave must walk, on top of the fixed floor) and `valu_issue` (VALU issue slots per pass) are the bounds
                 it is held to.  SURVEY.md section 8d's algorithmic bytes - what the REFERENCE touches - are kept under
                 `algorithmic` for the record; this kernel skips most of them, so that ratio may exceed 1.  `b_min`: compulsory
                 bytes; `time_split_us`: floor vs query work; `counters`: the committed rocprofv3 profile of this workload.
  "cpu_baseline" the reference's own Registration.cpp (oracle/_ref, kind "reference") timed on this box's host cores
                 at 1 thread (the reference's default) and at the best of several thread counts, on a bounded sample of
                 the same scans; the oracle port's figures beside it.
"""
import argparse
import json
import os
import sys
import time

# the CPU checkers' OpenMP teams: pin threads to cores (set before any OpenMP runtime starts).  Idle threads must SLEEP
# (the default passive policy): spinning ones eat the container's CPU quota and slowed the 1-thread sample 10x.
os.environ.setdefault("OMP_PROC_BIND", "close")
os.environ.setdefault("OMP_PLACES", "cores")
os.environ.setdefault("OMP_WAIT_POLICY", "passive")  # (libgomp spins without end by default once its threads are bound)

import numpy as np  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
SCLK_GHZ = 2.4          # same guide: max engine clock 2400 MHz (the sustained clock under load is lower: the bound is optimistic)
L2_PEAK_GBS = 34500.0   # same guide, "L2 (per XCD)": ~34.5 TB/s aggregate
MULTI_ITER_ERROR = (0.05, 0.5)  # extra odometry error of the multi-iteration workload: metres along x, degrees of yaw


class Workload:
    """one BASELINE configuration resident on this rank's GPU: scene, scans (this rank's shard of each), map, threshold"""

    def __init__(self, K, syn, name, n_scans, device, rank, world, replicas, torch=None):
        self.name = name
        self.cfg, self.scene, self.scans, rng = syn.make_case(name, n_scans=min(n_scans, 8))
        self.gmap = K.VoxelHashMap(self.cfg.voxel_size, self.cfg.max_range, self.cfg.max_points_per_voxel)
        # the map grows through VoxelHashMap::Update(points, identity) on the GPU - the path the pipeline's map update takes
        # (same map as the host-side AddPoints builds, tests/test_gpu_mapdev.py; seconds instead of a minute for cfg5)
        ident = np.array([0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0])
        syn.build_map_points(self.scene, self.cfg, lambda pts: self.gmap.UpdateDevice(K.DeviceFrame(pts, device=device), ident), self.gmap.num_points, rng)
        if n_scans > 8:
            # further scans from a generator of their own (the first eight and the map are those of every earlier round), ray-cast
            # on the GPU by torch: dozens of distinct 131 072-point scans in seconds
            cast = (lambda o, d: syn.raycast_torch(self.scene, o, d, "cuda")) if torch is not None else None
            self.scans += syn.extra_scans(self.cfg, self.scene, n_scans - 8, self.cfg.seed + 1000, raycast=cast)
        self.tau = self.cfg.first_frame_tau()
        self.gmap.sync(device)
        self.n_total = self.scans[0]["frame"].shape[0]
        self.lo, self.hi = (self.n_total * rank) // world, (self.n_total * (rank + 1)) // world  # contiguous shard of this rank
        if replicas:
            self.lo, self.hi = 0, self.n_total
        self.frames = [K.DeviceFrame(s["frame"][self.lo:self.hi], device=device) for s in self.scans]
        extra = syn.planar_pose(MULTI_ITER_ERROR[0], 0.0, np.deg2rad(MULTI_ITER_ERROR[1]))
        self.rel_single = [s["rel_odom"] for s in self.scans]
        self.rel_multi = [syn.pose_mul(s["rel_odom"], extra) for s in self.scans]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--scans-per-step", type=int, default=0,
                    help="registrations per step (one batch of synthetic scans); 0 (default) = chosen after a calibration batch so that the "
                         "timed region of the K steps lasts at least --min-timed-s, never fewer than 64; re-sized and re-run if the timed "
                         "region still comes in short")
    ap.add_argument("--min-timed-s", type=float, default=0.3, help="shortest acceptable timed region (auto batch size)")
    ap.add_argument("--no-pmc", action="store_true", help="skip the in-run rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE) behind roofline.traffic")
    ap.add_argument("--workload", default="cfg2")
    ap.add_argument("--set", action="append", default=[], metavar="NAME=VALUE", help="registration option set on every handle bench.py makes (experiments: kicp.h lists them)")
    ap.add_argument("--scans", type=int, default=64, help="distinct synthetic scans cycled through (SURVEY.md section 8d asks for >= 50; 64 x 3.1 MB "
                                                          "of scans + the map exceed what the caches hold)")
    ap.add_argument("--cpu-seconds", type=float, default=16.0, help="budget of the cpu_baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--pipeline-frames", type=int, default=40,
                    help="frames of the `pipeline` block (the whole drop-in KinematicICP::RegisterFrame - ingest, pre-steps, registration, map update - on a "
                         "synthetic drive of 131 072-point PointCloud2 messages, with the reference's own RegisterFrame timed beside it); 0: skip")
    ap.add_argument("--comm", default="rccl", choices=["rccl", "shm", "p2p", "torch"],
                    help="N>1 exchange of the per-iteration sums behind `value`: the built-in RCCL all-reduce (default, the north star's form; since "
                         "round 6 a batch call keeps several sharded scans in flight over it, a sub-communicator per lane), the host shared segment "
                         "(no device collective; measured beside it in the same run: value_shm), the one-shot peer-mailbox exchange over xGMI "
                         "mappings (value_p2p), or a torch.distributed all-reduce callback.  Ranks that share one GPU (tests) fall back to shm")
    ap.add_argument("--pg-backend", default="nccl", choices=["nccl", "gloo"], help="torch.distributed backend for barriers/timing")
    ap.add_argument("--mode", default="shard", choices=["shard", "replicas"],
                    help="N>1: 'shard' (default, the north star) splits every scan's points across the ranks and exchanges the sums each "
                         "iteration; 'replicas' lets every rank register whole scans on its own (one robot per GPU, no exchange; weak scaling)")
    ap.add_argument("--force-comm", action="store_true", help="exercise the multi-GPU code path (all-reduce + separate solve) even with one rank")
    ap.add_argument("--no-sharded-cfg5", action="store_true",
                    help="N>1: skip the second sharded workload (cfg5: 500 000-point scans, 62 500 per GPU at N = 8 - where sharding can pay)")
    args = ap.parse_args()

    # Everything except the final JSON line goes to stderr - also what C libraries print (RCCL writes a version banner to
    # the C stdout, possibly after Python's own output): fd 1 is pointed at fd 2 until the very end.
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs a torch.distributed.run launch with that many ranks" % args.gpus)
        raise SystemExit("WORLD_SIZE=%d does not match --gpus %d" % (world, args.gpus))

    # torch first: the process then shares ONE HIP runtime between torch and libkicp_amd.so
    import torch
    import torch.distributed as dist

    import __graft_entry__ as entry
    entry.build()
    import kinematic_icp_amd as K
    from kinematic_icp_amd import synthetic as syn

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback exists for the product path)")
    device = local_rank if world > 1 else 0
    if "KICP_BENCH_DEVICE" in os.environ:  # testing aid: several ranks on one GPU (works with --comm shm --pg-backend gloo)
        device = int(os.environ["KICP_BENCH_DEVICE"])
    torch.cuda.set_device(device)
    # the process runs on the NUMA node its GPU is attached to, as a deployment would start it (numactl --cpunodebind; INTEGRATION.md
    # section 6): the host side of a registration is polls of, and copies through, pinned memory the GPU writes over PCIe.  The CPU
    # baseline's threads inherit the binding (one socket's CPUs - more than the box's quota).  KICP_BENCH_PLACEMENT=0: left alone.
    host_placement = "not bound"
    if os.environ.get("KICP_BENCH_PLACEMENT", "1") != "0":
        try:
            # (OMP_PROC_BIND above binds the process's initial thread to ONE core as soon as an OpenMP runtime starts - torch's - and
            #  every child process inherited that: undo it before looking at what this process may use)
            os.sched_setaffinity(0, range(os.cpu_count() or 1))
            l3_only = os.environ.get("KICP_BENCH_PLACEMENT", "1") == "l3"
            near = K.cpus_near_gpu(device, one_l3_domain=l3_only)
            if near:
                os.sched_setaffinity(0, near)
                host_placement = "process bound to the %d CPUs of %sthe GPU's NUMA node (%d)" % (len(near), "one L3 domain of " if l3_only else "", K.device_locality(device)[0])
            else:
                host_placement = "not bound (GPU on NUMA node %d with %d CPUs, none of them among the %d this process may use)" % (
                    K.device_locality(device)[0], len(K.device_locality(device)[1]), len(os.sched_getaffinity(0)))
        except Exception as e:  # noqa: BLE001 - placement is an extra
            host_placement = "not bound (%s)" % e
    replicas = args.mode == "replicas" and world > 1
    use_comm = world > 1 or args.force_comm      # a process group exists (barriers, timing)
    exchange = use_comm and not replicas          # the registration itself exchanges sums
    if use_comm:
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29511")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        if args.pg_backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", device))
        else:
            dist.init_process_group(backend="gloo")
    pg_dev = "cuda" if args.pg_backend == "nccl" else "cpu"

    # ---- synthetic workload (identical on every rank: seeded) ---------------------------------------------------
    wl = Workload(K, syn, args.workload, max(1, args.scans), device, rank, world, replicas, torch)
    cfg, scans, gmap, tau, n_total, lo, hi = wl.cfg, wl.scans, wl.gmap, wl.tau, wl.n_total, wl.lo, wl.hi
    rel_single, rel_multi = wl.rel_single, wl.rel_multi

    def all_ranks_ok(err):
        """every rank learns whether ANY rank failed and all of them raise together (this rank's own error, or a stand-in)"""
        if use_comm:
            t = torch.tensor([0.0 if err is None else 1.0], dtype=torch.float64, device=pg_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            if float(t.item()) > 0.0 and err is None:
                err = K.KicpError(K.KICP_ERR_COMM, "a peer rank's registration failed")
        if err is not None:
            raise err

    def make_reg(comm, **kw):
        """a registration handle with the requested exchange attached (None: single GPU / replicas)"""
        reg = K.KinematicRegistration(device=device, **kw)  # reference defaults (KinematicICP.hpp:51-56) unless asked otherwise
        for o in args.set:
            reg.set_option(o.split("=")[0], float(o.split("=")[1]))
        keep = []
        if comm == "shm":
            name = "kicp_bench_%s_%s" % (os.environ.get("MASTER_PORT", "0"), os.environ.get("TORCHELASTIC_RUN_ID", "x"))
            err = None
            try:
                if rank == 0:
                    reg.shm_init(world, 0, name)  # removes a stale segment of that name, creates, zeroes and publishes the new one
            except K.KicpError as e:
                err = e
            all_ranks_ok(err)
            try:
                if rank != 0:
                    reg.shm_init(world, rank, name)
            except K.KicpError as e:
                err = e
            all_ranks_ok(err)
        elif comm == "rccl":
            uid = torch.zeros(K.COMM_ID_BYTES, dtype=torch.uint8, device=pg_dev)
            if rank == 0:
                uid.copy_(torch.frombuffer(bytearray(K.comm_unique_id()), dtype=torch.uint8))
            dist.broadcast(uid, 0)
            reg.comm_init(world, rank, bytes(uid.cpu().numpy().tobytes()))
        elif comm == "p2p":
            # (a set-up step that fails on ONE rank - an IPC handle that cannot be opened - must not leave the others at a barrier:
            #  every rank learns of it and all of them raise together)
            err, handle = None, b"\0" * K.P2P_HANDLE_BYTES
            try:
                handle = reg.p2p_export(world, rank)
            except K.KicpError as e:
                err = e
            mine = torch.frombuffer(bytearray(handle), dtype=torch.uint8).to(pg_dev)
            every = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(every, mine)
            all_ranks_ok(err)
            try:
                reg.p2p_connect([bytes(t.cpu().numpy().tobytes()) for t in every])
            except K.KicpError as e:
                err = e
            all_ranks_ok(err)
            dist.barrier()
        elif comm == "torch":
            def allreduce(ptr, count, stream):
                # wrap the device buffer without copying and reduce it in place on the registration's own stream
                class _Arr:
                    __cuda_array_interface__ = {"shape": (count,), "typestr": "<i8", "data": (ptr, False), "version": 2}
                t = torch.as_tensor(_Arr(), device="cuda")
                with torch.cuda.stream(torch.cuda.ExternalStream(stream)):
                    dist.all_reduce(t, op=dist.ReduceOp.SUM)
            reg.set_allreduce(allreduce)
            keep.append(allreduce)
        return reg, keep

    def release(reg, comm):
        if comm == "rccl":
            reg.comm_destroy()
        if comm in ("rccl", "shm", "torch", "p2p"):
            dist.barrier()
        if comm == "shm":
            reg.shm_destroy()
        if comm == "p2p":
            reg.p2p_destroy()

    def barrier():
        if use_comm:
            dist.barrier()
        torch.cuda.synchronize()
        K.lib().kicp_device_synchronize(device)

    def run_scan(reg, i, rels, stats_out=None, w=None):
        w = w or wl
        s = w.scans[i % len(w.scans)]
        pose = reg.ComputeRobotMotion(w.frames[i % len(w.scans)], w.gmap, s["last_pose"], rels[i % len(w.scans)], w.tau)
        if stats_out is not None:
            k = reg.last_stats.iterations
            stats_out.append((k, list(reg.last_stats.pass_ms[:k])))
        return pose

    state = {"B": max(1, args.scans_per_step) if args.scans_per_step > 0 else 64}  # (auto: fixed below, after the calibration batch)

    def timed(reg, rels, steps, warmup, per_call=False, w=None, B=None):
        """W untimed warm-up steps, then EXACTLY `steps` steps of B scans between barriers; max over ranks.
        A step is ONE kicp_register_device_batch call on the step's B independent scans (how many of them the library has in
        flight at a time is the handle's business: options batch_queues / batch_depth) - what a C++ caller of the C-ABI sees.
        per_call=True issues the B calls from Python instead, one scan at a time (adds the interpreter's ~3 us per call).  The host clock is
        read between steps (no synchronisation of any kind: a step ends when its last pose is back), for the median step."""
        w = w or wl
        B = B or state["B"]
        nsc = len(w.scans)
        # step k registers scans k*B .. (k+1)*B-1 of the cycle: with B not a multiple of the cycle every step starts elsewhere in it
        batches, batch_scans = [], []
        for k in range(max(1, min(steps, nsc)) if not per_call else 0):
            idx = [(k * B + i) % nsc for i in range(B)]
            batch_scans.append(idx)
            batches.append(reg.prepare_batch([w.frames[i] for i in idx], [w.scans[i]["last_pose"] for i in idx], [rels[i] for i in idx]))

        def step(k):
            if per_call:
                for i in range(k * B, (k + 1) * B):
                    run_scan(reg, i, rels, w=w)
            else:
                reg.ComputeRobotMotionBatch(batches[k % len(batches)], w.gmap, w.tau)
        # A registration that fails on one rank (an exchange that cannot complete on this box) must not leave the ranks at
        # different collectives: the failing rank still walks through the barriers, and all ranks leave together (all_ranks_ok).
        err = None
        try:
            for k in range(warmup):
                step(k)
        except K.KicpError as e:
            err = e
        barrier()
        marks = [time.perf_counter()]
        t0 = marks[0]
        if err is None:
            try:
                for k in range(steps):
                    step(k)
                    marks.append(time.perf_counter())
            except K.KicpError as e:
                err = e
        barrier()
        elapsed = time.perf_counter() - t0
        all_ranks_ok(err)
        if world > 1:
            t = torch.tensor([elapsed], dtype=torch.float64, device=pg_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        timed.last_iterations = float(np.mean(np.concatenate([b.iterations for b in batches]))) if not per_call else float("nan")  # ICP iterations per scan of this run
        # what the timed steps themselves returned (after the closing barrier: not timed): every batch's poses as its last step left
        # them, with the scans they belong to - compared bit for bit with one call per scan further down (check_timed_region)
        timed.last_results = [] if per_call else [(idx_k, b.out.copy(), np.array(b.iterations).copy()) for idx_k, b in zip(batch_scans, batches)]
        timed.last_step_s = np.diff(np.array(marks))  # this rank's clock
        return elapsed

    comm = args.comm if exchange else None
    comm_note = None
    if comm == "rccl" and (torch.cuda.device_count() < world or "KICP_BENCH_DEVICE" in os.environ):
        comm_note = "RCCL needs one GPU per rank (this run's ranks share a device): the host shared segment carries the exchange instead"
        comm = args.comm = "shm"
    try:
        reg, keep = make_reg(comm)
    except K.KicpError as e:  # e.g. RCCL cannot be initialised on this box: the scaling run still gets a number
        if comm != "rccl":
            raise
        comm_note = "rccl set-up failed (%s): fell back to the host shared segment" % e
        comm = args.comm = "shm"
        reg, keep = make_reg(comm)
    # RCCL with >= 2 ranks could never be tried on hardware in any round (no multi-GPU box): its first use here - a few single
    # registrations, then one small batch, which brings up the lanes' sub-communicators - is a probe.  Should it fail on ANY rank
    # (every wait inside is bounded by KICP_WAIT_TIMEOUT_S), all ranks move to the host shared segment together and say so; the
    # failed handle is kept out of reach of its destructor (a communicator with collectives outstanding may never come back from it).
    leaked = []
    if comm == "rccl" and world > 1:
        probe_err = None
        try:
            for i in range(4):
                run_scan(reg, i, rel_single)
            idx = [i % len(wl.scans) for i in range(8)]
            reg.ComputeRobotMotionBatch(reg.prepare_batch([wl.frames[i] for i in idx], [wl.scans[i]["last_pose"] for i in idx], [rel_single[i] for i in idx]), wl.gmap, wl.tau)
        except K.KicpError as e:
            probe_err = e
        try:
            all_ranks_ok(probe_err)
        except K.KicpError as e:
            leaked.append((reg, keep))
            state["rccl_failed"] = True
            comm_note = "rccl failed on first use (%s): every rank fell back to the host shared segment" % str(e)[:200]
            comm = args.comm = "shm"
            reg, keep = make_reg(comm)
    rccl_ranks = int(reg.get_option("comm_ranks")) if comm == "rccl" else None
    # ---- one-time settling (setup, not measurement): the HIP runtime finishes its lazy initialisation (signal pools,
    #      code objects, clocks) during the first few hundred launches of a process; a ~30 ms hiccup there would
    #      otherwise land inside a short timed region.
    #      With an exchange attached every registration is a collective, so every rank must run the SAME number of scans: the
    #      ranks agree on "another block?" (round 3 let each rank consult its own clock; when the 0.5 s mark fell between two
    #      ranks' checks one of them ran 50 scans more, which the peers never answered: the bounded wait of the exchange
    #      expired 20 s later - the one-in-fifteen failure of the two-rank test).
    if world > 1 and rank == world - 1 and os.environ.get("KICP_BENCH_RANK_SKEW_S"):
        time.sleep(float(os.environ["KICP_BENCH_RANK_SKEW_S"]))  # tests: the ranks' clocks start this far apart
    t_settle = time.perf_counter()
    while True:
        for i in range(50):
            run_scan(reg, i, rel_single)
        go_on = time.perf_counter() - t_settle < 0.5
        if use_comm:
            t = torch.tensor([1.0 if go_on else 0.0], dtype=torch.float64, device=pg_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            go_on = float(t.item()) > 0.0
        if not go_on:
            break
    if args.scans_per_step <= 0:
        # calibration (setup, not measurement): size the batch so that the K timed steps last >= --min-timed-s whatever K is;
        # identical on every rank (max over ranks)
        t_cal = timed(reg, rel_single, 4, 1) / (4 * state["B"])
        state["B"] = int(min(8192, max(64, -(-args.min_timed_s * 1.15 // (max(1, args.steps) * t_cal)))))
    resized = 0
    while True:
        elapsed = timed(reg, rel_single, args.steps, args.warmup)             # ---- the headline number
        step_s = timed.last_step_s.copy()
        results_timed = timed.last_results
        if args.scans_per_step > 0 or elapsed >= args.min_timed_s or state["B"] >= 8192 or resized >= 3:
            break
        # the timed region came in short of --min-timed-s AS MEASURED (a fresh box speeds up after the calibration batch):
        # re-size the batch from this run's own rate and time the K steps again (identical decision on every rank: `elapsed` is
        # the maximum over ranks)
        state["B"] = int(min(8192, max(state["B"] + 1, -(-state["B"] * args.min_timed_s * 1.2 // elapsed))))
        resized += 1
    B = state["B"]
    launch_path = ("direct AQL dispatch on the handle's own HSA queue (kicp_aql.hpp), kernel arguments in %s"
                   % {0.0: "host memory", 1.0: "device memory", 2.0: "device memory + HDP flush"}.get(reg.get_option("aql_kernarg"), "?")
                   if reg.get_option("aql_active") == 1.0 else "hipLaunchKernelGGL on the handle's stream")
    resident_batch = reg.get_option("batch_resident_passes") > 0
    batch_threads = int(max(0, reg.get_option("batch_threads_active")))  # (batches of small scans: resident kernels side by side, a host thread each)
    queued_batch = reg.get_option("batch_queue_passes") > 0
    queues = int(reg.get_option("batch_queues")) if queued_batch else 0
    depth = int(reg.get_option("batch_depth"))
    if queued_batch:
        launch_path += ("; inside a batch call %d scans are in flight at a time, each on a handle and HSA queue of its own (clones of the caller's), every pass "
                        "an ordinary launch of the pass kernel's four-waves-per-SIMD build, ONE host thread going round the scans in flight (rows complete -> "
                        "solve -> next launch): option batch_queues" % queues)
    elif resident_batch:
        launch_path += ("; inside a batch the pass kernel stays RESIDENT across the batch's scans (one launch per batch call, every pass - a scan's first "
                        "one included - started by a command the kernel polls: option batch_resident), with up to %d scans of the batch in flight "
                        "(option batch_depth)" % depth)
        if batch_threads > 1:
            launch_path += ("; %d such kernels side by side, each serving a contiguous part of the batch from a host thread of its own (option batch_threads)" % batch_threads)
    in_flight = queues if queued_batch else ((depth * max(1, batch_threads)) if resident_batch else 1)
    # the same batch calls with ONE scan in flight at a time (the batch's scans strictly one after the other: what a caller gets whose
    # next scan depends on the previous result) - informational, next to the headline
    elapsed_serial = None
    if in_flight > 1 and not exchange:
        reg_serial = reg.copy()
        reg_serial.set_option("batch_queues", 0), reg_serial.set_option("batch_depth", 1), reg_serial.set_option("batch_threads", 0)
        elapsed_serial = timed(reg_serial, rel_single, max(2, args.steps // 4), 1)
        serial_steps = max(2, args.steps // 4)
        del reg_serial
    small_kind = int(reg.get_option("small_active"))  # 0 generic pass kernel; small-scan path (kicp_small.hpp): 1 sub-lanes per query, 2 one wave per query
    small_active = small_kind > 0
    elapsed_multi = timed(reg, rel_multi, args.steps, min(args.warmup, 2))    # ---- same scans, several ICP iterations each
    results_timed_multi = timed.last_results
    elapsed_py = timed(reg, rel_single, args.steps, 1, per_call=True)         # ---- informational: one Python call per scan

    # ---- second pass over the same steps with HIP events around every pass-kernel launch (roofline) -------------
    # (with several scans in flight the timed region launches the four-waves-per-SIMD build of the pass kernel: that is the build
    #  timed here, one launch at a time, and the one the floor, the census and the PMC passes look at)
    latency_kernel_option = reg.get_option("latency_kernel")
    if queued_batch:
        reg.set_option("latency_kernel", 0)
    reg.set_option("timing", 2)
    per_call, per_call_multi = [], []
    n_ev = min(args.steps * B, 2048)
    for i in range(16):
        run_scan(reg, i, rel_single)
    for i in range(n_ev):
        run_scan(reg, i, rel_single, per_call)
    for i in range(min(n_ev, 512)):
        run_scan(reg, i, rel_multi, per_call_multi)
    barrier()
    # fixed floor of a pass and the census behind the latency model (visiting rounds per wave): the pass kernel's ablation switches,
    # which live in libkicp_amd_dbg.so only - tools/dbg_census.py measures both in a process of its own, after everything timed here
    floor_us, rounds_per_wave = None, None
    reg.set_option("timing", 0)
    reg.set_option("latency_kernel", latency_kernel_option)
    poses = [run_scan(reg, i, rel_single) for i in range(len(scans))]
    poses_multi = [run_scan(reg, i, rel_multi) for i in range(len(scans))]
    barrier()

    # ---- the timed region's OWN poses, checked: every pose a timed batch call returned must equal, bit for bit,
    #      the pose of the same scan registered alone by one call (which the epilogue compares with the oracle) - several scans in
    #      flight, another build of the pass kernel, another hand-over: exact integer sums make them the same doubles, or the line is void
    def check_timed_region(results, singles):
        checked = bad = 0
        for idx_k, out, _ in results:
            for j, i in enumerate(idx_k):
                checked += 1
                bad += 0 if np.array_equal(out[j], singles[i], equal_nan=True) else 1
        return checked, bad
    timed_checked, timed_bad = check_timed_region(results_timed, poses)
    timed_checked_multi, timed_bad_multi = check_timed_region(results_timed_multi, poses_multi)
    if timed_bad or timed_bad_multi:
        raise SystemExit("bench.py: %d of %d poses returned inside the timed region (%d of %d on the multi-iteration workload) differ from the pose of the "
                         "same scan registered alone: the measurement is void" % (timed_bad, timed_checked, timed_bad_multi, timed_checked_multi))
    # informational, never `value`: the same calls with the scan handed over as a HOST array (upload inside), as fp64 - the
    # reference's std::vector<Eigen::Vector3d> - and as float32, the wire format of the message the points came in
    host_rate = host_rate_f32 = None
    f32_pose_err = None
    if world == 1:
        nh = min(len(scans), 16)
        host_frames = [np.ascontiguousarray(s["frame"][lo:hi]) for s in scans[:nh]]
        for i in range(8):
            reg.ComputeRobotMotion(host_frames[i % nh], gmap, scans[i % nh]["last_pose"], scans[i % nh]["rel_odom"], tau)
        t1 = time.perf_counter()
        k_host = min(args.steps * B, 400)
        for i in range(k_host):
            reg.ComputeRobotMotion(host_frames[i % nh], gmap, scans[i % nh]["last_pose"], scans[i % nh]["rel_odom"], tau)
        host_rate = k_host / (time.perf_counter() - t1)
        f32_frames = [f.astype(np.float32) for f in host_frames]
        f32_poses = [reg.ComputeRobotMotion(f32_frames[i % nh], gmap, scans[i % nh]["last_pose"], scans[i % nh]["rel_odom"], tau) for i in range(8)]
        t1 = time.perf_counter()
        for i in range(k_host):
            reg.ComputeRobotMotion(f32_frames[i % nh], gmap, scans[i % nh]["last_pose"], scans[i % nh]["rel_odom"], tau)
        host_rate_f32 = k_host / (time.perf_counter() - t1)
    pass_kernel = 3  # (the 16-bit-mirror gather: the only pass kernel since round 6)
    # ---- latency model inputs: the time of one DEPENDENT load step under the pass kernel's own launch shape, far (a working set
    #      of the map's size: probes, buckets, winners) and near (the record next to the probed key: the same line again)
    #      UNLOADED - one wave per CU, so that what is measured is latency and nothing queues: no wave of the pass can take a
    #      dependent step faster than that, whatever else the machine is doing - and, for information, under the pass kernel's own
    #      launch shape with every lane chasing a line of its own (which is bandwidth bound: 131 072 random lines per step)
    lat_far_ns = lat_near_ns = lat_far_loaded_ns = lat_near_loaded_ns = None
    if world == 1 and not small_active:
        try:
            ws = max(gmap.device_bytes(), 1 << 20)
            cus = torch.cuda.get_device_properties(device).multi_processor_count
            idle = dict(workgroups=cus, block=64, steps=256, device=device)
            shape = dict(workgroups=max(1, -(-(hi - lo) // 256)), block=256, steps=64, device=device)
            lat_far_ns, lat_near_ns = K.probe_dependent_load(ws, **idle), K.probe_dependent_load(16 << 10, **idle)
            lat_far_loaded_ns, lat_near_loaded_ns = K.probe_dependent_load(ws, **shape), K.probe_dependent_load(16 << 10, **shape)
        except K.KicpError:
            pass
    if exchange:
        release(reg, comm)
    del reg
    # ---- N > 1: every exchange on the same box in the same run (the headline stays --comm's), each with its scans/s, its time
    #      per ICP iteration on the multi-iteration workload and what the exchange adds per iteration over the same shard
    #      registered WITHOUT any exchange (a plain handle on this rank's points: the kernel and hand-off alone)
    other = {}
    top_exchange = {}

    def measure_exchanges(w, steps, B_w, headline_comm=None, headline=None):
        """every exchange on workload `w`: scans/s, us per ICP iteration, what the exchange adds per iteration"""
        FIXED = dict(max_num_iteration=4, convergence_criterion=0.0)  # every scan runs exactly four iterations, exchange or not

        def rates(reg_x, reg_fixed):
            err = None
            try:
                for i in range(60):
                    run_scan(reg_x, i, w.rel_single, w=w)
            except K.KicpError as e:
                err = e
            all_ranks_ok(err)
            e1 = timed(reg_x, w.rel_single, steps, min(args.warmup, 2), w=w, B=B_w)
            em = timed(reg_x, w.rel_multi, steps, 1, w=w, B=B_w)
            its = timed.last_iterations
            ef = timed(reg_fixed(), w.rel_single, steps, 1, w=w, B=B_w)
            return {"scans_per_s": round(steps * B_w / e1, 1), "us_per_iteration": round(1e6 * em / (steps * B_w) / its, 3),
                    "iterations_per_scan": round(its, 3), "us_per_iteration_at_4_fixed_iterations": round(1e6 * ef / (steps * B_w) / 4.0, 3)}
        exch = {}
        plain = K.KinematicRegistration(device=device)
        held = {}

        def plain_fixed():
            held["r"] = K.KinematicRegistration(device=device, **FIXED)
            return held["r"]
        exch["none (this rank's shard alone, no exchange: NOT a registration of the scan)"] = base = rates(plain, plain_fixed)
        del plain
        held.clear()
        for alt in ("rccl", "shm", "p2p"):
            if alt == "rccl" and (torch.cuda.device_count() < world or "KICP_BENCH_DEVICE" in os.environ):
                exch[alt] = {"note": "skipped: RCCL needs one GPU per rank"}
                continue
            if alt == "rccl" and state.get("rccl_failed"):
                exch[alt] = {"note": "skipped: RCCL failed on its first use in this run (config.comm_note)"}
                continue
            try:
                reg2, keep2 = make_reg(alt)
            except K.KicpError as e:  # e.g. IPC mappings / RCCL unavailable: say so, keep the line
                exch[alt] = {"note": str(e)[:200]}
                continue
            try:
                def fixed_handle():
                    release(held.pop("x"), alt)
                    held["f"], held["k"] = make_reg(alt, **FIXED)
                    return held["f"]
                held["x"] = reg2
                ranks_seen = int(reg2.get_option("comm_ranks")) if alt == "rccl" else None
                r = rates(reg2, fixed_handle)
                r["exchange_us_per_iteration"] = round(r["us_per_iteration_at_4_fixed_iterations"] - base["us_per_iteration_at_4_fixed_iterations"], 3)
                if ranks_seen is not None:
                    r["rccl_ranks"] = ranks_seen
                exch[alt] = r
                release(held.pop("f"), alt)
                held.clear()
                if alt == "shm":  # the same exchange with ONE sharded scan in flight (round 4's rate: a scan at a time, a hand-off per pass)
                    held["s"], held["ks"] = make_reg(alt)
                    held["s"].set_option("batch_queues", 0)
                    es = timed(held["s"], w.rel_single, steps, 1, w=w, B=B_w)
                    r["scans_per_s_one_scan_in_flight"] = round(steps * B_w / es, 1)
                    r["scans_in_flight"] = int(reg2.get_option("batch_queues")) if reg2.get_option("batch_queue_passes") > 0 else 1
                    release(held.pop("s"), alt)
                    held.clear()
            except K.KicpError as e:
                exch[alt] = {"note": str(e)[:200]}
                held.clear()
        return exch

    sharded_cfg5 = None
    if exchange:
        exch = measure_exchanges(wl, args.steps, B)
        other["exchanges"] = exch
        for alt in ("rccl", "shm", "p2p"):
            top_exchange["value_" + alt] = exch.get(alt, {}).get("scans_per_s")
        top_exchange["value_" + comm] = None  # (filled with the headline below: the same exchange, the contract's timed region)
        for alt in ("shm", "p2p"):  # (kept under their round-2 names too)
            if "scans_per_s" in exch.get(alt, {}):
                other[alt + "_scans_per_s"] = exch[alt]["scans_per_s"]
        if world > 1 and not args.no_sharded_cfg5 and args.workload != "cfg5":
            # the workload where sharding the points can pay: cfg5's 500 000-point scans fill the machine (VALU-bound pass of ~56 us
            # on one GPU); 62 500 points per GPU at N = 8
            try:
                w5 = Workload(K, syn, "cfg5", 4, device, rank, world, False, torch)
                steps5 = max(3, min(args.steps, 10))
                ex5 = measure_exchanges(w5, steps5, 32)
                sharded_cfg5 = {"workload": "cfg5: %d-pt scan vs %d-pt / %d-voxel map, voxel %.2f m; %d points per GPU; %d steps of 32 scans"
                                            % (w5.n_total, w5.gmap.num_points(), w5.gmap.num_voxels(), w5.cfg.voxel_size, w5.hi - w5.lo, steps5),
                                "value_rccl": ex5.get("rccl", {}).get("scans_per_s"), "value_shm": ex5.get("shm", {}).get("scans_per_s"),
                                "value_p2p": ex5.get("p2p", {}).get("scans_per_s"), "exchanges": ex5}
                del w5
            except (K.KicpError, MemoryError) as e:
                sharded_cfg5 = {"note": str(e)[:300]}
    # ---- N > 1, informational: the SAME GPUs as independent replicas - every rank registers whole scans on its own (no exchange),
    #      the way a fleet localising in one map or a replayed log would use a node.  Sharding one 131 072-point scan splits a pass of
    #      a few microseconds and adds an exchange per ICP iteration; replicas multiply the single-GPU rate.  Both are printed: the
    #      headline stays the sharded (north-star) figure.
    value_replicas = None
    if exchange and world > 1:
        err, el = None, float("nan")
        rep_steps, rep_B, nrep = 8, 256, min(len(scans), 16)
        full = batch_rep = reg_rep = None
        try:
            full = [K.DeviceFrame(s["frame"], device=device) for s in scans[:nrep]]
            reg_rep = K.KinematicRegistration(device=device)
            batch_rep = reg_rep.prepare_batch([full[i % nrep] for i in range(rep_B)], [scans[i % nrep]["last_pose"] for i in range(rep_B)],
                                              [rel_single[i % nrep] for i in range(rep_B)])
            for _ in range(2):
                reg_rep.ComputeRobotMotionBatch(batch_rep, gmap, tau)
        except (K.KicpError, MemoryError) as e:
            err = e
        flag = torch.tensor([0.0 if err is None else 1.0], dtype=torch.float64, device=pg_dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        if float(flag.item()) == 0.0:
            barrier()
            t0 = time.perf_counter()
            try:
                for _ in range(rep_steps):
                    reg_rep.ComputeRobotMotionBatch(batch_rep, gmap, tau)
            except K.KicpError as e:
                err = e
            barrier()
            t = torch.tensor([time.perf_counter() - t0, 0.0 if err is None else 1.0], dtype=torch.float64, device=pg_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            if float(t[1].item()) == 0.0:
                el = float(t[0].item())
                value_replicas = {"scans_per_s": round(world * rep_steps * rep_B / el, 1), "scaling": "weak",
                                  "what": "every rank registers WHOLE scans on its own GPU, no exchange (%d batch calls of %d scans per rank between "
                                          "barriers, max over ranks; the batch call's default: scans in flight on queues of their own): what a node "
                                          "does with independent scans; `value` above is the north star's sharded registration of ONE scan at a "
                                          "time across the ranks" % (rep_steps, rep_B)}
        if value_replicas is None:
            value_replicas = {"note": "not measured: %s" % (str(err)[:200] if err is not None else "a peer rank failed")}
        del full, batch_rep, reg_rep
    if use_comm:  # all GPU work is done: tear the process group down before rank 0's CPU-only epilogue
        dist.barrier()
        dist.destroy_process_group()
    if rank != 0:
        return

    # ---- algorithmic bytes of the passes actually executed (counted by the oracle = the reference's own work) ---
    from oracle import okicp, rkicp
    omap = okicp.VoxelHashMap(cfg.voxel_size, cfg.max_range, cfg.max_points_per_voxel)
    map_points = gmap.Pointcloud()
    omap.AddPoints(map_points)
    oreg = okicp.KinematicRegistration(max_num_threads=0)
    balgo_pass, bmin_pass, iters_ref, iters_ref_multi, max_pose_err = [], [], [], [], 0.0
    vox_keys, vox_counts = _voxel_census(map_points, cfg.voxel_size)
    n_check = min(len(scans), 16)  # (the checker is the slow side: every scan's pose is compared on the multi-iteration workload, a sample carries the byte counts)
    for rels, ps, it_out, count in ((rel_single, poses, iters_ref, True), (rel_multi, poses_multi, iters_ref_multi, False)):
        for k_scan, (s, rel, pose) in enumerate(zip(scans, rels, ps)):
            if k_scan >= n_check and count:
                continue
            ref = oreg.ComputeRobotMotion(s["frame"], omap, s["last_pose"], rel, tau, count_work=count)
            st = oreg.last_stats
            it_out.append(st.iterations)
            max_pose_err = max(max_pose_err, float(np.max(np.abs(pose - ref))))
            if not count:
                continue
            for k in range(st.iterations):
                # B_algo(pass) = 12 N_q + 16 P + 12 S  (fp32 xyz per point, 16 B per probed slot)
                balgo_pass.append(12 * n_total + 16 * int(st.probes[k]) + 12 * int(st.points_scanned[k]))
            # compulsory bytes of the first pass: every query once, every touched slot and bucket point once
            q = okicp.se3_act(okicp.se3_mul(s["last_pose"], rel), s["frame"])
            v_touched, m_touched = _touched(q, cfg.voxel_size, vox_keys, vox_counts)
            bmin_pass.append(12 * n_total + 12 * m_touched + 16 * v_touched)
    if world == 1 and host_rate_f32 is not None:  # the float32 entry point against the checker on the widened frames
        f32_pose_err = 0.0
        for i in range(min(4, len(f32_poses))):
            wide = f32_frames[i].astype(np.float64)
            ref = oreg.ComputeRobotMotion(wide, omap, scans[i]["last_pose"], scans[i]["rel_odom"], tau)
            f32_pose_err = max(f32_pose_err, float(np.max(np.abs(f32_poses[i] - ref))))
    share = 1 if replicas else world  # each rank's launch covers its shard
    bytes_per_launch = float(np.mean(balgo_pass)) / share
    bmin_per_launch = float(np.mean(bmin_pass)) / share
    pass_ms = np.array([ms for _, lst in per_call for ms in lst], dtype=np.float64)
    iters_gpu = float(np.mean([it for it, _ in per_call]))
    iters_gpu_multi = float(np.mean([it for it, _ in per_call_multi])) if per_call_multi else float("nan")
    kernel_us = float(pass_ms.mean() * 1e3) if pass_ms.size else float("nan")
    kernel_time_source = "HIP events around every pass launch on the handle's stream (the event pair adds ~2 us to the ~kernel-trace duration)"
    if small_active and np.isfinite(iters_gpu_multi):
        # a resident kernel serves all passes of a call and cannot be bracketed per pass: wall clock per ICP iteration instead
        kernel_us = 1e6 * elapsed_multi / (args.steps * B) / iters_gpu_multi
        kernel_time_source = ("wall clock per ICP iteration of the multi-iteration run (the small-scan kernel stays resident for a call's passes; "
                              "includes the host-side solve and the command round trip)")
    achieved = bytes_per_launch / (kernel_us * 1e-6) / 1e9 if pass_ms.size else None
    pass_ms_multi = np.array([ms for _, lst in per_call_multi for ms in lst], dtype=np.float64)

    cpu = None if args.no_cpu_baseline else _cpu_baseline(args, cfg, scans[:8], rel_single[:8], tau, omap, map_points, okicp, rkicp)
    pipeline = _pipeline_block(args.pipeline_frames, with_reference=not args.no_cpu_baseline) if (world == 1 and args.pipeline_frames > 0 and args.workload == "cfg2") else None

    n_scans_timed = args.steps * B
    value = (world if replicas else 1) * n_scans_timed / elapsed  # replicas: every rank completed its own scans
    if exchange:
        top_exchange["value_" + comm] = round(value, 2)
    # ---- HBM traffic of the pass kernel, measured in THIS run: one rocprofv3 --pmc pass per counter over a bare loop of the
    #      same registrations (tools/prof_target.py), after everything timed is over.  Falls back to the committed profile
    #      (stamped with the commit it was taken at) where rocprofv3 cannot run.
    kernel_sub = {0: "k_pass_gather32", 1: "k_pass_small", 2: "k_pass_wave"}[small_kind]
    traffic, traffic_src = (None, "not measured (--no-pmc)") if (args.no_pmc or world != 1) else _pmc_traffic(args.workload, kernel_sub)
    # informational: INDEPENDENT scans with four in flight, a host thread each (kicp_register_device_concurrent: one handle, HSA queue
    # and host thread per lane) - what the device does when a workload has several scans to offer at a time (robots sharing a
    # map, replayed logs).  The batch call of the timed region keeps as many in flight with ONE host thread.
    conc_lanes = 4
    conc_rate = _concurrent_rate(args.workload, conc_lanes, min(len(scans), 8)) if (world == 1 and not use_comm) else None
    if not use_comm and not small_active and world == 1:
        floor_us, rounds_per_wave = _dbg_census(args.workload, 0 if queued_batch else int(latency_kernel_option))
    prof = _profile_counters(args.workload, world)
    if traffic is None and prof and prof.get("hbm_bytes_per_launch"):
        traffic = float(prof["hbm_bytes_per_launch"])
        traffic_src = "%s; in-run measurement unavailable: %s" % (prof.get("source"), traffic_src)
    t_kernel = kernel_us * 1e-6
    traffic_gbs = None if (traffic is None or not pass_ms.size) else traffic / t_kernel / 1e9
    # ---- the bound this kernel CAN be held to: its waves' chains of dependent accesses on top of the fixed floor.  A wave's chain:
    #      source point -> probe of the own voxel's slot -> per visiting round [bucket record (the line just probed: near) -> the
    #      bucket's points (far)] -> the winner's fp64 point (far).  Rounds per wave are counted by the kernel itself (dbg 10),
    #      the price of a dependent step by kicp_probe_dependent_load under the same launch shape, the floor by the same launch
    #      with every query switched off.
    latency = None
    if floor_us is not None and rounds_per_wave is not None and lat_far_ns and lat_near_ns and pass_ms.size:
        far_steps = 3.0 + rounds_per_wave
        bound_us = floor_us + (far_steps * lat_far_ns + rounds_per_wave * lat_near_ns) * 1e-3
        latency = {"floor_us": round(floor_us, 2), "rounds_per_wave": round(rounds_per_wave, 3), "dependent_far_steps_per_wave": round(far_steps, 3),
                   "dependent_near_steps_per_wave": round(rounds_per_wave, 3), "far_step_ns": round(lat_far_ns, 1), "near_step_ns": round(lat_near_ns, 1),
                   "far_step_ns_under_the_kernels_launch_shape": None if lat_far_loaded_ns is None else round(lat_far_loaded_ns, 1),
                   "near_step_ns_under_the_kernels_launch_shape": None if lat_near_loaded_ns is None else round(lat_near_loaded_ns, 1),
                   "far_working_set_bytes": int(gmap.device_bytes()), "latency_bound_us": round(bound_us, 2),
                   "what": "latency_bound_us = floor_us + (3 + rounds) x far_step_ns + rounds x near_step_ns: the pass cannot end before a wave "
                           "has walked source point -> probe -> rounds x (bucket record, bucket) -> winner, and no wave takes a dependent step "
                           "faster than the UNLOADED machine does: far / near step = kicp_probe_dependent_load with one wave per CU, every lane "
                           "chasing its own chain through a buffer of the map's size / of 16 KB (the same probe under the kernel's own launch "
                           "shape - 131 072 lanes, a random line each: bandwidth, not latency - is given beside it); rounds = mean visiting "
                           "rounds per wave counted by the kernel (dbg 10); floor = the same launch with every query off (launch, reduction, "
                           "hand-off; HIP events).  frac_latency = latency_bound_us / kernel_avg_us (1 = at the bound)"}
    # ---- the timed region's own figure: passes overlap there (several scans in flight), so what one pass costs is wall clock per
    #      pass, and what bounds THAT is the machine's throughput, not one wave's chain: the SIMDs' VALU issue slots.  A wave64 VALU
    #      instruction occupies its SIMD for 4 cycles; instructions per launch from the committed counter profile (SQ_INSTS_VALU).
    us_per_pass = 1e6 * elapsed / n_scans_timed / max(iters_gpu, 1e-9)
    valu_issue = None
    insts = ((prof or {}).get("raw", {}).get("SQ_INSTS_VALU", {}) or {}).get("mean_per_dispatch") if prof else None
    insts_src = (prof or {}).get("source", "profiles/")
    in_run = getattr(_pmc_traffic, "insts_valu", None)
    if in_run:  # measured in THIS run, at this commit, on this box (a third --pmc pass next to FETCH_SIZE and WRITE_SIZE)
        insts, insts_src = in_run[0], "this run: rocprofv3 --pmc SQ_INSTS_VALU over the same bare loop as the traffic passes, mean over %d dispatches" % in_run[1]
    if insts and world == 1:
        simds = 4 * torch.cuda.get_device_properties(device).multi_processor_count
        bound = insts * 4.0 / (simds * SCLK_GHZ * 1e3)
        valu_issue = {"insts_valu_per_launch": round(insts), "cycles_per_wave_instruction": 4, "simds": simds, "sclk_GHz": SCLK_GHZ,
                      "bound_us_per_pass": round(bound, 2), "frac": round(bound / us_per_pass, 4),
                      "what": "wave-level VALU instructions of one pass (SQ_INSTS_VALU per dispatch of the pass kernel, %s) x 4 cycles / (%d SIMDs x "
                              "%.1f GHz): the time the machine's VALU issue slots need for one pass if they never idle; frac = that / the wall clock "
                              "per pass of the timed region" % (insts_src, simds, SCLK_GHZ)}
    roof = {"bound": "hbm", "achieved": None if traffic_gbs is None else round(traffic_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": None if traffic_gbs is None else round(traffic_gbs / HBM_PEAK_GBS, 4),
            "frac_latency": None if latency is None else round(latency["latency_bound_us"] / kernel_us, 4),
            "latency_model": latency,
            "traffic": None if traffic is None else round(traffic), "traffic_source": traffic_src,
            "kernel": "fused association+accumulation pass (%s)" % kernel_sub, "kernel_avg_us": round(kernel_us, 2),
            "kernel_in_the_timed_region": None if not (resident_batch or queued_batch) else
            {"kernel": ("k_pass_gather32 in its four-waves-per-SIMD build, one launch per pass on %d HSA queues at a time (one per scan in flight): a kernel "
                        "trace of this command shows it with an average duration of about scans_in_flight x us_per_pass_wall_clock - the launches "
                        "overlap - while kernel_avg_us above is ONE launch at a time (the event-timed calls)" % queues) if queued_batch else
                       ("k_pass_resident (generic scans) / k_pass_wave (small scans): the same pass code, resident across the scans of one "
                        "kicp_register_device_batch call - ONE dispatch per step, so a kernel trace of this command shows it with an average duration "
                        "of about ms_per_step, and %s with the per-pass duration quoted here (the event-timed calls, one launch per pass)" % kernel_sub),
             "scans_in_flight": in_flight,
             "us_per_pass_wall_clock": round(us_per_pass, 3),
             "kernel_avg_us_expected_in_a_trace": round(in_flight * us_per_pass, 2) if queued_batch else None,
             "hbm_GBps_sustained": None if traffic is None else round(traffic / (us_per_pass * 1e-6) / 1e9, 1),
             "hbm_frac_sustained": None if traffic is None else round(traffic / (us_per_pass * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
             "valu_issue": valu_issue},
            "kernel_time_source": kernel_time_source, "launches_timed": int(pass_ms.size),
            "what": "achieved = HBM bytes the pass kernel moved per launch (2 x FETCH_SIZE + WRITE_SIZE, MI355X_MICROARCH.md's gfx950 correction) / its "
                    "average duration; frac = achieved / 8 TB/s.  The kernel is bound by its waves' dependent-load chains and a fixed launch + "
                    "reduction floor, not by HBM: frac_latency holds it to THAT bound (latency_model), time_split_us and counters show the rest",
            "algorithmic": {"bytes_per_launch": round(bytes_per_launch),
                            "GBps": None if achieved is None else round(achieved, 1),
                            "ratio_to_hbm_peak": None if achieved is None else round(achieved / HBM_PEAK_GBS, 4),
                            "what": "SURVEY.md section 8d: what the REFERENCE's search touches per pass (12 B per query + 16 B per probed slot, 27 per "
                                    "query, + 12 B per bucket point it scans, counted by the oracle) / the kernel's duration.  NOT a roofline fraction: "
                                    "this kernel skips provably irrelevant voxels and reads a 16-bit mirror, so it moves a fraction of these bytes and "
                                    "the ratio can exceed 1"},
            "b_min": {"bytes_per_launch": round(bmin_per_launch),
                      "achieved": None if achieved is None else round(bmin_per_launch / t_kernel / 1e9, 1),
                      "frac": None if achieved is None else round(bmin_per_launch / t_kernel / 1e9 / HBM_PEAK_GBS, 4),
                      "what": "compulsory bytes: 12 B per query + every touched table slot (16 B) and bucket point (12 B) once",
                      "traffic_over_b_min": None if traffic is None else round(traffic / bmin_per_launch, 2)},
            "time_split_us": None if floor_us is None else {"fixed_floor_launch_reduction_handoff": round(floor_us, 2),
                                                            "query_work": round(kernel_us - floor_us, 2)},
            "counters": prof or None}
    med_step = float(np.median(step_s)) if step_s.size else float("nan")
    out = {
        "metric": "scans/sec (ICP registration only), 128k-pt scan vs 1M-pt map",
        "value": round(value, 2),
        **({} if elapsed_serial is None else {"value_one_scan_in_flight": round(serial_steps * B / elapsed_serial, 2)}),
        "unit": "scans/s",
        "n_gpus": world,
        **({"comm": comm, "scaling_bound": _scaling_bound(world)} if exchange and world > 1 else {}),
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(1e3 * elapsed / args.steps, 5),
        "higher_is_better": True,
        "scaling": "weak" if replicas else "strong",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": "%s: %d-pt %d-beam scan vs %d-pt / %d-voxel map, voxel %.2f m, tau %.4f m, default ICP iterations "
                               "(mean %.2f per scan, reference %.2f); one step = one kicp_register_device_batch call = %d independent scans against the fixed map, %s; "
                               "%d distinct scans cycled"
                               % (cfg.name, n_total, cfg.n_beams, gmap.num_points(), gmap.num_voxels(), cfg.voxel_size, tau, iters_gpu,
                                  float(np.mean(iters_ref)), B,
                                  ("up to %d in flight at a time (every scan's result bit-identical to registering it alone)" % in_flight) if in_flight > 1
                                  else "registered one after the other", len(scans)),
                   "scans_in_flight": in_flight,
                   "scans_per_step": B, "distinct_scans": len(scans), "ms_per_scan": round(1e3 * elapsed / n_scans_timed, 5), "timed_region_s": round(elapsed, 4),
                   "batch_resized_after_short_timed_region": resized,
                   "ms_per_step_median": round(1e3 * med_step, 5),
                   "ms_per_step_min_max": [round(1e3 * float(step_s.min()), 5), round(1e3 * float(step_s.max()), 5)] if step_s.size else None,
                   "points_per_gpu": hi - lo,
                   "scans_per_s_one_python_call_per_scan": round((world if replicas else 1) * n_scans_timed / elapsed_py, 2),
                   "parallelism": ("%d independent replicas (one robot per GPU), no exchange" % world) if replicas else
                                  (("points sharded x%d, map replicated, %s exchange of the 24 limb words per ICP pass, %d sharded scan(s) in flight per rank"
                                    % (world, args.comm, in_flight)) if use_comm else "single GPU"),
                   "pass_kernel": pass_kernel, "launch_path": launch_path, "host_placement": host_placement, "max_pose_abs_diff_vs_oracle": max_pose_err,
                   "poses_checked_against_the_oracle": len(iters_ref_multi) + len(iters_ref),
                   "poses_of_the_timed_region_checked": timed_checked + timed_checked_multi,
                   "poses_of_the_timed_region_check": "every pose the timed batch calls returned (headline %d, multi-iteration %d; the batches as their last step left "
                                                      "them) is bit-identical to the pose of the same scan registered alone by one call; those single-call poses are "
                                                      "the ones compared with the oracle (max_abs_pose_diff_vs_oracle)" % (timed_checked, timed_checked_multi),
                   "multi_iteration": {
                       "workload": "same scans, odometry error +%.2f m / +%.1f deg: %.2f ICP iterations per scan (reference %.2f)"
                                   % (MULTI_ITER_ERROR[0], MULTI_ITER_ERROR[1], iters_gpu_multi, float(np.mean(iters_ref_multi))),
                       "scans_per_s": round((world if replicas else 1) * n_scans_timed / elapsed_multi, 2),
                       "ms_per_scan": round(1e3 * elapsed_multi / n_scans_timed, 5),
                       "ms_per_iteration": None if not np.isfinite(iters_gpu_multi) else round(1e3 * elapsed_multi / n_scans_timed / iters_gpu_multi, 5),
                       "pass_kernel_avg_us": round(float(pass_ms_multi.mean() * 1e3), 2) if pass_ms_multi.size else None},
                   "scans_per_s_with_host_input_incl_pcie": None if host_rate is None else round(host_rate, 1), **other,
                   **({"comm_note": comm_note} if comm_note else {})},
        "value_median_step": None if not np.isfinite(med_step) else
        {"scans_per_s": round((world if replicas else 1) * B / med_step, 2), "ms_per_step": round(1e3 * med_step, 5),
         "what": "the median of the %d timed steps (rank 0's clock between steps; `value` is the contract's K steps / elapsed)" % args.steps},
        "value_multi_iteration": {"scans_per_s": round((world if replicas else 1) * n_scans_timed / elapsed_multi, 2),
                                  "iterations_per_scan": None if not np.isfinite(iters_gpu_multi) else round(iters_gpu_multi, 3),
                                  "us_per_iteration": None if not np.isfinite(iters_gpu_multi) else round(1e6 * elapsed_multi / n_scans_timed / iters_gpu_multi, 3),
                                  "what": "the same scans with +%.2f m / +%.1f deg odometry error, same steps and batch" % MULTI_ITER_ERROR},
        "value_host_vector_input": None if host_rate is None else
        {"scans_per_s": round(host_rate, 1), "what": "kicp_register with the scan handed over as a HOST array, the reference's own signature "
                                                      "(Registration.hpp:39-43): upload over PCIe inside the call",
         "scans_per_s_float32": None if host_rate_f32 is None else round(host_rate_f32, 1),
         "float32_what": "kicp_register_f32: the same scans as float32 xyz (what the PointCloud2 carried, RosUtils.cpp:30-39), widened on the device",
         "float32_max_pose_abs_diff_vs_oracle_on_widened_frames": f32_pose_err},
        "value_concurrent_independent_scans": None if conc_rate is None else
        {"scans_per_s": round(conc_rate, 1), "lanes": conc_lanes,
         "what": "kicp_register_device_concurrent: the same scans as INDEPENDENT registrations, %d in flight (one handle + HSA queue + host thread each), "
                 "512 per call, median of 5 calls, measured by tools/bench_concurrent.py in a process of its own after everything timed here; "
                 "the batch call of the timed region keeps as many scans in flight from ONE host thread; what neither can serve is a caller whose next "
                 "scan depends on the previous result (the reference's own pipeline): value_one_scan_in_flight and bench_pipeline are that caller's rates" % conc_lanes},
        **top_exchange,
        **({} if value_replicas is None else {"value_replicas": value_replicas}),
        **({"rccl_ranks": rccl_ranks} if rccl_ranks is not None else {}),
        **({"sharded_cfg5": sharded_cfg5} if sharded_cfg5 is not None else {}),
        "roofline": roof,
        "cpu_baseline": cpu,
        **({} if pipeline is None else {"pipeline": pipeline}),
    }
    if elapsed < 0.25:
        out["config"]["warning"] = "timed region shorter than 0.25 s: raise --steps or --scans-per-step (0 = automatic)"
    import ctypes
    ctypes.CDLL(None).fflush(None)
    sys.stdout.flush()
    os.dup2(saved_stdout, 1)
    print(json.dumps(out), flush=True)
    if leaked:  # (see the probe above: no destructor for the failed RCCL handle)
        os._exit(0)


def _scaling_bound(world):
    """What sharding ONE scan's points over N GPUs can reach at best: a pass is a fixed floor (dispatch, wave launch, reduction,
    hand-off) plus query work that divides by N, so the speed-up is bounded by pass / (floor + (pass - floor) / N) even with a free
    exchange (Amdahl).  Floor and pass time are the single-GPU run's own (the newest committed profiles/r*_bench_n1.json:
    roofline.time_split_us); beyond that bound only independent replicas scale (value_replicas)."""
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_n1.json")), reverse=True) + sorted(glob.glob(os.path.join(ROOT, "profiles", "history", "r05_final_bench_n1.json"))):
        try:
            d = json.load(open(f))
            ts = d["roofline"]["time_split_us"]
            floor, work = float(ts["fixed_floor_launch_reduction_handoff"]), float(ts["query_work"])
            return {"speedup_bound_sharded": round((floor + work) / (floor + work / world), 2), "floor_us": floor, "query_work_us": work,
                    "from": os.path.basename(f), "what": "(floor + work) / (floor + work / N): one scan's pass sharded over N GPUs with a free exchange"}
        except Exception:  # noqa: BLE001
            continue
    return None


def _dbg_census(workload, latency_kernel):
    """floor of a pass (every query off) and visiting rounds per wave of the generic pass kernel, by tools/dbg_census.py on the library
    build that carries the kernels' ablation switches (libkicp_amd_dbg.so); (None, None) where that build is absent"""
    import subprocess
    if not os.path.exists(os.path.join(ROOT, "kinematic_icp_amd", "libkicp_amd_dbg.so")):
        return None, None
    try:
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "dbg_census.py"), "--workload", workload, "--latency-kernel", str(latency_kernel)],
                             capture_output=True, text=True, timeout=300)
        d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
        return float(d["floor_us"]), float(d["rounds_per_wave"])
    except Exception:  # noqa: BLE001
        return None, None


def _pipeline_block(frames, with_reference=True):
    """The frame, not the registration alone: the drop-in KinematicICP::RegisterFrame (pipeline/KinematicICP.cpp:48-85: pre-steps,
    registration, threshold, map update - all on the GPU) on a synthetic drive of 131 072-point frames that arrive as PointCloud2
    bytes, measured by tools/bench_pipeline.py in a process of its own (the C++ facade test binary is the caller, as a ROS node would
    be), after everything timed above.  Not part of `value`."""
    import subprocess
    cmd = [sys.executable, os.path.join(ROOT, "tools", "bench_pipeline.py"), "--frames", str(frames), "--json", "raw,raw_ahead",
           "--ref-frames", str(frames if with_reference else 0), "--ref-threads", "1", "16"]
    try:
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
        line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
        block = json.loads(line)
        block["what"] = ("whole frames through the drop-in KinematicICP (IngestCloud + RegisterIngestedFrame): wire-format ingest, deskew + crop + two voxel "
                         "downsamples, registration, adaptive threshold, map update, the two returned clouds back in host vectors; "
                         "ms_per_frame = the clock around those calls, frames_per_s = wall clock around the drive loop")
        return block
    except Exception as e:  # noqa: BLE001 - the block is an extra: its failure must not cost the run its headline line
        return {"error": "%s: %s" % (type(e).__name__, e)}


def _voxel_census(points, vs):
    """sorted packed voxel keys of the map and the number of points in each"""
    v = np.floor(points / vs).astype(np.int64) + (1 << 20)
    keys = (v[:, 0] << 42) | (v[:, 1] << 21) | v[:, 2]
    return np.unique(keys, return_counts=True)


def _touched(queries, vs, vox_keys, vox_counts):
    """distinct table slots probed by the 27-voxel searches of `queries`, and the map points in the occupied ones"""
    v = np.floor(queries / vs).astype(np.int64) + (1 << 20)
    sh = np.array([(dx, dy, dz) for dx in (-1, 0, 1) for dy in (-1, 0, 1) for dz in (-1, 0, 1)], dtype=np.int64)
    qk = np.unique((v[:, 0] << 42) | (v[:, 1] << 21) | v[:, 2])
    x, y, z = qk >> 42, (qk >> 21) & 0x1FFFFF, qk & 0x1FFFFF
    nb = np.unique((((x[:, None] + sh[:, 0]) << 42) | ((y[:, None] + sh[:, 1]) << 21) | (z[:, None] + sh[:, 2])).ravel())
    pos = np.searchsorted(vox_keys, nb)
    pos[pos >= len(vox_keys)] = len(vox_keys) - 1
    hit = vox_keys[pos] == nb
    return int(len(nb)), int(vox_counts[pos[hit]].sum())


def _cpu_baseline(args, cfg, scans, rels, tau, omap, map_points, okicp, rkicp):
    """The reference's own Registration.cpp (oracle/_ref) on this box's host cores, a bounded sample of the same scans:
    1 thread (the reference's default max_num_threads) and the best of several thread counts (its TBB stand-in cuts the
    scan into equal chunks on OpenMP threads); the oracle port the same way, for comparison."""
    ncores = okicp.lib().okicp_max_threads()
    counts = sorted({c for c in (1, 8, 16, 32, 64, ncores) if 1 <= c <= ncores})
    budget = max(2.0, args.cpu_seconds) / (2 * len(counts))

    def sample(fn):
        fn(0)  # warm the caches / the thread team
        t0, done = time.perf_counter(), 0
        while True:
            fn(done)
            done += 1
            if time.perf_counter() - t0 > budget or done >= 400:
                break
        return done / (time.perf_counter() - t0), done

    port, port_n = {}, 0
    for c in counts:
        oreg = okicp.KinematicRegistration(max_num_threads=c)
        port[c], k = sample(lambda i: oreg.ComputeRobotMotion(scans[i % len(scans)]["frame"], omap, scans[i % len(scans)]["last_pose"], rels[i % len(scans)], tau))
        port_n += k
    res = {"unit": "scans/s", "cpu_model": _cpu_model(), "host_cores": ncores, "cgroup_cpu_max": _cgroup_cpu_max(),
           "port_by_threads": {str(c): round(v, 3) for c, v in port.items()}}
    if rkicp.available():
        rmap = rkicp.VoxelHashMap(cfg.voxel_size, cfg.max_range, cfg.max_points_per_voxel)
        rmap.AddPoints(map_points)
        ref, ref_n = {}, 0
        for c in counts:
            rreg = rkicp.KinematicRegistration(max_num_threads=c)

            def call(i, rreg=rreg):  # ComputeRobotMotion alone: the array -> std::vector conversion is excluded
                s = scans[i % len(scans)]
                _, sec = rreg.timed(s["frame"], rmap, s["last_pose"], rels[i % len(scans)], tau, 1)
                call.seconds += sec
            call.seconds = 0.0
            _, k = sample(call)
            ref[c] = (k + 1) / call.seconds
            ref_n += k
        best = max(ref, key=ref.get)
        res.update({"value": round(ref[best], 3), "cores": best, "kind": "reference",
                    "sample": "%d ComputeRobotMotion calls of the reference's own Registration.cpp (oracle/_ref: compiled unmodified against "
                              "stand-in Eigen/Sophus/TBB/robin_map/kiss-icp headers, -O3, no -march) on the same %d scans, ~%.0f s; "
                              "threads tried %s, best reported" % (ref_n, len(scans), args.cpu_seconds / 2, counts),
                    "single_thread_value": round(ref[1], 3), "reference_by_threads": {str(c): round(v, 3) for c, v in ref.items()}})
        # THROUGHPUT mode, the CPU's twin of the GPU's scans-in-flight headline: min(cores the container may use, 16) INDEPENDENT
        # registrations at a time, one thread each (the reference's own default, ros/launch/offline_node.launch.py:60) - in a process
        # of its own (tools/bench_cpu_throughput.py)
        res["throughput"] = _cpu_throughput(cfg, scans, rels, tau, map_points, max(2.0, args.cpu_seconds / 4))
    else:
        best = max(port, key=port.get)
        res.update({"value": round(port[best], 3), "cores": best, "kind": "port",
                    "sample": "%d calls of the oracle port on the same %d scans (oracle/_ref not present); threads tried %s, best reported"
                              % (port_n, len(scans), counts), "single_thread_value": round(port[1], 3)})
    return res


def _cpu_throughput(cfg, scans, rels, tau, map_points, seconds):
    """scans/s of the reference build with several one-thread registrations running side by side, or a note"""
    import subprocess
    import tempfile
    quota = _cgroup_cpu_max()
    try:
        q, per = (quota or "max").split()[:2]
        cores = max(1, int(float(q) / float(per))) if q != "max" else (os.cpu_count() or 1)
    except Exception:  # noqa: BLE001
        cores = os.cpu_count() or 1
    threads = max(1, min(cores, 16))
    env = {k: v for k, v in os.environ.items() if k not in ("OMP_PROC_BIND", "OMP_PLACES")}
    env["OMP_NUM_THREADS"] = "1"
    try:
        with tempfile.TemporaryDirectory() as td:
            f = os.path.join(td, "cpu.npz")
            np.savez(f, map=map_points, frames=np.stack([s["frame"] for s in scans]), last=np.stack([s["last_pose"] for s in scans]), rel=np.stack(rels),
                     tau=tau, voxel=cfg.voxel_size, max_range=cfg.max_range, cap=cfg.max_points_per_voxel)
            out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "bench_cpu_throughput.py"), f, str(threads), "%.1f" % seconds], env=env,
                                 capture_output=True, text=True, timeout=120 + 4 * seconds, cwd=ROOT)
        line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
        line["what"] = ("%d independent ComputeRobotMotion calls of the reference build at a time, ONE thread each (its default), the same scans, ~%.0f s: "
                        "what a CPU does with independent scans - to be held against `value` (scans in flight); `value` / `single_thread_value` of this "
                        "object are the one-call-at-a-time rates, to be held against value_one_scan_in_flight" % (threads, seconds))
        return line
    except Exception as e:  # noqa: BLE001  (informational figure: a failure here must not cost the bench line)
        return {"note": "not measured: %s" % str(e)[:200]}


def _concurrent_rate(workload, lanes, n_scans):
    """scans/s of tools/bench_concurrent.py (its own process, without this one's OpenMP pinning), or None"""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("OMP_PROC_BIND", "OMP_PLACES")}
    try:
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "bench_concurrent.py"), "--workload", workload, "--lanes", str(lanes),
                              "--scans", str(n_scans)], env=env, capture_output=True, text=True, timeout=240, cwd=ROOT)
        line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
        return float(json.loads(line)["lanes_%d_median" % lanes])
    except Exception:  # noqa: BLE001  (informational figure: a failure here must not cost the bench line)
        return None


def _pmc_traffic(workload, kernel_sub, calls=200):
    """HBM bytes per launch of the pass kernel from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE: separate passes, as
    MI355X_MICROARCH.md prescribes) over tools/prof_target.py - the same registrations as the timed region, nothing else in
    the process.  bytes = (2 x FETCH_SIZE + WRITE_SIZE) KiB: the guide's gfx950 correction (FETCH_SIZE tallies 128-byte requests
    at 64 B).  Returns (bytes or None, provenance)."""
    import shutil
    import sqlite3
    import subprocess
    import tempfile
    rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rocprof):
        return None, "rocprofv3 not found on this box"
    means = {}
    for ctr in ("FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU"):  # (the third: the VALU issue bound of the timed region is priced with THIS run's count)
        d = tempfile.mkdtemp(prefix="kicp_pmc_", dir="/tmp")
        cmd = [rocprof, "--pmc", ctr, "-d", d, "-o", "pmc", "--", sys.executable, os.path.join(ROOT, "tools", "prof_target.py"), "--workload", workload,
               "--calls", str(calls)] + (["--batch", "64"] if kernel_sub == "k_pass_gather32" else [])
        try:
            r = subprocess.run(cmd, cwd="/tmp", env={**os.environ, "TMPDIR": "/tmp"}, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, timeout=300)
            vals = []
            for root, _, files in os.walk(d):
                for f in files:
                    if not f.endswith(".db"):
                        continue
                    c = sqlite3.connect(os.path.join(root, f))
                    cols = [row[1] for row in c.execute("pragma table_info('counters_collection')")]
                    name_col = "kernel_name" if "kernel_name" in cols else "name"
                    vals += [float(v) for (v,) in c.execute("select value from counters_collection where counter_name = ? and %s like ?" % name_col,
                                                             (ctr, "%" + kernel_sub + "%"))]
                    c.close()
            if not vals and ctr == "SQ_INSTS_VALU":
                continue  # (informational: the traffic figure does not depend on it)
            if not vals:
                return None, "rocprofv3 --pmc %s gave no rows for %s (rc %d: %s)" % (ctr, kernel_sub, r.returncode, r.stderr.decode(errors="replace")[-200:])
            means[ctr] = (float(np.mean(vals)), len(vals))
        except (OSError, subprocess.SubprocessError, sqlite3.Error) as e:
            if ctr == "SQ_INSTS_VALU":
                continue
            return None, "rocprofv3 --pmc %s failed: %s" % (ctr, str(e)[:200])
        finally:
            shutil.rmtree(d, ignore_errors=True)
    kib = 2.0 * means["FETCH_SIZE"][0] + means["WRITE_SIZE"][0]
    _pmc_traffic.insts_valu = means.get("SQ_INSTS_VALU")  # (mean per dispatch, dispatches) | None
    return kib * 1024.0, ("this run: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) over `tools/prof_target.py --workload %s --calls %d%s` "
                          "after the timed region; means over %d / %d dispatches of %s: FETCH_SIZE %.1f KiB, WRITE_SIZE %.1f KiB; bytes = (2 x FETCH_SIZE + "
                          "WRITE_SIZE) x 1024" % (workload, calls, " --batch 64" if kernel_sub == "k_pass_gather32" else "", means["FETCH_SIZE"][1], means["WRITE_SIZE"][1], kernel_sub, means["FETCH_SIZE"][0],
                                                   means["WRITE_SIZE"][0]))


def _profile_counters(workload, world):
    """Counters of the pass kernel from the committed rocprofv3 PMC passes of THIS workload (profiles/r02_counters_<workload>.json,
    written by tools/prof_counters_json.py from separate --pmc passes; HBM bytes = (2 x FETCH_SIZE + WRITE_SIZE) KiB per
    dispatch, the x2 being the gfx950 FETCH_SIZE correction of MI355X_MICROARCH.md).  None when this workload / GPU count has
    not been profiled: nothing is borrowed from another configuration."""
    if world != 1:
        return None
    for rnd, sub in (("r06", ""), ("r05", "history"), ("r04b", "history"), ("r04", "history"), ("r03", "history"), ("r02", "history")):
        try:
            with open(os.path.join(ROOT, "profiles", sub, "%s_counters_%s.json" % (rnd, workload))) as f:
                d = json.load(f)
            d["source"] = ("profiles/%s_counters_%s.json (offline rocprofv3 --pmc passes of this workload, NOT this run; taken at commit %s)"
                           % (rnd, workload, d.get("git_sha", "of round " + rnd[1:])))
            return d
        except (OSError, ValueError):
            continue
    return None


def _cgroup_cpu_max():
    """the container's CPU quota ("max" = unlimited), which caps what more threads can buy"""
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                return f.read().strip()
        except OSError:
            continue
    return None


def _cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


if __name__ == "__main__":
    main()
