#!/usr/bin/env python3
"""bench.py -- scans/sec of the ICP registration hot path (KinematicRegistration::ComputeRobotMotion) on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One "step" = one BATCH of --scans-per-step (default 64) ComputeRobotMotion calls (all ICP iterations of each scan; the batch
is ONE call of the C-ABI's kicp_register_device_batch, which registers the scans strictly one after the other - what a C++
caller sees; the rate with one Python call per scan is reported next to it in `config`) on
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
pose.  --comm selects the exchange: "rccl" (default, the north star: ncclAllReduce over xGMI on the registration's
stream), "shm" (every rank's host adds its GPU's rows and the ranks meet in a node-wide host shared segment: no device
collective), "p2p" (one-shot exchange: every pass kernel writes its totals into all ranks' HBM mailboxes over xGMI peer
mappings and adds the node's totals itself - no collective library), "torch" (torch.distributed all-reduce callback).  With
--comm rccl the shm and p2p figures are measured too and reported in `config`.  Total work is fixed -> "scaling": "strong".  (--mode replicas, not the default: every rank
registers whole scans on its own - one robot per GPU - no exchange, "scaling": "weak".)

Prints ONE JSON line on rank 0 with the contract's keys plus
  "roofline"     the dominant kernel (fused association+accumulation pass).  `achieved` follows the contract (SURVEY.md
                 section 8d algorithmic bytes per launch / live HIP-event duration on the kernel's own stream, vs the
                 8 TB/s HBM peak) and may exceed the peak: the algorithmic count is what the REFERENCE touches (27 probes
                 + every scanned bucket point per query) while this kernel skips provably irrelevant voxels and is served
                 by L1/L2 - the kernel is latency bound, not HBM bound.  Next to it: `b_min` (compulsory bytes, a true
                 lower bound), `time_split_us` (fixed launch+reduction floor measured live vs query work), and - when a
                 rocprofv3 profile of THIS workload is committed under profiles/ - measured HBM traffic, L2 / L1 request
                 traffic against their peaks, VALU busy and occupancy.
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
L2_PEAK_GBS = 34500.0   # same guide, "L2 (per XCD)": ~34.5 TB/s aggregate
MULTI_ITER_ERROR = (0.05, 0.5)  # extra odometry error of the multi-iteration workload: metres along x, degrees of yaw


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--scans-per-step", type=int, default=0,
                    help="registrations per step (one batch of synthetic scans); 0 (default) = chosen after a calibration batch so that the "
                         "timed region of the K steps lasts at least --min-timed-s, never fewer than 64")
    ap.add_argument("--min-timed-s", type=float, default=0.3, help="shortest acceptable timed region (auto batch size)")
    ap.add_argument("--no-pmc", action="store_true", help="skip the in-run rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE) behind roofline.traffic")
    ap.add_argument("--workload", default="cfg2")
    ap.add_argument("--scans", type=int, default=8, help="distinct synthetic scans cycled through")
    ap.add_argument("--cpu-seconds", type=float, default=16.0, help="budget of the cpu_baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--comm", default="rccl", choices=["rccl", "shm", "p2p", "torch"],
                    help="N>1 exchange of the per-iteration sums: built-in RCCL all-reduce (default), host shared segment (no device "
                         "collective), one-shot peer-mailbox exchange over xGMI mappings (no collective library), or torch.distributed "
                         "all-reduce callback")
    ap.add_argument("--pg-backend", default="nccl", choices=["nccl", "gloo"], help="torch.distributed backend for barriers/timing")
    ap.add_argument("--mode", default="shard", choices=["shard", "replicas"],
                    help="N>1: 'shard' (default, the north star) splits every scan's points across the ranks and exchanges the sums each "
                         "iteration; 'replicas' lets every rank register whole scans on its own (one robot per GPU, no exchange; weak scaling)")
    ap.add_argument("--force-comm", action="store_true", help="exercise the multi-GPU code path (all-reduce + separate solve) even with one rank")
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
    cfg, scene, scans, rng = syn.make_case(args.workload, n_scans=args.scans)
    gmap = K.VoxelHashMap(cfg.voxel_size, cfg.max_range, cfg.max_points_per_voxel)
    # the map grows through VoxelHashMap::Update(points, identity) on the GPU - the path the pipeline's map update takes
    # (same map as the host-side AddPoints builds, tests/test_gpu_mapdev.py; seconds instead of a minute for cfg5)
    ident = np.array([0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0])
    syn.build_map_points(scene, cfg, lambda pts: gmap.UpdateDevice(K.DeviceFrame(pts, device=device), ident), gmap.num_points, rng)
    tau = cfg.first_frame_tau()
    gmap.sync(device)
    n_total = scans[0]["frame"].shape[0]
    lo, hi = (n_total * rank) // world, (n_total * (rank + 1)) // world  # contiguous shard of this rank
    if replicas:
        lo, hi = 0, n_total
    frames = [K.DeviceFrame(s["frame"][lo:hi], device=device) for s in scans]
    extra = syn.planar_pose(MULTI_ITER_ERROR[0], 0.0, np.deg2rad(MULTI_ITER_ERROR[1]))
    rel_multi = [syn.pose_mul(s["rel_odom"], extra) for s in scans]

    def make_reg(comm, **kw):
        """a registration handle with the requested exchange attached (None: single GPU / replicas)"""
        reg = K.KinematicRegistration(device=device, **kw)  # reference defaults (KinematicICP.hpp:51-56) unless asked otherwise
        keep = []
        if comm == "shm":
            name = "kicp_bench_%s_%s" % (os.environ.get("MASTER_PORT", "0"), os.environ.get("TORCHELASTIC_RUN_ID", "x"))
            if rank == 0:
                reg.shm_init(world, 0, name)  # removes a stale segment of that name, creates, zeroes and publishes the new one
            dist.barrier()
            if rank != 0:
                reg.shm_init(world, rank, name)
            dist.barrier()
        elif comm == "rccl":
            uid = torch.zeros(K.COMM_ID_BYTES, dtype=torch.uint8, device=pg_dev)
            if rank == 0:
                uid.copy_(torch.frombuffer(bytearray(K.comm_unique_id()), dtype=torch.uint8))
            dist.broadcast(uid, 0)
            reg.comm_init(world, rank, bytes(uid.cpu().numpy().tobytes()))
        elif comm == "p2p":
            mine = torch.frombuffer(bytearray(reg.p2p_export(world, rank)), dtype=torch.uint8).to(pg_dev)
            every = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(every, mine)
            reg.p2p_connect([bytes(t.cpu().numpy().tobytes()) for t in every])
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

    def run_scan(reg, i, rels, stats_out=None):
        s = scans[i % len(scans)]
        pose = reg.ComputeRobotMotion(frames[i % len(scans)], gmap, s["last_pose"], rels[i % len(scans)], tau)
        if stats_out is not None:
            k = reg.last_stats.iterations
            stats_out.append((k, list(reg.last_stats.pass_ms[:k])))
        return pose

    B = max(1, args.scans_per_step) if args.scans_per_step > 0 else 64  # (auto: fixed below, after the calibration batch)

    def all_ranks_ok(err):
        """every rank learns whether ANY rank failed and all of them raise together (this rank's own error, or a stand-in)"""
        if use_comm:
            t = torch.tensor([0.0 if err is None else 1.0], dtype=torch.float64, device=pg_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            if float(t.item()) > 0.0 and err is None:
                err = K.KicpError(K.KICP_ERR_COMM, "a peer rank's registration failed")
        if err is not None:
            raise err

    def timed(reg, rels, steps, warmup, per_call=False):
        """W untimed warm-up steps, then EXACTLY `steps` steps of B scans between barriers; max over ranks.
        A step is ONE kicp_register_device_batch call: the library registers the step's B scans one after the other (a plain
        loop of ComputeRobotMotion, each scan run to completion before the next starts) - what a C++ caller of the C-ABI
        sees.  per_call=True issues the B calls from Python instead (adds the interpreter's ~3 us per call)."""
        batch = reg.prepare_batch([frames[i % len(scans)] for i in range(B)], [scans[i % len(scans)]["last_pose"] for i in range(B)],
                                  [rels[i % len(scans)] for i in range(B)])

        def step(k):
            if per_call:
                for i in range(k * B, (k + 1) * B):
                    run_scan(reg, i, rels)
            else:
                reg.ComputeRobotMotionBatch(batch, gmap, tau)
        # A registration that fails on one rank (an exchange that cannot complete on this box) must not leave the ranks at
        # different collectives: the failing rank still walks through the barriers, and all ranks leave together (all_ranks_ok).
        err = None
        try:
            for k in range(warmup):
                step(k)
        except K.KicpError as e:
            err = e
        barrier()
        t0 = time.perf_counter()
        if err is None:
            try:
                for k in range(steps):
                    step(k)
            except K.KicpError as e:
                err = e
        barrier()
        elapsed = time.perf_counter() - t0
        all_ranks_ok(err)
        if world > 1:
            t = torch.tensor([elapsed], dtype=torch.float64, device=pg_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        timed.last_iterations = float(np.mean(batch.iterations)) if not per_call else float("nan")  # ICP iterations per scan of this run
        return elapsed

    rel_single = [s["rel_odom"] for s in scans]
    comm = args.comm if exchange else None
    comm_note = None
    try:
        reg, keep = make_reg(comm)
    except K.KicpError as e:  # e.g. RCCL cannot be initialised on this box: the scaling run still gets a number
        if comm != "rccl":
            raise
        comm_note = "rccl set-up failed (%s): fell back to the host shared segment" % e
        comm = args.comm = "shm"
        reg, keep = make_reg(comm)
    # ---- one-time settling (setup, not measurement): the HIP runtime finishes its lazy initialisation (signal pools,
    #      code objects, clocks) during the first few hundred launches of a process; a ~30 ms hiccup there would
    #      otherwise land inside a short timed region.
    #      With an exchange attached every registration is a collective, so every rank must run the SAME number of scans: the
    #      ranks agree on "another block?" (round 3 let each rank consult its own clock; when the 0.5 s mark fell between two
    #      ranks' checks one of them ran 50 scans more, which the peers never answered: the bounded wait of the exchange
    #      expired 20 s later - the one-in-fifteen failure of the two-rank test).
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
        t_cal = timed(reg, rel_single, 4, 1) / (4 * B)
        B = int(min(8192, max(64, -(-args.min_timed_s // (max(1, args.steps) * t_cal)))))
    elapsed = timed(reg, rel_single, args.steps, args.warmup)                 # ---- the headline number
    launch_path = ("direct AQL dispatch on the handle's own HSA queue (kicp_aql.hpp), kernel arguments in %s"
                   % {0.0: "host memory", 1.0: "device memory", 2.0: "device memory + HDP flush"}.get(reg.get_option("aql_kernarg"), "?")
                   if reg.get_option("aql_active") == 1.0 else "hipLaunchKernelGGL on the handle's stream")
    small_kind = int(reg.get_option("small_active"))  # 0 generic pass kernel; small-scan path (kicp_small.hpp): 1 sub-lanes per query, 2 one wave per query
    small_active = small_kind > 0
    elapsed_multi = timed(reg, rel_multi, args.steps, min(args.warmup, 2))    # ---- same scans, several ICP iterations each
    elapsed_py = timed(reg, rel_single, args.steps, 1, per_call=True)         # ---- informational: one Python call per scan

    # ---- second pass over the same steps with HIP events around every pass-kernel launch (roofline) -------------
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
    # fixed floor of a pass, measured live: the same launch with every query switched off (launch + reduction + hand-off)
    floor_us = None
    if not use_comm and not small_active:  # (the dbg switches belong to the generic pass kernel)
        reg.set_option("dbg", 7)
        tmp = []
        for i in range(64 + 256):
            run_scan(reg, i, rel_single, tmp)
        reg.set_option("dbg", 0)
        fl = np.array([lst[0] for _, lst in tmp[64:] if lst], dtype=np.float64)  # (pass 0: with every query off the call ends there)
        floor_us = float(fl.mean() * 1e3) if fl.size else None
    reg.set_option("timing", 0)
    poses = [run_scan(reg, i, rel_single) for i in range(len(scans))]
    poses_multi = [run_scan(reg, i, rel_multi) for i in range(len(scans))]
    barrier()
    # informational, never `value`: the same calls with the scan handed over as a HOST array (upload inside)
    host_rate = None
    if world == 1:
        host_frames = [np.ascontiguousarray(s["frame"][lo:hi]) for s in scans]
        for i in range(8):
            reg.ComputeRobotMotion(host_frames[i % len(scans)], gmap, scans[i % len(scans)]["last_pose"], scans[i % len(scans)]["rel_odom"], tau)
        t1 = time.perf_counter()
        k_host = min(args.steps * B, 400)
        for i in range(k_host):
            reg.ComputeRobotMotion(host_frames[i % len(scans)], gmap, scans[i % len(scans)]["last_pose"], scans[i % len(scans)]["rel_odom"], tau)
        host_rate = k_host / (time.perf_counter() - t1)
    pass_kernel = int(reg.get_option("pass_kernel"))
    if exchange:
        release(reg, comm)
    del reg
    # ---- N > 1: every exchange on the same box in the same run (the headline stays --comm's), each with its scans/s, its time
    #      per ICP iteration on the multi-iteration workload and what the exchange adds per iteration over the same shard
    #      registered WITHOUT any exchange (a plain handle on this rank's points: the kernel and hand-off alone)
    other = {}
    if exchange:
        FIXED = dict(max_num_iteration=4, convergence_criterion=0.0)  # every scan runs exactly four iterations, exchange or not

        def measure(reg_x, reg_fixed):
            for i in range(60):
                run_scan(reg_x, i, rel_single)
            e1 = timed(reg_x, rel_single, args.steps, min(args.warmup, 2))
            em = timed(reg_x, rel_multi, args.steps, 1)
            its = timed.last_iterations
            ef = timed(reg_fixed, rel_single, args.steps, 1)
            return {"scans_per_s": round(args.steps * B / e1, 1), "us_per_iteration": round(1e6 * em / (args.steps * B) / its, 3),
                    "iterations_per_scan": round(its, 3), "us_per_iteration_at_4_fixed_iterations": round(1e6 * ef / (args.steps * B) / 4.0, 3)}
        exch = {}
        plain, plain_fixed = K.KinematicRegistration(device=device), K.KinematicRegistration(device=device, **FIXED)
        exch["none (this rank's shard alone, no exchange: NOT a registration of the scan)"] = base = measure(plain, plain_fixed)
        del plain, plain_fixed
        for alt in ("rccl", "shm", "p2p"):
            if alt == "rccl" and (torch.cuda.device_count() < world or "KICP_BENCH_DEVICE" in os.environ):
                exch[alt] = {"note": "skipped: RCCL needs one GPU per rank"}
                continue
            try:
                reg2, keep2 = make_reg(alt)
            except K.KicpError as e:  # e.g. IPC mappings / RCCL unavailable: say so, keep the line
                exch[alt] = {"note": str(e)[:200]}
                continue
            try:
                err = None
                try:
                    for i in range(60):
                        run_scan(reg2, i, rel_single)
                except K.KicpError as e:
                    err = e
                all_ranks_ok(err)
                e1 = timed(reg2, rel_single, args.steps, min(args.warmup, 2))
                em = timed(reg2, rel_multi, args.steps, 1)
                its = timed.last_iterations
                release(reg2, alt)
                del reg2
                reg3, keep3 = make_reg(alt, **FIXED)
                ef = timed(reg3, rel_single, args.steps, 1)
                fixed_us = 1e6 * ef / (args.steps * B) / 4.0
                exch[alt] = {"scans_per_s": round(args.steps * B / e1, 1), "us_per_iteration": round(1e6 * em / (args.steps * B) / its, 3),
                             "iterations_per_scan": round(its, 3), "us_per_iteration_at_4_fixed_iterations": round(fixed_us, 3),
                             # what the exchange adds to an iteration of this rank's shard (both sides run exactly four iterations per scan)
                             "exchange_us_per_iteration": round(fixed_us - base["us_per_iteration_at_4_fixed_iterations"], 3)}
                release(reg3, alt)
                del reg3
            except K.KicpError as e:
                exch[alt] = {"note": str(e)[:200]}
        other["exchanges"] = exch
        for alt in ("shm", "p2p"):  # (kept under their round-2 names too)
            if "scans_per_s" in exch.get(alt, {}):
                other[alt + "_scans_per_s"] = exch[alt]["scans_per_s"]
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
    for rels, ps, it_out, count in ((rel_single, poses, iters_ref, True), (rel_multi, poses_multi, iters_ref_multi, False)):
        for s, rel, pose in zip(scans, rels, ps):
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

    cpu = None if args.no_cpu_baseline else _cpu_baseline(args, cfg, scans, rel_single, tau, omap, map_points, okicp, rkicp)

    n_scans_timed = args.steps * B
    value = (world if replicas else 1) * n_scans_timed / elapsed  # replicas: every rank completed its own scans
    # ---- HBM traffic of the pass kernel, measured in THIS run: one rocprofv3 --pmc pass per counter over a bare loop of the
    #      same registrations (tools/prof_target.py), after everything timed is over.  Falls back to the committed profile
    #      (stamped with the commit it was taken at) where rocprofv3 cannot run.
    kernel_sub = {0: "k_pass_gather32", 1: "k_pass_small", 2: "k_pass_wave"}[small_kind]
    traffic, traffic_src = (None, "not measured (--no-pmc)") if (args.no_pmc or world != 1) else _pmc_traffic(args.workload, kernel_sub)
    # informational, never `value`: INDEPENDENT scans with four in flight (kicp_register_device_concurrent: one handle, HSA queue
    # and host thread per lane) - what the device does when a workload has several scans to offer at a time (robots sharing a
    # map, replayed logs).  The reference's sequential pipeline cannot use it, hence not the headline.
    conc_lanes = 4
    conc_rate = _concurrent_rate(args.workload, conc_lanes, len(scans)) if (world == 1 and not use_comm) else None
    prof = _profile_counters(args.workload, world)
    if traffic is None and prof and prof.get("hbm_bytes_per_launch"):
        traffic = float(prof["hbm_bytes_per_launch"])
        traffic_src = "%s; in-run measurement unavailable: %s" % (prof.get("source"), traffic_src)
    t_kernel = kernel_us * 1e-6
    traffic_gbs = None if (traffic is None or not pass_ms.size) else traffic / t_kernel / 1e9
    roof = {"bound": "hbm", "achieved": None if traffic_gbs is None else round(traffic_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": None if traffic_gbs is None else round(traffic_gbs / HBM_PEAK_GBS, 4),
            "traffic": None if traffic is None else round(traffic), "traffic_source": traffic_src,
            "kernel": "fused association+accumulation pass (%s)" % kernel_sub, "kernel_avg_us": round(kernel_us, 2),
            "kernel_time_source": kernel_time_source, "launches_timed": int(pass_ms.size),
            "what": "achieved = HBM bytes the pass kernel moved per launch (2 x FETCH_SIZE + WRITE_SIZE, MI355X_MICROARCH.md's gfx950 correction) / its "
                    "average duration; frac = achieved / 8 TB/s.  The kernel is bound by its waves' dependent-load chains and a fixed launch + "
                    "reduction floor, not by HBM: see time_split_us and counters",
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
    out = {
        "metric": "scans/sec (ICP registration only), 128k-pt scan vs 1M-pt map",
        "value": round(value, 2),
        "unit": "scans/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(1e3 * elapsed / args.steps, 5),
        "higher_is_better": True,
        "scaling": "weak" if replicas else "strong",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": "%s: %d-pt %d-beam scan vs %d-pt / %d-voxel map, voxel %.2f m, tau %.4f m, default ICP iterations "
                               "(mean %.2f per scan, reference %.2f); one step = one kicp_register_device_batch call = %d scans registered one after the other"
                               % (cfg.name, n_total, cfg.n_beams, gmap.num_points(), gmap.num_voxels(), cfg.voxel_size, tau, iters_gpu,
                                  float(np.mean(iters_ref)), B),
                   "scans_per_step": B, "ms_per_scan": round(1e3 * elapsed / n_scans_timed, 5), "timed_region_s": round(elapsed, 4),
                   "points_per_gpu": hi - lo,
                   "scans_per_s_one_python_call_per_scan": round((world if replicas else 1) * n_scans_timed / elapsed_py, 2),
                   "parallelism": ("%d independent replicas (one robot per GPU), no exchange" % world) if replicas else
                                  (("points sharded x%d, map replicated, %s all-reduce" % (world, args.comm)) if use_comm else "single GPU"),
                   "pass_kernel": pass_kernel, "launch_path": launch_path, "max_pose_abs_diff_vs_oracle": max_pose_err,
                   "multi_iteration": {
                       "workload": "same scans, odometry error +%.2f m / +%.1f deg: %.2f ICP iterations per scan (reference %.2f)"
                                   % (MULTI_ITER_ERROR[0], MULTI_ITER_ERROR[1], iters_gpu_multi, float(np.mean(iters_ref_multi))),
                       "scans_per_s": round((world if replicas else 1) * n_scans_timed / elapsed_multi, 2),
                       "ms_per_scan": round(1e3 * elapsed_multi / n_scans_timed, 5),
                       "ms_per_iteration": None if not np.isfinite(iters_gpu_multi) else round(1e3 * elapsed_multi / n_scans_timed / iters_gpu_multi, 5),
                       "pass_kernel_avg_us": round(float(pass_ms_multi.mean() * 1e3), 2) if pass_ms_multi.size else None},
                   "scans_per_s_with_host_input_incl_pcie": None if host_rate is None else round(host_rate, 1), **other,
                   **({"comm_note": comm_note} if comm_note else {})},
        "value_multi_iteration": {"scans_per_s": round((world if replicas else 1) * n_scans_timed / elapsed_multi, 2),
                                  "iterations_per_scan": None if not np.isfinite(iters_gpu_multi) else round(iters_gpu_multi, 3),
                                  "us_per_iteration": None if not np.isfinite(iters_gpu_multi) else round(1e6 * elapsed_multi / n_scans_timed / iters_gpu_multi, 3),
                                  "what": "the same scans with +%.2f m / +%.1f deg odometry error, same steps and batch" % MULTI_ITER_ERROR},
        "value_host_vector_input": None if host_rate is None else
        {"scans_per_s": round(host_rate, 1), "what": "kicp_register with the scan handed over as a HOST array, the reference's own signature "
                                                      "(Registration.hpp:39-43): upload over PCIe inside the call"},
        "value_concurrent_independent_scans": None if conc_rate is None else
        {"scans_per_s": round(conc_rate, 1), "lanes": conc_lanes,
         "what": "kicp_register_device_concurrent: the same scans as INDEPENDENT registrations, %d in flight (one handle + HSA queue + host thread each), "
                 "512 per call, median of 5 calls, measured by tools/bench_concurrent.py in a process of its own after everything timed here; "
                 "a throughput mode the reference's sequential pipeline cannot use - never the headline" % conc_lanes},
        "roofline": roof,
        "cpu_baseline": cpu,
    }
    if elapsed < 0.25:
        out["config"]["warning"] = "timed region shorter than 0.25 s: raise --steps or --scans-per-step (0 = automatic)"
    import ctypes
    ctypes.CDLL(None).fflush(None)
    sys.stdout.flush()
    os.dup2(saved_stdout, 1)
    print(json.dumps(out), flush=True)


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
    else:
        best = max(port, key=port.get)
        res.update({"value": round(port[best], 3), "cores": best, "kind": "port",
                    "sample": "%d calls of the oracle port on the same %d scans (oracle/_ref not present); threads tried %s, best reported"
                              % (port_n, len(scans), counts), "single_thread_value": round(port[1], 3)})
    return res


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
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="kicp_pmc_", dir="/tmp")
        cmd = [rocprof, "--pmc", ctr, "-d", d, "-o", "pmc", "--", sys.executable, os.path.join(ROOT, "tools", "prof_target.py"), "--workload", workload,
               "--calls", str(calls)]
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
            if not vals:
                return None, "rocprofv3 --pmc %s gave no rows for %s (rc %d: %s)" % (ctr, kernel_sub, r.returncode, r.stderr.decode(errors="replace")[-200:])
            means[ctr] = (float(np.mean(vals)), len(vals))
        except (OSError, subprocess.SubprocessError, sqlite3.Error) as e:
            return None, "rocprofv3 --pmc %s failed: %s" % (ctr, str(e)[:200])
        finally:
            shutil.rmtree(d, ignore_errors=True)
    kib = 2.0 * means["FETCH_SIZE"][0] + means["WRITE_SIZE"][0]
    return kib * 1024.0, ("this run: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) over `tools/prof_target.py --workload %s --calls %d` "
                          "after the timed region; means over %d / %d dispatches of %s: FETCH_SIZE %.1f KiB, WRITE_SIZE %.1f KiB; bytes = (2 x FETCH_SIZE + "
                          "WRITE_SIZE) x 1024" % (workload, calls, means["FETCH_SIZE"][1], means["WRITE_SIZE"][1], kernel_sub, means["FETCH_SIZE"][0],
                                                   means["WRITE_SIZE"][0]))


def _profile_counters(workload, world):
    """Counters of the pass kernel from the committed rocprofv3 PMC passes of THIS workload (profiles/r02_counters_<workload>.json,
    written by tools/prof_counters_json.py from separate --pmc passes; HBM bytes = (2 x FETCH_SIZE + WRITE_SIZE) KiB per
    dispatch, the x2 being the gfx950 FETCH_SIZE correction of MI355X_MICROARCH.md).  None when this workload / GPU count has
    not been profiled: nothing is borrowed from another configuration."""
    if world != 1:
        return None
    for rnd in ("r03", "r02"):
        try:
            with open(os.path.join(ROOT, "profiles", "%s_counters_%s.json" % (rnd, workload))) as f:
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
