"""Throughput of INDEPENDENT scans with several in flight (kicp_register_device_concurrent) next to the sequential queue
(kicp_register_device_batch): what the machine does when a workload has more than one scan to offer at a time - several robots
in one map, replayed logs.  The reference's pipeline cannot use this (a scan's initial guess is the previous result); the
headline of bench.py is the sequential rate.

    python tools/bench_concurrent.py --workload cfg2 --lanes 1 2 4 8 [--multi] [--count 512] [--repeats 5]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kinematic_icp_amd as K  # noqa: E402
from kinematic_icp_amd import synthetic as syn  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="cfg2")
ap.add_argument("--lanes", type=int, nargs="+", default=[1, 2, 4, 8])
ap.add_argument("--multi", action="store_true")
ap.add_argument("--count", type=int, default=512)
ap.add_argument("--repeats", type=int, default=5)
ap.add_argument("--with-torch", action="store_true", help="initialise torch's HIP context in the process first (what bench.py has)")
ap.add_argument("--scans", type=int, default=4, help="distinct scans cycled through")
args = ap.parse_args()

if args.with_torch:
    import torch
    torch.cuda.set_device(0)
    torch.zeros(8, device="cuda").sum().item()
cfg, scene, scans, rng = syn.make_case(args.workload, n_scans=args.scans)
gmap = K.VoxelHashMap(cfg.voxel_size, cfg.max_range, cfg.max_points_per_voxel)
ident = np.array([0, 0, 0, 1.0, 0, 0, 0])
syn.build_map_points(scene, cfg, lambda pts: gmap.UpdateDevice(K.DeviceFrame(pts), ident), gmap.num_points, rng)
gmap.sync(0)
tau = cfg.first_frame_tau()
frames = [K.DeviceFrame(s["frame"]) for s in scans]
extra = syn.planar_pose(0.2, 0.0, np.deg2rad(1.5)) if args.multi else syn.planar_pose(0.0, 0.0, 0.0)
rels = [syn.pose_mul(s["rel_odom"], extra) for s in scans]
regs = [K.KinematicRegistration() for _ in range(max(args.lanes))]
batch = regs[0].prepare_batch([frames[i % len(scans)] for i in range(args.count)], [scans[i % len(scans)]["last_pose"] for i in range(args.count)],
                              [rels[i % len(scans)] for i in range(args.count)])
want = regs[0].ComputeRobotMotionBatch(batch, gmap, tau).copy()
out = {"workload": args.workload, "multi": args.multi, "count": args.count, "iterations_mean": float(batch.iterations.mean())}
best = []
for _ in range(args.repeats):
    t0 = time.perf_counter()
    regs[0].ComputeRobotMotionBatch(batch, gmap, tau)
    best.append(time.perf_counter() - t0)
out["sequential_scans_per_s"] = round(args.count / min(best), 1)
for lanes in args.lanes:
    regs[0].ComputeRobotMotionConcurrent(regs[1:lanes], batch, gmap, tau)  # warm-up: every lane's buffers and queue
    best = []
    for _ in range(args.repeats):
        t0 = time.perf_counter()
        got = regs[0].ComputeRobotMotionConcurrent(regs[1:lanes], batch, gmap, tau)
        best.append(time.perf_counter() - t0)
    assert np.array_equal(got, want)
    out["lanes_%d_scans_per_s" % lanes] = round(args.count / min(best), 1)
    out["lanes_%d_median" % lanes] = round(args.count / float(np.median(best)), 1)
print(json.dumps(out))
