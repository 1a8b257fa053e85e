"""A/B of one registration option inside one process: two handles on the same map and frames, alternating blocks of calls,
the per-block rates compared by their medians (box-to-box and minute-to-minute noise on the GPU pool is ~10 %, far above the
1 us effects this is for).

    python tools/ab_option.py --workload cfg2 --option latency_kernel --values 0 1 [--multi] [--blocks 40] [--calls 250]
    python tools/ab_option.py --workload cfg2 --sets base latency_kernel=1 latency_kernel=1,nearest_first=2
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
ap.add_argument("--option")
ap.add_argument("--values", type=float, nargs="+", default=[])
ap.add_argument("--sets", nargs="+", default=[], help="comma-separated name=value lists, one handle each ('base' = defaults)")
ap.add_argument("--fixed", action="append", default=[], help="name=value options set on every handle")
ap.add_argument("--multi", action="store_true")
ap.add_argument("--blocks", type=int, default=40)
ap.add_argument("--calls", type=int, default=250)
ap.add_argument("--batch", action="store_true", help="a block is ONE kicp_register_device_batch call of --calls scans (what bench.py times)")
args = ap.parse_args()

cfg, scene, scans, rng = syn.make_case(args.workload, n_scans=4)
gmap = K.VoxelHashMap(cfg.voxel_size, cfg.max_range, cfg.max_points_per_voxel)
ident = np.array([0, 0, 0, 1.0, 0, 0, 0])
syn.build_map_points(scene, cfg, lambda pts: gmap.UpdateDevice(K.DeviceFrame(pts), ident), gmap.num_points, rng)
gmap.sync(0)
tau = cfg.first_frame_tau()
frames = [K.DeviceFrame(s["frame"]) for s in scans]
extra = syn.planar_pose(0.2, 0.0, np.deg2rad(1.5)) if args.multi else syn.planar_pose(0.0, 0.0, 0.0)
rels = [syn.pose_mul(s["rel_odom"], extra) for s in scans]
regs, labels = [], []
for spec in ["%s=%g" % (args.option, v) for v in args.values] + args.sets:
    reg = K.KinematicRegistration()
    for o in args.fixed + ([] if spec == "base" else spec.split(",")):
        k, x = o.split("=")
        reg.set_option(k, float(x))
    regs.append(reg)
    labels.append(spec if spec not in labels else "%s#%d" % (spec, len(labels)))
poses = []
for reg in regs:
    for i in range(50):
        pose = reg.ComputeRobotMotion(frames[i % 4], gmap, scans[i % 4]["last_pose"], rels[i % 4], tau)
    poses.append(pose)
K.lib().kicp_device_synchronize(0)
us = [[] for _ in regs]
batches = [reg.prepare_batch([frames[i % 4] for i in range(args.calls)], [scans[i % 4]["last_pose"] for i in range(args.calls)],
                             [rels[i % 4] for i in range(args.calls)]) for reg in regs] if args.batch else None
for b in range(args.blocks):
    for j, reg in enumerate(regs):
        t0 = time.perf_counter()
        if args.batch:
            reg.ComputeRobotMotionBatch(batches[j], gmap, tau)
        else:
            for i in range(args.calls):
                reg.ComputeRobotMotion(frames[i % 4], gmap, scans[i % 4]["last_pose"], rels[i % 4], tau)
        us[j].append((time.perf_counter() - t0) / args.calls * 1e6)
if args.batch:
    poses = [b.out.copy() for b in batches]
out = {"workload": args.workload, "multi": args.multi, "iterations": regs[0].last_stats.iterations,
       "same_pose": bool(all(np.array_equal(p, poses[0]) for p in poses))}
for v, u in zip(labels, us):
    out[v] = {"median_us": round(float(np.median(u)), 2), "p10_us": round(float(np.percentile(u, 10)), 2), "min_us": round(float(np.min(u)), 2)}
print(json.dumps(out))
