"""One batch setting per PROCESS (tools/ab_option.py keeps every handle's HSA queues alive side by side: with a dozen clones per handle
the hardware queue slots are oversubscribed and the comparison measures the queue scheduler): per-scan wall time of
kicp_register_device_batch on a BASELINE workload under the given options, with the counters that say which path served it.

    python tools/probe_batch.py --workload cfg1 --calls 1024 --blocks 12 [--multi] batch_threads=6
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
ap.add_argument("--workload", default="cfg1")
ap.add_argument("--multi", action="store_true")
ap.add_argument("--blocks", type=int, default=12)
ap.add_argument("--calls", type=int, default=1024)
ap.add_argument("options", nargs="*")
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
reg = K.KinematicRegistration()
for o in args.options:
    k, x = o.split("=")
    reg.set_option(k, float(x))
single = [reg.ComputeRobotMotion(frames[i], gmap, scans[i]["last_pose"], rels[i], tau).copy() for i in range(4)]
batch = reg.prepare_batch([frames[i % 4] for i in range(args.calls)], [scans[i % 4]["last_pose"] for i in range(args.calls)], [rels[i % 4] for i in range(args.calls)])
us = []
relaunches0 = reg.get_option("small_relaunches")
for b in range(args.blocks + 2):
    t0 = time.perf_counter()
    reg.ComputeRobotMotionBatch(batch, gmap, tau)
    if b >= 2:
        us.append((time.perf_counter() - t0) / args.calls * 1e6)
same = all(np.array_equal(batch.out[i], single[i % 4]) for i in range(args.calls))
print(json.dumps({"workload": args.workload, "multi": args.multi, "options": args.options, "scans_per_call": args.calls, "poses_equal_single_calls": bool(same),
                  "median_us": round(float(np.median(us)), 2), "p10_us": round(float(np.percentile(us, 10)), 2), "max_us": round(float(np.max(us)), 2),
                  "threads_active": reg.get_option("batch_threads_active"), "relaunches": reg.get_option("small_relaunches") - relaunches0,
                  "iterations_mean": round(float(np.mean(batch.iterations)), 3)}))
