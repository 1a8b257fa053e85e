"""Per-workgroup timeline of one pass in the MIDDLE of a batch call (kicp_register_device_batch, kernel resident across the
batch's scans; debug option "small_trace" = index of the pass to stamp): pass taken up, search done (all waves), row stored /
group row sent, next command seen (100 MHz device clock) - for every "batch_depth" given, next to the wall clock per scan.

    python tools/trace_batch.py [cfg2] [--depths 1 2 4] [--pass 40]
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kinematic_icp_amd as K  # noqa: E402
from kinematic_icp_amd import synthetic as syn  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("workload", nargs="?", default="cfg2")
ap.add_argument("--depths", type=int, nargs="+", default=[1, 2, 4])
ap.add_argument("--pass", dest="at", type=int, default=40)
ap.add_argument("--scans", type=int, default=128)
ap.add_argument("--fixed", action="append", default=[], help="name=value options set on every handle")
args = ap.parse_args()
cfg, scene, scans, rng = syn.make_case(args.workload, n_scans=8)
gmap = K.VoxelHashMap(cfg.voxel_size, cfg.max_range, cfg.max_points_per_voxel)
ident = np.array([0, 0, 0, 1.0, 0, 0, 0])
syn.build_map_points(scene, cfg, lambda pts: gmap.UpdateDevice(K.DeviceFrame(pts), ident), gmap.num_points, rng)
gmap.sync(0)
tau = cfg.first_frame_tau()
frames = [K.DeviceFrame(s["frame"]) for s in scans]
n = len(scans[0]["frame"])
B = args.scans
for depth in args.depths:
    reg = K.KinematicRegistration()
    reg.set_option("batch_queues", 0)  # (the resident kernel is what is traced: large scans would otherwise go out on several queues)
    reg.set_option("batch_depth", depth)
    for o in args.fixed:
        reg.set_option(o.split("=")[0], float(o.split("=")[1]))
    batch = reg.prepare_batch([frames[i % 8] for i in range(B)], [scans[i % 8]["last_pose"] for i in range(B)], [scans[i % 8]["rel_odom"] for i in range(B)])
    for _ in range(5):
        reg.ComputeRobotMotionBatch(batch, gmap, tau)
    t0 = time.perf_counter()
    for _ in range(20):
        reg.ComputeRobotMotionBatch(batch, gmap, tau)
    wall_us = (time.perf_counter() - t0) * 1e6 / (20 * B)
    reg.set_option("small_trace", args.at)
    rec = []
    grid = None
    for _ in range(30):
        reg.ComputeRobotMotionBatch(batch, gmap, tau)
        v = np.array([reg.get_option("trace_stamp_%d" % j) for j in range(4096)]).reshape(1024, 4)
        used = v[:, 0] > 0
        grid = int(used.sum())
        v = v[used] / 100.0
        rec.append(v - v[:, 0].min())
    r = np.array(rec)  # [calls][grid][4]
    print("%s, batch_depth %d: %d points, %d workgroups, %.2f us per scan (wall clock, %d-scan batch calls); pass %d of the launch stamped, %d calls" % (
        args.workload, depth, n, grid, wall_us, B, args.at, len(rec)))
    for j, label in enumerate(("pass taken up", "search done", "row stored", "next command seen")):
        x = r[:, :, j]
        print("    %-18s (us after the first workgroup took the pass up): first %6.2f  median %6.2f  p90 %6.2f  last %6.2f" % (
            label, x.min(axis=1).mean(), np.median(x, axis=1).mean(), np.percentile(x, 90, axis=1).mean(), x.max(axis=1).mean()))
    search, rows, wait = r[:, :, 1] - r[:, :, 0], r[:, :, 2] - r[:, :, 1], r[:, :, 3] - r[:, :, 2]
    print("    per workgroup: search mean %.2f / p90 %.2f / max %.2f us; reduction + row mean %.2f (max %.2f) us; wait for the next command mean %.2f / p10 %.2f / max %.2f us; "
          "whole cycle mean %.2f us" % (search.mean(), np.percentile(search, 90, axis=1).mean(), search.max(axis=1).mean(), rows.mean(), rows.max(axis=1).mean(),
                                        wait.mean(), np.percentile(wait, 10, axis=1).mean(), wait.max(axis=1).mean(), (r[:, :, 3] - r[:, :, 0]).mean()))
    slow = np.argsort(-search.mean(axis=0))[:6]
    print("    slowest workgroups (mean search us over the calls; the stamped pass registers the same scan in every call): " +
          ", ".join("%d: %.1f" % (int(b), search.mean(axis=0)[b]) for b in slow))
    late = np.argsort(-r[:, :, 0].mean(axis=0))[:8]
    print("    latest to take the pass up (mean us after the first): " + ", ".join("%d: %.1f" % (int(b), r[:, :, 0].mean(axis=0)[b]) for b in late) +
          "; workgroups more than 2 us late: %d" % int((r[:, :, 0].mean(axis=0) > 2.0).sum()))
