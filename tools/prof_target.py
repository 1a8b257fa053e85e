"""Profiling target: N registrations of one BASELINE workload with nothing else in the process (the map is built on the
device, which takes seconds even for cfg5's 10M points), for rocprofv3 --pmc / --kernel-trace passes.

    python tools/prof_target.py --workload cfg2 --calls 300 [--multi] [--build host] [--batch 64]
Prints one line of JSON with the wall-clock rate and the mean pass-kernel time from HIP events (un-profiled reference).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kinematic_icp_amd as K  # noqa: E402
if os.environ.get("KICP_AB_LIB"):  # A/B runs against another build of the library (tools/ab/)
    K.LIB_PATH = os.environ["KICP_AB_LIB"]
from kinematic_icp_amd import synthetic as syn  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="cfg2")
ap.add_argument("--calls", type=int, default=300)
ap.add_argument("--multi", action="store_true", help="the multi-iteration variant (+0.2 m / +1.5 deg odometry error)")
ap.add_argument("--build", default="device", choices=["device", "host"])
ap.add_argument("--events", action="store_true", help="HIP events around every pass (adds two event records per launch)")
ap.add_argument("--option", action="append", default=[], help="name=value registration options")
ap.add_argument("--batch", type=int, default=0, help="register through kicp_register_device_batch calls of this many scans (what bench.py times) instead of one call per scan")
args = ap.parse_args()

cfg, scene, scans, rng = syn.make_case(args.workload, n_scans=4)
gmap = K.VoxelHashMap(cfg.voxel_size, cfg.max_range, cfg.max_points_per_voxel)
ident = np.array([0, 0, 0, 1.0, 0, 0, 0])
t0 = time.time()
if args.build == "device":
    syn.build_map_points(scene, cfg, lambda pts: gmap.UpdateDevice(K.DeviceFrame(pts), ident), gmap.num_points, rng)
else:
    syn.build_map_points(scene, cfg, gmap.AddPoints, gmap.num_points, rng)
gmap.sync(0)
t_build = time.time() - t0
tau = cfg.first_frame_tau()
frames = [K.DeviceFrame(s["frame"]) for s in scans]
extra = syn.planar_pose(0.2, 0.0, np.deg2rad(1.5)) if args.multi else syn.planar_pose(0.0, 0.0, 0.0)
rels = [syn.pose_mul(s["rel_odom"], extra) for s in scans]
reg = K.KinematicRegistration()
for o in args.option:
    k, v = o.split("=")
    reg.set_option(k, float(v))
if args.events:
    reg.set_option("timing", 2)
pass_ms, iters = [], []
for i in range(0 if args.batch > 0 else 50):  # (batch mode: only the batch's own kernel build is dispatched in this process)
    reg.ComputeRobotMotion(frames[i % 4], gmap, scans[i % 4]["last_pose"], rels[i % 4], tau)
K.lib().kicp_device_synchronize(0)
t0 = time.perf_counter()
if args.batch > 0:
    B = args.batch
    batch = reg.prepare_batch([frames[i % 4] for i in range(B)], [scans[i % 4]["last_pose"] for i in range(B)], [rels[i % 4] for i in range(B)])
    done = 0
    while done < args.calls:
        reg.ComputeRobotMotionBatch(batch, gmap, tau)
        iters += list(batch.iterations)
        done += B
    args.calls = done
for i in range(0 if args.batch > 0 else args.calls):
    reg.ComputeRobotMotion(frames[i % 4], gmap, scans[i % 4]["last_pose"], rels[i % 4], tau)
    k = reg.last_stats.iterations
    iters.append(k)
    if args.events:
        pass_ms += list(reg.last_stats.pass_ms[:k])
K.lib().kicp_device_synchronize(0)
dt = time.perf_counter() - t0
print(json.dumps({"workload": args.workload, "multi": args.multi, "build": args.build, "map_points": gmap.num_points(), "map_voxels": gmap.num_voxels(),
                  "build_s": round(t_build, 2), "calls": args.calls, "scans_per_s": round(args.calls / dt, 1), "iterations_mean": float(np.mean(iters)),
                  "pass_us_events": round(float(np.mean(pass_ms)) * 1e3, 2) if pass_ms else None,
                  "batch_queue_passes": reg.get_option("batch_queue_passes"), "batch_resident_passes": reg.get_option("batch_resident_passes")}))
