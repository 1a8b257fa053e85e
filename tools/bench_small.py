"""Latency of one ComputeRobotMotion call on small scans (BASELINE.json config 4 and pipeline-sized sources): the small-scan
path (kicp_small.hpp) against the generic pass kernel, resident against one launch per pass, kernel arguments in host against
device memory.  Wall clock through kicp_register_device_batch (no Python between scans).

    python tools/bench_small.py [cfg4] [--scans 2000]
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
ap.add_argument("workload", nargs="?", default="cfg4")
ap.add_argument("--scans", type=int, default=3000)
ap.add_argument("--points", type=int, default=0, help="use only the first N points of every scan (0 = all)")
args = ap.parse_args()

cfg, scene, scans, rng = syn.make_case(args.workload, n_scans=8)
gmap = K.VoxelHashMap(cfg.voxel_size, cfg.max_range, cfg.max_points_per_voxel)
syn.build_map_points(scene, cfg, gmap.AddPoints, gmap.num_points, rng)
gmap.sync(0)
tau = cfg.first_frame_tau()
if args.points:
    for s in scans:
        s["frame"] = np.ascontiguousarray(s["frame"][np.linspace(0, len(s["frame"]) - 1, args.points).astype(int)])
frames = [K.DeviceFrame(s["frame"]) for s in scans]
extra = syn.planar_pose(0.05, 0.0, np.deg2rad(0.5))
rels = {"1-iteration scans": [s["rel_odom"] for s in scans], "multi-iteration scans": [syn.pose_mul(s["rel_odom"], extra) for s in scans]}
B = 64


def run(label, env=None, **opts):
    for k, v in (env or {}).items():
        os.environ[k] = v
    reg = K.KinematicRegistration()
    for k, v in opts.items():
        reg.set_option(k, v)
    out = [label]
    for name, rel in rels.items():
        batch = reg.prepare_batch([frames[i % 8] for i in range(B)], [scans[i % 8]["last_pose"] for i in range(B)], [rel[i % 8] for i in range(B)])
        for _ in range(6):
            reg.ComputeRobotMotionBatch(batch, gmap, tau)
        steps = max(1, args.scans // B)
        t0 = time.perf_counter()
        for _ in range(steps):
            reg.ComputeRobotMotionBatch(batch, gmap, tau)
        us = (time.perf_counter() - t0) / (steps * B) * 1e6
        iters = float(np.mean(batch.iterations))
        out.append("%s: %.2f us/scan, %.2f iterations, %.2f us/iteration" % (name, us, iters, us / iters))
    out.append("small %d aql %d kernarg %d cmd %.1f relaunches %d" % (reg.get_option("small_active"), reg.get_option("aql_active"), reg.get_option("aql_kernarg"),
                                                                       reg.get_option("small_cmd"), reg.get_option("small_relaunches")))
    for k in (env or {}):
        os.environ.pop(k, None)
    print(" | ".join(out), flush=True)


print("%s: %d-point scans vs %d-point map" % (cfg.name, len(scans[0]["frame"]), gmap.num_points()), flush=True)
fits = len(scans[0]["frame"]) <= 4096
run("generic pass kernel (round 2 path)", small=0)
run("generic, HIP launch               ", small=0, aql=0)
run("generic, kernargs in HBM          ", env={"KICP_KERNARG": "dev"}, small=0)
run("generic, kernargs in HBM + HDP    ", env={"KICP_KERNARG": "devhdp"}, small=0)
if fits:
    for b in (0, 256, 512, 1024):
        run("wave/query resident relay, block %4d" % b, wave_block=b)
    for b in (0, 512, 1024):
        run("wave/query resident BAR,   block %4d" % b, wave_block=b, small_cmd=1)
    run("wave per query, 1 launch per pass   ", small_resident=0)
    run("wave/query resident relay, HBM kargs", env={"KICP_KERNARG": "dev"})
    run("wave/query resident BAR, HBM kargs  ", env={"KICP_KERNARG": "dev"}, small_cmd=1)
    run("wave/query 1 launch/pass, HBM kargs ", env={"KICP_KERNARG": "dev"}, small_resident=0)
    run("wave/query resident relay, HIP launch", aql=0)
    run("sub-lanes, one launch per pass      ", small_wave=0, small_resident=0)
    run("sub-lanes, resident relay           ", small_wave=0)
    run("sub-lanes, resident BAR             ", small_wave=0, small_cmd=1)
    run("sub-lanes, resident BAR, HBM kargs  ", small_wave=0, small_cmd=1, env={"KICP_KERNARG": "dev"})
