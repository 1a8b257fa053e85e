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
fits = len(scans[0]["frame"]) <= 8192
run("generic pass kernel, kernargs in host memory (round 2 path)", env={"KICP_KERNARG": "host"}, small=0)
run("generic pass kernel, HIP launch                              ", small=0, aql=0)
run("generic pass kernel (kernargs in HBM: default)               ", small=0)
if fits:
    run("small: wave per query, resident, commands over BAR (default) ")
    run("small: wave per query, resident, kernargs in host memory     ", env={"KICP_KERNARG": "host"})
    run("small: wave per query, resident, commands relayed by wg 0    ", small_cmd=0)
    run("small: wave per query, always resident                       ", small_resident=2)
    run("small: wave per query, one launch per pass                   ", small_resident=0)
    run("small: wave per query, resident, HIP launch                  ", aql=0)
    for b in (256, 512, 1024):
        run("small: wave per query, resident, %4d-lane workgroups        " % b, wave_block=b)
    run("small: sub-lanes per query, resident                         ", small_wave=0)
    run("small: sub-lanes per query, one launch per pass              ", small_wave=0, small_resident=0)
