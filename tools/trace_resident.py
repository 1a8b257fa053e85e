"""Where a pass of the generic kernel goes when it is resident (k_pass_resident; debug option "small_trace"): every workgroup
stamps the second pass of a launch - command seen, search done (all its waves), row stored / group row sent, next command seen
(100 MHz device clock) - next to the host's wall clock per pass.  The scans here need several iterations each (+0.05 m / +0.5
deg odometry error), so a launch's second pass is an ordinary later iteration of one scan.

    python tools/trace_resident.py [cfg2]
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kinematic_icp_amd as K  # noqa: E402
from kinematic_icp_amd import synthetic as syn  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
cfg, scene, scans, rng = syn.make_case(name, n_scans=8)
gmap = K.VoxelHashMap(cfg.voxel_size, cfg.max_range, cfg.max_points_per_voxel)
ident = np.array([0, 0, 0, 1.0, 0, 0, 0])
syn.build_map_points(scene, cfg, lambda pts: gmap.UpdateDevice(K.DeviceFrame(pts), ident), gmap.num_points, rng)
gmap.sync(0)
tau = cfg.first_frame_tau()
frames = [K.DeviceFrame(s["frame"]) for s in scans]
rels = [syn.pose_mul(s["rel_odom"], syn.planar_pose(0.05, 0.0, np.deg2rad(0.5))) for s in scans]
n = len(scans[0]["frame"])
grid = -(-n // 256)
reg = K.KinematicRegistration()
reg.set_option("small_resident", 2)
for i in range(200):
    reg.ComputeRobotMotion(frames[i % 8], gmap, scans[i % 8]["last_pose"], rels[i % 8], tau)
t0 = time.perf_counter()
its = 0
for i in range(400):
    reg.ComputeRobotMotion(frames[i % 8], gmap, scans[i % 8]["last_pose"], rels[i % 8], tau)
    its += reg.last_stats.iterations
wall_us = (time.perf_counter() - t0) * 1e6 / its
reg.set_option("small_trace", 1)
rec = []
for i in range(200):
    reg.ComputeRobotMotion(frames[i % 8], gmap, scans[i % 8]["last_pose"], rels[i % 8], tau)
    if reg.last_stats.iterations >= 3 and reg.get_option("resident_passes") >= 3:
        v = np.array([reg.get_option("trace_stamp_%d" % j) for j in range(4 * grid)]).reshape(grid, 4) / 100.0
        rec.append(v - v[:, 0].min())
print("%s: %d points, %d workgroups; wall clock per ICP iteration %.2f us (resident from the first pass on)" % (name, n, grid, wall_us))
print("host: command sent -> rows seen %.2f us, rows seen -> command sent %.2f us, launch -> first rows %.2f us" % (
    reg.get_option("trace_device_us"), reg.get_option("trace_host_us"), reg.get_option("trace_first_us")))
r = np.array(rec)  # [calls][grid][4]
print("%d traced passes" % len(rec))
for j, label in enumerate(("command seen", "search done", "row stored", "next command seen")):
    x = r[:, :, j]
    print("    %-18s (us after the first workgroup saw the command): first %.2f  median %.2f  p90 %.2f  last %.2f" % (
        label, x.min(axis=1).mean(), np.median(x, axis=1).mean(), np.percentile(x, 90, axis=1).mean(), x.max(axis=1).mean()))
search = r[:, :, 1] - r[:, :, 0]
rows = r[:, :, 2] - r[:, :, 1]
print("    per workgroup: search mean %.2f / p90 %.2f / max %.2f us; reduction + row %.2f (max %.2f) us" % (
    search.mean(), np.percentile(search, 90, axis=1).mean(), search.max(axis=1).mean(), rows.mean(), rows.max(axis=1).mean()))
print("    pass on the device (first command seen -> last row stored) %.2f us; last row stored -> first next command seen %.2f us" % (
    r[:, :, 2].max(axis=1).mean(), (r[:, :, 3].min(axis=1) - r[:, :, 2].max(axis=1)).mean()))
