"""The generic pass kernel taken apart with the ablation switches of kicp_kernels.hpp (option "dbg").  The switches exist in
libkicp_amd_dbg.so only (make -C kinematic_icp_amd/csrc dbg, built by __graft_entry__.build()): the production library's kernels
carry none of them.  bench.py runs this script in a process of its own for two figures of its latency model:

    floor_us          the launch with every query switched off (dbg 7): dispatch + wave launch + reduction + hand-off, HIP events
    rounds_per_wave   visiting rounds per wave, counted by the kernel itself (dbg 10: the "count" sum carries every wave's rounds)

    python tools/dbg_census.py --workload cfg2 [--latency-kernel 0] [--ladder]
--ladder adds the ablation ladder: 0 full; 5 own + face voxels only; 3 own voxel only; 4 probe, no bucket visit; 2 no probe either;
7 no query work at all; 8 no reduction either.  (Result-preserving switches for in-process A/Bs, tools/ab_option.py --option dbg:
9 no exact fp64 fallback search; 11 idle lanes do not take over voxels; 12 all seven wave sums through the DPP reduction; 13 the exact
phase without the terms; 14 the plain launch hands its group rows over as round 4 did.)
Prints one line of JSON."""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import kinematic_icp_amd as K  # noqa: E402
K.LIB_PATH = os.environ.get("KICP_AB_LIB") or os.path.join(ROOT, "kinematic_icp_amd", "libkicp_amd_dbg.so")
from kinematic_icp_amd import synthetic as syn  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="cfg2")
ap.add_argument("--latency-kernel", type=int, default=1)
ap.add_argument("--lanes", type=int, default=0)
ap.add_argument("--ladder", action="store_true")
a = ap.parse_args()

cfg, scene, scans, rng = syn.make_case(a.workload, n_scans=4)
gmap = K.VoxelHashMap(cfg.voxel_size, cfg.max_range, cfg.max_points_per_voxel)
ident = np.array([0, 0, 0, 1.0, 0, 0, 0])
syn.build_map_points(scene, cfg, lambda pts: gmap.UpdateDevice(K.DeviceFrame(pts), ident), gmap.num_points, rng)
gmap.sync(0)
tau = cfg.first_frame_tau()
frames = [K.DeviceFrame(s["frame"]) for s in scans]
n = len(scans[0]["frame"])


def handle(dbg):
    reg = K.KinematicRegistration()
    reg.set_option("small", 0), reg.set_option("latency_kernel", a.latency_kernel), reg.set_option("lanes_per_query", a.lanes)
    reg.set_option("dbg", dbg), reg.set_option("timing", 2)
    reg.max_num_iterations_ = 1
    return reg


def pass_us(dbg, calls=320, skip=64):
    reg = handle(dbg)
    ms = []
    for i in range(calls):
        reg.ComputeRobotMotion(frames[i % 4], gmap, scans[i % 4]["last_pose"], scans[i % 4]["rel_odom"], tau)
        ms.append(reg.last_stats.pass_ms[0])
    return float(np.mean(ms[skip:]) * 1e3)


out = {"workload": a.workload, "points": n, "latency_kernel": a.latency_kernel, "floor_us": round(pass_us(7), 3)}
reg = handle(10)
lanes = a.lanes or (4 if n <= 4096 else (2 if n <= 32768 else 1))
waves = -(-n * lanes // 64)
rr = []
for i in range(16):
    reg.ComputeRobotMotion(frames[i % 4], gmap, scans[i % 4]["last_pose"], scans[i % 4]["rel_odom"], tau)
    rr.append(reg.last_stats.n_corr[0] / waves)
out["rounds_per_wave"] = round(float(np.mean(rr)), 4)
if a.ladder:
    out["ladder_us"] = {str(d): round(pass_us(d, 120, 20), 2) for d in (0, 5, 3, 4, 2, 7, 8)}
print(json.dumps(out))
