"""Exploration harness (not a test): parity + timing of the pass-kernel variants on one config.

usage: python tools/gpu_explore.py [cfg2] [--scans 30] [--order ring|azimuth|random]
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kinematic_icp_amd as K  # noqa: E402
from kinematic_icp_amd import synthetic as syn  # noqa: E402
from oracle import okicp  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("cfg", nargs="?", default="cfg2")
    ap.add_argument("--scans", type=int, default=30)
    ap.add_argument("--order", default="ring")
    ap.add_argument("--variants", default="0:128,3:64,3:128,3:256")
    ap.add_argument("--loops", default="0,1")
    ap.add_argument("--lanes", type=int, default=0, help="sub-lanes per query of variants 3/4 (0 = by scan size)")
    ap.add_argument("--big", type=float, default=0.0, help="extra initial-guess yaw error in degrees (more iterations)")
    args = ap.parse_args()

    t0 = time.time()
    cfg, scene, scans, rng = syn.make_case(args.cfg, n_scans=4, order=args.order if args.order in ("ring", "azimuth") else "ring")
    if args.order == "random":
        for s in scans:
            s["frame"] = np.ascontiguousarray(s["frame"][rng.permutation(len(s["frame"]))])
    if args.order in ("cell", "voxel"):
        for s in scans:
            guess = syn.pose_mul(s["last_pose"], s["rel_odom"])
            v = np.floor(syn.pose_act(guess, s["frame"]) / cfg.voxel_size).astype(np.int64) + 4096
            c = v >> 2 if args.order == "cell" else v
            key = (c[:, 2] << 40) | (c[:, 1] << 20) | c[:, 0]
            s["frame"] = np.ascontiguousarray(s["frame"][np.argsort(key, kind="stable")])
    if args.big:
        for s in scans:
            s["rel_odom"] = syn.pose_mul(s["rel_odom"], syn.planar_pose(0.15, 0.0, np.deg2rad(args.big)))
    gmap = K.VoxelHashMap(cfg.voxel_size, cfg.max_range, cfg.max_points_per_voxel)
    syn.build_map_points(scene, cfg, gmap.AddPoints, gmap.num_points, rng)
    omap = okicp.VoxelHashMap(cfg.voxel_size, cfg.max_range, cfg.max_points_per_voxel)
    omap.AddPoints(gmap.Pointcloud())  # same insertion order voxel by voxel -> identical buckets
    tau = cfg.first_frame_tau()
    print("case %s: %d pts/scan, map %d pts / %d voxels, tau %.4f, setup %.1fs" %
          (cfg.name, scans[0]["frame"].shape[0], gmap.num_points(), gmap.num_voxels(), tau, time.time() - t0), flush=True)

    oreg = okicp.KinematicRegistration()
    exp = []
    for s in scans:
        t1 = time.time()
        pose = oreg.ComputeRobotMotion(s["frame"], omap, s["last_pose"], s["rel_odom"], tau, count_work=True)
        st = oreg.last_stats
        guess = syn.pose_mul(s["last_pose"], s["rel_odom"])
        sums, cnt = okicp.icp_pass(omap, s["frame"], guess, tau)
        exp.append(dict(pose=pose, iters=st.iterations, sums=sums, cnt=cnt,
                        balgo=sum(12 * len(s["frame"]) + 16 * int(st.probes[i]) + 12 * int(st.points_scanned[i]) for i in range(st.iterations))))
        print(" oracle: iters %d conv %d t=%.3fs ncorr0 %.0f balgo %.1f MB" % (st.iterations, st.converged, time.time() - t1, st.n_corr[0], exp[-1]["balgo"] / 1e6))

    gmap.sync(0)
    dframes = [K.DeviceFrame(s["frame"]) for s in scans]
    for var in args.variants.split(","):
        kern, block = (int(x) for x in var.split(":"))
        reg = K.KinematicRegistration()
        reg.set_option("pass_kernel", kern)
        reg.set_option("block", block)
        reg.set_option("lanes_per_query", args.lanes)
        # parity: per-pass sums and final pose
        worst_sum, worst_pose, iters_ok = 0.0, 0.0, True
        for s, e in zip(scans, exp):
            guess = syn.pose_mul(s["last_pose"], s["rel_odom"])
            g = reg.pass_sums(s["frame"], gmap, guess, tau)
            worst_sum = max(worst_sum, float(np.max(np.abs(g - e["sums"]) / (np.abs(e["sums"]) + 1.0))))
            pose = reg.ComputeRobotMotion(s["frame"], gmap, s["last_pose"], s["rel_odom"], tau)
            worst_pose = max(worst_pose, float(np.max(np.abs(pose - e["pose"]))))
            iters_ok &= reg.last_stats.iterations == e["iters"]
        line = "kernel %d block %3d: sum relerr %.2e pose abserr %.2e iters_match %s |" % (kern, block, worst_sum, worst_pose, iters_ok)
        for loop in (int(x) for x in args.loops.split(",")):
            reg.set_option("loop", loop)
            reg.set_option("timing", 1)
            for i in range(5):
                reg.ComputeRobotMotion(dframes[i % 4], gmap, scans[i % 4]["last_pose"], scans[i % 4]["rel_odom"], tau)
            gpu_ms = []
            t1 = time.perf_counter()
            for i in range(args.scans):
                reg.ComputeRobotMotion(dframes[i % 4], gmap, scans[i % 4]["last_pose"], scans[i % 4]["rel_odom"], tau)
                gpu_ms.append(reg.last_stats.gpu_ms)
            wall = (time.perf_counter() - t1) / args.scans
            line += " loop%d wall %.1f us gpu %.1f us ns=%d |" % (loop, wall * 1e6, np.median(gpu_ms) * 1e3, reg.get_option("last_not_staged"))
        print(line, flush=True)


if __name__ == "__main__":
    main()
