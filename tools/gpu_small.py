"""Small-scan configs: effect of lanes_per_query and block size on cfg4 / cfg1."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kinematic_icp_amd as K
from kinematic_icp_amd import synthetic as syn
for name in ("cfg4", "cfg1"):
    cfg, scene, scans, rng = syn.make_case(name, n_scans=4)
    gmap = K.VoxelHashMap(cfg.voxel_size, cfg.max_range, cfg.max_points_per_voxel)
    syn.build_map_points(scene, cfg, gmap.AddPoints, gmap.num_points, rng)
    gmap.sync(0)
    tau = cfg.first_frame_tau()
    df = [K.DeviceFrame(s["frame"]) for s in scans]
    for G in (1, 2, 4):
        for block in (64, 128, 256):
            reg = K.KinematicRegistration()
            reg.set_option("lanes_per_query", G); reg.set_option("block", block); reg.set_option("timing", 2)
            ms = []
            for i in range(30):
                reg.ComputeRobotMotion(df[i % 4], gmap, scans[i % 4]["last_pose"], scans[i % 4]["rel_odom"], tau)
                ms.append(reg.last_stats.pass_ms[0])
            reg.set_option("timing", 0)
            t0 = time.perf_counter()
            for i in range(200):
                reg.ComputeRobotMotion(df[i % 4], gmap, scans[i % 4]["last_pose"], scans[i % 4]["rel_odom"], tau)
            wall = (time.perf_counter() - t0) / 200 * 1e6
            print("%s G %d block %3d: pass %.1f us, wall %.1f us/scan (iters %d)" % (name, G, block, np.median(ms[5:]) * 1e3, wall, reg.last_stats.iterations), flush=True)
