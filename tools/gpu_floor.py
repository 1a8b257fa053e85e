"""Latency floor of one registration call vs scan size (variant 3, host solve)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kinematic_icp_amd as K
from kinematic_icp_amd import synthetic as syn
cfg, scene, scans, rng = syn.make_case("cfg2", n_scans=1)
gmap = K.VoxelHashMap(cfg.voxel_size, cfg.max_range, cfg.max_points_per_voxel)
syn.build_map_points(scene, cfg, gmap.AddPoints, gmap.num_points, rng)
gmap.sync(0)
tau = cfg.first_frame_tau()
s = scans[0]
for n in (64, 1024, 4096, 16384, 65536, 131072):
    idx = np.linspace(0, len(s["frame"]) - 1, n).astype(int)
    df = K.DeviceFrame(s["frame"][idx])
    reg = K.KinematicRegistration()
    reg.set_option("timing", 2)
    ms = []
    for i in range(40):
        reg.ComputeRobotMotion(df, gmap, s["last_pose"], s["rel_odom"], tau)
        ms.append(reg.last_stats.pass_ms[0])
    reg.set_option("timing", 0)
    for i in range(300):
        reg.ComputeRobotMotion(df, gmap, s["last_pose"], s["rel_odom"], tau)
    t0 = time.perf_counter()
    for i in range(500):
        reg.ComputeRobotMotion(df, gmap, s["last_pose"], s["rel_odom"], tau)
    wall = (time.perf_counter() - t0) / 500 * 1e6
    print("n %6d: pass(events) %.1f us, wall %.1f us/call, iters %d, G %d" % (n, np.median(ms[5:]) * 1e3, wall, reg.last_stats.iterations, 4 if n <= 4096 else (2 if n <= 32768 else 1)), flush=True)
