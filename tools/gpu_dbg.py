"""Experiment: split of the binned pass kernel's time (dbg modes)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kinematic_icp_amd as K
from kinematic_icp_amd import synthetic as syn

cfg, scene, scans, rng = syn.make_case("cfg2", n_scans=2)
gmap = K.VoxelHashMap(cfg.voxel_size, cfg.max_range, cfg.max_points_per_voxel)
syn.build_map_points(scene, cfg, gmap.AddPoints, gmap.num_points, rng)
gmap.sync(0)
tau = cfg.first_frame_tau()
df = [K.DeviceFrame(s["frame"]) for s in scans]
for wpc in (4, 8, 16):
    for dbg in (0, 1, 2):
        reg = K.KinematicRegistration()
        reg.set_option("dbg", dbg)
        reg.set_option("waves_per_cu", wpc)
        reg.set_option("timing", 1)
        reg.max_num_iterations_ = 1
        ms = []
        for i in range(12):
            reg.ComputeRobotMotion(df[i % 2], gmap, scans[i % 2]["last_pose"], scans[i % 2]["rel_odom"], tau)
            ms.append(reg.last_stats.gpu_ms)
        print("waves_per_cu %2d dbg %d: gpu %.1f us" % (wpc, dbg, np.median(ms[2:]) * 1e3), flush=True)
