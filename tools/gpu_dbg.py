"""Experiment: where the pass kernels' time goes (dbg 0 = full, 1 = search without accumulation, 2 = neither)."""
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
for kern, block, G in ((3, 128, 1), (3, 256, 1)):
    for dbg in (0, 3, 2):
        reg = K.KinematicRegistration()
        reg.set_option("lanes_per_query", G)
        reg.set_option("pass_kernel", kern); reg.set_option("block", block)
        reg.set_option("dbg", dbg)
        reg.set_option("timing", 2)
        reg.max_num_iterations_ = 1
        ms = []
        for i in range(14):
            reg.ComputeRobotMotion(df[i % 2], gmap, scans[i % 2]["last_pose"], scans[i % 2]["rel_odom"], tau)
            ms.append(reg.last_stats.pass_ms[0])
        print("kernel %d block %3d G %d dbg %d: pass %.1f us" % (kern, block, G, dbg, np.median(ms[4:]) * 1e3), flush=True)
