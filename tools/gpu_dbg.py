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
for kern, block, G, rows in ((3, 128, 1, 1), (3, 128, 1, 0)):
    # 0 full; 1 search without accumulation; 5 own + face voxels; 3 own voxel only; 4 probe + masks, no bucket visit;
    # 2 no probe either; 7 no query work at all (launch + reduction + hand-off); 8 no reduction either (launch + hand-off)
    for dbg in (0, 7, 8):
        reg = K.KinematicRegistration()
        reg.set_option("lanes_per_query", G)
        reg.set_option("pass_kernel", kern); reg.set_option("block", block)
        reg.set_option("dbg", dbg)
        reg.set_option("group_rows", rows)
        reg.set_option("timing", 2)
        reg.max_num_iterations_ = 1
        ms = []
        for i in range(14):
            reg.ComputeRobotMotion(df[i % 2], gmap, scans[i % 2]["last_pose"], scans[i % 2]["rel_odom"], tau)
            ms.append(reg.last_stats.pass_ms[0])
        print("kernel %d block %3d G %d group_rows %d dbg %d: pass %.1f us" % (kern, block, G, rows, dbg, np.median(ms[4:]) * 1e3), flush=True)
