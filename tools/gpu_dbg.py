"""Experiment: where the pass kernel's time goes (ablation switches of kicp_kernels.hpp, option "dbg"):
0 full; 1 search without the exact phase / accumulation; 5 own + face voxels only; 3 own voxel only; 4 probe, no bucket visit;
2 no probe either; 7 no query work at all (launch + reduction + hand-off); 8 no reduction either (launch + hand-off).
Switches that leave the RESULT alone (for in-process A/Bs, tools/ab_option.py --option dbg): 9 no exact fp64 fallback search for three
near-equal candidates (changes poses in rare ties: diagnosis only); 10 the rounds census of bench.py's latency model (the count sum
carries every wave's visiting rounds); 11 idle lanes do NOT take over voxels of loaded queries (four-waves build); 12 all seven wave
sums through the DPP reduction (none derived from a ballot); 13 the exact phase without the terms (tools/valu_attribution.sh); 14 the plain
launch hands its group rows over as round 4 did (rows -> ticket -> reload) instead of through the counting accumulators."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kinematic_icp_amd as K
from kinematic_icp_amd import synthetic as syn

wl = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
cfg, scene, scans, rng = syn.make_case(wl, n_scans=2)
gmap = K.VoxelHashMap(cfg.voxel_size, cfg.max_range, cfg.max_points_per_voxel)
ident = np.array([0, 0, 0, 1.0, 0, 0, 0])
syn.build_map_points(scene, cfg, lambda pts: gmap.UpdateDevice(K.DeviceFrame(pts), ident), gmap.num_points, rng)
gmap.sync(0)
tau = cfg.first_frame_tau()
df = [K.DeviceFrame(s["frame"]) for s in scans]
for block in (256,):
    for dbg in (0, 1, 5, 3, 4, 2, 7, 8):
        reg = K.KinematicRegistration()
        reg.set_option("block", block)
        reg.set_option("dbg", dbg)
        reg.set_option("timing", 2)
        reg.max_num_iterations_ = 1
        ms = []
        for i in range(60):
            reg.ComputeRobotMotion(df[i % 2], gmap, scans[i % 2]["last_pose"], scans[i % 2]["rel_odom"], tau)
            ms.append(reg.last_stats.pass_ms[0])
        print("%s block %3d dbg %d: pass %.2f us (events, median of 50)" % (wl, block, dbg, np.median(ms[10:]) * 1e3), flush=True)
