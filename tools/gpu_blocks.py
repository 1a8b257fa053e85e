"""Experiment: pass kernel time (events) and wall time per scan vs workgroup size, with the ablation switches 0 / 7 / 8."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kinematic_icp_amd as K
from kinematic_icp_amd import synthetic as syn

wl = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
blocks = [int(b) for b in sys.argv[2].split(",")] if len(sys.argv) > 2 else [128, 256, 512]
cfg, scene, scans, rng = syn.make_case(wl, n_scans=4)
gmap = K.VoxelHashMap(cfg.voxel_size, cfg.max_range, cfg.max_points_per_voxel)
ident = np.array([0, 0, 0, 1.0, 0, 0, 0])
syn.build_map_points(scene, cfg, lambda pts: gmap.UpdateDevice(K.DeviceFrame(pts), ident), gmap.num_points, rng)
gmap.sync(0)
tau = cfg.first_frame_tau()
df = [K.DeviceFrame(s["frame"]) for s in scans]
ref = None
for block, aql in [(b, a) for b in blocks for a in (1, 0)]:
    for dbg in (0, 7, 8):
        reg = K.KinematicRegistration()
        reg.set_option("aql", aql)
        reg.set_option("block", block)
        reg.set_option("dbg", dbg)
        reg.set_option("timing", 2)
        ms = []
        for i in range(200):
            reg.ComputeRobotMotion(df[i % 4], gmap, scans[i % 4]["last_pose"], scans[i % 4]["rel_odom"], tau)
            ms.append(reg.last_stats.pass_ms[0])
        reg.set_option("timing", 0)
        batch = reg.prepare_batch([df[i % 4] for i in range(64)], [scans[i % 4]["last_pose"] for i in range(64)], [scans[i % 4]["rel_odom"] for i in range(64)])
        for i in range(5):
            out = reg.ComputeRobotMotionBatch(batch, gmap, tau)
        t0 = time.perf_counter()
        for i in range(20):
            out = reg.ComputeRobotMotionBatch(batch, gmap, tau)
        wall = (time.perf_counter() - t0) / (20 * 64) * 1e6
        if dbg == 0:
            if ref is None:
                ref = out.copy()
            assert np.array_equal(ref, out), "results differ between workgroup sizes"
        print("%s block %3d aql %d(active %d) dbg %d: pass %.2f us (events, median), wall %.2f us/scan, iters %.2f" % (wl, block, aql, int(reg.get_option("aql_active")), dbg, np.median(ms[20:]) * 1e3, wall, batch.iterations.mean()), flush=True)
