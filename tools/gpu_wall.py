"""Wall-time per call with/without torch imported first (which HIP runtime serves libkicp_amd.so)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "torch":
    import torch
    torch.cuda.init()
import kinematic_icp_amd as K
from kinematic_icp_amd import synthetic as syn
cfg, scene, scans, rng = syn.make_case("cfg2", n_scans=4)
gmap = K.VoxelHashMap(cfg.voxel_size, cfg.max_range, cfg.max_points_per_voxel)
syn.build_map_points(scene, cfg, gmap.AddPoints, gmap.num_points, rng)
gmap.sync(0)
tau = cfg.first_frame_tau()
df = [K.DeviceFrame(s["frame"]) for s in scans]
with open("/proc/self/maps") as f:
    libs = sorted({l.split()[-1] for l in f if "libamdhip64" in l or "librccl" in l})
print("runtime libs:", libs)
for loop, wait, qe in ((1, 0, 64), (1, 0, 64), (1, 0, 64), (1, 0, 16), (1, 1, 64), (1, 0, 64)):
    reg = K.KinematicRegistration()
    reg.set_option("loop", loop); reg.set_option("wait", wait); reg.set_option("query_every", qe)
    t0 = time.perf_counter()
    for i in range(20):
        reg.ComputeRobotMotion(df[i % 4], gmap, scans[i % 4]["last_pose"], scans[i % 4]["rel_odom"], tau)
    print("  first 20 calls: %.1f us/call" % ((time.perf_counter() - t0) / 20 * 1e6))
    for blk in range(3):
        t0 = time.perf_counter()
        for i in range(100):
            reg.ComputeRobotMotion(df[i % 4], gmap, scans[i % 4]["last_pose"], scans[i % 4]["rel_odom"], tau)
        print("  block %d: %.1f us/call" % (blk, (time.perf_counter() - t0) / 100 * 1e6))
    t0 = time.perf_counter()
    for i in range(300):
        reg.ComputeRobotMotion(df[i % 4], gmap, scans[i % 4]["last_pose"], scans[i % 4]["rel_odom"], tau)
    print("loop %d wait %d query_every %d: %.1f us/call" % (loop, wait, qe, (time.perf_counter() - t0) / 300 * 1e6), flush=True)
