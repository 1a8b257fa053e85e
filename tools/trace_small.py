"""Where a resident pass of the small-scan path spends its time (debug option "small_trace"): every workgroup stamps its second
pass (command seen, search done, row stored, next command seen; 100 MHz device clock), next to the host's view (launch -> first
rows, command sent -> rows seen, rows seen -> command sent)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kinematic_icp_amd as K  # noqa: E402
from kinematic_icp_amd import synthetic as syn  # noqa: E402

cfg, scene, scans, rng = syn.make_case(sys.argv[1] if len(sys.argv) > 1 else "cfg4", n_scans=8)
gmap = K.VoxelHashMap(cfg.voxel_size, cfg.max_range, cfg.max_points_per_voxel)
syn.build_map_points(scene, cfg, gmap.AddPoints, gmap.num_points, rng)
gmap.sync(0)
tau = cfg.first_frame_tau()
frames = [K.DeviceFrame(s["frame"]) for s in scans]
rels = [syn.pose_mul(s["rel_odom"], syn.planar_pose(0.05, 0.0, np.deg2rad(0.5))) for s in scans]
n = len(scans[0]["frame"])
for opts in (dict(), dict(small_cmd=0)):
    os.environ["KICP_KERNARG"] = "dev"
    reg = K.KinematicRegistration()
    for k, v in opts.items():
        reg.set_option(k, v)
    grid = -(-n // ((opts.get("wave_block") or (256 if n <= 512 else (512 if n <= 2176 else 1024))) // 64))  # (wave_block 0 = by scan size)
    for i in range(200):
        reg.ComputeRobotMotion(frames[i % 8], gmap, scans[i % 8]["last_pose"], rels[i % 8], tau)
    reg.set_option("small_trace", 1)
    rec = []
    for i in range(300):
        reg.ComputeRobotMotion(frames[i % 8], gmap, scans[i % 8]["last_pose"], rels[i % 8], tau)
        if reg.last_stats.iterations >= 3:
            v = np.array([reg.get_option("trace_stamp_%d" % j) for j in range(4 * grid)]).reshape(grid, 4) / 100.0
            rec.append(v - v[:, 0].min())
    print("%-40s host: launch->rows %.2f us, command->rows %.2f us, rows->command %.2f us" % (
        opts, reg.get_option("trace_first_us"), reg.get_option("trace_device_us"), reg.get_option("trace_host_us")), flush=True)
    if rec:
        r = np.array(rec)  # [calls][grid][4]
        for j, name in enumerate(("command seen", "search done", "row stored", "next command seen")):
            x = r[:, :, j]
            print("    %-18s (us after the first workgroup saw the command): first %.2f  median %.2f  p90 %.2f  last %.2f" % (
                name, x.min(axis=1).mean(), np.median(x, axis=1).mean(), np.percentile(x, 90, axis=1).mean(), x.max(axis=1).mean()), flush=True)
        print("    per workgroup: search %.2f (max %.2f), row phase %.2f (max %.2f) us" % (
            (r[:, :, 1] - r[:, :, 0]).mean(), (r[:, :, 1] - r[:, :, 0]).max(axis=1).mean(), (r[:, :, 2] - r[:, :, 1]).mean(), (r[:, :, 2] - r[:, :, 1]).max(axis=1).mean()))
