"""Where the time of the pre-steps goes for one pipeline-sized frame (131 072 points): upload alone, deskew + crop with and
without stamps, the two downsamples; median of 200 calls each (wall clock around the C-ABI call).
    python tools/time_presteps.py [--points 131072]"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kinematic_icp_amd as K  # noqa: E402
from kinematic_icp_amd import synthetic as syn  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--points", type=int, default=131072)
args = ap.parse_args()
cfg, scene, scans, rng = syn.make_case("cfg2", n_scans=1)
frame = np.ascontiguousarray(scans[0]["frame"][:args.points])
ts = np.linspace(0.0, 1.0, len(frame))
rel = syn.pose_mul(syn.planar_pose(0.6, 0.05, 0.04), np.array([0.004, -0.003, 0, np.sqrt(1 - 25e-6), 0, 0, 0.01]))
ext = np.array([0, 0, 0, 1.0, 0.3, 0.1, 0.9])
pre = K.PreSteps()


def med(fn, reps=200):
    for _ in range(10):
        fn()
    t = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        t.append(time.perf_counter() - t0)
    return round(float(np.median(t)) * 1e6, 1)


out = {"points": len(frame), "direct_upload": os.environ.get("KICP_DIRECT_UPLOAD", "0")}
out["upload_xyz_us"] = med(lambda: pre.upload(0, frame))
out["preprocess_deskew_us"] = med(lambda: pre.Preprocess(frame, ts, rel, ext, 100.0, 1.0, 1, dst=0))
out["preprocess_no_stamps_us"] = med(lambda: pre.Preprocess(frame, None, rel, ext, 100.0, 1.0, 0, dst=0))
n0 = pre.Preprocess(frame, ts, rel, ext, 100.0, 1.0, 1, dst=0)
out["downsample_half_voxel_us"] = med(lambda: pre.VoxelDownsample(0, cfg.voxel_size * 0.5, 1))
out["downsample_1p5_voxel_us"] = med(lambda: pre.VoxelDownsample(1, cfg.voxel_size * 1.5, 2))
out["survivors"] = [int(n0), int(pre.VoxelDownsample(0, cfg.voxel_size * 0.5, 1)), int(pre.VoxelDownsample(1, cfg.voxel_size * 1.5, 2))]
out["download_down_us"] = med(lambda: pre.download(1))
print(json.dumps(out))
