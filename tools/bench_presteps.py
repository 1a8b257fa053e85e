"""Throughput of the GPU pre-steps vs the CPU oracle on a cfg2-sized raw scan (131 072 points)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kinematic_icp_amd as K
from kinematic_icp_amd import synthetic as syn
from oracle import okicp
cfg, scene, scans, rng = syn.make_case("cfg2", n_scans=1)
frame = scans[0]["frame"]
ts = np.linspace(0, 1, len(frame))
rel = syn.planar_pose(0.6, 0.05, 0.04)
ext = syn.planar_pose(0.3, 0.0, 0.02, z=0.5)
pre = K.PreSteps()
def gpu_all(deskew):
    n0 = pre.Preprocess(frame, ts, rel, ext, 100.0, 0.0, deskew, dst=0)
    n1 = pre.VoxelDownsample(0, 0.5 * cfg.voxel_size, 1)
    n2 = pre.VoxelDownsample(1, 1.5 * cfg.voxel_size, 2)
    return n0, n1, n2
for deskew in (0, 1):
    for _ in range(20): gpu_all(deskew)
    t0 = time.perf_counter()
    for _ in range(100): counts = gpu_all(deskew)
    g = (time.perf_counter() - t0) / 100
    t0 = time.perf_counter()
    for _ in range(3):
        a = okicp.se3_act(ext, okicp.preprocess(frame, ts, rel, 100.0, 0.0, bool(deskew)))
        b = okicp.voxel_downsample(a, 0.5 * cfg.voxel_size)
        c = okicp.voxel_downsample(b, 1.5 * cfg.voxel_size)
    c_t = (time.perf_counter() - t0) / 3
    print("deskew %d: GPU %.1f us per frame incl. 4.2 MB upload (%s survivors) | CPU oracle (1 thread) %.2f ms | x%.0f" % (deskew, g * 1e6, counts, c_t * 1e3, c_t / g), flush=True)
# downsample alone on a resident buffer
pre.Preprocess(frame, None, rel, ext, 100.0, 0.0, 0, dst=0)
for _ in range(20): pre.VoxelDownsample(0, 0.5, 1)
t0 = time.perf_counter()
for _ in range(200): pre.VoxelDownsample(0, 0.5, 1)
print("VoxelDownsample(131072 resident points, 0.5 m): %.1f us" % ((time.perf_counter() - t0) / 200 * 1e6))
