"""Per-frame cost of VoxelHashMap::Update(points, pose) on the device against a LARGE local map (the cfg2 map: ~1M points,
75k voxels, 2M-slot table) - what a long drive with max_range 100 m builds up - next to the oracle on one host core.
The frame is what the pipeline feeds: the scan downsampled at 0.5 voxel sizes (~8k points), moved along a slow trajectory."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kinematic_icp_amd as K
from kinematic_icp_amd import synthetic as syn
from oracle import okicp

cfg, scene, scans, rng = syn.make_case("cfg2", n_scans=1)
gmap = K.VoxelHashMap(cfg.voxel_size, cfg.max_range, cfg.max_points_per_voxel)
ident = np.array([0, 0, 0, 1.0, 0, 0, 0])
syn.build_map_points(scene, cfg, lambda pts: gmap.UpdateDevice(K.DeviceFrame(pts), ident), gmap.num_points, rng)
omap = okicp.VoxelHashMap(cfg.voxel_size, cfg.max_range, cfg.max_points_per_voxel)
omap.AddPoints(gmap.Pointcloud())
down = okicp.voxel_downsample(scans[0]["frame"], 0.5 * cfg.voxel_size)
df = K.DeviceFrame(down)
print("map %d points / %d voxels, frame %d points" % (gmap.num_points(), gmap.num_voxels(), len(down)))
poses = [syn.planar_pose(0.05 * k, 0.01 * k, 0.002 * k) for k in range(60)]
t_gpu, t_cpu = [], []
for k, pose in enumerate(poses):
    t0 = time.perf_counter()
    assert gmap.UpdateDevice(df, pose)
    t_gpu.append(time.perf_counter() - t0)
    if k < 12:
        t0 = time.perf_counter()
        omap.Update(down, pose)
        t_cpu.append(time.perf_counter() - t0)
        assert (gmap.num_points(), gmap.num_voxels()) == (omap.num_points(), omap.num_voxels()), k
print("device Update: median %.1f us per frame (first %.1f us) | oracle (1 core): median %.1f us | map now %d points" %
      (np.median(t_gpu[5:]) * 1e6, t_gpu[0] * 1e6, np.median(t_cpu[2:]) * 1e6, gmap.num_points()))
