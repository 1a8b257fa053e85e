"""BASELINE.json configurations at their stated sizes on the GPU: cfg2 (131 072-pt scan vs ~1M-pt map, the headline),
cfg4 (1 080-pt 2-D scan vs 50k-pt map, voxel 0.2), cfg5 (500k-pt scan vs 10M-pt map, voxel 0.1 - the 10M-point map is
built ON THE DEVICE through the map's Update(points, pose) path, which takes seconds instead of the host map's minute).
Direct parity against the oracle and the reference build (they finish in seconds at these sizes) plus size-independent properties:
permutation invariance (bitwise, exact accumulation), shard-sum exactness, determinism."""
import os

import numpy as np
import pytest

import kinematic_icp_amd as K
from checkers import okicp, ref_available, ref_map_like
from kinematic_icp_amd import sharding as sh
from kinematic_icp_amd import synthetic as syn

pytestmark = pytest.mark.gpu
QUICK = os.environ.get("KICP_QUICK", "0") == "1"  # developer switch: quarter-size cfg5


def build(name, n_scans=2, **override):
    if override:
        base = syn.CONFIGS[name]
        syn.CONFIGS[name + "_lite"] = syn.Config(**{**base.__dict__, **override, "name": name + "_lite"})
        name = name + "_lite"
    cfg, scene, scans, rng = syn.make_case(name, n_scans=n_scans)
    gmap = K.VoxelHashMap(cfg.voxel_size, cfg.max_range, cfg.max_points_per_voxel)
    syn.build_map_points(scene, cfg, gmap.AddPoints, gmap.num_points, rng)
    omap = okicp.VoxelHashMap(cfg.voxel_size, cfg.max_range, cfg.max_points_per_voxel)
    omap.AddPoints(gmap.Pointcloud())
    return cfg, scans, gmap, omap


def check_case(cfg, scans, gmap, omap, extra_yaw_deg=0.0, rmap=None):
    reg, oreg = K.KinematicRegistration(), okicp.KinematicRegistration(max_num_threads=0)
    tau = cfg.first_frame_tau()
    for s in scans:
        rel = syn.pose_mul(s["rel_odom"], syn.planar_pose(0.0, 0.0, np.deg2rad(extra_yaw_deg)))
        a = reg.ComputeRobotMotion(s["frame"], gmap, s["last_pose"], rel, tau)
        b = oreg.ComputeRobotMotion(s["frame"], omap, s["last_pose"], rel, tau)
        k = reg.last_stats.iterations
        assert k == oreg.last_stats.iterations and reg.last_stats.converged == oreg.last_stats.converged
        np.testing.assert_allclose(a, b, rtol=0, atol=1e-9)
        if rmap is not None:  # the reference's own Registration.cpp (all host cores through the TBB stand-in)
            from checkers import rkicp
            c = rkicp.KinematicRegistration(max_num_threads=0).ComputeRobotMotion(s["frame"], rmap, s["last_pose"], rel, tau)
            np.testing.assert_allclose(a, c, rtol=0, atol=1e-9)
        np.testing.assert_array_equal(np.array(reg.last_stats.n_corr[:k]), np.array(oreg.last_stats.n_corr[:k]))
        # properties that do not need the oracle: order of the scan does not matter (bitwise), repeat runs are bitwise equal,
        # the limb words of disjoint shards add up exactly
        perm = np.random.default_rng(1).permutation(len(s["frame"]))
        assert np.array_equal(reg.ComputeRobotMotion(s["frame"][perm], gmap, s["last_pose"], rel, tau), a)
        assert np.array_equal(reg.ComputeRobotMotion(K.DeviceFrame(s["frame"]), gmap, s["last_pose"], rel, tau), a)
        guess = syn.pose_mul(s["last_pose"], rel)
        full = reg.pass_words(s["frame"], gmap, guess, tau)
        parts = np.sum([reg.pass_words(s["frame"][slice(*sh.shard_bounds(len(s["frame"]), 8, r))], gmap, guess, tau) for r in range(8)], 0)
        assert [sh.from_limbs(parts[3 * i:3 * i + 3]) for i in range(7)] == [sh.from_limbs(full[3 * i:3 * i + 3]) for i in range(7)]
    return reg


def test_cfg2_full_size():
    cfg, scans, gmap, omap = build("cfg2")
    assert scans[0]["frame"].shape == (131072, 3) and gmap.num_points() > 950_000
    rmap = ref_map_like(omap) if ref_available() else None
    check_case(cfg, scans, gmap, omap, rmap=rmap)
    check_case(cfg, scans[:1], gmap, omap, extra_yaw_deg=1.5, rmap=rmap)  # a bad initial guess: several iterations


def test_cfg4_small_scan():
    cfg, scans, gmap, omap = build("cfg4", n_scans=3)
    assert scans[0]["frame"].shape == (1080, 3) and 45_000 < gmap.num_points() < 56_000
    rmap = ref_map_like(omap) if ref_available() else None
    check_case(cfg, scans, gmap, omap, rmap=rmap)
    check_case(cfg, scans[:1], gmap, omap, extra_yaw_deg=2.0, rmap=rmap)


def build_on_device(name, n_scans=1, **override):
    """Like build(), but the map grows through VoxelHashMap::Update(points, pose) on the GPU (identity pose; nothing is
    farther than max_range, so nothing is pruned) - the same map the host path builds (tests/test_gpu_mapdev.py)."""
    if override:
        base = syn.CONFIGS[name]
        syn.CONFIGS[name + "_lite"] = syn.Config(**{**base.__dict__, **override, "name": name + "_lite"})
        name = name + "_lite"
    cfg, scene, scans, rng = syn.make_case(name, n_scans=n_scans)
    gmap = K.VoxelHashMap(cfg.voxel_size, cfg.max_range, cfg.max_points_per_voxel)
    on_device = []
    syn.build_map_points(scene, cfg, lambda pts: on_device.append(gmap.UpdateDevice(K.DeviceFrame(pts), okicp.IDENTITY)), gmap.num_points, rng)
    assert on_device and all(on_device)
    omap = okicp.VoxelHashMap(cfg.voxel_size, cfg.max_range, cfg.max_points_per_voxel)
    omap.AddPoints(gmap.Pointcloud())
    assert (omap.num_points(), omap.num_voxels()) == (gmap.num_points(), gmap.num_voxels())
    return cfg, scans, gmap, omap


def test_cfg5_dense_full_size():
    if QUICK:
        cfg, scans, gmap, omap = build_on_device("cfg5", n_az=1000, map_points=2_500_000)
        assert scans[0]["frame"].shape == (125000, 3)
    else:
        cfg, scans, gmap, omap = build_on_device("cfg5")
        assert scans[0]["frame"].shape == (500000, 3) and gmap.num_points() > 9_500_000
    rmap = ref_map_like(omap) if ref_available() else None
    check_case(cfg, scans, gmap, omap, rmap=rmap)
    check_case(cfg, scans, gmap, omap, extra_yaw_deg=0.3, rmap=rmap)  # several iterations at this voxel size
