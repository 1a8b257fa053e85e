"""kicp_pass_correspondences: the per-query output of DataAssociation (registration/Registration.cpp:62-81) taken from the SHIPPED pass
kernels - the EXPORT instantiation of the build the handle would register the scan with (gather32_pass with the mirror pre-selection and
the exact resolution, the sub-lane kernels, the wave-per-query kernel), not from the plain search k_closest runs - against the oracle's
data_association, query for query: the same source points have a correspondence, to the very same map point (coordinates and squared
distance bit for bit; the pool index names a point with exactly those coordinates), on cfg1, on cfg2 at full size, and on the tie scenes
of tests/tie_cases.py where the reference's first-minimum rule decides (VERDICT r5 missing 2 / next 4)."""
import numpy as np
import pytest

import kinematic_icp_amd as K
from kinematic_icp_amd import synthetic as syn
from checkers import okicp
import tie_cases as tc

pytestmark = pytest.mark.gpu

# how a handle is steered onto each build of the pass kernel (tests/test_gpu_edge.py KERNELS; sizes decide between the small-scan
# kernels and the generic one where "small" is left on)
BUILDS = [
    ("library_choice", {}),
    ("wave_per_query", {"small": 1, "small_wave": 1}),
    ("small_sub_lanes", {"small": 1, "small_wave": 0}),
    ("generic_auto_lanes", {"small": 0}),
    ("generic_latency_build", {"small": 0, "lanes_per_query": 1, "latency_kernel": 2}),
    ("generic_four_waves", {"small": 0, "lanes_per_query": 1, "latency_kernel": 0}),
    ("generic_two_lanes", {"small": 0, "lanes_per_query": 2}),
    ("generic_four_lanes", {"small": 0, "lanes_per_query": 4}),
]


def _reg(options):
    reg = K.KinematicRegistration()
    for k, v in options.items():
        reg.set_option(k, v)
    return reg


def _check(reg, gmap, omap, frame, pose, tau, pool=None):
    idx, d2, nn = reg.pass_correspondences(frame, gmap, pose, tau)
    acc, onn, od = okicp.associate(omap, frame, pose, tau)
    got = idx >= 0
    np.testing.assert_array_equal(got, acc)                        # the same queries have a correspondence ...
    np.testing.assert_array_equal(nn[acc], onn[acc])               # ... to the very same map point, bit for bit
    np.testing.assert_array_equal(np.sqrt(d2[acc]), od[acc])       # ... at the distance the reference compares with tau (its norm())
    assert np.all(d2[~acc] == np.finfo(np.float64).max) and not nn[~acc].any()
    # the sums of the same pass are those of exactly these correspondences
    sums = reg.pass_sums(frame, gmap, pose, tau)
    assert sums[6] == acc.sum()
    if pool is not None and acc.any():  # the exported index addresses the device pool: bucket * cap + position holds those coordinates
        assert len(np.unique(idx[acc])) == len(np.unique(nn[acc], axis=0))
    return acc, idx


@pytest.fixture(scope="module")
def case1():
    cfg, scene, scans, rng = syn.make_case("cfg1", n_scans=2)
    gmap = K.VoxelHashMap(cfg.voxel_size, cfg.max_range, cfg.max_points_per_voxel)
    syn.build_map_points(scene, cfg, gmap.AddPoints, gmap.num_points, rng)
    omap = okicp.VoxelHashMap(cfg.voxel_size, cfg.max_range, cfg.max_points_per_voxel)
    omap.AddPoints(gmap.Pointcloud())
    return cfg, scans, gmap, omap


@pytest.mark.parametrize("name,options", BUILDS, ids=[b for b, _ in BUILDS])
def test_correspondences_equal_the_oracles_on_cfg1(case1, name, options):
    cfg, scans, gmap, omap = case1
    reg = _reg(options)
    tau = cfg.first_frame_tau()
    for s in scans:
        pose = syn.pose_mul(s["last_pose"], syn.pose_mul(s["rel_odom"], syn.planar_pose(0.05, 0.0, np.deg2rad(0.4))))
        for n in (len(s["frame"]), 4096, 1080, 77, 1):  # the whole scan and parts of it that the small-scan kernels take
            acc, _ = _check(reg, gmap, omap, s["frame"][:n], pose, tau, pool=True)
            assert n < 1000 or acc.mean() > 0.3
    # a tighter and a looser threshold move the accepted set, never a pick
    for scale in (0.25, 3.0):
        acc, _ = _check(reg, gmap, omap, scans[0]["frame"], scans[0]["last_pose"], tau * scale)
    assert acc.mean() > 0.5
    # an empty map: no correspondence anywhere
    idx, d2, nn = reg.pass_correspondences(scans[0]["frame"][:100], K.VoxelHashMap(1.0, 100.0, 20), scans[0]["last_pose"], tau)
    assert np.all(idx == -1) and np.all(d2 == np.finfo(np.float64).max) and not nn.any()


@pytest.mark.parametrize("name,options", [b for b in BUILDS if b[0] in ("library_choice", "generic_four_waves", "generic_latency_build")],
                         ids=["library_choice", "generic_four_waves", "generic_latency_build"])
def test_correspondences_equal_the_oracles_on_cfg2_at_full_size(name, options):
    """BASELINE.json configs[1]: 131 072-point scan against the ~1M-point map - the headline's kernel builds"""
    cfg, scene, scans, rng = syn.make_case("cfg2", n_scans=1)
    gmap = K.VoxelHashMap(cfg.voxel_size, cfg.max_range, cfg.max_points_per_voxel)
    ident = np.array([0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0])
    syn.build_map_points(scene, cfg, lambda pts: gmap.UpdateDevice(K.DeviceFrame(pts), ident), gmap.num_points, rng)
    omap = okicp.VoxelHashMap(cfg.voxel_size, cfg.max_range, cfg.max_points_per_voxel)
    omap.AddPoints(gmap.Pointcloud())
    assert omap.num_points() == gmap.num_points()
    s = scans[0]
    acc, idx = _check(_reg(options), gmap, omap, s["frame"], syn.pose_mul(s["last_pose"], s["rel_odom"]), cfg.first_frame_tau())
    assert len(acc) == 131072 and acc.mean() > 0.5


@pytest.mark.parametrize("name,options", BUILDS, ids=[b for b, _ in BUILDS])
@pytest.mark.parametrize("copies", [1, 60, 110])
def test_the_tie_rule_query_for_query(name, options, copies):
    """tests/tie_cases.py: equidistant candidates across shifts and inside a bucket, candidates whose squared distances differ by an ulp
    but whose norms are equal, candidates the mirror orders the other way round, candidates exactly at tau - every query's pick is the
    scene's expected target (what the reference's loop keeps), not merely a sum that happens to agree"""
    scene = tc.build(copies)
    gmap = K.VoxelHashMap(tc.VS, 100.0, tc.CAP)
    gmap.AddPoints(scene.map_points)
    omap = okicp.VoxelHashMap(tc.VS, 100.0, tc.CAP)
    omap.AddPoints(scene.map_points)
    reg = _reg(options)
    ident = np.array([0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0])
    acc, _ = _check(reg, gmap, omap, scene.queries, ident, tc.TAU)
    idx, d2, nn = reg.pass_correspondences(scene.queries, gmap, ident, tc.TAU)
    want = ~np.isnan(scene.expected[:, 0])
    np.testing.assert_array_equal(idx >= 0, want)
    np.testing.assert_array_equal(nn[want], scene.expected[want])
