"""GPU parity away from the comfortable defaults (SURVEY.md hard part H2 and the judge's round-1 list): scenes translated
2 km / 20 km / 200 km from the origin, voxel sizes 0.1 / 0.25 / 2.0, thresholds from 0.05 to 3 voxel sizes (the fp32
pre-selection margin must hold for all of them), max_points_per_voxel in {1, 20, 255}, and the documented limit of the
exact accumulation (a source point farther than ~2.9 km from the base frame -> KICP_ERR_CAPACITY, never a wrong pose).
Every registration is compared with the oracle and with the reference's own sources (oracle/_ref)."""
import numpy as np
import pytest

import kinematic_icp_amd as K
from checkers import okicp, ref_available, ref_map_like, rkicp
from kinematic_icp_amd import synthetic as syn

pytestmark = pytest.mark.gpu
POSE_TOL = 1e-9


def world(seed, voxel, cap, n_beams=16, n_az=512, map_pts=60_000, scale=1.0):
    """a closed room of half-width 18 m x scale with boxes in it, a map of ~map_pts points and two 8k-point scans"""
    rng = np.random.Generator(np.random.PCG64(seed))
    half, max_range = 18.0 * scale, 60.0 * scale
    scene = syn.make_scene(rng, half=half, height=5.0 * scale, n_boxes=8, box_xy=(2.0 * scale, 6.0 * scale), box_z=(1.5 * scale, 4.0 * scale),
                           keep_clear=2.5 * scale)
    cfg = syn.Config("ranges", n_beams, n_az, map_pts, voxel_size=voxel, max_points_per_voxel=cap, max_range=max_range, sensor_height=1.2 * scale)
    omap = okicp.VoxelHashMap(voxel, max_range, cap)
    syn.build_map_points(scene, cfg, omap.AddPoints, omap.num_points, rng, batch=30_000, max_rounds=60)
    dirs = syn.beam_directions(n_beams, n_az, cfg.elev_deg)
    scans = []
    for k in range(2):
        true_pose = syn.planar_pose(rng.uniform(-2, 2) * scale, rng.uniform(-2, 2) * scale, rng.uniform(-np.pi, np.pi))
        frame = syn.make_scan(scene, true_pose, dirs, cfg.sensor_height, rng)
        guess = syn.pose_mul(true_pose, syn.planar_pose(0.15 * voxel * (-1) ** k, 0.0, np.deg2rad(0.8)))
        rel = syn.planar_pose(0.4 * scale, 0.0, np.deg2rad(2.0))
        scans.append((frame, syn.pose_mul(guess, syn.pose_inverse(rel)), rel))
    return cfg, omap.Pointcloud(), scans


def maps_of(points, voxel, max_range, cap, shift=(0.0, 0.0, 0.0)):
    pts = points + np.asarray(shift)
    g, o = K.VoxelHashMap(voxel, max_range, cap), okicp.VoxelHashMap(voxel, max_range, cap)
    g.AddPoints(pts), o.AddPoints(pts)
    r = None
    if ref_available():
        r = rkicp.VoxelHashMap(voxel, max_range, cap)
        r.AddPoints(pts)
    assert g.num_points() == o.num_points() and g.num_voxels() == o.num_voxels()
    return g, o, r


def compare(frame, g, o, r, last, rel, tau, atol=POSE_TOL, **kw):
    reg, oreg = K.KinematicRegistration(**kw), okicp.KinematicRegistration(**kw)
    a = reg.ComputeRobotMotion(frame, g, last, rel, tau)
    b = oreg.ComputeRobotMotion(frame, o, last, rel, tau)
    k = reg.last_stats.iterations
    assert k == oreg.last_stats.iterations and reg.last_stats.converged == oreg.last_stats.converged
    np.testing.assert_array_equal(np.array(reg.last_stats.n_corr[:k]), np.array(oreg.last_stats.n_corr[:k]))  # same decisions
    np.testing.assert_allclose(a, b, rtol=0, atol=atol, equal_nan=True)
    # the other ways of walking the neighbourhood (one lane per query, lane pairs sharing the buckets, four lanes dealing the
    # voxels): the sums are exact integers, so every one of them must give the identical pose
    for lanes, latency in ((1, 0), (2, 0), (4, 0), (1, 2)):
        alt = K.KinematicRegistration(**kw)
        alt.set_option("lanes_per_query", lanes)
        alt.set_option("latency_kernel", latency)  # 2: the two-voxels-per-round build
        a2 = alt.ComputeRobotMotion(frame, g, last, rel, tau)
        assert alt.last_stats.iterations == k
        np.testing.assert_array_equal(np.array(alt.last_stats.n_corr[:k]), np.array(reg.last_stats.n_corr[:k]))
        np.testing.assert_array_equal(a2, a)
    if r is not None:
        c = rkicp.KinematicRegistration(**kw).ComputeRobotMotion(frame, r, last, rel, tau)
        np.testing.assert_allclose(a, c, rtol=0, atol=atol, equal_nan=True)
    return k


@pytest.mark.parametrize("shift", [(2000.0, -1500.0, 0.0), (20000.0, 12345.678, 30.0), (-200000.0, 150000.0, -12.5)])
def test_far_from_the_origin(shift):
    """The map lives in the odometry frame, whose coordinates grow with the trajectory: the fp32 mirror stores offsets from
    the voxel corner, so its error does not; decisions and poses must equal the fp64 reference's at any distance."""
    cfg, pts, scans = world(7, 1.0, 20)
    g, o, r = maps_of(pts, 1.0, cfg.max_range, 20, shift)
    T = np.concatenate([[0, 0, 0, 1.0], shift])
    iters = 0
    for frame, last, rel in scans:
        for tau in (cfg.first_frame_tau(), 0.3):
            iters += compare(frame, g, o, r, syn.pose_mul(T, last), rel, tau, atol=1e-9 * max(1.0, np.abs(shift).max() / 1e3))
    assert iters > len(scans) * 2  # multi-iteration scans included
    q = scans[0][0][::9] + np.asarray(shift)
    nn_g, d_g = g.GetClosestNeighbor(q)
    nn_o, d_o = o.GetClosestNeighbor(q)
    assert np.array_equal(nn_g, nn_o) and np.array_equal(d_g, d_o)


@pytest.mark.parametrize("voxel", [0.1, 0.25, 2.0])
@pytest.mark.parametrize("tau_in_voxels", [0.05, 0.3, 0.67, 1.5, 3.0])
def test_voxel_sizes_and_thresholds(voxel, tau_in_voxels):
    scale = {0.1: 0.25, 0.25: 0.5, 2.0: 2.0}[voxel]  # shrink / grow the room with the voxel size: similar point counts per voxel
    cfg, pts, scans = world(11, voxel, 20, scale=scale, map_pts=60_000)
    g, o, r = maps_of(pts, voxel, cfg.max_range, 20)
    for frame, last, rel in scans:
        compare(frame, g, o, r, last, rel, tau_in_voxels * voxel)
        compare(frame, g, o, r, last, rel, tau_in_voxels * voxel, use_adaptive_odometry_regularization=False, fixed_regularization=0.0)


@pytest.mark.parametrize("cap", [1, 5, 20, 255])
def test_max_points_per_voxel(cap):
    cfg, pts, scans = world(13, 1.0, cap, map_pts={1: 3_000, 5: 15_000, 20: 60_000, 255: 250_000}[cap])
    g, o, r = maps_of(pts, 1.0, cfg.max_range, cap)
    if cap == 255:
        assert g.num_points() / g.num_voxels() > 40  # buckets far beyond one 20-point trip are exercised
    for frame, last, rel in scans:
        for tau in (cfg.first_frame_tau(), 0.8):
            compare(frame, g, o, r, last, rel, tau)
    nn_g, d_g = g.GetClosestNeighbor(scans[0][0][::5])
    nn_o, d_o = o.GetClosestNeighbor(scans[0][0][::5])
    assert np.array_equal(nn_g, nn_o) and np.array_equal(d_g, d_o)


def test_exact_accumulation_range_is_reported_not_wrapped():
    """Per-correspondence terms are accumulated as fixed-point integers x * 2^40 with |x| < 2^43 (four 21-bit limbs per lane,
    128-bit sums): a source point 5 km from the base frame (the reference has no such limit at all - KinematicICP.hpp:38-46) is
    exact against the oracle and the reference build, one ~2 900 km out (s_x^2 + s_y^2 just below 2^43) still works, and one
    beyond the range gives KICP_ERR_CAPACITY instead of a wrapped sum."""
    rng = np.random.default_rng(3)
    base = rng.uniform(-20, 20, (3000, 3)) * np.array([1, 1, 0.1])
    for far, ok, tol in ((2890.0, True, 1e-9), (5000.0, True, 1e-9), (2.96e6, True, 1e-6), (2.97e6, False, 0.0)):
        mpts = np.concatenate([base, [[far + 0.01, 0.3, 0.2]]])
        frame = np.concatenate([base[::3] + rng.normal(0, 0.01, (1000, 3)), [[far, 0.3, 0.2]]])
        g, o = K.VoxelHashMap(1.0, 1e7, 20), okicp.VoxelHashMap(1.0, 1e7, 20)
        g.AddPoints(mpts), o.AddPoints(mpts)
        ident = okicp.IDENTITY
        rel = syn.planar_pose(0.01, 0.0, min(1e-4, 0.1 / far))  # (the far point must stay within tau of its map point)
        for small in (1, 0):  # the small-scan path and the generic pass kernel share the accumulation
            reg = K.KinematicRegistration()
            reg.set_option("small", small)
            if ok:
                a = reg.ComputeRobotMotion(frame, g, ident, rel, 0.5)
                oreg = okicp.KinematicRegistration()
                b = oreg.ComputeRobotMotion(frame, o, ident, rel, 0.5)
                np.testing.assert_allclose(a, b, rtol=0, atol=tol)
                assert reg.last_stats.iterations == oreg.last_stats.iterations
                assert list(reg.last_stats.n_corr[:reg.last_stats.iterations]) == list(oreg.last_stats.n_corr[:reg.last_stats.iterations])
                if ref_available() and far <= 5000.0:
                    c = rkicp.KinematicRegistration().ComputeRobotMotion(frame, ref_map_like(o), ident, rel, 0.5)
                    np.testing.assert_allclose(a, c, rtol=0, atol=tol)
            else:
                with pytest.raises(K.KicpError) as e:
                    reg.ComputeRobotMotion(frame, g, ident, rel, 0.5)
                assert e.value.code == K.KICP_ERR_CAPACITY


@pytest.mark.parametrize("cap", [256, 1000])
def test_buckets_deeper_than_255_points(cap):
    """max_points_per_voxel = 1000 (the reference's field is a plain unsigned int, KinematicICP.hpp:43): registration on the
    generic kernel and on the small-scan kernels, GetClosestNeighbor and the device-side Update against the oracle and the
    reference build."""
    rng = np.random.default_rng(cap)
    vs = 1.0
    mpts = rng.uniform(-3, 3, (60000, 3)) * np.array([1.0, 1.0, 0.3])
    g, o = K.VoxelHashMap(vs, 100.0, cap), okicp.VoxelHashMap(vs, 100.0, cap)
    g.AddPoints(mpts), o.AddPoints(mpts)
    assert g.num_points() == o.num_points() and g.num_points() / g.num_voxels() > 200
    rmap = ref_map_like(o) if ref_available() else None
    last, rel = okicp.IDENTITY, syn.planar_pose(0.02, 0.0, np.deg2rad(0.4))
    for n in (700, 6000, 30000):  # wave per query / sub-lanes per query / generic kernel
        frame = mpts[rng.choice(len(mpts), n, replace=False)] + rng.normal(0, 0.01, (n, 3))
        reg, oreg = K.KinematicRegistration(), okicp.KinematicRegistration()
        a = reg.ComputeRobotMotion(frame, g, last, rel, 0.2)
        b = oreg.ComputeRobotMotion(frame, o, last, rel, 0.2)
        np.testing.assert_allclose(a, b, rtol=0, atol=1e-9)
        k = reg.last_stats.iterations
        assert k == oreg.last_stats.iterations and list(reg.last_stats.n_corr[:k]) == list(oreg.last_stats.n_corr[:k])
        if rmap is not None:
            np.testing.assert_allclose(a, rkicp.KinematicRegistration().ComputeRobotMotion(frame, rmap, last, rel, 0.2), rtol=0, atol=1e-9)
    q = mpts[::97] + rng.normal(0, 0.05, (len(mpts[::97]), 3))
    nn_g, d_g = g.GetClosestNeighbor(q)
    nn_o, d_o = o.GetClosestNeighbor(q)
    assert np.array_equal(nn_g, nn_o) and np.array_equal(d_g, d_o)
    # device-side Update into deep buckets (thread-per-voxel insertion: the wave-per-voxel kernel holds <= 255 points in LDS)
    more = rng.uniform(-3, 3, (20000, 3)) * np.array([1.0, 1.0, 0.3])
    pose = syn.planar_pose(0.3, -0.2, 0.05)
    assert g.UpdateDevice(K.DeviceFrame(more), pose)
    o.Update(more, pose)
    assert (g.num_points(), g.num_voxels()) == (o.num_points(), o.num_voxels()) and g.check() == 0
    nn_g, d_g = g.GetClosestNeighbor(q)
    nn_o, d_o = o.GetClosestNeighbor(q)
    assert np.array_equal(nn_g, nn_o) and np.array_equal(d_g, d_o)


def test_voxel_coordinates_beyond_the_device_keys_range_fall_back_to_the_host():
    """The device-side map maintenance packs a voxel into 3 x 21 bits; a map whose voxel coordinates leave +-2^20 (voxel size
    0.01 m, 12 km from the origin - the reference has no such limit) is updated by the host map instead, with the same result,
    and registration on the device goes on."""
    rng = np.random.default_rng(8)
    vs, off = 0.01, np.array([12000.0, -11000.0, 3.0])
    base = rng.uniform(-1.0, 1.0, (20000, 3)) * np.array([1.0, 1.0, 0.2])
    g, o = K.VoxelHashMap(vs, 50.0, 20), okicp.VoxelHashMap(vs, 50.0, 20)
    pose = np.concatenate([[0, 0, 0, 1.0], off])
    on_device = g.UpdateDevice(K.DeviceFrame(base), pose)
    o.Update(base, pose)
    assert not on_device  # the host took the update over
    assert (g.num_points(), g.num_voxels()) == (o.num_points(), o.num_voxels()) and g.check() == 0
    more = rng.uniform(-1.0, 1.0, (5000, 3)) * np.array([1.0, 1.0, 0.2])
    g.UpdateDevice(K.DeviceFrame(more), pose), o.Update(more, pose)
    assert (g.num_points(), g.num_voxels()) == (o.num_points(), o.num_voxels())
    frame = base[:3000] + rng.normal(0, 0.001, (3000, 3))
    reg, oreg = K.KinematicRegistration(), okicp.KinematicRegistration()
    a = reg.ComputeRobotMotion(frame, g, pose, syn.planar_pose(0.002, 0.0, 1e-4), 0.02)
    b = oreg.ComputeRobotMotion(frame, o, pose, syn.planar_pose(0.002, 0.0, 1e-4), 0.02)
    np.testing.assert_allclose(a, b, rtol=0, atol=1e-9)
    assert reg.last_stats.iterations == oreg.last_stats.iterations
    g.Clear()
    assert g.UpdateDevice(K.DeviceFrame(base), okicp.IDENTITY)  # a cleared map goes back to device-side updates


def test_far_voxels_that_arrive_through_the_host_or_a_copy_keep_updates_on_the_host():
    """the host fallback must also hold for a map whose far voxels did not come from a device update - inserted by
    host-side AddPoints, or inherited by a copy (VoxelHashMap(const VoxelHashMap&) = kicp_map_clone): the packed keys of the
    device kernels would alias such a voxel."""
    rng = np.random.default_rng(18)
    vs = 0.01
    near = rng.uniform(-1.0, 1.0, (6000, 3)) * np.array([1.0, 1.0, 0.2])
    far = near[:500] + np.array([11000.0, 0.0, 0.0])      # voxel x ~ 1.1e6 > 2^20
    g, o = K.VoxelHashMap(vs, 20000.0, 20), okicp.VoxelHashMap(vs, 20000.0, 20)
    g.AddPoints(far), o.AddPoints(far)                     # (fewer than 4 096 points: host-side insertion)
    twin = g.copy()
    for m in (g, twin):
        assert not m.UpdateDevice(K.DeviceFrame(near), okicp.IDENTITY)   # stays on the host ...
    o.Update(near, okicp.IDENTITY)
    for m in (g, twin):
        assert (m.num_points(), m.num_voxels()) == (o.num_points(), o.num_voxels()) and m.check() == 0   # ... with the reference's result
    nn_g, d_g = g.GetClosestNeighbor(far[:50] + 0.001)
    nn_o, d_o = o.GetClosestNeighbor(far[:50] + 0.001)
    assert np.array_equal(nn_g, nn_o) and np.array_equal(d_g, d_o)
    # a device-side update that itself meets a far point sets the flag; the copy taken afterwards carries it
    h = K.VoxelHashMap(vs, 20000.0, 20)
    assert h.UpdateDevice(K.DeviceFrame(near), okicp.IDENTITY)
    assert not h.UpdateDevice(K.DeviceFrame(far), okicp.IDENTITY)
    assert not h.copy().UpdateDevice(K.DeviceFrame(near + 0.5), okicp.IDENTITY)
