"""The reference's edge cases, sent THROUGH THE HIP PATH (round 3 pinned them on the CPU checkers only, tests/test_ref.py):

  KAT-0  empty map                      -> the prediction last * delta                       Registration.cpp:156-157
  KAT-1  source == map points           -> dx == 0 exactly, theta == 0 drops the translation Registration.cpp:159-167, beta = 1 / DBL_MIN :56-58
  KAT-1' the same under a pose          -> residuals of rounding size only
  KAT-2  small shift, fixed beta = 0    -> the shift is recovered                            Registration.cpp:119-125
  KAT-3  small shift, adaptive beta     -> odometry is trusted                               Registration.cpp:171-177
  max_num_iterations = 0                -> the loop body never runs                          Registration.cpp:179,189
  a tau that accepts exactly ONE correspondence, one that accepts none (NaN pose, runs on to max_num_iterations)  :75, :179-187

on every pass kernel the library can pick - one wave per query, sub-lanes per query, the generic thread-per-query kernel in its
four-waves and latency-oriented builds and with 2 / 4 sub-lanes, the plain fp64 gather, the generic kernel resident across a
call's passes - through kicp_register (host frame), kicp_register_f32, kicp_register_device and kicp_register_device_batch,
against the oracle and the reference's own Registration.cpp (oracle/_ref)."""
import numpy as np
import pytest

import kinematic_icp_amd as K
from checkers import okicp, ref_available, ref_map_like, rkicp
from kinematic_icp_amd import synthetic as syn

pytestmark = pytest.mark.gpu
I = okicp.IDENTITY
DBL_MIN = np.finfo(np.float64).tiny

# (name, options): how a handle is steered onto one particular pass kernel
KERNELS = [
    ("default", {}),                                                        # <= 4 096 points: one wave per query, resident
    ("sub_lanes", {"small_wave": 0}),                                       # k_pass_small
    ("small_one_launch_per_pass", {"small_resident": 0}),
    ("generic_auto_lanes", {"small": 0}),                                   # k_pass_gather32, 4 / 2 / 1 sub-lanes by scan size
    ("generic_latency_build", {"small": 0, "lanes_per_query": 1}),          # <.., LAT>
    ("generic_four_waves", {"small": 0, "lanes_per_query": 1, "latency_kernel": 0}),
    ("generic_two_lanes", {"small": 0, "lanes_per_query": 2}),
    ("generic_four_lanes", {"small": 0, "lanes_per_query": 4}),
    ("generic_hip_launch", {"small": 0, "aql": 0}),
    ("generic_stream_sync", {"small": 0, "wait": 1}),
]


def _reg(options, **cfg):
    reg = K.KinematicRegistration(**cfg)
    for k, v in options.items():
        reg.set_option(k, v)
    return reg


def _maps(points, voxel=1.0, cap=20):
    g = K.VoxelHashMap(voxel, 100.0, cap)
    g.AddPoints(points)
    o = okicp.VoxelHashMap(voxel, 100.0, cap)
    o.AddPoints(points)
    assert g.num_points() == o.num_points()
    return g, o, (ref_map_like(o) if ref_available() else None)


def _same(a, b, how):
    if how == "bits":      # results that involve no rounding at all: the prediction, the identity
        assert np.array_equal(a, b), (a, b)
    elif how == "nan":     # 0 / 0: both poses are NaN (which components carry the NaN is the solver's business)
        assert np.isnan(a).any() and np.isnan(b).any(), (a, b)
    else:                  # demanded 1e-4, asserted 1e-9 (the HIP path rounds each term once to 2^-40, the checkers sum in fp64)
        np.testing.assert_allclose(a, b, rtol=0, atol=1e-9)


def _check(reg, src, maps, last, rel, tau, cfg, how="close", via="host"):
    """one registration through the HIP path against the oracle (and the reference build): pose, iteration count, stop reason,
    per-pass correspondence counts, beta; returns (pose, oracle stats)"""
    g, o, r = maps
    if via == "device":
        a = reg.ComputeRobotMotion(K.DeviceFrame(src, device=0), g, last, rel, tau)
    elif via == "f32":
        a = reg.ComputeRobotMotion(src.astype(np.float32), g, last, rel, tau)
        src = src.astype(np.float32).astype(np.float64)
    elif via == "batch":
        batch = reg.prepare_batch([K.DeviceFrame(src, device=0)] * 9, [last] * 9, [rel] * 9)  # (eight and more: the kernel stays across the scans)
        poses = reg.ComputeRobotMotionBatch(batch, g, tau)
        assert all(np.array_equal(poses[0], poses[k], equal_nan=True) for k in range(1, 9))
        a = poses[0].copy()
    else:
        a = reg.ComputeRobotMotion(src, g, last, rel, tau)
    oreg = okicp.KinematicRegistration(**cfg)
    b = oreg.ComputeRobotMotion(src, o, last, rel, tau)
    _same(a, b, how)
    if via != "batch":
        k = oreg.last_stats.iterations
        assert reg.last_stats.iterations == k and reg.last_stats.converged == oreg.last_stats.converged
        kk = min(k, 32)
        np.testing.assert_array_equal(np.array(reg.last_stats.n_corr[:kk]), np.array(oreg.last_stats.n_corr[:kk]))
        if kk and np.isfinite(oreg.last_stats.beta) and cfg["use_adaptive_odometry_regularization"]:
            # beta = 1 / (mean squared residual + DBL_MIN), Registration.cpp:56-58.  The HIP path carries every squared residual at a
            # resolution of 2^-40 m^2, so what can be compared is the mean squared residual itself: residuals of rounding size (1e-16 m)
            # give beta = 1 / DBL_MIN there and ~1e32 in fp64 - either way a displacement weight beyond anything JTJ / N (~1) can answer
            np.testing.assert_allclose(1.0 / reg.last_stats.beta, 1.0 / oreg.last_stats.beta, rtol=1e-9, atol=2.0 ** -40)
    else:
        assert [int(x) for x in batch.iterations] == [oreg.last_stats.iterations] * 9
    if r is not None:
        c = rkicp.KinematicRegistration(**cfg).ComputeRobotMotion(src, r, last, rel, tau)
        _same(a, c, how)
    return a, oreg.last_stats


CFG = dict(max_num_iteration=10, convergence_criterion=1e-3, max_num_threads=1, use_adaptive_odometry_regularization=True,
           fixed_regularization=0.0)


@pytest.fixture(scope="module")
def world():
    m1 = np.random.default_rng(4).uniform(-10, 10, (800, 3))       # KAT-1 of tests/test_ref.py
    m2 = np.random.default_rng(6).uniform(-10, 10, (1500, 3))      # KAT-2 / KAT-3
    return _maps(m1), _maps(m2)


@pytest.mark.parametrize("name,options", KERNELS, ids=[k for k, _ in KERNELS])
@pytest.mark.parametrize("via", ["host", "f32", "device", "batch"])
def test_known_answers_through_the_hip_path(world, name, options, via):
    maps1, maps2 = world
    last, rel = syn.planar_pose(1, 2, 0.3), syn.planar_pose(0.5, 0, 0.1)
    # KAT-0: empty map -> prediction, whatever the frame
    reg = _reg(options, **CFG)
    empty = K.VoxelHashMap(1.0, 100.0, 20)
    p = reg.ComputeRobotMotion(np.zeros((10, 3)), empty, last, rel, 1.0)
    np.testing.assert_array_equal(p, okicp.se3_mul(last, rel))
    assert reg.last_stats.empty_map == 1
    # KAT-1: source == the map's own points at the identity: every residual is exactly zero -> dx == 0 -> theta == 0.0 -> the
    # motion model's quirk returns the identity; beta = 1 / (0 + DBL_MIN)
    src = maps1[1].Pointcloud()
    if via != "f32":  # (float32 cannot carry the map's doubles: KAT-1 is a float64 case)
        p, st = _check(reg, src, maps1, I, I, 0.5, CFG, how="bits", via=via)
        np.testing.assert_array_equal(p, I)
        if via != "batch":
            assert st.iterations == 1 and reg.last_stats.n_corr[0] == len(src)
            assert reg.last_stats.beta == 1.0 / DBL_MIN and list(reg.last_stats.dx[0]) == [0.0, 0.0]
            assert all(s == 0.0 for s in reg.last_stats.sums[0][3:6])  # JTr and the residual sum: exactly zero, not 2^-40 dust
        # KAT-1': the same cloud seen from a pose: T * (T^-1 * m) differs from m by rounding only
        T = syn.planar_pose(0.7, -0.4, 0.25)
        src_t = okicp.se3_act(okicp.se3_inverse(T), src)
        _check(reg, src_t, maps1, T, I, 0.5, CFG, via=via)
    # KAT-2 / KAT-3: a 2 cm shift along x, recovered without regularisation, suppressed by the adaptive one
    d = 0.02
    src = maps2[1].Pointcloud() - np.array([d, 0, 0])
    cfg2 = dict(CFG, use_adaptive_odometry_regularization=False)
    p, _ = _check(_reg(options, **cfg2), src, maps2, I, I, 0.5, cfg2, via=via)
    np.testing.assert_allclose(p[4], d, atol=1e-5 if via == "f32" else 1e-6)
    p, _ = _check(_reg(options, **CFG), src, maps2, I, I, 0.5, CFG, via=via)
    assert abs(p[4]) < d * 1e-2
    # max_num_iterations = 0: the loop body never runs, the prediction comes back untouched
    cfg0 = dict(CFG, max_num_iteration=0)
    p, st = _check(_reg(options, **cfg0), src, maps2, last, rel, 0.5, cfg0, how="bits", via=via)
    np.testing.assert_array_equal(p, okicp.se3_mul(last, rel))
    # no correspondence at all: 0 / 0 -> NaN pose, warning code, the reference runs on to max_num_iterations
    reg = _reg(options, **CFG)
    p, st = _check(reg, np.full((50, 3), 500.0), maps2, I, I, 0.5, CFG, how="nan", via=via)
    assert np.isnan(p).any()
    if via != "batch":
        assert reg.last_status == K.KICP_WARN_NO_CORRESPONDENCES and reg.last_stats.iterations == 10


@pytest.mark.parametrize("name,options", KERNELS, ids=[k for k, _ in KERNELS])
def test_a_threshold_that_accepts_exactly_one_correspondence(world, name, options):
    """49 source points nowhere near the map and one 3 cm from a map point; tau between: N = 1 in every pass"""
    maps = world[1]
    target = maps[1].Pointcloud()[17]
    src = np.concatenate([np.full((20, 3), 300.0) + np.arange(20)[:, None], (target + np.array([0.03, 0.0, 0.0]))[None], np.full((29, 3), -400.0)])
    for cfg in (CFG, dict(CFG, use_adaptive_odometry_regularization=False, fixed_regularization=0.3)):
        reg = _reg(options, **cfg)
        p, st = _check(reg, src, maps, I, I, 0.2, cfg)
        assert st.n_corr[0] == 1.0 and np.isfinite(p).all()
        _check(reg, src, maps, syn.planar_pose(0.01, 0.0, 0.002), syn.planar_pose(-0.005, 0.0, 0.001), 0.2, cfg, via="device")


def _big_world(n_map=60000, n_src=12000, seed=11):
    rng = np.random.default_rng(seed)
    mp = np.concatenate([rng.uniform(-40, 40, (n_map, 2)), rng.uniform(0, 0.05, (n_map, 1))], 1)  # a noisy ground plane: full voxels
    maps = _maps(mp)
    src = maps[1].Pointcloud()[:n_src]
    return maps, src


def test_known_answers_on_the_resident_generic_kernel():
    """Scans beyond the small-scan kernels (here 12 000 points: k_pass_resident serves a call's later passes; small_resident = 2:
    from the first pass on) and the thread-per-query kernel on a scan that fills more than one group of workgroups."""
    maps, src = _big_world()
    for resident in (2, 1, 0):
        reg = _reg({"small_resident": resident}, **CFG)
        p, st = _check(reg, src, maps, I, I, 0.5, CFG, how="bits")               # KAT-1: exact zero residuals over 12 000 lanes
        np.testing.assert_array_equal(p, I)
        assert reg.last_stats.beta == 1.0 / DBL_MIN and reg.last_stats.n_corr[0] == len(src)
        d = 0.02
        cfg2 = dict(CFG, use_adaptive_odometry_regularization=False)
        reg2 = _reg({"small_resident": resident}, **cfg2)
        shifted = src - np.array([d, 0, 0])
        p, st = _check(reg2, shifted, maps, I, I, 0.5, cfg2)                      # KAT-2: several passes
        np.testing.assert_allclose(p[4], d, atol=1e-6)
        assert st.iterations >= 2
        if resident == 2:
            assert reg2.get_option("resident_passes") == st.iterations
        elif resident == 0:
            assert reg2.get_option("resident_passes") == 0
        p, st = _check(reg2, shifted, maps, I, I, 0.5, cfg2, via="batch")
        p, st = _check(_reg({"small_resident": resident}, **CFG), shifted, maps, I, I, 0.5, CFG, via="f32")   # KAT-3, float32 on the wire
        cfg0 = dict(CFG, max_num_iteration=0)
        p, st = _check(_reg({"small_resident": resident}, **cfg0), shifted, maps, syn.planar_pose(1, 2, 0.3), I, 0.5, cfg0, how="bits")
        # one correspondence among 12 000 lanes
        lonely = np.concatenate([np.full((11999, 3), 300.0), (maps[1].Pointcloud()[5] + np.array([0.0, 0.03, 0.0]))[None]])
        p, st = _check(_reg({"small_resident": resident}, **CFG), lonely, maps, I, I, 0.2, CFG)
        assert st.n_corr[0] == 1.0


def test_copies_of_a_registration_handle_are_independent():
    """kicp_reg_clone: same parameters and options, workspaces of its own; the original may go away"""
    maps, src = _big_world(n_map=20000, n_src=3000, seed=3)
    reg = _reg({"small_wave": 0}, **dict(CFG, max_num_iteration=7))
    twin = reg.copy()
    assert twin.max_num_iterations_ == 7 and twin.get_option("small_wave") == 0.0
    a = reg.ComputeRobotMotion(src - np.array([0.03, 0, 0]), maps[0], I, I, 0.5)
    del reg
    b = twin.ComputeRobotMotion(src - np.array([0.03, 0, 0]), maps[0], I, I, 0.5)
    np.testing.assert_array_equal(a, b)
    twin.max_num_iterations_ = 1
    twin.ComputeRobotMotion(src - np.array([0.03, 0, 0]), maps[0], I, I, 0.5)
    assert twin.last_stats.iterations == 1


@pytest.mark.parametrize("depth,rotate", [(1, 1), (2, 0), (3, 1), (4, 1)])
def test_batch_with_the_kernel_resident_across_scans_equals_the_plain_loop(depth, rotate):
    """kicp_register_device_batch keeps the generic kernel on the device across the scans of a batch ("batch_resident", here without
    the several-queues mode that large scans take by default), with `depth` scans in flight: scans of different sizes and
    iteration counts, the zero-correspondence scan in the middle, a kernel that gives up half-way (the plain loop takes over) -
    all bit-equal to one call per scan."""
    maps, src = _big_world(n_map=60000, n_src=20000, seed=21)
    g = maps[0]
    shifts = [0.0, 0.02, -0.05, 0.08, 0.0, 0.03, 0.01, -0.02, 0.04]  # (nine scans: the mode is for batches of eight and more)
    sizes = [20000, 12000, 9000, 20000, 15000, 10000, 16000, 9500, 20000]
    frames = [src[:k] - np.array([d, 0.0, 0.0]) for k, d in zip(sizes, shifts)]
    frames[4] = np.full((9000, 3), 400.0)  # no correspondence at all
    lasts = [syn.planar_pose(0.01 * i, 0.0, 0.001 * i) for i in range(9)]
    rels = [syn.planar_pose(-0.004 * i, 0.0, 0.0005) for i in range(9)]
    dev = [K.DeviceFrame(f, device=0) for f in frames]
    plain = _reg({"batch_resident": 0, "batch_queues": 0}, **CFG)
    b0 = plain.prepare_batch(dev, lasts, rels)
    want = plain.ComputeRobotMotionBatch(b0, g, 0.5).copy()
    assert plain.get_option("batch_resident_passes") == 0.0 and plain.last_status == K.KICP_WARN_NO_CORRESPONDENCES
    reg = _reg({"batch_queues": 0, "batch_depth": depth, "batch_rotate": rotate}, **CFG)
    b1 = reg.prepare_batch(dev, lasts, rels)
    for _ in range(3):
        got = reg.ComputeRobotMotionBatch(b1, g, 0.5).copy()
        assert np.array_equal(got, want, equal_nan=True) and list(b1.iterations) == list(b0.iterations)
        assert reg.last_status == K.KICP_WARN_NO_CORRESPONDENCES
    assert reg.get_option("batch_resident_passes") >= 3 * (sum(b0.iterations) - 10) and max(b0.iterations[:4]) >= 2
    for k, f in enumerate(frames[:4]):  # ... and to the oracle
        o = okicp.KinematicRegistration(**CFG).ComputeRobotMotion(f, maps[1], lasts[k], rels[k], 0.5)
        np.testing.assert_allclose(got[k], o, rtol=0, atol=1e-9)
    # the kernel gives up waiting for a late host: the scan in hand and the rest run through the plain loop, same bits
    reg.set_option("small_timeout_us", 300.0), reg.set_option("debug_stall_us", 3000.0)
    before = reg.get_option("small_relaunches")
    got = reg.ComputeRobotMotionBatch(b1, g, 0.5).copy()
    assert np.array_equal(got, want, equal_nan=True) and list(b1.iterations) == list(b0.iterations)
    assert reg.get_option("small_relaunches") == before + 1
    reg.set_option("small_timeout_us", 20000.0)
    got = reg.ComputeRobotMotionBatch(b1, g, 0.5).copy()
    assert np.array_equal(got, want, equal_nan=True)


def test_batch_of_small_scans_with_the_wave_kernel_resident_across_scans():
    """the same for scans of at most 4 096 points (one wave per query, k_pass_wave): sizes that differ inside one launch, a scan
    without correspondences, a give-up half-way; a batch that mixes small and large scans takes the plain loop"""
    maps, src = _big_world(n_map=60000, n_src=20000, seed=23)
    g = maps[0]
    sizes, shifts = [3000, 1080, 4000, 700, 2048, 512, 3500, 1900, 4096], [0.02, -0.04, 0.0, 0.06, 0.03, 0.01, -0.03, 0.05, 0.02]
    frames = [src[i * 100:i * 100 + k] - np.array([d, 0.0, 0.0]) for i, (k, d) in enumerate(zip(sizes, shifts))]
    frames[3] = np.full((700, 3), -300.0)
    lasts = [syn.planar_pose(0.01 * i, 0.0, 0.001 * i) for i in range(9)]
    rels = [syn.planar_pose(-0.004 * i, 0.0, 0.0005) for i in range(9)]
    dev = [K.DeviceFrame(f, device=0) for f in frames]
    plain = _reg({"batch_resident": 0, "batch_queues": 0}, **CFG)
    b0 = plain.prepare_batch(dev, lasts, rels)
    want = plain.ComputeRobotMotionBatch(b0, g, 0.5).copy()
    reg = _reg({"batch_queues": 0}, **CFG)
    b1 = reg.prepare_batch(dev, lasts, rels)
    for _ in range(3):
        got = reg.ComputeRobotMotionBatch(b1, g, 0.5).copy()
        assert np.array_equal(got, want, equal_nan=True) and list(b1.iterations) == list(b0.iterations)
    assert reg.get_option("batch_resident_passes") >= 3 * (sum(b0.iterations) - 10) and reg.get_option("small_active") == 2.0
    for k in (0, 1, 2, 4, 8):
        o = okicp.KinematicRegistration(**CFG).ComputeRobotMotion(frames[k], maps[1], lasts[k], rels[k], 0.5)
        np.testing.assert_allclose(got[k], o, rtol=0, atol=1e-9)
    reg.set_option("small_timeout_us", 300.0), reg.set_option("debug_stall_us", 3000.0)
    before = reg.get_option("small_relaunches")
    got = reg.ComputeRobotMotionBatch(b1, g, 0.5).copy()
    assert np.array_equal(got, want, equal_nan=True) and reg.get_option("small_relaunches") == before + 1
    reg.set_option("small_timeout_us", 20000.0)
    # small and large scans in one batch: no kernel serves both - the plain loop runs, same results
    mixed = [dev[0], K.DeviceFrame(src[:12000] - np.array([0.02, 0, 0]), device=0)] + dev[2:]
    bm0, bm1 = plain.prepare_batch(mixed, lasts, rels), reg.prepare_batch(mixed, lasts, rels)
    served = reg.get_option("batch_resident_passes")
    assert np.array_equal(reg.ComputeRobotMotionBatch(bm1, g, 0.5), plain.ComputeRobotMotionBatch(bm0, g, 0.5), equal_nan=True)
    assert reg.get_option("batch_resident_passes") == served


@pytest.mark.parametrize("queues,kind", [(2, "large"), (4, "large"), (8, "large"), (4, "mostly small"), (3, "mixed")])
def test_batch_with_several_scans_in_flight_on_queues_of_their_own_equals_the_plain_loop(queues, kind):
    """kicp_register_device_batch, default: "batch_queues" scans in flight at a time, each on a handle and HSA queue of its own,
    one host thread going round them - scans of different sizes and iteration counts (large: the generic pass kernel; small:
    one wave per query, a launch per pass; both kinds in one batch), a zero-correspondence scan, a change of
    max_num_iterations between calls (the lanes follow the caller's configuration): bit-equal to one call per scan."""
    maps, src = _big_world(n_map=60000, n_src=20000, seed=29)
    g = maps[0]
    count = 19
    sizes = [20000, 12000, 9000, 20000, 15000, 10000, 16000, 9500, 20000, 17000] * 2
    if kind == "mostly small":  # (a batch of small scans ONLY keeps the resident kernel: the other tests)
        sizes = [3000, 1080, 4000, 700, 2048, 512, 3500, 1900, 9000, 64] * 2
    elif kind == "mixed":
        sizes = [20000, 1080, 9000, 700, 15000, 512, 16000, 1900, 4096, 17000] * 2
    shifts = [0.0, 0.02, -0.05, 0.08, 0.0, 0.03, 0.01, -0.02, 0.04, 0.06] * 2
    frames = [src[:k] - np.array([d, 0.0, 0.0]) for k, d in zip(sizes[:count], shifts[:count])]
    frames[6] = np.full((sizes[6], 3), 400.0)  # no correspondence at all
    lasts = [syn.planar_pose(0.01 * i, 0.0, 0.001 * i) for i in range(count)]
    rels = [syn.planar_pose(-0.004 * i, 0.0, 0.0005) for i in range(count)]
    dev = [K.DeviceFrame(f, device=0) for f in frames]
    plain = _reg({"batch_resident": 0, "batch_queues": 0}, **CFG)
    b0 = plain.prepare_batch(dev, lasts, rels)
    want = plain.ComputeRobotMotionBatch(b0, g, 0.5).copy()
    reg = _reg({"batch_queues": queues}, **CFG)
    b1 = reg.prepare_batch(dev, lasts, rels)
    for _ in range(3):
        got = reg.ComputeRobotMotionBatch(b1, g, 0.5).copy()
        assert np.array_equal(got, want, equal_nan=True) and list(b1.iterations) == list(b0.iterations)
        assert reg.last_status == K.KICP_WARN_NO_CORRESPONDENCES
    assert reg.get_option("batch_queue_passes") >= 3 * count and reg.get_option("batch_resident_passes") == 0.0
    assert max(b0.iterations[:6]) >= 2
    for k in (0, 1, 3):  # ... and to the oracle
        o = okicp.KinematicRegistration(**CFG).ComputeRobotMotion(frames[k], maps[1], lasts[k], rels[k], 0.5)
        np.testing.assert_allclose(got[k], o, rtol=0, atol=1e-9)
    plain.max_num_iterations_ = reg.max_num_iterations_ = 1
    want1 = plain.ComputeRobotMotionBatch(b0, g, 0.5).copy()
    got1 = reg.ComputeRobotMotionBatch(b1, g, 0.5).copy()
    assert np.array_equal(got1, want1, equal_nan=True) and set(b1.iterations) == {1}
    # a batch too short for the mode (fewer than two scans per queue) goes the other ways, same bits
    served = reg.get_option("batch_queue_passes")
    short0, short1 = plain.prepare_batch(dev[:3], lasts[:3], rels[:3]), reg.prepare_batch(dev[:3], lasts[:3], rels[:3])
    assert np.array_equal(reg.ComputeRobotMotionBatch(short1, g, 0.5), plain.ComputeRobotMotionBatch(short0, g, 0.5), equal_nan=True)
    assert reg.get_option("batch_queue_passes") == served


@pytest.mark.parametrize("n_src", [1500, 9000])
def test_a_batch_longer_than_one_resident_launch_serves(n_src):
    """a resident launch serves at most 1 024 passes (= tags): a batch of 320 scans x ~4 iterations is relaunched on the way, in the
    middle of a scan if that is where the budget ends - same bits as one call per scan (wave-per-query and generic kernel)"""
    maps, src = _big_world(n_map=60000, n_src=20000, seed=29)
    g = maps[0]
    rng = np.random.default_rng(7)
    base = [K.DeviceFrame(src[i * 50:i * 50 + n_src] - np.array([0.04 + 0.01 * i, 0.0, 0.0]), device=0) for i in range(4)]
    count = 320
    dev = [base[i % 4] for i in range(count)]
    lasts = [syn.planar_pose(rng.uniform(-0.02, 0.02), 0.0, rng.uniform(-0.002, 0.002)) for _ in range(count)]
    rels = [syn.planar_pose(rng.uniform(-0.01, 0.01), 0.0, 0.0005) for _ in range(count)]
    four = dict(CFG, max_num_iteration=4, convergence_criterion=0.0)  # every scan runs exactly four iterations
    plain, reg = _reg({"batch_resident": 0, "batch_queues": 0}, **four), _reg({"batch_queues": 0}, **four)
    b0, b1 = plain.prepare_batch(dev, lasts, rels), reg.prepare_batch(dev, lasts, rels)
    want = plain.ComputeRobotMotionBatch(b0, g, 0.5).copy()
    got = reg.ComputeRobotMotionBatch(b1, g, 0.5).copy()
    assert sum(b0.iterations) == 1280  # more passes than one launch's budget of 1 024
    assert np.array_equal(got, want) and list(b1.iterations) == list(b0.iterations)
    assert reg.get_option("batch_resident_passes") == sum(b0.iterations)


@pytest.mark.parametrize("kind,threads", [("wave", 2), ("wave", 3), ("generic", 3), ("generic", 4)])
def test_batch_of_small_scans_on_several_resident_kernels_side_by_side(kind, threads):
    """kicp_register_device_batch on scans that leave most of the device empty (round 5, option "batch_threads"): the batch is cut into
    contiguous parts, each served by a resident kernel of its own from a host thread of the library's pool - sizes that differ, iteration
    counts 1 .. 10, a scan without correspondences in every part: bit-equal to one call per scan, counters added up on the caller's
    handle.  `wave`: one wave per query (<= 4 096 points); `generic`: the generic kernel's resident build on 16 500 .. 20 000-point scans
    (65 .. 79 workgroups each: up to six such kernels fit the device; fewer than three would not beat the four queues, which then stay)."""
    maps, src = _big_world(n_map=60000, n_src=20000, seed=31)
    g = maps[0]
    count = 16 * threads + 7
    rng = np.random.default_rng(3)
    if kind == "wave":
        sizes = [int(rng.integers(64, 1300)) for _ in range(count)]  # (three kernels of <= 1 365 waves fit the device side by side; a 4 096-point scan's kernel fills it alone)
    else:
        sizes = [int(rng.integers(16500, 20001)) for _ in range(count)]
    shifts = [float(rng.uniform(-0.06, 0.08)) for _ in range(count)]
    starts = [(17 * i) % (4000 if kind == "wave" else len(src) - k + 1) for i, k in enumerate(sizes)]
    frames = [src[a:a + k] - np.array([d, 0.0, 0.0]) for a, k, d in zip(starts, sizes, shifts)]
    for t in range(threads):
        frames[5 + 16 * t] = np.full((700 if kind == "wave" else 17000, 3), 400.0 + t)  # no correspondence at all
    lasts = [syn.planar_pose(0.002 * i, 0.0, 0.0005 * i) for i in range(count)]
    rels = [syn.planar_pose(-0.001 * i, 0.0, 0.0005) for i in range(count)]
    dev = [K.DeviceFrame(f, device=0) for f in frames]
    plain = _reg({"batch_resident": 0, "batch_queues": 0, "batch_threads": 0}, **CFG)
    b0 = plain.prepare_batch(dev, lasts, rels)
    want = plain.ComputeRobotMotionBatch(b0, g, 0.5).copy()
    assert plain.get_option("batch_threads_active") == 0.0 and max(b0.iterations) >= 3
    reg = _reg({"batch_threads": threads}, **CFG)
    b1 = reg.prepare_batch(dev, lasts, rels)
    for _ in range(3):
        before = reg.get_option("batch_resident_passes")
        got = reg.ComputeRobotMotionBatch(b1, g, 0.5).copy()
        assert np.array_equal(got, want, equal_nan=True) and list(b1.iterations) == list(b0.iterations)
        assert reg.get_option("batch_threads_active") == float(threads) and reg.last_status == K.KICP_WARN_NO_CORRESPONDENCES
        assert reg.get_option("batch_resident_passes") - before == sum(b0.iterations) - 10 * threads + threads  # (a NaN scan's later passes are accounted, not run)
    one = _reg({"batch_threads": 0}, **CFG)
    b2 = one.prepare_batch(dev, lasts, rels)
    assert np.array_equal(one.ComputeRobotMotionBatch(b2, g, 0.5), want, equal_nan=True) and one.get_option("batch_threads_active") == 0.0
    if kind == "wave":  # ... and a batch of scans whose kernel fills the device alone stays on one kernel
        big = [K.DeviceFrame(src[:4000], device=0)] * count
        bb = reg.prepare_batch(big, lasts, rels)
        reg.ComputeRobotMotionBatch(bb, g, 0.5)
        assert reg.get_option("batch_threads_active") == 0.0
    else:  # ... and two kernels of the generic build would not beat the four queues: those stay
        two = _reg({"batch_threads": 2}, **CFG)
        b3 = two.prepare_batch(dev, lasts, rels)
        before = two.get_option("batch_queue_passes")
        assert np.array_equal(two.ComputeRobotMotionBatch(b3, g, 0.5), want, equal_nan=True)
        assert two.get_option("batch_threads_active") == 0.0 and two.get_option("batch_queue_passes") > before
    # too few scans per thread: one kernel, the caller's thread
    short = reg.prepare_batch(dev[:20], lasts[:20], rels[:20])
    assert np.array_equal(reg.ComputeRobotMotionBatch(short, g, 0.5), want[:20], equal_nan=True) and reg.get_option("batch_threads_active") == 0.0


