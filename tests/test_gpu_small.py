"""The small-scan registration path (kicp_small.hpp: <= 16 workgroups, rows straight to the host, kernel resident for the
call's iterations) against the oracle, the reference build and the generic path - on BASELINE.json's config 4 (1 080-point 2-D
scan vs 50k-point map) and on sources of the size the pipeline registers (pipeline/KinematicICP.cpp:38-44,68-72)."""
import numpy as np
import pytest

import kinematic_icp_amd as K
from checkers import okicp, ref_available, ref_map_like, rkicp
from kinematic_icp_amd import synthetic as syn

pytestmark = pytest.mark.gpu
POSE_TOL = 1e-9


@pytest.fixture(scope="module")
def case4():
    cfg, scene, scans, rng = syn.make_case("cfg4", n_scans=4)
    gmap = K.VoxelHashMap(cfg.voxel_size, cfg.max_range, cfg.max_points_per_voxel)
    syn.build_map_points(scene, cfg, gmap.AddPoints, gmap.num_points, rng)
    omap = okicp.VoxelHashMap(cfg.voxel_size, cfg.max_range, cfg.max_points_per_voxel)
    omap.AddPoints(gmap.Pointcloud())
    rmap = ref_map_like(omap) if ref_available() else None
    return cfg, scans, gmap, omap, rmap


def _rels(scans, dx=0.0, yaw_deg=0.0):
    return [syn.pose_mul(s["rel_odom"], syn.planar_pose(dx, 0.0, np.deg2rad(yaw_deg))) for s in scans]


@pytest.mark.parametrize("wave", [1, 0])
@pytest.mark.parametrize("err", [(0.0, 0.0), (0.05, 0.5), (0.1, 2.0)])
def test_cfg4_small_path_matches_oracle_and_reference(case4, err, wave):
    cfg, scans, gmap, omap, rmap = case4
    tau = cfg.first_frame_tau()
    reg, oreg = K.KinematicRegistration(), okicp.KinematicRegistration()
    reg.set_option("small_wave", wave)
    iters = []
    for s, rel in zip(scans, _rels(scans, *err)):
        a = reg.ComputeRobotMotion(s["frame"], gmap, s["last_pose"], rel, tau)
        assert reg.get_option("small_active") == (2.0 if wave else 1.0)  # 2: one wave per query, 1: sub-lanes per query
        b = oreg.ComputeRobotMotion(s["frame"], omap, s["last_pose"], rel, tau)
        k = reg.last_stats.iterations
        iters.append(k)
        assert k == oreg.last_stats.iterations and reg.last_stats.converged == oreg.last_stats.converged
        np.testing.assert_allclose(a, b, rtol=0, atol=POSE_TOL)
        np.testing.assert_array_equal(np.array(reg.last_stats.n_corr[:k]), np.array(oreg.last_stats.n_corr[:k]))
        np.testing.assert_allclose(np.array([list(reg.last_stats.dx[i]) for i in range(k)]), np.array([list(oreg.last_stats.dx[i]) for i in range(k)]), rtol=0, atol=1e-10)
        if rmap is not None:
            c = rkicp.KinematicRegistration().ComputeRobotMotion(s["frame"], rmap, s["last_pose"], rel, tau)
            np.testing.assert_allclose(a, c, rtol=0, atol=POSE_TOL)
    if err != (0.0, 0.0):
        assert max(iters) > 1  # the resident loop really served later passes
    assert reg.get_option("small_relaunches") == 0.0


def test_small_path_bits_equal_generic_path(case4):
    """Exact accumulation: the small kernel (resident or one launch per pass, AQL or HIP launch, 1 / 2 / 4 sub-lanes) and the
    generic pass kernel give the same bits, iteration counts and per-pass counts."""
    cfg, scans, gmap, omap, rmap = case4
    tau = cfg.first_frame_tau()
    rels = _rels(scans, 0.08, 1.2)
    frames = [K.DeviceFrame(s["frame"]) for s in scans]
    base = K.KinematicRegistration()
    base.set_option("small", 0)
    want = []
    for fr, s, rel in zip(frames, scans, rels):
        want.append((base.ComputeRobotMotion(fr, gmap, s["last_pose"], rel, tau), base.last_stats.iterations, list(base.last_stats.n_corr[:base.last_stats.iterations])))
        assert base.get_option("small_active") == 0.0
    assert max(w[1] for w in want) > 1
    variants = [dict(small_wave=0, small_resident=r, aql=a, lanes_per_query=l) for r in (1, 0) for a in (1, 0) for l in (0, 1, 2, 4)]
    variants += [dict(small_wave=1, small_resident=r, aql=a) for r in (1, 0) for a in (1, 0)]
    variants += [dict(small_wave=w, small_cmd=0, aql=a) for w in (1, 0) for a in (1, 0)]  # test hook: workgroup 0 relays the commands (what a platform without a CPU-writable BAR runs)
    for opts in variants:
        reg = K.KinematicRegistration()
        for k, v in opts.items():
            reg.set_option(k, v)
        for rounds in range(2):
            for fr, s, rel, (pose, k, ncorr) in zip(frames, scans, rels, want):
                got = reg.ComputeRobotMotion(fr, gmap, s["last_pose"], rel, tau)
                assert reg.get_option("small_active") == (2.0 if opts["small_wave"] else 1.0)
                assert np.array_equal(got, pose), opts
                assert reg.last_stats.iterations == k and list(reg.last_stats.n_corr[:k]) == ncorr, opts
        if opts.get("aql", 1):
            assert reg.get_option("aql_active") == 1.0, opts


def test_small_path_sizes_and_limits(case4):
    """1 point, one workgroup's worth, the largest scans the two small kernels take and one point more (next path)."""
    cfg, scans, gmap, omap, rmap = case4
    tau = cfg.first_frame_tau()
    big = np.concatenate([s["frame"] for s in scans] * 8)  # 34 560 points of the same scene
    reg, gen, oreg = K.KinematicRegistration(), K.KinematicRegistration(), okicp.KinematicRegistration()
    gen.set_option("small", 0)
    s = scans[0]
    rel = _rels(scans, 0.05, 0.8)[0]
    for n, lanes, small in ((1, 0, 2), (3, 0, 2), (4, 0, 2), (5, 0, 2), (255, 0, 2), (1088, 0, 2), (1089, 0, 2), (2177, 0, 2), (4096, 0, 2), (4097, 0, 1), (8192, 0, 1),
                            (8193, 0, 0), (16384, 1, 1), (16385, 1, 0)):
        reg.set_option("lanes_per_query", lanes), gen.set_option("lanes_per_query", lanes)
        fr = np.ascontiguousarray(big[:n])
        a = reg.ComputeRobotMotion(fr, gmap, s["last_pose"], rel, tau)
        assert reg.get_option("small_active") == float(small), n
        b = gen.ComputeRobotMotion(fr, gmap, s["last_pose"], rel, tau)
        assert np.array_equal(a, b, equal_nan=True) and reg.last_stats.iterations == gen.last_stats.iterations, n
        c = oreg.ComputeRobotMotion(fr, omap, s["last_pose"], rel, tau)
        np.testing.assert_allclose(a, c, rtol=0, atol=POSE_TOL)
        assert reg.last_stats.iterations == oreg.last_stats.iterations


@pytest.mark.parametrize("cmd", [0, 1])
@pytest.mark.parametrize("wave", [1, 0])
def test_resident_kernel_gives_up_and_the_host_relaunches(case4, wave, cmd):
    """A host that is late with its next command (descheduled thread): the resident workgroups leave after their time-out and
    mark the pass they did not run; the host sees the marks, launches afresh and still returns the same bits."""
    cfg, scans, gmap, omap, rmap = case4
    tau = cfg.first_frame_tau()
    s, rel = scans[1], _rels(scans, 0.08, 1.2)[1]
    ref = K.KinematicRegistration()
    want = ref.ComputeRobotMotion(s["frame"], gmap, s["last_pose"], rel, tau)
    assert ref.last_stats.iterations > 2
    reg = K.KinematicRegistration()
    reg.set_option("small_wave", wave), reg.set_option("small_cmd", cmd)
    reg.set_option("small_timeout_us", 200.0)
    for stall in (2000.0, 150.0, 260.0):  # far beyond, just inside and just beyond the time-out (either outcome must give the same bits)
        before = reg.get_option("small_relaunches")
        reg.set_option("debug_stall_us", stall)
        got = reg.ComputeRobotMotion(s["frame"], gmap, s["last_pose"], rel, tau)
        assert np.array_equal(got, want) and reg.last_stats.iterations == ref.last_stats.iterations
        if stall == 2000.0:
            assert reg.get_option("small_relaunches") == before + 1
    # and the handle is fine afterwards
    assert np.array_equal(reg.ComputeRobotMotion(s["frame"], gmap, s["last_pose"], rel, tau), want)


def test_many_passes_and_tag_wraparound(case4):
    """More iterations than one launch serves (kSmallMaxPasses = 48: the loop relaunches), max_num_iterations reached, and the
    16-bit tag's wrap-around inside a reserved tag range."""
    cfg, scans, gmap, omap, rmap = case4
    tau = cfg.first_frame_tau()
    s, rel = scans[2], _rels(scans, 0.1, 2.0)[2]
    for max_it in (1, 2, 3, 60):
        kw = dict(max_num_iteration=max_it, convergence_criterion=0.0)  # never converges: runs to max_num_iterations
        reg, gen, oreg = K.KinematicRegistration(**kw), K.KinematicRegistration(**kw), okicp.KinematicRegistration(**kw)
        gen.set_option("small", 0)
        a = reg.ComputeRobotMotion(s["frame"], gmap, s["last_pose"], rel, tau)
        assert reg.get_option("small_active") == 2.0 and reg.last_stats.iterations == max_it and reg.last_stats.converged == 0
        assert np.array_equal(a, gen.ComputeRobotMotion(s["frame"], gmap, s["last_pose"], rel, tau))
        np.testing.assert_allclose(a, oreg.ComputeRobotMotion(s["frame"], omap, s["last_pose"], rel, tau), rtol=0, atol=POSE_TOL)
        assert oreg.last_stats.iterations == max_it
    reg = K.KinematicRegistration()
    reg.set_option("small_resident", 2)  # always resident: every launch reserves max_num_iterations tags
    want = reg.ComputeRobotMotion(s["frame"], gmap, s["last_pose"], rel, tau)
    reg.set_option("debug_tag", 65535 - 12)
    for _ in range(6):
        assert np.array_equal(reg.ComputeRobotMotion(s["frame"], gmap, s["last_pose"], rel, tau), want)
    assert reg.get_option("debug_tag") < 1000


def test_zero_correspondences_iteration_count(case4):
    """N_corr == 0: NaN pose, and - like the reference, whose loop cannot stop on a NaN (Registration.cpp:179-187) -
    max_num_iterations iterations are reported, on the small path and on the generic one."""
    cfg, scans, gmap, omap, rmap = case4
    tau = cfg.first_frame_tau()
    s = scans[0]
    far = s["frame"] + np.array([0.0, 0.0, 300.0])
    oreg = okicp.KinematicRegistration()
    assert np.isnan(oreg.ComputeRobotMotion(far, omap, s["last_pose"], s["rel_odom"], tau)).any()
    for small in (1, 0):
        reg = K.KinematicRegistration()
        reg.set_option("small", small)
        pose = reg.ComputeRobotMotion(far, gmap, s["last_pose"], s["rel_odom"], tau)
        assert np.isnan(pose).any() and reg.last_status == K.KICP_WARN_NO_CORRESPONDENCES
        assert reg.last_stats.iterations == oreg.last_stats.iterations == 10
        assert list(reg.last_stats.n_corr[:10]) == list(oreg.last_stats.n_corr[:10])


def test_pipeline_sized_sources(case4):
    """Sources of the size the drop-in pipeline registers: a double-downsampled 3-D frame (a few thousand points) against a 3-D
    map, through kicp_register_device as KinematicICP::RegisterFrame calls it."""
    cfg, scene, scans, rng = syn.make_case("cfg1", n_scans=2)
    gmap = K.VoxelHashMap(cfg.voxel_size, cfg.max_range, cfg.max_points_per_voxel)
    syn.build_map_points(scene, cfg, gmap.AddPoints, gmap.num_points, rng)
    omap = okicp.VoxelHashMap(cfg.voxel_size, cfg.max_range, cfg.max_points_per_voxel)
    omap.AddPoints(gmap.Pointcloud())
    rmap = ref_map_like(omap) if ref_available() else None
    reg, oreg = K.KinematicRegistration(), okicp.KinematicRegistration()
    for s in scans:
        src = okicp.voxel_downsample(okicp.voxel_downsample(s["frame"], cfg.voxel_size * 0.5), cfg.voxel_size * 1.5)
        assert 200 < len(src) < 4096
        rel = syn.pose_mul(s["rel_odom"], syn.planar_pose(0.06, 0.0, np.deg2rad(0.7)))
        a = reg.ComputeRobotMotion(K.DeviceFrame(src), gmap, s["last_pose"], rel, cfg.first_frame_tau())
        assert reg.get_option("small_active") == 2.0
        np.testing.assert_allclose(a, oreg.ComputeRobotMotion(src, omap, s["last_pose"], rel, cfg.first_frame_tau()), rtol=0, atol=POSE_TOL)
        assert reg.last_stats.iterations == oreg.last_stats.iterations
        if rmap is not None:
            np.testing.assert_allclose(a, rkicp.KinematicRegistration().ComputeRobotMotion(src, rmap, s["last_pose"], rel, cfg.first_frame_tau()), rtol=0, atol=POSE_TOL)


@pytest.mark.parametrize("cap", [1, 20, 21, 60, 255])
def test_wave_kernel_on_deep_and_shallow_buckets(cap):
    """max_points_per_voxel 1 ... 255: buckets of one point and buckets that take several 20-point trips of the wave kernel
    (the mirror's bucket stride is 20, 40, 60, 260), dense enough to fill them; against the oracle, the reference build and the
    generic kernel."""
    rng = np.random.default_rng(cap)
    vs = 0.5
    mpts = rng.uniform(-4, 4, (60000, 3)) * np.array([1.0, 1.0, 0.15])
    gmap, omap = K.VoxelHashMap(vs, 100.0, cap), okicp.VoxelHashMap(vs, 100.0, cap)
    gmap.AddPoints(mpts), omap.AddPoints(mpts)
    assert gmap.num_points() == omap.num_points()
    frame = mpts[rng.choice(len(mpts), 900, replace=False)] + rng.normal(0, 0.02, (900, 3))
    last, rel = okicp.IDENTITY, syn.planar_pose(0.03, 0.0, np.deg2rad(0.6))
    reg, gen, oreg = K.KinematicRegistration(), K.KinematicRegistration(), okicp.KinematicRegistration()
    gen.set_option("small", 0)
    for tau in (0.3, 0.05):
        a = reg.ComputeRobotMotion(frame, gmap, last, rel, tau)
        assert reg.get_option("small_active") == 2.0
        assert np.array_equal(a, gen.ComputeRobotMotion(frame, gmap, last, rel, tau)) and reg.last_stats.iterations == gen.last_stats.iterations
        b = oreg.ComputeRobotMotion(frame, omap, last, rel, tau)
        np.testing.assert_allclose(a, b, rtol=0, atol=POSE_TOL)
        k = reg.last_stats.iterations
        assert k == oreg.last_stats.iterations and list(reg.last_stats.n_corr[:k]) == list(oreg.last_stats.n_corr[:k])
        if ref_available():
            np.testing.assert_allclose(a, rkicp.KinematicRegistration().ComputeRobotMotion(frame, ref_map_like(omap), last, rel, tau), rtol=0, atol=POSE_TOL)


def test_small_host_frames_through_the_bar(case4):
    """kicp_register (the reference's own signature: a host vector) with frames of up to 8 192 points: written straight into HBM
    through the PCIe BAR.  Alternating frames, sizes around the limit and the staged path give the same bits as device-resident
    frames; a stale cache line anywhere would show as a wrong pose."""
    cfg, scans, gmap, omap, rmap = case4
    tau = cfg.first_frame_tau()
    rels = _rels(scans, 0.05, 0.8)
    big = np.concatenate([s["frame"] for s in scans] * 3)
    reg, staged, dev = K.KinematicRegistration(), K.KinematicRegistration(), K.KinematicRegistration()
    staged.set_option("bar_frame", 0)
    for rounds in range(3):
        for i, (s, rel) in enumerate(zip(scans, rels)):
            n = (1080, 777, 8192, 8193, 1, 4097)[(i + rounds) % 6]
            fr = np.ascontiguousarray(np.roll(big, 131 * (i + 4 * rounds), axis=0)[:n])
            a = reg.ComputeRobotMotion(fr, gmap, s["last_pose"], rel, tau)
            b = staged.ComputeRobotMotion(fr, gmap, s["last_pose"], rel, tau)
            c = dev.ComputeRobotMotion(K.DeviceFrame(fr), gmap, s["last_pose"], rel, tau)
            assert np.array_equal(a, b, equal_nan=True) and np.array_equal(a, c, equal_nan=True), (rounds, i, n)
            assert reg.last_stats.iterations == dev.last_stats.iterations
    assert reg.get_option("bar_frame") == 1.0 and staged.get_option("bar_frame") == 0.0
    np.testing.assert_allclose(a, okicp.KinematicRegistration().ComputeRobotMotion(fr, omap, s["last_pose"], rel, tau), rtol=0, atol=POSE_TOL)
