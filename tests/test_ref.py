"""CPU tests (no GPU) of oracle/_ref: the reference's OWN sources (registration/Registration.cpp,
correspondence_threshold/CorrespondenceThreshold.cpp, pipeline/KinematicICP.cpp, compiled unmodified against the
stand-in headers of oracle/ref_shim) as the anchor of parity.

  * the stand-ins are pinned first: the Sophus shim against scipy, the tsl::robin_map shim against the container's
    published behaviour, the kiss-icp stand-in against the survey's known-answer tests;
  * then oracle/kicp_oracle.cpp (the restatement every GPU test is compared with) is required to equal the reference
    build BIT FOR BIT in serial mode - poses, neighbour queries, thresholds, pre-steps, whole pipeline;
  * tests/golden/ref_outputs.npz (outputs of the reference build, generator committed) must be reproduced by the
    reference build where it exists and by the oracle everywhere (the GPU box has no /root/reference).
"""
import os

import numpy as np
import pytest

import ref_numpy as rn
from checkers import GOLDEN, okicp, ref, ref_map_like
from conftest import sort_rows
from kinematic_icp_amd import synthetic as syn
from test_oracle import rand_pose, small_world

DBL_MAX = np.finfo(np.float64).max
REG_VARIANTS = (("default", dict()), ("fixed0", dict(use_adaptive_odometry_regularization=False, fixed_regularization=0.0)),
                ("fixed5", dict(use_adaptive_odometry_regularization=False, fixed_regularization=5.0)),
                ("it3", dict(max_num_iteration=3)), ("loose", dict(convergence_criterion=1e-2)))


# ---------------------------------------------------------------- the stand-ins ---------------------------------------
def test_reference_build_names_its_sources():
    r = ref()
    src = r.lib().rkicp_sources().decode()
    for f in ("registration/Registration.cpp", "correspondence_threshold/CorrespondenceThreshold.cpp", "pipeline/KinematicICP.cpp"):
        assert f in src
    if r.reference_present():  # the library is newer than every reference source it was built from
        lib_t = os.path.getmtime(os.path.join(os.path.dirname(r.__file__), "_ref", "libkicp_ref.so"))
        for f in ("registration/Registration.cpp", "correspondence_threshold/CorrespondenceThreshold.cpp", "pipeline/KinematicICP.cpp"):
            assert os.path.getmtime(os.path.join(r.REFERENCE, "cpp", "kinematic_icp", f)) < lib_t


def test_sophus_shim_matches_scipy_and_the_oracle_bitwise():
    r = ref()
    rng = np.random.default_rng(0)
    for _ in range(50):
        a, b = rand_pose(rng), rand_pose(rng)
        pts = rng.normal(size=(7, 3))
        np.testing.assert_allclose(r.se3_act(a, pts), rn.act(rn.from_qt(a), pts), atol=1e-13)
        np.testing.assert_allclose(r.se3_act(r.se3_mul(a, b), pts), rn.act(rn.mul(rn.from_qt(a), rn.from_qt(b)), pts), atol=1e-12)
        np.testing.assert_allclose(r.se3_act(r.se3_inverse(a), r.se3_act(a, pts)), pts, atol=1e-12)
        assert np.array_equal(r.se3_act(a, pts), okicp.se3_act(a, pts))
        assert np.array_equal(r.se3_mul(a, b), okicp.se3_mul(a, b))
        assert np.array_equal(r.se3_inverse(a), okicp.se3_inverse(a))
    for scale in (1.0, 1e-3, 1e-9, 1e-12, 0.0):
        for _ in range(10):
            xi = rng.normal(size=6) * np.array([1, 1, 1, scale, scale, scale])
            xi[3:] *= min(1.0, 2.5 / max(np.linalg.norm(xi[3:]), 1e-300))
            T = r.se3_exp(xi)
            np.testing.assert_allclose(r.se3_act(T, pts), rn.act(rn.se3_exp(xi), pts), atol=1e-12)
            np.testing.assert_allclose(r.se3_log(T), xi, atol=1e-9 if scale else 1e-12)
            assert np.array_equal(T, okicp.se3_exp(xi))
            assert np.array_equal(r.se3_log(T), okicp.se3_log(T))


def _robin_reference_order(keys_hash, n_reserve):
    """tsl::robin_map iteration order after reserve(n_reserve) + insertion of distinct keys in the given order,
    re-enacted in plain Python from the container's published rule (independent of both C++ implementations)."""
    want = int(np.ceil(np.float32(n_reserve) / np.float32(0.5)))
    B = 1
    while B < want:
        B <<= 1
    slot = [None] * B  # (dist, item)
    for item, h in enumerate(keys_hash):
        b, dist, carry = int(h) & (B - 1), 0, item
        while slot[b] is not None and dist <= slot[b][0]:
            b, dist = (b + 1) & (B - 1), dist + 1
        while slot[b] is not None:
            if dist > slot[b][0]:
                (dist, carry), slot[b] = slot[b], (dist, carry)
            b, dist = (b + 1) & (B - 1), dist + 1
        slot[b] = (dist, carry)
    return [s[1] for s in slot if s is not None]


def test_robin_map_shim_iteration_order():
    """VoxelDownsample's output order = the grid's iteration order: shim container, oracle restatement and a plain
    Python re-enactment of robin-hood insertion agree, on hashes that collide heavily (planar data)."""
    r = ref()
    rng = np.random.default_rng(5)
    for n, vs, scale in ((3000, 0.5, (20, 20, 0.4)), (700, 1.5, (30, 30, 30)), (5000, 0.25, (4, 4, 4)), (64, 1.0, (2, 2, 2))):
        pts = rng.uniform(-1, 1, (n, 3)) * np.array(scale)
        a, b = r.voxel_downsample(pts, vs), okicp.voxel_downsample(pts, vs)
        assert np.array_equal(a, b)
        vox = np.floor(pts / vs).astype(np.int64)
        _, first = np.unique(vox, axis=0, return_index=True)
        first = np.sort(first)  # voxels in first-seen order = the insertion order of the grid
        v = vox[first].astype(np.uint32)
        h = (v[:, 0] * np.uint32(73856093)) ^ (v[:, 1] * np.uint32(19349669)) ^ (v[:, 2] * np.uint32(83492791))
        order = _robin_reference_order(h, n)
        assert np.array_equal(a, pts[first[order]])
        assert len(a) < n or n < 100  # some voxel saw more than one point


def test_kiss_icp_stand_in_known_answers():
    """SURVEY.md App. B.4 KAT-4 / KAT-5 / remove-far on the kiss-icp stand-in (same expectations as for the oracle)."""
    r = ref()
    m = r.VoxelHashMap(1.0, 100.0, 20)
    m.AddPoints(np.array([[-0.1, -0.1, -0.1]]))
    nn, d = m.GetClosestNeighbor(np.array([[0.05, 0.05, 0.05]]))
    np.testing.assert_allclose(nn[0], [-0.1, -0.1, -0.1])
    np.testing.assert_allclose(d[0], np.sqrt(3 * 0.15**2))
    nn, d = m.GetClosestNeighbor(np.array([[5.5, 5.5, 5.5], [1.5, 0.5, 0.5]]))
    assert d[0] == DBL_MAX and np.all(nn[0] == 0.0) and d[1] == DBL_MAX
    t = r.VoxelHashMap(1.0, 100.0, 20)
    t.AddPoints(np.array([[-0.25, 0.5, 0.5], [1.25, 0.5, 0.5]]))
    np.testing.assert_array_equal(t.GetClosestNeighbor(np.array([[0.5, 0.5, 0.5]]))[0][0], [1.25, 0.5, 0.5])  # earlier shift wins
    t2 = r.VoxelHashMap(1.0, 100.0, 20)
    t2.AddPoints(np.array([[0.75, 0.5, 0.5], [0.25, 0.5, 0.5]]))
    np.testing.assert_array_equal(t2.GetClosestNeighbor(np.array([[0.5, 0.5, 0.5]]))[0][0], [0.75, 0.5, 0.5])  # first inserted wins
    cap, vs = 20, 1.0
    res = vs / np.sqrt(cap)
    gx, gy = np.meshgrid(np.arange(5) * 0.23 + 0.02, np.arange(4) * 0.24 + 0.02)
    m = r.VoxelHashMap(vs, 100.0, cap)
    m.AddPoints(np.stack([gx.ravel(), gy.ravel(), np.full(20, 0.5)], 1))
    m.AddPoints(np.array([[0.5, 0.5, 0.95]]))
    assert m.num_points() == 20
    m2 = r.VoxelHashMap(vs, 100.0, cap)
    m2.AddPoints(np.array([[0.1, 0.1, 0.1], [0.1 + 0.9 * res, 0.1, 0.1], [0.1 + 2.0 * res, 0.1, 0.1]]))
    assert m2.num_points() == 2
    m3 = r.VoxelHashMap(2.0, 100.0, 16)
    m3.AddPoints(np.array([[0.25, 0.25, 0.25], [0.75, 0.25, 0.25]]))
    assert m3.num_points() == 2  # distance == map_resolution is kept (strict <)
    m5 = r.VoxelHashMap(vs, 100.0, cap)
    m5.AddPoints(np.array([[-0.5, -0.5, -0.5], [-1.0, -1.0, -1.0], [-1e-9, 0.0, 0.0], [-1.0 - 1e-9, -1.0, -1.0]]))
    assert m5.num_voxels() == 3 and m5.num_points() == 4
    f = r.VoxelHashMap(1.0, 10.0, 20)
    f.AddPoints(np.array([[9.95, 0.5, 0.5], [9.55, 0.5, 0.5], [-9.55, 0.5, 0.5], [-9.95, 0.5, 0.5]]))
    f.RemovePointsFarFromLocation(np.array([-0.04, 0.5, 0.5]))
    assert f.num_points() == 4
    f.RemovePointsFarFromLocation(np.array([-0.06, 0.5, 0.5]))
    np.testing.assert_array_equal(sort_rows(f.Pointcloud()), sort_rows(np.array([[-9.55, 0.5, 0.5], [-9.95, 0.5, 0.5]])))


# ---------------------------------------------------------------- the reference's own code ----------------------------
def test_reference_known_answers():
    """SURVEY.md App. B.4 on the reference build itself: KAT-0 (empty map), KAT-1 (perfect alignment), KAT-2/3 (small
    shift without / with adaptive regularisation), the NaN convention, the theta == 0 quirk, max_num_iterations = 0."""
    r = ref()
    I = okicp.IDENTITY
    last, rel = syn.planar_pose(1, 2, 0.3), syn.planar_pose(0.5, 0, 0.1)
    p = r.KinematicRegistration().ComputeRobotMotion(np.zeros((10, 3)), r.VoxelHashMap(1.0, 100.0, 20), last, rel, 1.0)
    np.testing.assert_array_equal(p, r.se3_mul(last, rel))                                   # KAT-0
    rng = np.random.default_rng(4)
    m = r.VoxelHashMap(1.0, 100.0, 20)
    m.AddPoints(rng.uniform(-10, 10, (800, 3)))
    np.testing.assert_array_equal(r.KinematicRegistration().ComputeRobotMotion(m.Pointcloud(), m, I, I, 0.5), I)  # KAT-1 (+ theta == 0)
    m = r.VoxelHashMap(1.0, 100.0, 20)
    m.AddPoints(np.random.default_rng(6).uniform(-10, 10, (1500, 3)))
    d = 0.02
    src = m.Pointcloud() - np.array([d, 0, 0])
    p = r.KinematicRegistration(10, 1e-3, 1, False, 0.0).ComputeRobotMotion(src, m, I, I, 0.5)
    np.testing.assert_allclose(p[4], d, atol=1e-6)                                            # KAT-2
    p = r.KinematicRegistration(10, 1e-3, 1, True, 0.0).ComputeRobotMotion(src, m, I, I, 0.5)
    assert abs(p[4]) < d * 1e-2                                                               # KAT-3: odometry is trusted
    reg = r.KinematicRegistration()
    p = reg.ComputeRobotMotion(np.full((50, 3), 500.0), m, I, I, 0.5)
    assert reg.last_status == 1 and np.isnan(p).any()                                         # 0/0 -> NaN
    p = r.KinematicRegistration(0).ComputeRobotMotion(src, m, last, rel, 0.5)
    np.testing.assert_array_equal(p, r.se3_mul(last, rel))                                   # loop body never runs
    t = r.CorrespondenceThreshold(1.0 / np.sqrt(20), 100.0, True, 1.0)
    np.testing.assert_allclose(t.ComputeThreshold(), 3.0 / np.sqrt(20), rtol=1e-15)           # KAT-6
    t.UpdateOdometryError(np.array([0, 0, 0, 1, 0.3, 0, 0]))
    np.testing.assert_allclose(t.ComputeThreshold(), 3.0 * (1 / np.sqrt(20) + np.sqrt(0.09 / (1 + 1e-8))), rtol=1e-14)
    assert r.CorrespondenceThreshold(0.2, 50.0, False, 1.25).ComputeThreshold() == 1.25


@pytest.mark.parametrize("seed,voxel,cap", [(9, 1.0, 20), (21, 0.5, 20), (33, 2.0, 5), (45, 1.0, 1)])
def test_oracle_equals_reference_bitwise_registration(seed, voxel, cap):
    """ComputeRobotMotion: the restatement must return the reference build's bits (serial mode, same libstdc++)."""
    r = ref()
    mpts, f = small_world(seed=seed, n_map=5000, n_frame=700)
    o = okicp.VoxelHashMap(voxel, 100.0, cap)
    o.AddPoints(mpts)
    m = ref_map_like(o)
    assert m.num_points() == o.num_points() and m.num_voxels() == o.num_voxels()
    np.testing.assert_array_equal(sort_rows(m.Pointcloud()), sort_rows(o.Pointcloud()))
    q = np.concatenate([f, f + np.random.default_rng(seed).normal(0, 0.4, f.shape), np.full((3, 3), 400.0)])
    nn_o, d_o = o.GetClosestNeighbor(q)
    nn_r, d_r = m.GetClosestNeighbor(q)
    assert np.array_equal(nn_o, nn_r) and np.array_equal(d_o, d_r)
    n_multi = 0
    for vname, kw in REG_VARIANTS:
        for tau in (0.67 * voxel, 0.25 * voxel, 2.5 * voxel):
            for last, rel in ((syn.planar_pose(0.1, 0.05, 0.02), syn.planar_pose(0.08, 0.0, 0.015)),
                              (syn.planar_pose(-0.3, 0.2, -0.05), syn.planar_pose(0.25, 0.0, -0.04))):
                oreg = okicp.KinematicRegistration(**kw)
                a = oreg.ComputeRobotMotion(f, o, last, rel, tau)
                b = r.KinematicRegistration(**kw).ComputeRobotMotion(f, m, last, rel, tau)
                assert np.array_equal(a, b, equal_nan=True), (vname, tau, a, b)
                n_multi += oreg.last_stats.iterations > 1
    assert n_multi >= 4  # the multi-iteration loop, stop rule and re-association are exercised


def test_oracle_equals_reference_bitwise_on_a_baseline_config():
    """cfg1 of BASELINE.json (16 384-pt scan vs 100k-pt map) at full size, default and multi-iteration settings."""
    r = ref()
    cfg, scene, scans, rng = syn.make_case("cfg1", n_scans=2)
    o = okicp.VoxelHashMap(cfg.voxel_size, cfg.max_range, cfg.max_points_per_voxel)
    syn.build_map_points(scene, cfg, o.AddPoints, o.num_points, rng)
    m = ref_map_like(o)
    o2 = okicp.VoxelHashMap(cfg.voxel_size, cfg.max_range, cfg.max_points_per_voxel)
    o2.AddPoints(o.Pointcloud())
    tau = cfg.first_frame_tau()
    for s in scans:
        for extra in (syn.planar_pose(0, 0, 0), syn.planar_pose(0.2, 0.0, np.deg2rad(1.5))):
            rel = syn.pose_mul(s["rel_odom"], extra)
            a = okicp.KinematicRegistration().ComputeRobotMotion(s["frame"], o2, s["last_pose"], rel, tau)
            b = r.KinematicRegistration().ComputeRobotMotion(s["frame"], m, s["last_pose"], rel, tau)
            assert np.array_equal(a, b)


def test_reference_threads_agree_with_serial():
    """The stand-in TBB with several threads (chunked parallel_for / parallel_reduce, unordered concurrent_vector) gives
    the serial result up to summation order - the property the reference itself has with max_num_threads > 1 (F10)."""
    r = ref()
    mpts, f = small_world(seed=17, n_map=6000, n_frame=3000)
    m = r.VoxelHashMap(1.0, 100.0, 20)
    m.AddPoints(mpts)
    last, rel = syn.planar_pose(0.1, 0.05, 0.02), syn.planar_pose(0.08, 0.0, 0.015)
    a = r.KinematicRegistration(max_num_threads=1).ComputeRobotMotion(f, m, last, rel, 0.67)
    for nt in (2, 4, 0):
        b = r.KinematicRegistration(max_num_threads=nt).ComputeRobotMotion(f, m, last, rel, 0.67)
        np.testing.assert_allclose(a, b, rtol=0, atol=1e-12)


def test_oracle_equals_reference_bitwise_threshold_and_presteps():
    r = ref()
    rng = np.random.default_rng(12)
    to, tr = okicp.CorrespondenceThreshold(0.2, 80.0, True, 1.0), r.CorrespondenceThreshold(0.2, 80.0, True, 1.0)
    for k in range(20):
        e = rand_pose(rng) if k % 3 else rand_pose(rng, planar=True)
        e[4:] *= 0.02
        if k % 5 == 0:
            e[:4] = -e[:4]  # w < 0 branch of logAndTheta
        to.UpdateOdometryError(e), tr.UpdateOdometryError(e)
        assert to.ComputeThreshold() == tr.ComputeThreshold()
    to.Reset(), tr.Reset()
    assert to.ComputeThreshold() == tr.ComputeThreshold() == 0.6000000000000001 or to.ComputeThreshold() == tr.ComputeThreshold()
    pts = rng.uniform(-30, 30, (4000, 3))
    ts = rng.uniform(0, 1, len(pts))
    rel = syn.pose_mul(syn.planar_pose(0.5, 0.1, 0.05), np.array([0.01, -0.02, 0, np.sqrt(1 - 5e-4), 0, 0, 0.02]))
    for deskew, stamps in ((True, ts), (False, ts), (True, None)):
        a, b = okicp.preprocess(pts, stamps, rel, 25.0, 2.0, deskew), r.preprocess(pts, stamps, rel, 25.0, 2.0, deskew)
        assert np.array_equal(a, b) and 0 < len(a) < len(pts)
    for vs in (0.5, 1.5, 0.05):
        assert np.array_equal(okicp.voxel_downsample(pts, vs), r.voxel_downsample(pts, vs))
    # Update(points, pose) sequences with pruning: same voxels, same points
    o, m = okicp.VoxelHashMap(0.5, 6.0, 10), r.VoxelHashMap(0.5, 6.0, 10)
    for k in range(4):
        pose = syn.planar_pose(1.5 * k, -0.5 * k, 0.2 * k)
        chunk = rng.uniform(-7, 7, (1500, 3)) * np.array([1, 1, 0.2])
        o.Update(chunk, pose), m.Update(chunk, pose)
        assert o.num_points() == m.num_points() and o.num_voxels() == m.num_voxels()
    assert np.array_equal(sort_rows(o.Pointcloud()), sort_rows(m.Pointcloud()))


# ---------------------------------------------------------------- frozen outputs of the reference build ---------------
def _registration_outputs(mod):
    g = np.load(os.path.join(GOLDEN, "registration_small.npz"))
    out = {}
    for name in ("a", "b", "c"):
        m = mod.VoxelHashMap(float(g[name + "_voxel"]), float(g[name + "_maxrange"]), 20)
        m.AddPoints(g[name + "_map"])
        for vname, kw in REG_VARIANTS:
            for tau_scale in (1.0, 0.4):
                out["reg_%s_%s_%g" % (name, vname, tau_scale)] = mod.KinematicRegistration(**kw).ComputeRobotMotion(
                    g[name + "_frame"], m, g[name + "_last"], g[name + "_rel"], float(g[name + "_tau"]) * tau_scale)
        q = g[name + "_frame"][::7]
        out["nn_%s" % name], out["nnd_%s" % name] = m.GetClosestNeighbor(mod.se3_act(mod.se3_mul(g[name + "_last"], g[name + "_rel"]), q))
    return out


def _pipeline_outputs(mod_pipeline_factory, deskew):
    p = np.load(os.path.join(GOLDEN, "pipeline_small.npz"))
    L = [int(v) for v in p["layout"]]
    icp = mod_pipeline_factory(max_range=float(p["max_range"]), min_range=float(p["min_range"]), voxel_size=float(p["voxel"]), deskew=deskew)
    out = {}
    for k in range(int(p["n_frames"])):
        raw = p["raw%d" % k]
        xyz, stamps, _ = okicp.ingest(raw.tobytes(), len(raw) // L[0], L[0], L[1], L[2], L[3], L[4], L[5])
        tag = "pipe%d_%d" % (deskew, k)
        out[tag + "_tau"] = np.array(icp.tau())
        frame, source = icp.RegisterFrame(xyz, stamps, p["ext"], p["delta%d" % k])
        out[tag + "_pose"], out[tag + "_nframe"], out[tag + "_source"] = icp.pose(), np.array(len(frame)), source
        out[tag + "_nmap"] = np.array(len(icp.LocalMap()))
    pc = icp.LocalMap()
    out["pipe%d_final_map_sorted" % deskew] = pc[np.lexsort((pc[:, 2], pc[:, 1], pc[:, 0]))]
    return out


class OraclePipeline:
    """pipeline/KinematicICP.cpp:48-85 re-enacted with the oracle's pieces (the restatement of RegisterFrame)."""

    def __init__(self, max_range, min_range, voxel_size, deskew, max_points_per_voxel=20):
        self.max_range, self.min_range, self.voxel_size, self.deskew = max_range, min_range, voxel_size, deskew
        self.map = okicp.VoxelHashMap(voxel_size, max_range, max_points_per_voxel)
        self.thr = okicp.CorrespondenceThreshold(voxel_size / np.sqrt(max_points_per_voxel), max_range, True, 1.0)
        self.reg = okicp.KinematicRegistration()
        self.last = okicp.IDENTITY.copy()

    def tau(self):
        return self.thr.ComputeThreshold()

    def pose(self):
        return self.last.copy()

    def LocalMap(self):
        return self.map.Pointcloud()

    def RegisterFrame(self, frame, stamps, ext, delta):
        rel_lidar = okicp.se3_mul(okicp.se3_mul(okicp.se3_inverse(ext), delta), ext)
        pre = okicp.preprocess(frame, stamps, rel_lidar, self.max_range, self.min_range, self.deskew)
        in_base = okicp.se3_act(ext, pre)
        down = okicp.voxel_downsample(in_base, self.voxel_size * 0.5)
        source = okicp.voxel_downsample(down, self.voxel_size * 1.5)
        new = self.reg.ComputeRobotMotion(source, self.map, self.last, delta, self.thr.ComputeThreshold())
        self.thr.UpdateOdometryError(okicp.se3_mul(okicp.se3_inverse(okicp.se3_mul(self.last, delta)), new))
        self.map.Update(down, new)
        self.last = new
        return in_base, source


def test_golden_reference_outputs_reproduced_by_the_reference_build():
    r = ref()
    g = np.load(os.path.join(GOLDEN, "ref_outputs.npz"))
    got = _registration_outputs(r)
    for deskew in (0, 1):
        got.update(_pipeline_outputs(r.KinematicICP, deskew))
    for k, v in got.items():
        assert np.array_equal(np.asarray(v), g[k], equal_nan=True), k
    to = r.CorrespondenceThreshold(1.0 / np.sqrt(20), 100.0, True, 1.0)
    taus = [to.ComputeThreshold()]
    for e in g["thr_errs"]:
        to.UpdateOdometryError(e)
        taus.append(to.ComputeThreshold())
    assert np.array_equal(np.array(taus), g["thr_taus"])


def test_oracle_reproduces_the_reference_builds_frozen_outputs_bitwise():
    """Runs everywhere (also without oracle/_ref): the restatement against the frozen outputs of the reference build."""
    g = np.load(os.path.join(GOLDEN, "ref_outputs.npz"))
    got = _registration_outputs(okicp)
    for deskew in (0, 1):
        got.update(_pipeline_outputs(OraclePipeline, deskew))
    for k, v in got.items():
        assert np.array_equal(np.asarray(v), g[k], equal_nan=True), k
    to = okicp.CorrespondenceThreshold(1.0 / np.sqrt(20), 100.0, True, 1.0)
    taus = [to.ComputeThreshold()]
    for e in g["thr_errs"]:
        to.UpdateOdometryError(e)
        taus.append(to.ComputeThreshold())
    assert np.array_equal(np.array(taus), g["thr_taus"])
