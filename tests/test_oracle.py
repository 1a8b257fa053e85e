"""CPU tests (no GPU): the C++ oracle against (i) an independent numpy/scipy restatement, (ii) analytic known-answer
tests derived from the cited reference formulas, (iii) the frozen regression vectors.

PARITY UNPINNED: the reference has no tests/fixtures and cannot be built offline; two independent restatements
agreeing + analytic KATs are the strongest pin available."""
import os

import numpy as np
import pytest

import ref_numpy as rn
from kinematic_icp_amd import synthetic as syn
from oracle import okicp

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "registration_small.npz")
DBL_MAX = np.finfo(np.float64).max


def rand_pose(rng, planar=False):
    if planar:
        return syn.planar_pose(rng.uniform(-5, 5), rng.uniform(-5, 5), rng.uniform(-3, 3))
    q = rng.normal(size=4)
    q /= np.linalg.norm(q)
    return np.concatenate([q, rng.uniform(-5, 5, 3)])


def small_world(seed=5, n_map=4000, n_frame=500):
    rng = np.random.default_rng(seed)
    # two planes + clutter, so that neighbours exist in several of the 27 voxels
    g = np.stack([rng.uniform(-8, 8, n_map // 2), rng.uniform(-8, 8, n_map // 2), rng.normal(0, 0.02, n_map // 2)], 1)
    w = np.stack([rng.uniform(-8, 8, n_map // 4), np.full(n_map // 4, 6.0) + rng.normal(0, 0.02, n_map // 4), rng.uniform(0, 3, n_map // 4)], 1)
    c = rng.uniform(-8, 8, (n_map // 4, 3)) * np.array([1, 1, 0.3])
    mpts = np.concatenate([g, w, c])
    rng.shuffle(mpts)
    f = np.concatenate([np.stack([rng.uniform(-7, 7, n_frame // 2), rng.uniform(-7, 7, n_frame // 2), rng.normal(0, 0.02, n_frame // 2)], 1),
                        np.stack([rng.uniform(-7, 7, n_frame // 2), np.full(n_frame // 2, 6.0), rng.uniform(0, 3, n_frame // 2)], 1)])
    return mpts, f


# ---------------------------------------------------------------- Lie group ----------------------------------------
def test_se3_ops_match_scipy():
    rng = np.random.default_rng(0)
    for _ in range(50):
        a, b = rand_pose(rng), rand_pose(rng)
        pts = rng.normal(size=(7, 3))
        np.testing.assert_allclose(okicp.se3_act(a, pts), rn.act(rn.from_qt(a), pts), atol=1e-13)
        ab = okicp.se3_mul(a, b)
        np.testing.assert_allclose(okicp.se3_act(ab, pts), rn.act(rn.mul(rn.from_qt(a), rn.from_qt(b)), pts), atol=1e-12)
        ai = okicp.se3_inverse(a)
        np.testing.assert_allclose(okicp.se3_act(ai, okicp.se3_act(a, pts)), pts, atol=1e-12)
        assert abs(np.linalg.norm(ab[:4]) - 1.0) < 1e-15


def test_se3_exp_log():
    rng = np.random.default_rng(1)
    for scale in (1.0, 1e-3, 1e-9, 1e-12, 0.0):
        for _ in range(10):
            xi = rng.normal(size=6) * np.array([1, 1, 1, scale, scale, scale])
            xi[3:] *= min(1.0, 2.5 / max(np.linalg.norm(xi[3:]), 1e-300))  # keep the rotation angle below pi (log is principal)
            T = okicp.se3_exp(xi)
            Tr = rn.se3_exp(xi)
            pts = rng.normal(size=(5, 3))
            np.testing.assert_allclose(okicp.se3_act(T, pts), rn.act(Tr, pts), atol=1e-12)
            np.testing.assert_allclose(okicp.se3_log(T), xi, atol=1e-9 if scale else 1e-12)


def test_motion_model_and_theta_zero_quirk():
    # Registration.cpp:159-167.  theta == 0.0 exactly drops the translation: mirrored, not "fixed".
    T = okicp.motion_model([0.3, 0.0])
    np.testing.assert_array_equal(T, [0, 0, 0, 1, 0, 0, 0])
    d, th = 0.25, 0.1
    T = okicp.motion_model([d, th])
    # closed form: translation = 2 d (1-cos th)/th^2 * (cos th, sin th)  
    k = 2 * d * (1 - np.cos(th)) / th**2
    np.testing.assert_allclose(T[4:], [k * np.cos(th), k * np.sin(th), 0.0], atol=1e-15)
    np.testing.assert_allclose(T[:4], [0, 0, np.sin(th / 2), np.cos(th / 2)], atol=1e-16)
    np.testing.assert_allclose(okicp.se3_act(T, np.zeros((1, 3))), rn.act(rn.motion_model([d, th]), np.zeros((1, 3))), atol=1e-14)


# ---------------------------------------------------------------- voxel map -----------------------------------------
def test_map_matches_numpy_restatement():
    mpts, f = small_world()
    o = okicp.VoxelHashMap(1.0, 100.0, 20)
    r = rn.VoxelHashMap(1.0, 100.0, 20)
    o.AddPoints(mpts), r.AddPoints(mpts)
    from conftest import sort_rows
    np.testing.assert_array_equal(sort_rows(o.Pointcloud()), sort_rows(r.Pointcloud()))
    nn_o, d_o = o.GetClosestNeighbor(f)
    nn_r, d_r = r.closest(f)
    np.testing.assert_allclose(d_o, d_r, rtol=0, atol=1e-14)
    np.testing.assert_array_equal(nn_o, nn_r)
    # Update = transform + AddPoints + RemovePointsFarFromLocation (App. A.5/A.6)
    o2, r2 = okicp.VoxelHashMap(0.5, 4.0, 10), rn.VoxelHashMap(0.5, 4.0, 10)
    pose = syn.planar_pose(1.0, -2.0, 0.4)
    o2.Update(mpts[:1500], pose), r2.Update(mpts[:1500], pose)
    o2.Update(mpts[1500:2500], syn.planar_pose(3.0, 1.0, -0.2)), r2.Update(mpts[1500:2500], syn.planar_pose(3.0, 1.0, -0.2))
    a, b = sort_rows(o2.Pointcloud()), sort_rows(r2.Pointcloud())
    assert a.shape == b.shape and a.shape[0] > 50
    np.testing.assert_allclose(a, b, atol=1e-12)


def test_kat4_closest_neighbor_corners_ties_and_misses():
    m = okicp.VoxelHashMap(1.0, 100.0, 20)
    m.AddPoints(np.array([[-0.1, -0.1, -0.1]]))                # lives in voxel (-1,-1,-1)
    nn, d = m.GetClosestNeighbor(np.array([[0.05, 0.05, 0.05]]))  # query in voxel (0,0,0): diagonal neighbour must be found
    np.testing.assert_allclose(nn[0], [-0.1, -0.1, -0.1])
    np.testing.assert_allclose(d[0], np.sqrt(3 * 0.15**2))
    nn, d = m.GetClosestNeighbor(np.array([[5.5, 5.5, 5.5]]))   # nothing in the 27 voxels
    assert d[0] == DBL_MAX and np.all(nn[0] == 0.0)
    nn, d = m.GetClosestNeighbor(np.array([[1.5, 0.5, 0.5]]))   # two voxels away on x: outside the neighbourhood
    assert d[0] == DBL_MAX
    # tie: equidistant points, the one in the earlier shift wins: shift order starts (0,0,0),(1,0,0),(-1,0,0)
    t = okicp.VoxelHashMap(1.0, 100.0, 20)
    t.AddPoints(np.array([[-0.25, 0.5, 0.5], [1.25, 0.5, 0.5]]))  # voxels (-1,0,0) and (1,0,0), both 0.75 from the query
    nn, d = t.GetClosestNeighbor(np.array([[0.5, 0.5, 0.5]]))
    np.testing.assert_array_equal(nn[0], [1.25, 0.5, 0.5])
    # tie inside one bucket: first inserted wins (std::min_element)
    t2 = okicp.VoxelHashMap(1.0, 100.0, 20)
    t2.AddPoints(np.array([[0.75, 0.5, 0.5], [0.25, 0.5, 0.5]]))
    nn, d = t2.GetClosestNeighbor(np.array([[0.5, 0.5, 0.5]]))
    np.testing.assert_array_equal(nn[0], [0.75, 0.5, 0.5])


def test_kat5_add_points_rules():
    cap, vs = 20, 1.0
    res = vs / np.sqrt(cap)
    m = okicp.VoxelHashMap(vs, 100.0, cap)
    # 5x4 grid with spacing > res inside voxel (0,0,0): exactly 20 accepted, the 21st distinct point is dropped
    gx, gy = np.meshgrid(np.arange(5) * 0.23 + 0.02, np.arange(4) * 0.24 + 0.02)
    pts = np.stack([gx.ravel(), gy.ravel(), np.full(20, 0.5)], 1)
    m.AddPoints(pts)
    assert m.num_points() == 20
    m.AddPoints(np.array([[0.5, 0.5, 0.95]]))
    assert m.num_points() == 20
    m2 = okicp.VoxelHashMap(vs, 100.0, cap)
    m2.AddPoints(np.array([[0.1, 0.1, 0.1]]))
    m2.AddPoints(np.array([[0.1 + 0.9 * res, 0.1, 0.1]]))   # closer than map_resolution -> dropped
    assert m2.num_points() == 1
    m2.AddPoints(np.array([[0.1 + 2.0 * res, 0.1, 0.1]]))   # far enough -> kept
    assert m2.num_points() == 2
    m3 = okicp.VoxelHashMap(2.0, 100.0, 16)                 # map_resolution = 2/4 = 0.5 exactly representable
    m3.AddPoints(np.array([[0.25, 0.25, 0.25], [0.75, 0.25, 0.25]]))  # distance == resolution: kept (strict <)
    assert m3.num_points() == 2
    m4 = okicp.VoxelHashMap(vs, 100.0, cap)                 # spacing only applies inside one voxel
    m4.AddPoints(np.array([[0.99, 0.5, 0.5], [1.01, 0.5, 0.5]]))
    assert m4.num_points() == 2 and m4.num_voxels() == 2
    m5 = okicp.VoxelHashMap(vs, 100.0, cap)                 # floor voxelisation of negatives
    m5.AddPoints(np.array([[-0.5, -0.5, -0.5], [-1.0, -1.0, -1.0], [-1e-9, 0.0, 0.0], [-1.0 - 1e-9, -1.0, -1.0]]))
    assert m5.num_voxels() == 3 and m5.num_points() == 4  # floor(-1.0) = -1 (same voxel as -0.5); -1e-9 -> -1; -1-1e-9 -> -2


def test_remove_far_uses_first_point_of_voxel():
    m = okicp.VoxelHashMap(1.0, 10.0, 20)
    m.AddPoints(np.array([[9.95, 0.5, 0.5], [9.55, 0.5, 0.5],      # voxel (9,0,0): first point at 9.95
                          [-9.55, 0.5, 0.5], [-9.95, 0.5, 0.5]]))   # voxel (-10,0,0): first point at -9.55
    m.RemovePointsFarFromLocation(np.array([0.0, 0.5, 0.5]))
    assert m.num_points() == 4
    m.RemovePointsFarFromLocation(np.array([-0.04, 0.5, 0.5]))   # 9.95 + 0.04 = 9.99 < 10, still kept
    assert m.num_points() == 4
    m.RemovePointsFarFromLocation(np.array([-0.06, 0.5, 0.5]))   # first point of voxel 9 now >= 10 -> whole voxel goes
    from conftest import sort_rows
    np.testing.assert_array_equal(sort_rows(m.Pointcloud()), sort_rows(np.array([[-9.55, 0.5, 0.5], [-9.95, 0.5, 0.5]])))


# ---------------------------------------------------------------- registration -------------------------------------
def test_pass_sums_and_registration_match_numpy_restatement():
    mpts, f = small_world(seed=9)
    o, r = okicp.VoxelHashMap(1.0, 100.0, 20), rn.VoxelHashMap(1.0, 100.0, 20)
    o.AddPoints(mpts), r.AddPoints(mpts)
    rng = np.random.default_rng(2)
    for tau in (0.67, 0.25):
        T = syn.planar_pose(0.05, -0.03, 0.01)
        so, cnt = okicp.icp_pass(o, f, T, tau)
        sr = rn.pass_sums(r, f, rn.from_qt(T), tau)
        assert so[6] == sr[6] and 0 < so[6] <= len(f)
        np.testing.assert_allclose(so, sr, rtol=1e-11, atol=1e-10)
        assert cnt[0] == 27 * len(f)
    for adaptive, fixed in ((True, 0.0), (False, 0.0), (False, 5.0)):
        last, rel = syn.planar_pose(0.1, 0.05, 0.02), syn.planar_pose(0.08, 0.0, 0.015)
        reg = okicp.KinematicRegistration(10, 1e-3, 1, adaptive, fixed)
        po = reg.ComputeRobotMotion(f, o, last, rel, 0.67)
        pr, it = rn.compute_robot_motion(f, r, last, rel, 0.67, adaptive=adaptive, fixed_reg=fixed)
        assert reg.last_stats.iterations == it
        np.testing.assert_allclose(po, pr, atol=1e-10)


def test_literal_jacobian_equals_simplified_form():
    # JTJ = [[1,-sy],[-sy,sx^2+sy^2]], JTr = [ex, sx ey - sy ex], e = R^T r
    mpts, f = small_world(seed=11)
    o = okicp.VoxelHashMap(1.0, 100.0, 20)
    o.AddPoints(mpts)
    T = rand_pose(np.random.default_rng(3), planar=True) * np.array([1, 1, 1, 1, 0.01, 0.01, 0.0])
    T[:4] /= np.linalg.norm(T[:4])
    sums, _ = okicp.icp_pass(o, f, T, 0.67)
    q = okicp.se3_act(T, f)
    nn, d = o.GetClosestNeighbor(q)
    keep = d < 0.67
    s, t = f[keep], nn[keep]
    e = okicp.se3_act(okicp.se3_inverse(T), t)  # T^-1 t
    e = s - e
    expect = [keep.sum(), np.sum(-s[:, 1]), np.sum(s[:, 0]**2 + s[:, 1]**2), np.sum(e[:, 0]), np.sum(s[:, 0] * e[:, 1] - s[:, 1] * e[:, 0]),
              np.sum(e * e), keep.sum()]
    np.testing.assert_allclose(sums, expect, rtol=1e-11, atol=1e-9)


def test_kat0_empty_map_returns_prediction():
    last, rel = syn.planar_pose(1, 2, 0.3), syn.planar_pose(0.5, 0, 0.1)
    reg = okicp.KinematicRegistration()
    p = reg.ComputeRobotMotion(np.zeros((10, 3)), okicp.VoxelHashMap(1.0, 100.0, 20), last, rel, 1.0)
    np.testing.assert_array_equal(p, okicp.se3_mul(last, rel))
    assert reg.last_stats.empty_map == 1 and reg.last_stats.iterations == 0


def test_kat1_perfect_alignment_stops_at_once():
    rng = np.random.default_rng(4)
    pts = rng.uniform(-10, 10, (800, 3))
    m = okicp.VoxelHashMap(1.0, 100.0, 20)
    m.AddPoints(pts)
    src = m.Pointcloud()
    reg = okicp.KinematicRegistration()
    p = reg.ComputeRobotMotion(src, m, okicp.IDENTITY, okicp.IDENTITY, 0.5)
    np.testing.assert_array_equal(p, okicp.IDENTITY)  # e = 0 -> b = 0 -> dx = 0 -> theta == 0 -> identity update
    assert reg.last_stats.iterations == 1 and reg.last_stats.converged == 1
    assert np.isfinite(reg.last_stats.beta) and reg.last_stats.beta > 1e300  # 1/epsilon


def test_kat2_kat3_small_shift():
    rng = np.random.default_rng(6)
    pts = rng.uniform(-10, 10, (1500, 3))
    m = okicp.VoxelHashMap(1.0, 100.0, 20)
    m.AddPoints(pts)
    tgt = m.Pointcloud()
    d = 0.02  # << point spacing, so NN recovers the true pairing
    src = tgt - np.array([d, 0, 0])
    # KAT-2: no regularisation -> first step dx = (d, ~0), second step ~0 -> converged after 2 iterations
    reg = okicp.KinematicRegistration(10, 1e-3, 1, False, 0.0)
    p = reg.ComputeRobotMotion(src, m, okicp.IDENTITY, okicp.IDENTITY, 0.5)
    st = reg.last_stats
    assert st.n_corr[0] == len(src)
    np.testing.assert_allclose(st.dx[0][0], d, atol=1e-6)
    assert abs(st.dx[0][1]) < 1e-6
    np.testing.assert_allclose(p[4], d, atol=1e-6)
    assert st.iterations == 2 and st.converged == 1
    # KAT-3: adaptive beta = 1/(d^2+eps); (1+beta) dx0 - ybar dx1 = d  -> displacement is trusted to odometry
    reg = okicp.KinematicRegistration(10, 1e-3, 1, True, 0.0)
    reg.ComputeRobotMotion(src, m, okicp.IDENTITY, okicp.IDENTITY, 0.5)
    st = reg.last_stats
    np.testing.assert_allclose(st.beta, 1.0 / (d * d), rtol=1e-9)
    s5 = np.array(st.sums[0][:5])
    n = st.n_corr[0]
    A = np.array([[s5[0] / n + st.beta, s5[1] / n], [s5[1] / n, s5[2] / n]])
    np.testing.assert_allclose(A @ np.array(st.dx[0]), -s5[3:5] / n, atol=1e-12)
    assert abs(st.dx[0][0]) < d * 1e-2


def test_zero_correspondences_give_nan():
    m = okicp.VoxelHashMap(1.0, 100.0, 20)
    m.AddPoints(np.random.default_rng(7).uniform(-5, 5, (200, 3)))
    reg = okicp.KinematicRegistration()
    p = reg.ComputeRobotMotion(np.full((50, 3), 500.0), m, okicp.IDENTITY, okicp.IDENTITY, 0.5)
    assert reg.last_status == 1 and np.isnan(p).any()


def test_kat6_threshold():
    t = okicp.CorrespondenceThreshold(1.0 / np.sqrt(20), 100.0, True, 1.0)
    np.testing.assert_allclose(t.ComputeThreshold(), 3.0 / np.sqrt(20), rtol=1e-15)  # 0.670820...
    e = 0.3
    t.UpdateOdometryError(np.array([0, 0, 0, 1, e, 0, 0]))
    np.testing.assert_allclose(t.ComputeThreshold(), 3.0 * (1 / np.sqrt(20) + np.sqrt(e * e / (1 + 1e-8))), rtol=1e-14)
    err = syn.planar_pose(0.1, -0.2, 0.05)
    t2 = okicp.CorrespondenceThreshold(0.2, 50.0, True, 1.0)
    t2.UpdateOdometryError(err)
    x = rn.odometry_error_in_point_space(err, 50.0)
    np.testing.assert_allclose(t2.ComputeThreshold(), rn.compute_threshold(0.2, x * x, 1.0 + 1e-8), rtol=1e-12)
    t2.Reset()
    np.testing.assert_allclose(t2.ComputeThreshold(), 0.6, rtol=1e-15)
    assert okicp.CorrespondenceThreshold(0.2, 50.0, False, 1.25).ComputeThreshold() == 1.25


def test_kat7_sharding_invariance_of_the_sums():
    mpts, f = small_world(seed=13, n_frame=1200)
    o = okicp.VoxelHashMap(1.0, 100.0, 20)
    o.AddPoints(mpts)
    T = syn.planar_pose(0.02, 0.01, 0.005)
    full, _ = okicp.icp_pass(o, f, T, 0.67)
    for g in (2, 4, 8):
        parts = [okicp.icp_pass(o, f[len(f) * r // g: len(f) * (r + 1) // g], T, 0.67)[0] for r in range(g)]
        np.testing.assert_allclose(np.sum(parts, 0), full, rtol=1e-12, atol=1e-12)


def test_omp_path_equals_serial():
    mpts, f = small_world(seed=17, n_frame=3000)
    o = okicp.VoxelHashMap(1.0, 100.0, 20)
    o.AddPoints(mpts)
    last, rel = syn.planar_pose(0.1, 0.05, 0.02), syn.planar_pose(0.08, 0.0, 0.015)
    a = okicp.KinematicRegistration(max_num_threads=1).ComputeRobotMotion(f, o, last, rel, 0.67)
    b = okicp.KinematicRegistration(max_num_threads=4).ComputeRobotMotion(f, o, last, rel, 0.67)
    np.testing.assert_allclose(a, b, atol=1e-12)


def test_pre_steps_voxel_downsample_and_preprocess():
    rng = np.random.default_rng(8)
    pts = rng.uniform(-20, 20, (5000, 3))
    ds = okicp.voxel_downsample(pts, 1.5)
    keys = np.floor(pts / 1.5).astype(np.int64)
    _, first = np.unique(keys, axis=0, return_index=True)       # first point per voxel wins (App. A.7)
    from conftest import sort_rows
    np.testing.assert_array_equal(sort_rows(ds), sort_rows(pts[first]))
    rel = syn.pose_mul(syn.planar_pose(0.5, 0.1, 0.05), np.array([0.01, -0.02, 0, np.sqrt(1 - 5e-4), 0, 0, 0.02]))
    ts = rng.uniform(0, 1, len(pts))
    out = okicp.preprocess(pts, ts, rel, 25.0, 2.0, True)
    om = rn.se3_log(rn.from_qt(rel))
    Tinv = rn.inv(rn.from_qt(rel))
    exp = np.array([rn.act(rn.mul(Tinv, rn.se3_exp(t * om)), p[None])[0] for t, p in zip(ts[:300], pts[:300])])
    r = np.linalg.norm(exp, axis=1)
    exp = exp[(r < 25.0) & (r > 2.0)]
    np.testing.assert_allclose(out[:len(exp)], exp, atol=1e-9)
    out2 = okicp.preprocess(pts, None, rel, 25.0, 2.0, False)
    r2 = np.linalg.norm(pts, axis=1)
    np.testing.assert_array_equal(out2, pts[(r2 < 25.0) & (r2 > 2.0)])


# ---------------------------------------------------------------- frozen vectors -----------------------------------
@pytest.mark.parametrize("name", ["a", "b", "c"])
def test_oracle_reproduces_golden(name):
    g = np.load(GOLD)
    m = okicp.VoxelHashMap(float(g[name + "_voxel"]), float(g[name + "_maxrange"]), 20)
    m.AddPoints(g[name + "_map"])
    reg = okicp.KinematicRegistration()
    p = reg.ComputeRobotMotion(g[name + "_frame"], m, g[name + "_last"], g[name + "_rel"], float(g[name + "_tau"]))
    st = reg.last_stats
    assert st.iterations == int(g[name + "_iters"]) and st.converged == int(g[name + "_converged"])
    np.testing.assert_allclose(p, g[name + "_pose"], atol=1e-12)
    np.testing.assert_array_equal(np.array(st.n_corr[:st.iterations]), g[name + "_ncorr"])


def test_golden_case_a_matches_numpy_restatement():
    g = np.load(GOLD)
    r = rn.VoxelHashMap(float(g["a_voxel"]), float(g["a_maxrange"]), 20)
    r.AddPoints(g["a_map"])
    p, it = rn.compute_robot_motion(g["a_frame"], r, g["a_last"], g["a_rel"], float(g["a_tau"]))
    assert it == int(g["a_iters"])
    np.testing.assert_allclose(p, g["a_pose"], atol=1e-9)


def test_association_per_query_is_the_pass_the_sums_come_from():
    """okicp.associate (DataAssociation per query, Registration.cpp:73-77) - the checker of kicp_pass_correspondences: its accepted set is
    the correspondence count of the fused pass, its neighbours and distances are GetClosestNeighbor's, acceptance is strict."""
    rng = np.random.default_rng(5)
    pts = rng.uniform(-6, 6, (3000, 3))
    m = okicp.VoxelHashMap(1.0, 100.0, 20)
    m.AddPoints(pts)
    frame = rng.uniform(-6, 6, (500, 3))
    pose = np.array([0.0, 0.0, np.sin(0.05), np.cos(0.05), 0.1, -0.2, 0.0])
    acc, nn, d = okicp.associate(m, frame, pose, 0.4)
    sums, _ = okicp.icp_pass(m, frame, pose, 0.4)
    assert acc.sum() == sums[6] and 0 < acc.sum() < len(frame)
    nn2, d2 = m.GetClosestNeighbor(okicp.se3_act(pose, frame))
    assert np.array_equal(nn, nn2) and np.array_equal(d, d2) and np.array_equal(acc, d < 0.4)
    k = int(np.argmax(acc))
    assert not okicp.associate(m, frame[k:k + 1], pose, d[k])[0][0] and okicp.associate(m, frame[k:k + 1], pose, np.nextafter(d[k], 1.0))[0][0]
