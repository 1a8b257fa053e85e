"""GPU parity tests proper: the HIP path (through the C-ABI) against the CPU checkers on identical seeded inputs: the
oracle (oracle/kicp_oracle.cpp) and - wherever a registration result is compared - the reference's own sources
(oracle/_ref/libkicp_ref.so, prebuilt; tests/checkers.py), plus the frozen outputs of that build (tests/golden/ref_outputs.npz).

Tolerances: the north star asks for poses within 1e-4 m / 1e-4 rad.  The HIP path computes in fp64 in the
reference's operation order, so we assert far tighter: 1e-9 on poses, 1e-10 relative on the per-pass sums
(only the summation order differs)."""
import numpy as np
import pytest

import kinematic_icp_amd as K
from checkers import GOLDEN, okicp, ref, ref_available, ref_map_like
from kinematic_icp_amd import synthetic as syn

pytestmark = pytest.mark.gpu

POSE_TOL = 1e-9
SUM_RTOL = 1e-10
# The builds of the generic pass kernel (kicp_reg_launch.hip launch_pass; all of 256-thread workgroups since round 6) as
# (sub-lanes per query, latency_kernel): the scan-size default (two sub-lanes sharing every bucket on this 16k scan), one lane per query
# at four waves per SIMD (what large scans and batches in flight run), the same as the two-voxels-per-round build (what scans of up
# to 131 072 points run one call at a time), and four sub-lanes per query (what very small scans run); None = the library's choice
VARIANTS = [(None, None), (1, 0), (1, 2), (2, None), (4, None)]


@pytest.fixture(scope="module")
def case1():
    cfg, scene, scans, rng = syn.make_case("cfg1", n_scans=3)
    gmap = K.VoxelHashMap(cfg.voxel_size, cfg.max_range, cfg.max_points_per_voxel)
    syn.build_map_points(scene, cfg, gmap.AddPoints, gmap.num_points, rng)
    omap = okicp.VoxelHashMap(cfg.voxel_size, cfg.max_range, cfg.max_points_per_voxel)
    omap.AddPoints(gmap.Pointcloud())
    return cfg, scans, gmap, omap


@pytest.fixture(scope="module")
def case1_ref(case1):
    """the same map inside the reference build (None when oracle/_ref is absent: the test then says so and skips that part)"""
    return ref_map_like(case1[3]) if ref_available() else None


def test_reference_build_is_present():
    """The GPU box must carry oracle/_ref/libkicp_ref.so (built where /root/reference exists, shipped with the snapshot)."""
    ref()


def _reg(lanes=None, latency=None, **kw):
    reg = K.KinematicRegistration(**kw)
    reg.set_option("small", 0)  # (the generic pass kernel: the small-scan kernels have tests/test_gpu_small.py)
    if lanes is not None:
        reg.set_option("lanes_per_query", lanes)
    if latency is not None:
        reg.set_option("latency_kernel", latency)
    return reg


def test_device_present():
    assert K.device_count() >= 1


def test_closest_neighbor_matches_oracle(case1):
    cfg, scans, gmap, omap = case1
    rng = np.random.default_rng(7)
    q = np.concatenate([scans[0]["frame"][:4000] + rng.normal(0, 0.3, (4000, 3)), rng.uniform(-200, 200, (500, 3))])
    nn_g, d_g = gmap.GetClosestNeighbor(q)
    nn_o, d_o = omap.GetClosestNeighbor(q)
    assert np.array_equal(d_g, d_o)  # bit-exact: same fp64 operations
    assert np.array_equal(nn_g, nn_o)
    assert (d_o == np.finfo(np.float64).max).any()  # some queries have no candidate


@pytest.mark.parametrize("variant", VARIANTS)
def test_pass_sums_match_oracle(case1, variant):
    cfg, scans, gmap, omap = case1
    reg = _reg(*variant)
    for s in scans:
        guess = syn.pose_mul(s["last_pose"], s["rel_odom"])
        for tau in (cfg.first_frame_tau(), 0.2):
            g = reg.pass_sums(s["frame"], gmap, guess, tau)
            o, _ = okicp.icp_pass(omap, s["frame"], guess, tau)
            assert g[6] == o[6]  # identical accept/reject decisions
            np.testing.assert_allclose(g, o, rtol=SUM_RTOL, atol=1e-9)


@pytest.mark.parametrize("variant", VARIANTS)
@pytest.mark.parametrize("wait", [0, 1])
def test_registration_matches_oracle(case1, case1_ref, variant, wait):
    cfg, scans, gmap, omap = case1
    reg = _reg(*variant)
    reg.set_option("wait", wait)  # 0 (default): poll the tagged rows in host memory; 1: hipStreamSynchronize
    oreg = okicp.KinematicRegistration()
    for s in scans:
        # make the initial guess bad enough to need several iterations
        rel = syn.pose_mul(s["rel_odom"], syn.planar_pose(0.2, 0.0, np.deg2rad(1.5)))
        pose = reg.ComputeRobotMotion(s["frame"], gmap, s["last_pose"], rel, cfg.first_frame_tau())
        ref = oreg.ComputeRobotMotion(s["frame"], omap, s["last_pose"], rel, cfg.first_frame_tau())
        assert reg.last_stats.iterations == oreg.last_stats.iterations
        assert reg.last_stats.converged == oreg.last_stats.converged
        np.testing.assert_allclose(pose, ref, rtol=0, atol=POSE_TOL)
        k = reg.last_stats.iterations
        np.testing.assert_allclose(np.array(reg.last_stats.n_corr[:k]), np.array(oreg.last_stats.n_corr[:k]))
        if case1_ref is not None:  # the reference's own Registration.cpp
            theirs = rkicp_registration().ComputeRobotMotion(s["frame"], case1_ref, s["last_pose"], rel, cfg.first_frame_tau())
            np.testing.assert_allclose(pose, theirs, rtol=0, atol=POSE_TOL)


def rkicp_registration(**kw):
    return ref().KinematicRegistration(**kw)


def test_empty_map_returns_prediction(case1):
    cfg, scans, gmap, omap = case1
    s = scans[0]
    reg = K.KinematicRegistration()
    empty = K.VoxelHashMap(1.0, 100.0, 20)
    pose = reg.ComputeRobotMotion(s["frame"], empty, s["last_pose"], s["rel_odom"], 1.0)
    ref = okicp.KinematicRegistration().ComputeRobotMotion(s["frame"], okicp.VoxelHashMap(1.0, 100.0, 20), s["last_pose"], s["rel_odom"], 1.0)
    np.testing.assert_allclose(pose, ref, rtol=0, atol=1e-15)
    assert reg.last_stats.empty_map == 1
    if ref_available():
        r = checkers_ref()
        assert np.array_equal(pose, r.KinematicRegistration().ComputeRobotMotion(s["frame"], r.VoxelHashMap(1.0, 100.0, 20), s["last_pose"], s["rel_odom"], 1.0))


def checkers_ref():
    return ref()


def test_zero_correspondences_gives_nan_like_reference(case1, case1_ref):
    cfg, scans, gmap, omap = case1
    s = scans[0]
    reg = K.KinematicRegistration()
    far = s["frame"] + np.array([0.0, 0.0, 500.0])
    pose = reg.ComputeRobotMotion(far, gmap, s["last_pose"], s["rel_odom"], cfg.first_frame_tau())
    assert reg.last_status == K.KICP_WARN_NO_CORRESPONDENCES
    assert np.isnan(pose).any()
    oreg = okicp.KinematicRegistration()
    ref = oreg.ComputeRobotMotion(far, omap, s["last_pose"], s["rel_odom"], cfg.first_frame_tau())
    assert np.isnan(ref).any()
    if case1_ref is not None:
        rreg = rkicp_registration()
        assert np.isnan(rreg.ComputeRobotMotion(far, case1_ref, s["last_pose"], s["rel_odom"], cfg.first_frame_tau())).any() and rreg.last_status == 1


def test_device_frame_equals_host_frame(case1):
    cfg, scans, gmap, omap = case1
    s = scans[1]
    reg = K.KinematicRegistration()
    a = reg.ComputeRobotMotion(s["frame"], gmap, s["last_pose"], s["rel_odom"], cfg.first_frame_tau())
    b = reg.ComputeRobotMotion(K.DeviceFrame(s["frame"]), gmap, s["last_pose"], s["rel_odom"], cfg.first_frame_tau())
    assert np.array_equal(a, b)  # deterministic: fixed-order reductions


def test_all_variants_bit_identical(case1):
    """Exact (integer) accumulation: every build of the pass kernel - generic and small-scan -, either way of waiting and either
    launch path give the same bits."""
    cfg, scans, gmap, omap = case1
    s = scans[2]
    rel = syn.pose_mul(s["rel_odom"], syn.planar_pose(0.1, 0.0, np.deg2rad(0.8)))
    poses = []
    for variant in VARIANTS:
        for wait in (0, 1):
            for aql in (1, 0):
                reg = _reg(*variant)
                reg.set_option("wait", wait), reg.set_option("aql", aql)
                poses.append(reg.ComputeRobotMotion(s["frame"], gmap, s["last_pose"], rel, cfg.first_frame_tau()))
                assert reg.last_stats.iterations > 1
    small = K.KinematicRegistration()  # (this 16k scan does not take the small-scan path by default: a 4 096-point part of it does)
    part = s["frame"][:4096]
    a = small.ComputeRobotMotion(part, gmap, s["last_pose"], rel, cfg.first_frame_tau())
    assert small.get_option("small_active") != 0.0
    poses_part = [_reg(*v).ComputeRobotMotion(part, gmap, s["last_pose"], rel, cfg.first_frame_tau()) for v in VARIANTS]
    for p in poses[1:]:
        assert np.array_equal(p, poses[0])
    for p in poses_part:
        assert np.array_equal(p, a)


def test_random_order_input(case1):
    cfg, scans, gmap, omap = case1
    s = scans[0]
    perm = np.random.default_rng(3).permutation(len(s["frame"]))
    reg = K.KinematicRegistration()
    a = reg.ComputeRobotMotion(s["frame"], gmap, s["last_pose"], s["rel_odom"], cfg.first_frame_tau())
    b = reg.ComputeRobotMotion(s["frame"][perm], gmap, s["last_pose"], s["rel_odom"], cfg.first_frame_tau())
    assert np.array_equal(a, b)  # order-independent sums


GOLD = __import__("os").path.join(GOLDEN, "registration_small.npz")
REF_GOLD = __import__("os").path.join(GOLDEN, "ref_outputs.npz")
REG_VARIANTS = (("default", dict()), ("fixed0", dict(use_adaptive_odometry_regularization=False, fixed_regularization=0.0)),
                ("fixed5", dict(use_adaptive_odometry_regularization=False, fixed_regularization=5.0)),
                ("it3", dict(max_num_iteration=3)), ("loose", dict(convergence_criterion=1e-2)))


@pytest.mark.parametrize("name", ["a", "b", "c"])
def test_frozen_outputs_of_the_reference_build(name):
    """tests/golden/ref_outputs.npz = what the reference's own Registration.cpp returned (tests/golden/make_ref_golden.py):
    five parameter sets x two thresholds per case, and the neighbour queries of the first pass."""
    g, r = np.load(GOLD), np.load(REF_GOLD)
    m = K.VoxelHashMap(float(g[name + "_voxel"]), float(g[name + "_maxrange"]), 20)
    m.AddPoints(g[name + "_map"])
    for vname, kw in REG_VARIANTS:
        for tau_scale in (1.0, 0.4):
            reg = K.KinematicRegistration(**kw)
            p = reg.ComputeRobotMotion(g[name + "_frame"], m, g[name + "_last"], g[name + "_rel"], float(g[name + "_tau"]) * tau_scale)
            np.testing.assert_allclose(p, r["reg_%s_%s_%g" % (name, vname, tau_scale)], rtol=0, atol=POSE_TOL)
    q = okicp.se3_act(okicp.se3_mul(g[name + "_last"], g[name + "_rel"]), g[name + "_frame"][::7])
    nn, d = m.GetClosestNeighbor(q)
    assert np.array_equal(nn, r["nn_" + name]) and np.array_equal(d, r["nnd_" + name])


@pytest.mark.parametrize("name", ["a", "b", "c"])
@pytest.mark.parametrize("kernel", [(None, None), (1, 0)])
def test_golden_vectors(name, kernel):
    """The committed fixtures (multi-iteration small cases, incl. two that exhaust max_num_iterations)."""
    g = np.load(GOLD)
    m = K.VoxelHashMap(float(g[name + "_voxel"]), float(g[name + "_maxrange"]), 20)
    m.AddPoints(g[name + "_map"])
    reg = _reg(*kernel) if kernel[0] is not None else K.KinematicRegistration()  # the library's own choice (small-scan kernels) / the generic kernel
    p = reg.ComputeRobotMotion(g[name + "_frame"], m, g[name + "_last"], g[name + "_rel"], float(g[name + "_tau"]))
    st = reg.last_stats
    assert st.iterations == int(g[name + "_iters"]) and st.converged == int(g[name + "_converged"])
    np.testing.assert_allclose(p, g[name + "_pose"], rtol=0, atol=POSE_TOL)
    np.testing.assert_array_equal(np.array(st.n_corr[:st.iterations]), g[name + "_ncorr"])
    np.testing.assert_allclose(np.array([list(st.dx[i]) for i in range(st.iterations)]), g[name + "_dx"], rtol=0, atol=1e-10)
    np.testing.assert_allclose(st.beta, float(g[name + "_beta"]), rtol=1e-10)
    s0 = reg.pass_sums(g[name + "_frame"], m, syn.pose_mul(g[name + "_last"], g[name + "_rel"]), float(g[name + "_tau"]))
    np.testing.assert_allclose(s0, g[name + "_sums0"], rtol=SUM_RTOL, atol=1e-9)


@pytest.mark.parametrize("kernel", [(None, None), (1, 0)])
def test_shard_words_add_up_exactly(case1, kernel):
    """G-GPU emulation on one device: the limb words of disjoint shards sum to the words' value of the whole scan,
    bit for bit, for G in {2,4,8} -- the property that makes the multi-GPU pose independent of G."""
    from kinematic_icp_amd import sharding as sh
    cfg, scans, gmap, omap = case1
    s = scans[0]
    guess = syn.pose_mul(s["last_pose"], s["rel_odom"])
    reg = _reg(*kernel)
    tau = cfg.first_frame_tau()
    full = reg.pass_words(s["frame"], gmap, guess, tau)
    total = [sh.from_limbs(full[3 * i:3 * i + 3]) for i in range(7)]
    assert total[6] == int(reg.pass_sums(s["frame"], gmap, guess, tau)[6]) << 40
    for g in (2, 4, 8):
        words = np.sum([reg.pass_words(s["frame"][slice(*sh.shard_bounds(len(s["frame"]), g, r))], gmap, guess, tau) for r in range(g)], axis=0)
        assert [sh.from_limbs(words[3 * i:3 * i + 3]) for i in range(7)] == total
    np.testing.assert_allclose(sh.unpack(full), okicp.icp_pass(omap, s["frame"], guess, tau)[0], rtol=1e-11, atol=1e-9)


def test_single_rank_communicator_and_callback(case1):
    """The multi-GPU code path (limb publish -> all-reduce -> separate solve kernel) with world size 1:
    built-in RCCL communicator and user callback both reproduce the single-GPU bits."""
    cfg, scans, gmap, omap = case1
    s = scans[1]
    rel = syn.pose_mul(s["rel_odom"], syn.planar_pose(0.1, 0.0, np.deg2rad(0.8)))
    tau = cfg.first_frame_tau()
    base = K.KinematicRegistration().ComputeRobotMotion(s["frame"], gmap, s["last_pose"], rel, tau)
    reg = K.KinematicRegistration()
    reg.comm_init(1, 0, K.comm_unique_id())
    a = reg.ComputeRobotMotion(s["frame"], gmap, s["last_pose"], rel, tau)
    assert np.array_equal(a, base)
    # a BATCH over the communicator: four scans in flight, lane j's all-reduces on a sub-communicator of its own (ncclCommSplit on first
    # use), every pass = device-side tree -> all-reduce -> totals to the host on the lane's stream; the bits of the single-GPU batch
    frames = [K.DeviceFrame(x["frame"]) for x in scans]
    rels = [syn.pose_mul(x["rel_odom"], syn.planar_pose(0.02 * k, 0.0, np.deg2rad(0.2 * k))) for k, x in enumerate(scans)]
    order = [k % len(scans) for k in range(13)]
    plain = K.KinematicRegistration()
    want_batch = plain.prepare_batch([frames[k] for k in order], [scans[k]["last_pose"] for k in order], [rels[k] for k in order])
    want = plain.ComputeRobotMotionBatch(want_batch, gmap, tau).copy()
    batch = reg.prepare_batch([frames[k] for k in order], [scans[k]["last_pose"] for k in order], [rels[k] for k in order])
    for _ in range(2):
        before = reg.get_option("batch_queue_passes")
        got = reg.ComputeRobotMotionBatch(batch, gmap, tau).copy()
        assert np.array_equal(got, want) and list(batch.iterations) == list(want_batch.iterations)
        assert reg.get_option("batch_queue_passes") >= before + sum(want_batch.iterations)  # (the lanes served every pass)
    assert np.array_equal(reg.ComputeRobotMotion(s["frame"], gmap, s["last_pose"], rel, tau), base)  # single calls go on over the handle's own communicator
    reg.comm_destroy()
    assert np.array_equal(reg.ComputeRobotMotion(s["frame"], gmap, s["last_pose"], rel, tau), base)
    calls = []
    reg2 = K.KinematicRegistration()
    reg2.set_allreduce(lambda ptr, count, stream: calls.append((ptr != 0, count)))  # world size 1: in-place sum is a no-op
    b = reg2.ComputeRobotMotion(s["frame"], gmap, s["last_pose"], rel, tau)
    assert np.array_equal(b, base)
    assert calls and all(ok and c == 24 for ok, c in calls) and len(calls) >= reg2.last_stats.iterations


@pytest.mark.parametrize("kernel", [(None, None), (1, 2)])
def test_registration_after_updates_with_pruning(kernel):
    """Map built by a sequence of Update(points, pose) calls that also prune (RemovePointsFarFromLocation), re-using freed
    buckets: the HBM mirror, its halo entries and neighbour-occupancy masks must track every change."""
    rng = np.random.Generator(np.random.PCG64(99))
    scene = syn.make_scene(rng, half=30.0, height=5.0, n_boxes=14, box_xy=(2.0, 6.0), box_z=(1.5, 4.0), keep_clear=3.0)
    dirs = syn.beam_directions(16, 512, (-22.0, 6.0))
    vs, max_range = 0.5, 12.0  # small range: voxels leave the map as the robot drives
    gmap, omap = K.VoxelHashMap(vs, max_range, 20), okicp.VoxelHashMap(vs, max_range, 20)
    rmap = ref().VoxelHashMap(vs, max_range, 20) if ref_available() else None
    reg, oreg = _reg(*kernel), okicp.KinematicRegistration()
    pose = syn.planar_pose(-18.0, -15.0, 0.6)
    removed_any = False
    uploads = []
    for k in range(14):
        step = syn.planar_pose(2.2, 0.0, np.deg2rad(3.0))
        true_next = syn.pose_mul(pose, step)
        scan = syn.make_scan(scene, true_next, dirs, 1.0, rng)
        scan = scan[np.linalg.norm(scan, axis=1) < max_range]
        if k > 0:
            rel = syn.pose_mul(step, syn.planar_pose(0.05, 0.0, np.deg2rad(0.4)))
            a = reg.ComputeRobotMotion(scan, gmap, pose, rel, 3 * vs / np.sqrt(20))
            b = oreg.ComputeRobotMotion(scan, omap, pose, rel, 3 * vs / np.sqrt(20))
            assert reg.last_stats.iterations == oreg.last_stats.iterations
            np.testing.assert_allclose(a, b, rtol=0, atol=POSE_TOL)
            k_it = reg.last_stats.iterations
            np.testing.assert_array_equal(np.array(reg.last_stats.n_corr[:k_it]), np.array(oreg.last_stats.n_corr[:k_it]))
            if rmap is not None:
                np.testing.assert_allclose(a, rkicp_registration().ComputeRobotMotion(scan, rmap, pose, rel, 3 * vs / np.sqrt(20)), rtol=0, atol=POSE_TOL)
        if k > 0:
            uploads.append(gmap.last_upload())
        before = gmap.num_voxels()
        gmap.Update(scan, true_next), omap.Update(scan, true_next)
        if rmap is not None:
            rmap.Update(scan, true_next)
            assert (rmap.num_points(), rmap.num_voxels()) == (gmap.num_points(), gmap.num_voxels())
        removed_any |= gmap.num_voxels() < before + 1 and k > 3
        assert (gmap.num_points(), gmap.num_voxels()) == (omap.num_points(), omap.num_voxels())
        pose = true_next
    nn_g, d_g = gmap.GetClosestNeighbor(scan[:2000])
    nn_o, d_o = omap.GetClosestNeighbor(scan[:2000])
    assert np.array_equal(d_g, d_o) and np.array_equal(nn_g, nn_o)
    assert omap.num_points() < 14 * len(scan) // 4  # the sliding window really dropped old voxels
    # most per-frame mirror updates were deltas (changed slots/buckets only), far smaller than a full re-send
    deltas = [b for b, full in uploads if not full]
    fulls = [b for b, full in uploads if full]
    assert len(deltas) >= len(uploads) // 2 and fulls and max(deltas) < max(fulls)


def test_hand_off_modes_and_tag_wraparound(case1):
    """The tagged-row hand-off polled in host memory (default) and the stream-sync wait give the same bits, also across the 16-bit
    pass tag's wrap-around (where every buffer holding tagged words is cleared) and for scans of changing size."""
    cfg, scans, gmap, omap = case1
    tau = cfg.first_frame_tau()
    rel = [syn.pose_mul(s["rel_odom"], syn.planar_pose(0.1, 0.0, np.deg2rad(0.8))) for s in scans]
    ref = K.KinematicRegistration()
    sizes = [len(scans[0]["frame"]), 5000, 64, 12345, 1]
    expected = [ref.ComputeRobotMotion(scans[i % 3]["frame"][:n], gmap, scans[i % 3]["last_pose"], rel[i % 3], tau) for i, n in enumerate(sizes)]
    assert ref.last_stats.iterations >= 1
    for wait in (0, 1):
        reg = K.KinematicRegistration()
        reg.set_option("wait", wait)
        reg.set_option("debug_tag", 65535 - 7)  # a few passes before the wrap
        assert reg.get_option("debug_tag") == 65528
        for rounds in range(4):
            for i, n in enumerate(sizes):
                pose = reg.ComputeRobotMotion(scans[i % 3]["frame"][:n], gmap, scans[i % 3]["last_pose"], rel[i % 3], tau)
                assert np.array_equal(pose, expected[i], equal_nan=True), (wait, rounds, i)
        assert 0 < reg.get_option("debug_tag") < 65528  # the wrap happened inside the loop


def test_batch_call_equals_a_loop_of_single_calls(case1):
    """kicp_register_device_batch = ComputeRobotMotion scan after scan (bits), incl. a scan that needs several iterations,
    an empty frame (warning code) and the iteration counts."""
    cfg, scans, gmap, omap = case1
    tau = cfg.first_frame_tau()
    reg = K.KinematicRegistration()
    far = syn.pose_mul(scans[1]["rel_odom"], syn.planar_pose(0.1, 0.0, np.deg2rad(1.0)))
    items = [(scans[0]["frame"], scans[0]["last_pose"], scans[0]["rel_odom"]), (scans[1]["frame"], scans[1]["last_pose"], far),
             (scans[2]["frame"][:777], scans[2]["last_pose"], scans[2]["rel_odom"]), (scans[0]["frame"], scans[2]["last_pose"], far)]
    frames = [K.DeviceFrame(f) for f, _, _ in items]
    single, iters = [], []
    for fr, (_, last, rel) in zip(frames, items):
        single.append(reg.ComputeRobotMotion(fr, gmap, last, rel, tau))
        iters.append(reg.last_stats.iterations)
    batch = reg.prepare_batch(frames, [i[1] for i in items], [i[2] for i in items])
    out = reg.ComputeRobotMotionBatch(batch, gmap, tau)
    assert np.array_equal(out, np.array(single)) and list(batch.iterations) == iters and max(iters) > 1
    for (f, last, rel), pose in zip(items, out):
        np.testing.assert_allclose(pose, okicp.KinematicRegistration().ComputeRobotMotion(f, omap, last, rel, tau), rtol=0, atol=POSE_TOL)
    # a frame without points: NaN pose and the warning code, as the single call reports it; the rest of the batch still runs
    empty = K.DeviceFrame(np.zeros((0, 3)))
    b2 = reg.prepare_batch([empty, frames[0]], [items[0][1], items[0][1]], [items[0][2], items[0][2]])
    out2 = reg.ComputeRobotMotionBatch(b2, gmap, tau)
    assert reg.last_status == K.KICP_WARN_NO_CORRESPONDENCES and np.isnan(out2[0]).any() and np.array_equal(out2[1], single[0])


def test_resident_generic_kernel_equals_the_plain_launches(case1):
    """Scans beyond the small-scan kernels (here 16 384 points) keep the generic kernel RESIDENT for a call's later iterations
    (k_pass_resident, commands through the BAR): bits, iteration counts and per-iteration statistics equal one launch per
    iteration (option small = 0) and the oracle; adaptive and forced residency; more iterations than one launch serves; a
    host that is late with a command (the workgroups leave, the marked group rows make the host launch afresh); tag wrap."""
    cfg, scans, gmap, omap = case1
    tau = cfg.first_frame_tau()
    rels = [syn.pose_mul(s["rel_odom"], syn.planar_pose(0.2, 0.0, np.deg2rad(1.5))) for s in scans]
    frames = [K.DeviceFrame(s["frame"]) for s in scans]
    plain = K.KinematicRegistration()
    plain.set_option("small", 0)
    want = [plain.ComputeRobotMotion(f, gmap, s["last_pose"], rel, tau) for f, s, rel in zip(frames, scans, rels)]
    iters = []
    for f, s, rel, w in zip(frames, scans, rels, want):
        np.testing.assert_allclose(w, okicp.KinematicRegistration().ComputeRobotMotion(s["frame"], omap, s["last_pose"], rel, tau), rtol=0, atol=POSE_TOL)
        plain.ComputeRobotMotion(f, gmap, s["last_pose"], rel, tau)
        iters.append(plain.last_stats.iterations)
    assert min(iters) > 2
    for resident in (1, 2):
        reg = K.KinematicRegistration()
        reg.set_option("small_resident", resident)
        for rounds in range(2):
            for f, s, rel, w, k in zip(frames, scans, rels, want, iters):
                got = reg.ComputeRobotMotion(f, gmap, s["last_pose"], rel, tau)
                assert np.array_equal(got, w) and reg.last_stats.iterations == k
                assert reg.get_option("small_active") == 0.0 and reg.get_option("resident_passes") >= k - 1  # (adaptive: the first pass may be a plain launch)
        # a scan that converges at once right after: no resident launch is wasted on it when adaptive
        easy = reg.ComputeRobotMotion(frames[0], gmap, scans[0]["last_pose"], scans[0]["rel_odom"], tau)
        assert np.array_equal(easy, plain.ComputeRobotMotion(frames[0], gmap, scans[0]["last_pose"], scans[0]["rel_odom"], tau))
    # never converging: max_num_iterations reached, 60 > the 48 passes one launch serves
    for max_it in (1, 2, 60):
        kw = dict(max_num_iteration=max_it, convergence_criterion=0.0)
        a, b = K.KinematicRegistration(**kw), K.KinematicRegistration(**kw)
        b.set_option("small", 0), a.set_option("small_resident", 2)
        assert np.array_equal(a.ComputeRobotMotion(frames[1], gmap, scans[1]["last_pose"], rels[1], tau), b.ComputeRobotMotion(frames[1], gmap, scans[1]["last_pose"], rels[1], tau))
        assert a.last_stats.iterations == b.last_stats.iterations == max_it
        assert a.get_option("resident_passes") == (max_it if max_it > 1 else 0)  # (a launch that can serve one pass only goes out as the plain kernel)
    # a late host
    reg = K.KinematicRegistration()
    reg.set_option("small_resident", 2), reg.set_option("small_timeout_us", 200.0)
    for cmd in (1, 0):
        reg.set_option("small_cmd", cmd)
        for stall in (2000.0, 150.0, 260.0):
            before = reg.get_option("small_relaunches")
            reg.set_option("debug_stall_us", stall)
            assert np.array_equal(reg.ComputeRobotMotion(frames[2], gmap, scans[2]["last_pose"], rels[2], tau), want[2]) and reg.last_stats.iterations == iters[2]
            if stall == 2000.0:
                assert reg.get_option("small_relaunches") == before + 1
    assert np.array_equal(reg.ComputeRobotMotion(frames[2], gmap, scans[2]["last_pose"], rels[2], tau), want[2])
    # the 16-bit tag wraps inside a reserved range
    reg.set_option("debug_tag", 65535 - 12)
    for _ in range(6):
        assert np.array_equal(reg.ComputeRobotMotion(frames[0], gmap, scans[0]["last_pose"], rels[0], tau), want[0])
    assert reg.get_option("debug_tag") < 1000
    # explicit kernel-shape options keep the plain kernels
    shaped = K.KinematicRegistration()
    shaped.set_option("lanes_per_query", 2)
    assert np.array_equal(shaped.ComputeRobotMotion(frames[0], gmap, scans[0]["last_pose"], rels[0], tau), want[0]) and shaped.get_option("resident_passes") == 0.0


def test_concurrent_lanes_equal_the_sequential_batch(case1):
    """kicp_register_device_concurrent (independent scans, several in flight, one host thread and one handle per lane): every
    pose and iteration count equals the sequential batch's, bit for bit, for 1, 2, 3 and 8 lanes, with scans of both the
    small-scan and the generic path mixed; the argument checks."""
    cfg, scans, gmap, omap = case1
    tau = cfg.first_frame_tau()
    rng = np.random.default_rng(3)
    items = []
    for k in range(40):
        s = scans[k % 3]
        rel = syn.pose_mul(s["rel_odom"], syn.planar_pose(rng.uniform(-0.15, 0.15), 0.0, np.deg2rad(rng.uniform(-1.0, 1.0))))
        items.append((s["frame"][:(900 if k % 5 == 0 else len(s["frame"]))], s["last_pose"], rel))
    frames = [K.DeviceFrame(f) for f, _, _ in items]
    regs = [K.KinematicRegistration() for _ in range(8)]
    batch = regs[0].prepare_batch(frames, [i[1] for i in items], [i[2] for i in items])
    want, want_it = regs[0].ComputeRobotMotionBatch(batch, gmap, tau).copy(), batch.iterations.copy()
    assert max(want_it) > 2 and min(want_it) >= 1
    for lanes in (1, 2, 3, 8):
        batch.out[:] = 0.0
        batch.iterations[:] = 0
        out = regs[0].ComputeRobotMotionConcurrent(regs[1:lanes], batch, gmap, tau)
        assert np.array_equal(out, want) and np.array_equal(batch.iterations, want_it), lanes
    with pytest.raises(K.KicpError) as e:  # a handle may serve one lane only
        regs[0].ComputeRobotMotionConcurrent([regs[1], regs[1]], batch, gmap, tau)
    assert e.value.code == K.KICP_ERR_ARG
    # a frame without points: the warning code comes back, the other scans are unaffected
    b2 = regs[0].prepare_batch([K.DeviceFrame(np.zeros((0, 3))), frames[1], frames[2]], [items[0][1], items[1][1], items[2][1]], [items[0][2], items[1][2], items[2][2]])
    out2 = regs[0].ComputeRobotMotionConcurrent(regs[1:3], b2, gmap, tau)
    assert regs[0].last_status == K.KICP_WARN_NO_CORRESPONDENCES and np.isnan(out2[0]).any() and np.array_equal(out2[1:], want[1:3])


def test_aql_dispatch_equals_hip_launch(case1):
    """The pass kernel dispatched with hand-written AQL packets on the handle's own HSA queue (default, kicp_aql.hpp) and the
    same kernel launched through hipLaunchKernelGGL: the same bits, for every sub-lane variant, across switches between the
    two paths, with a host frame in between (HIP upload on the stream, then AQL again)."""
    cfg, scans, gmap, omap = case1
    tau = cfg.first_frame_tau()
    far = syn.pose_mul(scans[1]["rel_odom"], syn.planar_pose(0.1, 0.0, np.deg2rad(1.0)))
    frames = [K.DeviceFrame(s["frame"]) for s in scans]
    for lanes in (None, 1, 2, 4):
        a, b = _reg(lanes), _reg(lanes)
        b.set_option("aql", 0)
        for k, (fr, s) in enumerate(zip(frames, scans)):
            rel = far if k == 1 else s["rel_odom"]
            pa, pb = a.ComputeRobotMotion(fr, gmap, s["last_pose"], rel, tau), b.ComputeRobotMotion(fr, gmap, s["last_pose"], rel, tau)
            assert np.array_equal(pa, pb) and a.last_stats.iterations == b.last_stats.iterations
            assert b.get_option("aql_active") == 0.0
        assert a.get_option("aql_active") == 1.0, "the AQL path did not come up on this box (KICP_TRACE=1 says why)"
        # host frame (staged upload through the HIP stream), then device frames again; toggling the option on one handle
        ph = a.ComputeRobotMotion(scans[0]["frame"], gmap, scans[0]["last_pose"], scans[0]["rel_odom"], tau)
        a.set_option("aql", 0)
        p0 = a.ComputeRobotMotion(frames[0], gmap, scans[0]["last_pose"], scans[0]["rel_odom"], tau)
        a.set_option("aql", 1)
        p1 = a.ComputeRobotMotion(frames[0], gmap, scans[0]["last_pose"], scans[0]["rel_odom"], tau)
        assert np.array_equal(ph, p0) and np.array_equal(p0, p1) and a.get_option("aql_active") == 1.0
    np.testing.assert_allclose(p1, okicp.KinematicRegistration().ComputeRobotMotion(scans[0]["frame"], omap, scans[0]["last_pose"], scans[0]["rel_odom"], tau),
                               rtol=0, atol=POSE_TOL)
