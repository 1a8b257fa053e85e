"""The pipeline's pre-steps on the GPU against the oracle's restatements of kiss-icp v1.2.0
Preprocessor::Preprocess / VoxelDownsample and pipeline/KinematicICP.cpp:31-44,54-62.
fp64 throughout: points agree to 1e-12 (device vs host sin/cos); VoxelDownsample returns the very same points in the
reference's order (its hash table's iteration order)."""
import numpy as np
import pytest

import kinematic_icp_amd as K
from kinematic_icp_amd import synthetic as syn
from boundary_case import boundary_frame, EXT, REL
from oracle import okicp, rkicp

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def raw():
    cfg, scene, scans, rng = syn.make_case("cfg1", n_scans=1)
    ext = np.concatenate([[0.01, -0.02, np.sin(0.05), np.sqrt(1 - np.sin(0.05) ** 2 - 5e-4)], [0.3, 0.1, 0.9]])
    ext[:4] /= np.linalg.norm(ext[:4])
    frame = okicp.se3_act(okicp.se3_inverse(ext), scans[0]["frame"])  # sensor-frame points
    ts = np.linspace(0.0, 1.0, len(frame))
    rel = syn.pose_mul(syn.planar_pose(0.6, 0.05, 0.04), np.array([0.004, -0.003, 0, np.sqrt(1 - 25e-6), 0, 0, 0.01]))
    return frame, ts, rel, ext


@pytest.mark.parametrize("deskew", [0, 1])
def test_preprocess_matches_oracle(raw, deskew):
    frame, ts, rel, ext = raw
    pre = K.PreSteps()
    n = pre.Preprocess(frame, ts, rel, ext, 30.0, 3.0, deskew, dst=0)
    ref = okicp.se3_act(ext, okicp.preprocess(frame, ts, rel, 30.0, 3.0, bool(deskew)))
    assert 0 < n == len(ref) < len(frame)  # the crop removed something on both sides
    np.testing.assert_allclose(pre.download(0), ref, rtol=0, atol=1e-11)
    # no timestamps -> no deskew even when asked (Preprocessing.cpp: deskew_ && !timestamps.empty())
    n2 = pre.Preprocess(frame, None, rel, ext, 30.0, 3.0, 1, dst=1)
    ref2 = okicp.se3_act(ext, okicp.preprocess(frame, None, rel, 30.0, 3.0, True))
    assert n2 == len(ref2)
    np.testing.assert_allclose(pre.download(1), ref2, rtol=0, atol=1e-12)


def test_voxel_downsample_matches_oracle(raw):
    frame, ts, rel, ext = raw
    pre = K.PreSteps()
    pre.upload(0, frame)
    for vs, (src, dst) in ((0.5, (0, 1)), (1.5, (1, 2))):  # the pipeline's two levels: 0.5 * voxel, then 1.5 * voxel
        n = pre.VoxelDownsample(src, vs, dst)
        got = pre.download(dst)
        inp = pre.download(src)
        ref = okicp.voxel_downsample(inp, vs)
        assert n == len(ref)
        np.testing.assert_array_equal(got, ref)  # the very same points in the reference's (hash table iteration) order
        if rkicp.available():
            np.testing.assert_array_equal(got, rkicp.voxel_downsample(inp, vs))
    # determinism and the all-in-one-voxel / all-distinct edge cases
    assert pre.VoxelDownsample(0, 1e6, 3) == len(np.unique(np.floor(frame / 1e6), axis=0)) <= 8  # one voxel per octant
    np.testing.assert_array_equal(pre.download(3), okicp.voxel_downsample(frame, 1e6))
    assert pre.VoxelDownsample(0, 1e-3, 3) == len(np.unique(np.floor(frame / 1e-3), axis=0))  # (almost) all distinct: load 0.5
    np.testing.assert_array_equal(pre.download(3), okicp.voxel_downsample(frame, 1e-3))
    for m in (1, 2, 3, 5, 64, 1000):  # tiny tables (2 .. 2048 buckets), incl. clusters that wrap around the table's end
        pre.upload(1, frame[:m])
        assert pre.VoxelDownsample(1, 0.7, 3) == len(okicp.voxel_downsample(frame[:m], 0.7))
        np.testing.assert_array_equal(pre.download(3), okicp.voxel_downsample(frame[:m], 0.7))
    with pytest.raises(K.KicpError) as e:  # documented limit: voxel coordinates must fit +-2^20
        pre.VoxelDownsample(0, 1e-6, 3)
    assert e.value.code == K.KICP_ERR_CAPACITY


def test_large_uploads_arrive_intact():
    """Uploads travel through the handle's pinned staging buffer in 1 MB pieces (kicp_core.hip): every byte arrives, for sizes
    around the piece boundaries, back to back on one handle, also when a kernel that reads the destination is still queued on
    the handle's stream."""
    pre = K.PreSteps()
    rng = np.random.default_rng(2)
    for n in (87381, 87382, 131072, 43690, 300001, 87382, 5):  # 2 MB - 8 B, 2 MB + 16 B, ..., back down to one piece
        pts = rng.normal(size=(n, 3))
        pre.upload(0, pts)
        np.testing.assert_array_equal(pre.download(0), pts)
    ts = np.linspace(0.0, 1.0, 300001)
    ident = np.array([0, 0, 0, 1.0, 0, 0, 0])
    for _ in range(3):  # back to back: the next upload overwrites what the previous call's kernels read
        pts = rng.normal(size=(300001, 3))
        assert pre.Preprocess(pts, ts, ident, ident, 1e9, -1.0, 1, dst=1) == 300001  # both arrays in one call
        np.testing.assert_allclose(pre.download(1), pts, rtol=0, atol=1e-15)


def test_background_download_overlaps_the_next_steps(raw):
    frame, ts, rel, ext = raw
    pre = K.PreSteps()
    n = pre.Preprocess(frame, ts, rel, ext, 30.0, 3.0, 1, dst=0)
    want = pre.download(0)
    pre.download_begin(0)                      # the copy runs while the pipeline goes on ...
    pre.VoxelDownsample(0, 0.5, 1), pre.VoxelDownsample(1, 1.5, 2)
    got = pre.download_finish(0)               # ... and is collected afterwards
    assert len(got) == n and np.array_equal(got, want)
    with pytest.raises(K.KicpError):           # nothing in flight any more
        pre.download_finish(0)
    pre.download_begin(2)
    assert np.array_equal(pre.download_finish(2), pre.download(2))


def test_presteps_feed_registration_without_leaving_the_gpu(raw):
    frame, ts, rel, ext = raw
    cfg, scene, scans, rng = syn.make_case("cfg1", n_scans=1)
    gmap = K.VoxelHashMap(cfg.voxel_size, cfg.max_range, cfg.max_points_per_voxel)
    syn.build_map_points(scene, cfg, gmap.AddPoints, gmap.num_points, rng)
    s = scans[0]
    pre = K.PreSteps()
    pre.Preprocess(frame, ts, syn.IDENTITY, ext, 100.0, 0.0, 0, dst=0)
    pre.VoxelDownsample(0, 0.5 * cfg.voxel_size, 1)
    pre.VoxelDownsample(1, 1.5 * cfg.voxel_size, 2)
    reg = K.KinematicRegistration()
    a = reg.ComputeRobotMotion(pre.frame(2), gmap, s["last_pose"], s["rel_odom"], cfg.first_frame_tau())
    b = reg.ComputeRobotMotion(pre.download(2), gmap, s["last_pose"], s["rel_odom"], cfg.first_frame_tau())
    assert np.array_equal(a, b)
    omap = okicp.VoxelHashMap(cfg.voxel_size, cfg.max_range, cfg.max_points_per_voxel)
    omap.AddPoints(gmap.Pointcloud())
    src = okicp.voxel_downsample(okicp.voxel_downsample(okicp.se3_act(ext, frame), 0.5), 1.5)
    ref = okicp.KinematicRegistration().ComputeRobotMotion(src, omap, s["last_pose"], s["rel_odom"], cfg.first_frame_tau())
    np.testing.assert_allclose(a, ref, rtol=0, atol=1e-9)


def test_decisions_next_to_a_boundary_equal_the_reference_builds():
    """points the reference build's deskew puts within 1e-12 / 1e-11 / 1e-9 m of max_range, min_range or a
    face of the 0.5 * voxel_size grid are cropped / kept / deduplicated by the device exactly as by the reference build."""
    if not rkicp.available():
        pytest.skip("oracle/_ref not built")
    max_range, min_range, v = 30.0, 3.0, 0.5
    ext, rel = EXT, REL
    raw, ts, kind = boundary_frame(rel, ext, max_range, min_range, v, (1e-12, 1e-11, 1e-9))
    # where the reference build really put them
    d = rkicp.preprocess(raw, ts, rel, 1e300, -1.0, True)
    r = np.sqrt((d * d).sum(axis=1))
    off = np.minimum(np.abs(r - max_range), np.abs(r - min_range))[kind == 0]
    assert (off < 3e-9).all() and (off[:48] < 3e-12).all() and (off[:48] > 2e-13).all()  # the 1e-12 groups sit where intended
    base = rkicp.se3_act(ext, d)
    g = base[kind == 2] / v
    face_off = np.abs(g - np.rint(g)).min(axis=1) * v
    assert (face_off[:36] < 3e-12).all() and (face_off[:36] > 2e-13).all()
    ref_kept = rkicp.preprocess(raw, ts, rel, max_range, min_range, True)
    ref_base = rkicp.se3_act(ext, ref_kept)
    ref_down = rkicp.voxel_downsample(ref_base, v)
    n_ring = int((kind == 0).sum())
    assert n_ring // 2 - 12 < len(ref_kept) - int((kind != 0).sum()) < n_ring // 2 + 12  # about half of the ring points fall on either side
    assert len(ref_down) < len(ref_kept)  # some boundary points were on the + side and lost to their companion
    pre = K.PreSteps()
    n = pre.Preprocess(raw, ts, rel, ext, max_range, min_range, 1, dst=0)
    assert n == len(ref_kept)
    np.testing.assert_allclose(pre.download(0), ref_base, rtol=0, atol=1e-11)  # the same points survive, in the same order
    nd = pre.VoxelDownsample(0, v, 1)
    assert nd == len(ref_down)
    np.testing.assert_allclose(pre.download(1), ref_down, rtol=0, atol=1e-11)


def test_a_probe_beyond_the_containers_limit_is_reported_not_hidden():
    """Adversarial frame for the reference's table: voxel x-coordinates that are multiples of the bucket count make every key's
    ideal bucket depend on (y, z) alone, so a column of such voxels piles up in ONE run of buckets and the last insertions walk
    hundreds of buckets.  tsl::robin_map would grow its table at that point (probe > 128 in 0.6.x, > 8192 in 1.x) and re-insert
    everything - an order the parallel replay does not model.  The survivors are still exactly the reference's first-per-voxel
    points; the call says KICP_WARN_TABLE_ORDER instead of vouching for their order; with the limit of robin-map 1.x (or an
    ordinary frame) it does not."""
    import kinematic_icp_amd as K
    n = 300
    buckets = 1024  # reserve(300) -> ceil(300 / 0.5) = 600 -> 1024 buckets
    vox = np.stack([np.arange(n) * buckets - 150 * buckets, np.full(n, 7), np.full(n, -3)], 1)
    pts = (vox + 0.25) * 0.5 + np.random.default_rng(5).uniform(0.0, 0.2, (n, 3))
    pre = K.PreSteps()
    pre.upload(0, pts)
    assert pre.VoxelDownsample(0, 0.5, 1) == n
    assert pre.last_max_probe() >= n - 2 and pre.last_status == K.KICP_WARN_TABLE_ORDER
    assert "ORDER" in K.lib().kicp_last_error().decode()
    got = pre.download(1)
    np.testing.assert_array_equal(got[np.argsort(got[:, 0])], pts[np.argsort(pts[:, 0])])  # every voxel's first (here: only) point
    pre.set_probe_limit(8192)
    assert pre.VoxelDownsample(0, 0.5, 1) == n and pre.last_status == K.KICP_OK
    # an ordinary frame never comes near either limit
    pre.set_probe_limit(128)
    pre.upload(0, np.random.default_rng(6).uniform(-50, 50, (20000, 3)))
    pre.VoxelDownsample(0, 0.5, 1)
    assert pre.last_status == K.KICP_OK and pre.last_max_probe() < 64


@pytest.mark.parametrize("deskew", [0, 1])
@pytest.mark.parametrize("source", ["host", "ingested"])
def test_the_chained_frame_call_equals_the_separate_calls(deskew, source):
    """kicp_pre_frame / kicp_pre_frame_ingested: Preprocess + two downsamples behind one synchronisation, the counts never leaving the
    device in between - element for element what Preprocess -> VoxelDownsample -> VoxelDownsample give, the background copy of the
    preprocessed frame included; also on an empty frame, a frame the crop empties, and with the table reused across calls."""
    import kinematic_icp_amd as K
    rng = np.random.default_rng(31 + deskew)
    ext = np.concatenate([[0, 0, np.sin(0.05), np.cos(0.05)], [0.3, 0.0, 0.9]])
    rel = syn.planar_pose(0.3, 0.0, 0.02)
    pre, sep = K.PreSteps(), K.PreSteps()
    for n, max_range in ((40000, 60.0), (0, 60.0), (5000, 60.0), (3000, 1e-3), (40000, 25.0)):
        pts = (rng.uniform(-50, 50, (n, 3)) * np.array([1.0, 1.0, 0.1])).astype(np.float32).astype(np.float64)
        ts = np.linspace(0.0, 1.0, n).astype(np.float32).astype(np.float64) if n else np.zeros(0)
        if source == "ingested":
            rec = np.zeros(n, dtype=[("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("t", "<f4")])
            rec["x"], rec["y"], rec["z"], rec["t"] = pts[:, 0], pts[:, 1], pts[:, 2], ts
            for h in (pre, sep):
                h.Ingest(rec.tobytes(), n, 16, 0, 4, 8, 7, 12)
            counts, frame = pre.Frame(None, None, rel, ext, max_range, 0.5, deskew, 0.5, 1.5)
            n0 = sep.PreprocessIngested(rel, ext, max_range, 0.5, deskew, 0)
        else:
            counts, frame = pre.Frame(pts, ts, rel, ext, max_range, 0.5, deskew, 0.5, 1.5)
            n0 = sep.Preprocess(pts, ts, rel, ext, max_range, 0.5, deskew, 0)
        n1 = sep.VoxelDownsample(0, 0.5, 1)
        n2 = sep.VoxelDownsample(1, 1.5, 2)
        assert counts == [n0, n1, n2], (n, max_range)
        if n:
            np.testing.assert_array_equal(frame, sep.download(0))
        for b in (0, 1, 2):
            np.testing.assert_array_equal(pre.download(b), sep.download(b))
    # the frames above went through the five-launch chain (kicp_pre.hpp k_frame_*); the last one's crop left a count in another
    # power-of-two bracket than the chain had guessed its first table for: found on the device, finished by the unfused downsamples
    assert pre.get_option("fused_frames") == 4 and pre.get_option("guess_misses") >= 1


@pytest.mark.parametrize("guess", [0, 100, 5000, 70000])
def test_the_fused_chain_equals_the_unfused_one_whatever_it_guesses(guess):
    """The five-launch chain sizes its first table from a GUESS of the crop's survivor count (the previous frame's): a right guess,
    a guess in a smaller and in a larger power-of-two bracket, and no guess at all give the unfused chain's buffers bit for bit;
    the tables are left clean for the next frame either way (two frames per handle)."""
    rng = np.random.default_rng(77)
    ext = np.concatenate([[0, 0, np.sin(0.05), np.cos(0.05)], [0.3, 0.0, 0.9]])
    rel = syn.planar_pose(0.3, 0.0, 0.02)
    fused, plain = K.PreSteps(), K.PreSteps()
    plain.set_option("fused", 0)
    for n, max_range in ((9000, 40.0), (9000, 60.0), (300, 60.0)):
        pts = (rng.uniform(-50, 50, (n, 3)) * np.array([1.0, 1.0, 0.1])).astype(np.float32).astype(np.float64)
        ts = np.linspace(0.0, 1.0, n)
        fused.set_option("guess", guess)
        c1, f1 = fused.Frame(pts, ts, rel, ext, max_range, 0.5, 1, 0.5, 1.5)
        c0, f0 = plain.Frame(pts, ts, rel, ext, max_range, 0.5, 1, 0.5, 1.5)
        assert c1 == c0 and 0 < c1[2] <= c1[1] <= c1[0] <= n
        np.testing.assert_array_equal(f1, f0)
        for b in (0, 1, 2):
            np.testing.assert_array_equal(fused.download(b), plain.download(b))
        ref = okicp.voxel_downsample(okicp.voxel_downsample(f0, 0.5), 1.5)
        np.testing.assert_array_equal(fused.download(2), ref)
    assert plain.get_option("fused_frames") == 0 and fused.get_option("fused_frames") == 3
    assert fused.get_option("guess_misses") >= (1 if guess == 100 else 0)


def test_long_clusters_of_the_downsample_table():
    """The chained pre-steps replay a table's clusters through an LDS window of 256 + 32 buckets per tile (kicp_pre.hpp replay_tile):
    voxels whose reference hashes fall into a few consecutive buckets make clusters that (a) sit inside one window with long robin-hood
    displacements, (b) start near a tile's end and leave the window (the global walk takes over), (c) wrap around the table's end.
    6 000 points -> 16 384 buckets at the first level; everything equals the unfused chain and the oracle bit for bit."""
    rng = np.random.default_rng(99)
    n_total, mask = 6000, 16383

    def voxels_hashing_into(lo, count, want):
        """`want` distinct voxel x coordinates (y = z = 0) whose hash & mask lies in [lo, lo + count) cyclically"""
        x = np.arange(-(1 << 19), 1 << 19, dtype=np.int64)
        h = ((x * 73856093) & 0xFFFFFFFF) & mask
        hit = x[((h - lo) & mask) < count]
        assert len(hit) >= want
        return rng.choice(hit, want, replace=False)

    crafted = np.concatenate([voxels_hashing_into(5 * 256 + 100, 8, 24),     # (a) 24 keys for 8 home buckets, mid-tile
                              voxels_hashing_into(9 * 256 + 236, 12, 70),    # (b) 70 keys from bucket 236 of a tile on: leaves the window
                              voxels_hashing_into(mask - 6, 10, 30)])        # (c) across the table's end
    pts = np.zeros((n_total, 3))
    pts[:len(crafted), 0] = (crafted + 0.5) * 0.5
    pts[:len(crafted), 1:] = 0.25
    filler = rng.uniform(-400, 400, (n_total - len(crafted), 3))
    filler[:, 2] = rng.uniform(5, 40, len(filler))  # (away from the crafted row)
    pts[len(crafted):] = filler
    pts = pts[rng.permutation(n_total)].astype(np.float32).astype(np.float64)
    ident = np.array([0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0])
    fused, plain = K.PreSteps(), K.PreSteps()
    plain.set_option("fused", 0)
    for _ in range(2):  # (the second frame finds the tables as the first one left them)
        c1, f1 = fused.Frame(pts, np.zeros(n_total), ident, ident, 1e9, 0.0, 0, 0.5, 1.5)
        c0, f0 = plain.Frame(pts, np.zeros(n_total), ident, ident, 1e9, 0.0, 0, 0.5, 1.5)
        assert c1 == c0 and c1[0] == n_total and c1[1] > 5900  # (distinct voxels: the table has 16 384 buckets)
        np.testing.assert_array_equal(f1, f0)
        for b in (0, 1, 2):
            np.testing.assert_array_equal(fused.download(b), plain.download(b))
        first = okicp.voxel_downsample(f0, 0.5)
        np.testing.assert_array_equal(fused.download(1), first)
        np.testing.assert_array_equal(fused.download(2), okicp.voxel_downsample(first, 1.5))
    assert fused.get_option("fused_frames") == 2 and fused.get_option("guess_misses") == 0
    assert fused.last_max_probe() >= 32  # (the long clusters were there: 70 keys for 12 home buckets)
