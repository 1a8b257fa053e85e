"""Wire-format ingest: PointCloud2 bytes -> fp64 points + normalised per-point stamps.
Reference: ros/src/kinematic_icp_ros/utils/RosUtils.cpp:30-39 (PointCloud2ToEigen) and TimeStampHandler.cpp:57-128.
CPU part: the oracle restatement against an independent numpy decode (structured dtypes).  GPU part: the HIP decode
against the oracle, bit for bit (float->double widening, one subtraction and one division in fp64: nothing to round
differently), and the deskewing pipeline fed from raw bytes against the one fed from host arrays."""
import numpy as np
import pytest

from oracle import okicp

U32, F32, F64 = 6, 7, 8


def make_cloud(rng, n, stamp, layout="ouster", scale=1.0, base=0.0):
    """A PointCloud2-like record array.  layout 'packed': x y z [t]; 'ouster': x y z pad intensity t ... (48 B, like the
    Ouster driver's point type); 'odd': unaligned offsets."""
    st = {None: None, U32: "<u4", F32: "<f4", F64: "<f8"}[stamp]
    if layout == "packed":
        fields = [("x", "<f4"), ("y", "<f4"), ("z", "<f4")] + ([("t", st)] if st else [])
        dt = np.dtype(fields)
    elif layout == "ouster":
        names, fmts, offs = ["x", "y", "z", "intensity"], ["<f4"] * 4, [0, 4, 8, 16]
        if st:
            names.append("t"), fmts.append(st), offs.append(24 if stamp == F64 else 20)
        dt = np.dtype({"names": names, "formats": fmts, "offsets": offs, "itemsize": 48})
    else:
        names, fmts, offs = ["z", "x", "y"], ["<f4"] * 3, [1, 7, 13]  # deliberately unaligned and permuted
        if st:
            names.append("t"), fmts.append(st), offs.append(19)
        dt = np.dtype({"names": names, "formats": fmts, "offsets": offs, "itemsize": 31})
    rec = np.zeros(n, dtype=dt)
    rec.view(np.uint8)[:] = rng.integers(0, 255, rec.view(np.uint8).shape, dtype=np.uint8)  # junk in the padding / other fields
    for k in "xyz":
        rec[k] = rng.uniform(-80, 80, n).astype(np.float32)
    if st:
        t = base + scale * np.sort(rng.uniform(0.0, 0.1, n))
        rec["t"] = t.astype(dt.fields["t"][0])
    off = {k: dt.fields[k][1] for k in dt.names}
    return rec, dt.itemsize, off


def numpy_decode(rec, stamp):
    xyz = np.stack([rec["x"].astype(np.float64), rec["y"].astype(np.float64), rec["z"].astype(np.float64)], axis=1)
    if stamp is None or len(rec) == 0:
        return xyz, None, (0.0, 0.0)
    t = rec["t"].astype(np.float64)
    t = np.where(np.round(t) >= 1e10, t * 1e-9, t)  # more than 10 integer digits -> nanoseconds
    lo, hi = t.min(), t.max()
    return xyz, (t - lo) / (hi - lo), (lo, hi)


CASES = [("packed", None, 1.0, 0.0), ("packed", F32, 1.0, 0.0), ("ouster", U32, 1e9, 0.0),      # relative ns in a uint32
         ("ouster", F64, 1.0, 1.7e9),                                                           # absolute epoch seconds
         ("ouster", F64, 1e9, 1.7e18),                                                          # absolute epoch nanoseconds
         ("odd", F32, 1.0, 0.0), ("odd", F64, 1.0, 12345.0), ("odd", U32, 1e6, 0.0)]


@pytest.mark.parametrize("layout,stamp,scale,base", CASES)
def test_oracle_ingest_matches_numpy_decode(layout, stamp, scale, base):
    rng = np.random.Generator(np.random.PCG64(11))
    rec, step, off = make_cloud(rng, 5000, stamp, layout, scale, base)
    xyz, st, mm = okicp.ingest(rec.tobytes(), len(rec), step, off["x"], off["y"], off["z"], stamp or 0, off.get("t", 0))
    exp_xyz, exp_st, exp_mm = numpy_decode(rec, stamp)
    np.testing.assert_array_equal(xyz, exp_xyz)
    if stamp is None:
        assert st is None and mm == (0.0, 0.0)
    else:
        np.testing.assert_array_equal(st, exp_st)
        assert mm == exp_mm and st.min() == 0.0 and st.max() == 1.0
        if base > 1e17:
            assert 1.6e9 < mm[0] < 1.8e9  # converted to seconds


def test_oracle_ingest_nanosecond_rule_boundary_and_errors():
    # 9 999 999 999 has 10 integer digits (kept), 10 000 000 000 has 11 (nanoseconds): TimeStampHandler.cpp:60-63,76-78
    rec = np.zeros(3, dtype=np.dtype([("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("t", "<f8")]))
    rec["t"] = [9_999_999_999.4, 10_000_000_000.0, 9_999_999_999.5]  # the last one rounds up to 1e10
    _, st, mm = okicp.ingest(rec.tobytes(), 3, 20, 0, 4, 8, F64, 12)
    assert mm == (9_999_999_999.5 * 1e-9, 9_999_999_999.4)  # the third is the smallest after its conversion
    with pytest.raises(RuntimeError):
        okicp.ingest(rec.tobytes(), 3, 20, 0, 4, 8, 2, 12)  # UINT8 stamps: "timestamp field type not supported"
    xyz, st, mm = okicp.ingest(b"", 0, 20, 0, 4, 8, F64, 12)
    assert xyz.shape == (0, 3) and st is None and mm == (0.0, 0.0)


@pytest.mark.gpu
@pytest.mark.parametrize("layout,stamp,scale,base", CASES)
def test_gpu_ingest_equals_oracle(layout, stamp, scale, base):
    import kinematic_icp_amd as kicp
    rng = np.random.Generator(np.random.PCG64(12))
    pre = kicp.PreSteps()
    for n in (1, 63, 257, 70_001, 200_003):  # (1 MB and more: a helper thread shares the copy into the staging buffer, in 128 KB sub-pieces)
        rec, step, off = make_cloud(rng, n, stamp, layout, scale, base)
        mm = pre.Ingest(rec.tobytes(), n, step, off["x"], off["y"], off["z"], stamp or 0, off.get("t", 0))
        exp_xyz, exp_st, exp_mm = okicp.ingest(rec.tobytes(), n, step, off["x"], off["y"], off["z"], stamp or 0, off.get("t", 0))
        xyz, st = pre.ingested()
        np.testing.assert_array_equal(xyz, exp_xyz)
        assert mm == exp_mm
        if stamp is None:
            assert st is None
        else:
            np.testing.assert_array_equal(st, exp_st)  # NaN == NaN positions too (n = 1: 0/0)


@pytest.mark.gpu
def test_gpu_ingest_with_sensor_pose_and_errors():
    import kinematic_icp_amd as kicp
    from kinematic_icp_amd import synthetic as syn
    rng = np.random.Generator(np.random.PCG64(13))
    pre = kicp.PreSteps()
    rec, step, off = make_cloud(rng, 4097, F32, "ouster")
    T = syn.pose_mul(syn.planar_pose(0.3, -0.2, 0.4, 1.1), np.array([np.sin(0.1), 0, 0, np.cos(0.1), 0, 0, 0]))
    pre.Ingest(rec.tobytes(), len(rec), step, off["x"], off["y"], off["z"], F32, off["t"], sensor_pose=T)
    exp_xyz, _, _ = okicp.ingest(rec.tobytes(), len(rec), step, off["x"], off["y"], off["z"], F32, off["t"], sensor_pose_qt=T)
    np.testing.assert_allclose(pre.ingested()[0], exp_xyz, rtol=0, atol=1e-12)
    with pytest.raises(kicp.KicpError):
        pre.Ingest(rec.tobytes(), len(rec), step, off["x"], off["y"], off["z"], 2, off["t"])   # unsupported stamp type
    with pytest.raises(kicp.KicpError):
        pre.Ingest(rec.tobytes(), len(rec), step, off["x"], off["y"], 46, F32, off["t"])       # z would cross the record end
    with pytest.raises(kicp.KicpError):
        kicp.PreSteps().PreprocessIngested(syn.IDENTITY, syn.IDENTITY, 100.0, 0.0, True)       # nothing ingested yet
    assert pre.Ingest(b"", 0, step, 0, 4, 8) == (0.0, 0.0)
    assert pre.PreprocessIngested(syn.IDENTITY, syn.IDENTITY, 100.0, 0.0, True) == 0


@pytest.mark.gpu
@pytest.mark.parametrize("deskew", [False, True])
def test_gpu_pipeline_from_raw_bytes_equals_pipeline_from_host_arrays(deskew):
    """Preprocess fed by the ingested cloud == Preprocess fed by what PointCloud2ToEigen / ProcessTimestamps would hand over."""
    import kinematic_icp_amd as kicp
    from kinematic_icp_amd import synthetic as syn
    rng = np.random.Generator(np.random.PCG64(14))
    rec, step, off = make_cloud(rng, 50_000, U32, "ouster", scale=1e9)
    rel = syn.planar_pose(0.4, 0.01, np.deg2rad(3.0))
    ext = syn.planar_pose(0.2, 0.0, 0.05, 0.7)
    a, b = kicp.PreSteps(), kicp.PreSteps()
    a.Ingest(rec.tobytes(), len(rec), step, off["x"], off["y"], off["z"], U32, off["t"])
    na = a.PreprocessIngested(rel, ext, 60.0, 1.0, deskew, dst=0)
    xyz, st, _ = okicp.ingest(rec.tobytes(), len(rec), step, off["x"], off["y"], off["z"], U32, off["t"])
    nb = b.Preprocess(xyz, st, rel, ext, 60.0, 1.0, deskew, dst=0)
    assert na == nb and 0 < na < len(rec)
    np.testing.assert_array_equal(a.download(0), b.download(0))
    ref = okicp.se3_act(ext, okicp.preprocess(xyz, st, rel, 60.0, 1.0, deskew))
    np.testing.assert_allclose(a.download(0), ref, rtol=0, atol=1e-11)
    # (the wire-format decoding itself - RosUtils.cpp / TimeStampHandler.cpp - needs ROS headers and is not part of the reference
    # build; what follows it is: kiss_icp::Preprocessor::Preprocess of oracle/_ref on the decoded cloud)
    from oracle import rkicp
    if rkicp.available():
        theirs = rkicp.preprocess(xyz, st, rel, 60.0, 1.0, deskew)
        assert len(theirs) == na
        np.testing.assert_allclose(a.download(0), okicp.se3_act(ext, theirs), rtol=0, atol=1e-11)


@pytest.mark.gpu
@pytest.mark.parametrize("stamp,layout", [(U32, "ouster"), (F32, "packed"), (None, "packed"), (F64, "odd")])
def test_gpu_look_ahead_ingest_equals_the_plain_ingest(stamp, layout):
    """kicp_pre_ingest_ahead (round 5): message k + 1 is uploaded and decoded by the chained pre-steps of message k; its Ingest call
    then takes the slot.  Decoded clouds, stamp extrema and every output of the chained pre-steps equal the plain sequence's, bit for
    bit - over four messages of different sizes; an announcement that is not followed by its message is void."""
    import kinematic_icp_amd as kicp
    from kinematic_icp_amd import synthetic as syn
    rng = np.random.Generator(np.random.PCG64(21))
    scale = 1e9 if stamp == U32 else 1.0
    msgs = []
    for n in (41_000, 30_000, 12_345, 41_000, 70_000):  # (the last one outgrows the buffers: it is left to its own Ingest call)
        rec, step, off = make_cloud(rng, n, stamp, layout, scale=scale)
        msgs.append((np.frombuffer(rec.tobytes(), dtype=np.uint8).copy(), n, step, off))
    rel = syn.planar_pose(0.4, 0.01, np.deg2rad(3.0))
    ext = syn.planar_pose(0.2, 0.0, 0.05, 0.7)

    def args(m):
        raw, n, step, off = m
        return (raw, n, step, off["x"], off["y"], off["z"], stamp or 0, off.get("t", 0))

    def chain(pre):
        counts, frame = pre.Frame(None, None, rel, ext, 60.0, 1.0, True, 0.5, 1.5)
        return counts, frame, pre.download(1), pre.download(2)

    plain, ahead = kicp.PreSteps(), kicp.PreSteps()
    want = []
    for m in msgs:
        lohi = plain.Ingest(*args(m))
        want.append((lohi, plain.ingested(), chain(plain)))
    for k, m in enumerate(msgs):
        lohi = ahead.Ingest(*args(m))
        got_cloud = ahead.ingested()
        if k + 1 < len(msgs):
            ahead.IngestAhead(*args(msgs[k + 1]))
        got = chain(ahead)  # (uploads message k + 1 behind its own kernels)
        assert lohi == want[k][0]
        np.testing.assert_array_equal(got_cloud[0], want[k][1][0])
        if want[k][1][1] is None:
            assert got_cloud[1] is None
        else:
            np.testing.assert_array_equal(got_cloud[1], want[k][1][1])
        assert got[0] == want[k][2][0]
        for a, b in zip(got[1:], want[k][2][1:]):
            np.testing.assert_array_equal(a, b)
    assert ahead.ahead_hits() == 3 and plain.ahead_hits() == 0  # (messages 1 .. 3 were found decoded; message 4 did not fit the buffers of its predecessors)
    # an announcement followed by ANOTHER message: void - the other message is ingested as usual
    ahead.IngestAhead(*args(msgs[0]))
    chain(ahead)
    assert ahead.Ingest(*args(msgs[2])) == want[2][0]
    np.testing.assert_array_equal(ahead.ingested()[0], want[2][1][0])
    assert chain(ahead)[0] == want[2][2][0] and ahead.ahead_hits() == 3
    # a receive buffer that is REUSED against the contract: message 0 announced and decoded ahead, then ANOTHER message of the same size
    # written over it at the same address - the first and last bytes no longer match what was uploaded: ingested afresh, no stale cloud
    buf = msgs[0][0].copy()
    reused = (buf,) + tuple(args(msgs[0])[1:])
    ahead.Ingest(*args(msgs[1]))
    ahead.IngestAhead(*reused)
    chain(ahead)
    buf[:] = msgs[3][0]  # (message 3 has message 0's size and layout, other points)
    assert ahead.Ingest(*reused) == want[3][0] and ahead.ahead_hits() == 3
    np.testing.assert_array_equal(ahead.ingested()[0], want[3][1][0])


def test_ordered_integer_keys_of_doubles_are_monotone():
    """k_ingest merges the stamps' extrema with integer atomics on an order-preserving map double -> uint64
    (kicp_pre.hpp ordered_key / ordered_value); re-enacted here: monotone over negatives, zeros, subnormals and infinities,
    and invertible."""
    def ordered_key(v):
        b = np.asarray(v, dtype=np.float64).view(np.uint64)
        return np.where(b >> np.uint64(63), ~b, b | np.uint64(1 << 63))

    def ordered_value(k):
        b = np.where(k >> np.uint64(63), k & np.uint64((1 << 63) - 1), ~k)
        return b.view(np.float64)

    rng = np.random.Generator(np.random.PCG64(21))
    v = np.concatenate([rng.normal(0, 1e9, 5000), rng.normal(0, 1e-300, 100), [0.0, -0.0, 5e-324, -5e-324, np.inf, -np.inf, 1.7e9, 1.7e18]])
    k = ordered_key(v)
    order = np.argsort(k, kind="stable")
    assert np.all(np.diff(v[order]) >= 0)
    assert np.array_equal(ordered_value(k).view(np.uint64), v.view(np.uint64))
    assert ordered_value(np.array([k.min()]))[0] == v.min() and ordered_value(np.array([k.max()]))[0] == v.max()
