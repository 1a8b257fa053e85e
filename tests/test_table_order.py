"""Table-order VoxelDownsample (kiss-icp v1.2.0 core/VoxelUtils.cpp): the survivors come out in the
iteration order of the reference's robin-hood table.  CPU side of it:
  * the integer core of the device kernels (csrc/kicp_table_order.hpp: claim in any order + per-cluster replay) against a
    sequential robin-hood table (tests/cpp/downsample_order_test.cpp, compiled with g++ from the very header the kernels use);
  * the drop-in HOST VoxelDownsample against the oracle and, where present, the reference build.
The kernels themselves: tests/test_gpu_presteps.py, tests/test_golden_pipeline.py, tests/test_facade.py (GPU)."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT
from kinematic_icp_amd import synthetic as syn
from oracle import okicp

CPP = os.path.join(ROOT, "kinematic_icp_amd", "cpp")


def test_cluster_replay_equals_sequential_robin_hood(tmp_path):
    exe = str(tmp_path / "downsample_order_test")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wall", "-Wextra", os.path.join(ROOT, "tests", "cpp", "downsample_order_test.cpp"), "-o", exe])
    out = subprocess.check_output([exe], text=True).strip()
    assert out.startswith("ok "), out
    assert int(out.split()[1]) > 80


def _build_host_filter(exe):
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wall", "-Wextra", "-I", CPP, "-I", os.path.join(CPP, "compat"), "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "host_downsample_test.cpp"), "-o", exe, "-Wl,--unresolved-symbols=ignore-all"])


def _clouds():
    rng = np.random.Generator(np.random.PCG64(5))
    cfg, scene, scans, _ = syn.make_case("cfg1", n_scans=1)
    yield "scan", scans[0]["frame"], 0.5
    yield "scan_coarse", scans[0]["frame"], 1.5
    yield "dense_duplicates", rng.uniform(-3, 3, (5000, 3)), 1.0
    yield "all_distinct", rng.uniform(-200, 200, (4096, 3)), 0.05  # load factor 0.5: long clusters
    yield "negative_side", rng.uniform(-50, -49, (300, 3)), 0.1
    yield "single", np.array([[0.1, -0.2, 0.3]]), 1.0


def test_host_drop_in_downsample_has_the_reference_order(tmp_path):
    exe = str(tmp_path / "host_downsample_test")
    _build_host_filter(exe)
    try:
        from oracle import rkicp
        ref = rkicp if rkicp.available() else None
    except Exception:  # noqa: BLE001
        ref = None
    for name, pts, vs in _clouds():
        f = tmp_path / (name + ".bin")
        np.ascontiguousarray(pts, dtype=np.float64).tofile(f)
        got = np.frombuffer(subprocess.check_output([exe, "downsample", str(f), "%.17g" % vs]), dtype=np.float64).reshape(-1, 3)
        want = okicp.voxel_downsample(pts, vs)
        assert np.array_equal(got, want), name
        if ref is not None:
            assert np.array_equal(got, ref.voxel_downsample(pts, vs)), name
        # and it IS order sensitive: first-seen order is a different sequence on anything but trivial inputs
        if len(want) > 50:
            keys = np.floor(pts / vs).astype(np.int64)
            _, first = np.unique(keys, axis=0, return_index=True)
            assert not np.array_equal(pts[np.sort(first)], want), name


def test_host_drop_in_preprocess_matches_the_oracle(tmp_path):
    """kiss_icp::Preprocessor::Preprocess of the drop-in headers (constant-velocity deskew to the scan end + strict range
    crop, order kept) against the oracle and the reference build's stand-in: same survivors, points to 1e-12 (the compat
    Sophus exp / log against the oracle's)."""
    exe = str(tmp_path / "host_filter")
    _build_host_filter(exe)
    cfg, scene, scans, _ = syn.make_case("cfg1", n_scans=1)
    frame = scans[0]["frame"][:6000]
    stamps = np.linspace(0.0, 1.0, len(frame))
    rel = syn.pose_mul(syn.planar_pose(0.6, 0.05, 0.04), np.array([0.004, -0.003, 0, np.sqrt(1 - 25e-6), 0, 0, 0.01]))
    np.ascontiguousarray(frame).tofile(tmp_path / "p.bin"), stamps.tofile(tmp_path / "t.bin"), rel.tofile(tmp_path / "r.bin")
    np.zeros(0).tofile(tmp_path / "none.bin")
    for deskew, tfile in ((1, "t.bin"), (0, "t.bin"), (1, "none.bin")):  # (no stamps: no deskew even when asked)
        got = np.frombuffer(subprocess.check_output([exe, "preprocess", str(tmp_path / "p.bin"), str(tmp_path / tfile), str(tmp_path / "r.bin"),
                                                     "4.6", "3.9", str(deskew)]), dtype=np.float64).reshape(-1, 3)
        want = okicp.preprocess(frame, stamps if tfile == "t.bin" else None, rel, 4.6, 3.9, bool(deskew))
        assert 0 < len(got) == len(want) < len(frame)
        np.testing.assert_allclose(got, want, rtol=0, atol=1e-12)


def test_host_drop_in_preprocess_next_to_the_crop_radii_equals_the_reference_build(tmp_path):
    """The host twin runs the reference's operation sequence on the host libm, so on points the reference build's deskew puts
    within 1e-12 m of max_range / min_range (tests/boundary_case.py) it keeps the very same points, bit for bit."""
    from oracle import rkicp
    if not rkicp.available():
        pytest.skip("oracle/_ref not built")
    from boundary_case import boundary_frame, EXT, REL
    exe = str(tmp_path / "host_filter")
    _build_host_filter(exe)
    raw, ts, kind = boundary_frame(REL, EXT, 30.0, 3.0, 0.5, (1e-12, 1e-11, 1e-9))
    np.ascontiguousarray(raw).tofile(tmp_path / "p.bin"), ts.tofile(tmp_path / "t.bin"), REL.tofile(tmp_path / "r.bin")
    got = np.frombuffer(subprocess.check_output([exe, "preprocess", str(tmp_path / "p.bin"), str(tmp_path / "t.bin"), str(tmp_path / "r.bin"), "30", "3", "1"]),
                        dtype=np.float64).reshape(-1, 3)
    want = rkicp.preprocess(raw, ts, REL, 30.0, 3.0, True)
    assert 0 < len(want) < len(raw) and np.array_equal(got, want)


def test_host_drop_in_threshold_reproduces_the_reference_builds_sequence(tmp_path):
    """kinematic_icp::CorrespondenceThreshold of the drop-in headers (host scalar code) on the error
    sequence frozen in tests/golden/ref_outputs.npz: the taus the reference build's CorrespondenceThreshold.cpp returned."""
    exe = str(tmp_path / "host_filter")
    _build_host_filter(exe)
    g = np.load(os.path.join(ROOT, "tests", "golden", "ref_outputs.npz"))
    np.ascontiguousarray(g["thr_errs"], dtype=np.float64).tofile(tmp_path / "e.bin")
    got = np.frombuffer(subprocess.check_output([exe, "threshold", str(tmp_path / "e.bin"), "%.17g" % (1.0 / np.sqrt(20)), "100.0"]), dtype=np.float64)
    assert len(got) == len(g["thr_taus"]) > 3
    np.testing.assert_allclose(got, g["thr_taus"], rtol=1e-15, atol=0)
