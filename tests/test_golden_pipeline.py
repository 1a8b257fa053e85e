"""tests/golden/pipeline_small.npz: four PointCloud2-style frames through RegisterFrame (ingest -> deskew + crop ->
two-level downsample in the reference's table order -> registration -> map update), frozen from the oracle
(tests/golden/make_golden.py pipeline) and equal, bit for bit, to what the reference build's own KinematicICP::RegisterFrame
returned on the same frames (tests/golden/ref_outputs.npz, frozen from oracle/_ref).
CPU: the oracle still reproduces the fixture (drift guard) and the fixture still equals the reference build's outputs.
GPU: every device stage against the frozen values - the device pipeline equals the reference's frame by frame."""
import os

import numpy as np
import pytest

from conftest import sort_rows
from oracle import okicp

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "pipeline_small.npz"))
STEP, OX, OY, OZ, STAMP_TYPE, OT = (int(v) for v in G["layout"])
VOXEL, MAX_RANGE, MIN_RANGE = float(G["voxel"]), float(G["max_range"]), float(G["min_range"])
N = int(G["n_frames"])


def test_oracle_reproduces_pipeline_fixture():
    ext = G["ext"]
    omap = okicp.VoxelHashMap(VOXEL, MAX_RANGE, 20)
    thr = okicp.CorrespondenceThreshold(VOXEL / np.sqrt(20), MAX_RANGE, True, 1.0)
    reg = okicp.KinematicRegistration()
    last = okicp.IDENTITY.copy()
    for k in range(N):
        raw, delta = G["raw%d" % k], G["delta%d" % k]
        xyz, stamps, mm = okicp.ingest(raw.tobytes(), len(raw) // STEP, STEP, OX, OY, OZ, STAMP_TYPE, OT)
        assert np.array_equal(np.array(mm), G["minmax%d" % k])
        rel_lidar = okicp.se3_mul(okicp.se3_mul(okicp.se3_inverse(ext), delta), ext)
        in_base = okicp.se3_act(ext, okicp.preprocess(xyz, stamps, rel_lidar, MAX_RANGE, MIN_RANGE, True))
        if k == 0:
            assert np.array_equal(xyz, G["xyz0"]) and np.array_equal(stamps, G["stamps0"]) and np.array_equal(in_base, G["in_base0"])
        down = okicp.voxel_downsample(in_base, VOXEL * 0.5)
        source = okicp.voxel_downsample(down, VOXEL * 1.5)
        assert np.array_equal(down, G["down%d" % k]) and np.array_equal(source, G["source%d" % k])
        new = reg.ComputeRobotMotion(source, omap, last, delta, thr.ComputeThreshold())
        np.testing.assert_allclose(new, G["pose%d" % k], rtol=0, atol=1e-12)
        thr.UpdateOdometryError(okicp.se3_mul(okicp.se3_inverse(okicp.se3_mul(last, delta)), new))
        omap.Update(down, new)
        last = new
        assert (omap.num_points(), omap.num_voxels()) == (int(G["map_points%d" % k]), int(G["map_voxels%d" % k]))
    np.testing.assert_allclose(sort_rows(omap.Pointcloud()), G["final_map_sorted"], rtol=0, atol=1e-12)


def test_pipeline_fixture_is_the_reference_builds_output():
    R = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_outputs.npz"))
    for k in range(N):
        assert np.array_equal(G["pose%d" % k], R["pipe1_%d_pose" % k]) and np.array_equal(G["source%d" % k], R["pipe1_%d_source" % k])
        assert int(G["n_in_base%d" % k]) == int(R["pipe1_%d_nframe" % k]) and int(G["map_points%d" % k]) == int(R["pipe1_%d_nmap" % k])
    assert np.array_equal(G["final_map_sorted"], R["pipe1_final_map_sorted"])


class ProductThreshold:
    """kinematic_icp::CorrespondenceThreshold of the PRODUCT's drop-in header (kinematic_icp_amd/cpp/kinematic_icp/
    correspondence_threshold/CorrespondenceThreshold.hpp), driven through tests/cpp/host_downsample_test.cpp: the tau after the
    odometry errors seen so far."""

    def __init__(self, tmp_path, map_discretization_error, max_range):
        from test_table_order import _build_host_filter
        self.exe, self.file = str(tmp_path / "host_filter"), tmp_path / "errs.bin"
        _build_host_filter(self.exe)
        self.args, self.errs = ["%.17g" % map_discretization_error, "%.17g" % max_range], []

    def UpdateOdometryError(self, err_qt):
        self.errs.append(np.asarray(err_qt, dtype=np.float64))

    def ComputeThreshold(self):
        import subprocess
        np.array(self.errs, dtype=np.float64).reshape(-1).tofile(self.file)
        return float(np.frombuffer(subprocess.check_output([self.exe, "threshold", str(self.file)] + self.args), dtype=np.float64)[-1])


@pytest.mark.gpu
def test_device_pipeline_reproduces_pipeline_fixture(tmp_path):
    import kinematic_icp_amd as K
    ext = G["ext"]
    pre, gmap, reg = K.PreSteps(), K.VoxelHashMap(VOXEL, MAX_RANGE, 20), K.KinematicRegistration()
    thr = ProductThreshold(tmp_path, VOXEL / np.sqrt(20), MAX_RANGE)  # the product's own header, not the oracle's restatement
    othr = okicp.CorrespondenceThreshold(VOXEL / np.sqrt(20), MAX_RANGE, True, 1.0)
    last = okicp.IDENTITY.copy()
    for k in range(N):
        raw, delta = G["raw%d" % k], G["delta%d" % k]
        mm = pre.Ingest(raw.tobytes(), len(raw) // STEP, STEP, OX, OY, OZ, STAMP_TYPE, OT)
        assert np.array_equal(np.array(mm), G["minmax%d" % k])
        if k == 0:
            xyz, stamps = pre.ingested()
            assert np.array_equal(xyz, G["xyz0"]) and np.array_equal(stamps, G["stamps0"])
        rel_lidar = okicp.se3_mul(okicp.se3_mul(okicp.se3_inverse(ext), delta), ext)
        n_in = pre.PreprocessIngested(rel_lidar, ext, MAX_RANGE, MIN_RANGE, True, dst=0)
        assert n_in == int(G["n_in_base%d" % k])
        if k == 0:
            np.testing.assert_allclose(pre.download(0), G["in_base0"], rtol=0, atol=1e-11)
        n_down, n_src = pre.VoxelDownsample(0, VOXEL * 0.5, 1), pre.VoxelDownsample(1, VOXEL * 1.5, 2)
        assert (n_down, n_src) == (len(G["down%d" % k]), len(G["source%d" % k]))
        np.testing.assert_allclose(pre.download(1), G["down%d" % k], rtol=0, atol=1e-11)    # same survivors, in the reference's order
        np.testing.assert_allclose(pre.download(2), G["source%d" % k], rtol=0, atol=1e-11)
        tau = thr.ComputeThreshold()
        assert tau == othr.ComputeThreshold()  # the product header and the oracle agree to the bit on this sequence
        new = reg.ComputeRobotMotion(pre.frame(2), gmap, last, delta, tau)
        np.testing.assert_allclose(new, G["pose%d" % k], rtol=0, atol=1e-9)
        err = okicp.se3_mul(okicp.se3_inverse(okicp.se3_mul(last, delta)), new)
        thr.UpdateOdometryError(err), othr.UpdateOdometryError(err)
        assert gmap.UpdateDevice(pre.frame(1), new)
        last = new
        assert (gmap.num_points(), gmap.num_voxels()) == (int(G["map_points%d" % k]), int(G["map_voxels%d" % k]))
    np.testing.assert_allclose(sort_rows(gmap.Pointcloud()), G["final_map_sorted"], rtol=0, atol=1e-9)
    assert gmap.check() == 0
