"""The drop-in C++ headers (kinematic_icp_amd/cpp): compiled here on CPU against the Eigen/Sophus stand-ins, run on
the GPU box against the golden registration vector, against a python re-enactment of RegisterFrame built from the
oracle's pieces (pipeline/KinematicICP.cpp:48-85) and against the reference build's own RegisterFrame (oracle/_ref)."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT
from kinematic_icp_amd import synthetic as syn
from oracle import okicp, rkicp

CPP = os.path.join(ROOT, "kinematic_icp_amd", "cpp")
BIN = os.path.join(ROOT, "tests", "cpp", "facade_test")
GOLD = os.path.join(ROOT, "tests", "golden", "registration_small.npz")


def build_facade():
    src = os.path.join(ROOT, "tests", "cpp", "facade_test.cpp")
    deps = [src] + [os.path.join(dp, f) for dp, _, fs in os.walk(CPP) for f in fs] + [os.path.join(ROOT, "include", "kicp.h")]
    if not os.path.exists(BIN) or any(os.path.getmtime(d) > os.path.getmtime(BIN) for d in deps):
        libdir = os.path.join(ROOT, "kinematic_icp_amd")
        subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wall", "-Wextra", "-I", CPP, "-I", os.path.join(CPP, "compat"),
                               "-I", os.path.join(ROOT, "include"), src, "-o", BIN, "-L", libdir, "-lkicp_amd",
                               "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-Wl,--allow-shlib-undefined"])
    return BIN


def test_facade_compiles_and_links(tmp_path):
    assert os.path.exists(build_facade())
    # the KICP_HOST_PRESTEPS variant of RegisterFrame (host Preprocess / VoxelDownsample, registration and map on the GPU) compiles too
    tu = tmp_path / "host_presteps.cpp"
    tu.write_text('#define KICP_HOST_PRESTEPS\n#include "kinematic_icp/pipeline/KinematicICP.hpp"\n'
                  'int main() { kinematic_icp::pipeline::Config c; return c.max_num_iterations == 10 ? 0 : 1; }\n')
    subprocess.check_call(["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-Wextra", "-I", CPP, "-I", os.path.join(CPP, "compat"),
                           "-I", os.path.join(ROOT, "include"), str(tu)])
    # the reference's include lines for this path resolve inside the drop-in tree
    for inc in ("kinematic_icp/pipeline/KinematicICP.hpp", "kinematic_icp/registration/Registration.hpp",
                "kinematic_icp/correspondence_threshold/CorrespondenceThreshold.hpp", "kiss_icp/core/VoxelHashMap.hpp",
                "kiss_icp/core/Preprocessing.hpp"):
        assert os.path.exists(os.path.join(CPP, inc))


@pytest.mark.gpu
def test_facade_registration_matches_golden(tmp_path):
    g = np.load(GOLD)
    f = tmp_path / "reg.bin"
    with open(f, "wb") as fh:
        np.array([len(g["a_map"]), len(g["a_frame"]), float(g["a_voxel"]), float(g["a_maxrange"]), float(g["a_tau"])]).tofile(fh)
        for k in ("a_map", "a_frame", "a_last", "a_rel"):
            np.ascontiguousarray(g[k], dtype=np.float64).tofile(fh)
    out = subprocess.check_output([build_facade(), "reg", str(f)], text=True).splitlines()
    pose = np.array([float(x) for x in out[0].split()[1:]])
    np.testing.assert_allclose(pose, g["a_pose"], rtol=0, atol=1e-9)
    assert out[1] == "iterations %d converged %d" % (int(g["a_iters"]), int(g["a_converged"]))
    m = okicp.VoxelHashMap(float(g["a_voxel"]), float(g["a_maxrange"]), 20)
    m.AddPoints(g["a_map"])
    nn, d = m.GetClosestNeighbor(g["a_frame"][:1])
    np.testing.assert_array_equal(np.array([float(x) for x in out[2].split()[1:]]), np.concatenate([nn[0], d]))
    assert out[3] == "iterations_after_edit 1"
    np.testing.assert_array_equal(np.array([float(x) for x in out[4].split()[1:]]), pose)  # registration against a copy of the map
    assert out[5] == "original_empty 1 copy_points %d" % m.num_points()
    # copy / move / assignment of the registration (value semantics of the reference's struct) and of the pipeline object
    assert out[6] == "copied_fields 7 7 7 1 1"
    for k in (7, 8, 9):
        np.testing.assert_array_equal(np.array([float(x) for x in out[k].split()[1:]]), pose)
    f32_pose, widened_pose = (np.array([float(x) for x in out[k].split()[1:]]) for k in (10, 11))
    np.testing.assert_array_equal(f32_pose, widened_pose)  # float32 on the wire, widened on the device == widened on the host
    wide = g["a_frame"].astype(np.float32).astype(np.float64)
    ref = okicp.KinematicRegistration().ComputeRobotMotion(wide, m, g["a_last"], g["a_rel"], float(g["a_tau"]))
    np.testing.assert_allclose(f32_pose, ref, rtol=0, atol=1e-9)
    assert out[12] == "pipeline_copy %d 0" % m.num_points()
    assert out[13] == "threads_field 1 1 4"  # max_num_threads <= 0 -> the hardware's thread count, as Registration.cpp:141-142 stores it


@pytest.mark.gpu
@pytest.mark.parametrize("deskew", [0, 1])
def test_facade_pipeline_matches_oracle_pipeline(tmp_path, deskew):
    rng = np.random.Generator(np.random.PCG64(77))
    scene = syn.make_scene(rng, half=16.0, height=4.0, n_boxes=6, box_xy=(2.0, 5.0), box_z=(1.5, 3.5), keep_clear=3.0)
    dirs = syn.beam_directions(12, 512, (-20.0, 8.0))
    ext = np.concatenate([[0, 0, np.sin(0.05), np.cos(0.05)], [0.3, 0.0, 0.9]])  # lidar_to_base
    voxel, max_range = 0.5, 30.0
    poses, frames, stamps, deltas = [syn.planar_pose(0.0, 0.0, 0.1)], [], [], []
    for k in range(6):
        delta_true = syn.planar_pose(0.25, 0.0, np.deg2rad(2.0 + k))
        poses.append(syn.pose_mul(poses[-1], delta_true))
        world_from_lidar = syn.pose_mul(poses[-1], ext)
        R = syn.quat_to_matrix(world_from_lidar[:4])
        t = scene.raycast(world_from_lidar[4:], dirs @ R.T) + rng.normal(0, 0.01, len(dirs))
        frames.append(dirs * t[:, None])                                   # points in the LIDAR frame
        stamps.append(np.linspace(0.0, 1.0, len(dirs)))
        deltas.append(syn.pose_mul(delta_true, syn.planar_pose(0.01 * (-1) ** k, 0.0, np.deg2rad(0.15))))  # noisy wheel odometry
    f = tmp_path / "pipe.bin"
    with open(f, "wb") as fh:
        np.array([len(frames), voxel, max_range, float(deskew)]).tofile(fh)
        ext.tofile(fh)
        for fr, st, dl in zip(frames, stamps, deltas):
            np.array([float(len(fr))]).tofile(fh)
            np.ascontiguousarray(fr).tofile(fh), st.tofile(fh), dl.tofile(fh)
    out = subprocess.check_output([build_facade(), "pipeline", str(f)], text=True).splitlines()
    # python re-enactment of KinematicICP::RegisterFrame with the oracle's pieces
    omap = okicp.VoxelHashMap(voxel, max_range, 20)
    thr = okicp.CorrespondenceThreshold(voxel / np.sqrt(20), max_range, True, 1.0)
    reg = okicp.KinematicRegistration()
    last = okicp.IDENTITY.copy()
    # ... and the reference's own KinematicICP.cpp (oracle/_ref), where the build is present
    ref_icp = rkicp.KinematicICP(max_range=max_range, min_range=0.0, voxel_size=voxel, deskew=bool(deskew)) if rkicp.available() else None
    for k, (fr, st, dl) in enumerate(zip(frames, stamps, deltas)):
        rel_lidar = okicp.se3_mul(okicp.se3_mul(okicp.se3_inverse(ext), dl), ext)
        pre = okicp.preprocess(fr, st, rel_lidar, max_range, 0.0, bool(deskew))
        in_base = okicp.se3_act(ext, pre)
        down = okicp.voxel_downsample(in_base, voxel * 0.5)
        source = okicp.voxel_downsample(down, voxel * 1.5)
        tau = thr.ComputeThreshold()
        new = reg.ComputeRobotMotion(source, omap, last, dl, tau)
        thr.UpdateOdometryError(okicp.se3_mul(okicp.se3_inverse(okicp.se3_mul(last, dl)), new))
        omap.Update(down, new)
        last = new
        pose = np.array([float(x) for x in out[2 * k].split()[1:]])
        np.testing.assert_allclose(pose, new, rtol=0, atol=1e-9, err_msg="frame %d" % k)
        sizes = [int(x) for x in out[2 * k + 1].split()[1:]]
        assert sizes == [len(in_base), len(source), omap.num_points()], "frame %d" % k
        if ref_icp is not None:  # the drop-in pipeline against the reference's RegisterFrame, frame by frame
            ref_frame, ref_source = ref_icp.RegisterFrame(fr, st, ext, dl)
            np.testing.assert_allclose(pose, ref_icp.pose(), rtol=0, atol=1e-9, err_msg="frame %d vs reference build" % k)
            assert sizes == [len(ref_frame), len(ref_source), len(ref_icp.LocalMap())], "frame %d vs reference build" % k
    assert out[-1] == "after_setpose 0 1"


@pytest.mark.gpu
def test_facade_pipeline_from_pointcloud2_bytes(tmp_path):
    """IngestCloud + RegisterIngestedFrame on float32 records == RegisterFrame on the same values as fp64 vectors."""
    rng = np.random.Generator(np.random.PCG64(78))
    scene = syn.make_scene(rng, half=16.0, height=4.0, n_boxes=6, box_xy=(2.0, 5.0), box_z=(1.5, 3.5), keep_clear=3.0)
    dirs = syn.beam_directions(12, 512, (-20.0, 8.0))
    ext = np.concatenate([[0, 0, np.sin(0.05), np.cos(0.05)], [0.3, 0.0, 0.9]])
    pose = syn.planar_pose(0.0, 0.0, 0.1)
    f = tmp_path / "pipe.bin"
    with open(f, "wb") as fh:
        np.array([5.0, 0.5, 30.0, 1.0]).tofile(fh)
        ext.tofile(fh)
        for k in range(5):
            delta = syn.planar_pose(0.25, 0.0, np.deg2rad(2.0 + k))
            pose = syn.pose_mul(pose, delta)
            wl = syn.pose_mul(pose, ext)
            t = scene.raycast(wl[4:], dirs @ syn.quat_to_matrix(wl[:4]).T) + rng.normal(0, 0.01, len(dirs))
            fr = (dirs * t[:, None]).astype(np.float32).astype(np.float64)  # what a float32 message can carry
            np.array([float(len(fr))]).tofile(fh)
            np.ascontiguousarray(fr).tofile(fh), np.linspace(0.0, 1.0, len(dirs)).tofile(fh), delta.tofile(fh)
    host = subprocess.check_output([build_facade(), "pipeline", str(f)], text=True).splitlines()
    raw = subprocess.check_output([build_facade(), "pipeline_raw", str(f)], text=True).splitlines()
    assert raw[2] == "stamps 1 0 1"
    raw = raw[:2] + raw[3:]
    assert raw == host[:len(raw)] and len(raw) == 10
    # AnnounceNextCloud (round 5): message k + 1 uploaded and decoded while frame k's pre-steps run - the same poses, sizes and map
    keep = lambda lines: [l for l in lines if l.startswith("pose") or l.startswith("map")]  # noqa: E731  (the timing lines differ)
    plain = subprocess.check_output([build_facade(), "pipeline_timed_raw", str(f)], text=True).splitlines()
    ahead = subprocess.check_output([build_facade(), "pipeline_timed_raw_ahead", str(f)], text=True).splitlines()
    assert keep(ahead) == keep(plain) and len(keep(plain)) == 6
    sizes = lambda lines: [l.split()[l.split().index("in"):l.split().index("map_on_device")] for l in lines if l.startswith("frame")]  # noqa: E731
    assert sizes(ahead) == sizes(plain)
