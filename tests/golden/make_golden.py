"""Generates tests/golden/*.npz.  The reference ships no fixtures and cannot be built or imported offline
(SURVEY.md F5-F7), so these are REGRESSION vectors: inputs from the seeded synthetic generator, outputs from the
CPU oracle (oracle/kicp_oracle.cpp) at the time of freezing, cross-checked against tests/ref_numpy.py by
tests/test_oracle.py.  Re-run only deliberately:  python tests/golden/make_golden.py            (registration_small.npz)
                                                 python tests/golden/make_golden.py pipeline   (pipeline_small.npz)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from kinematic_icp_amd import synthetic as syn  # noqa: E402
from oracle import okicp  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def small_case(seed, n_beams=8, n_az=256, map_pts=6000, voxel=1.0):
    rng = np.random.Generator(np.random.PCG64(seed))
    scene = syn.make_scene(rng, half=14.0, height=4.0, n_boxes=5, box_xy=(2.0, 5.0), box_z=(1.5, 3.5), keep_clear=2.5)
    cfg = syn.Config("golden", n_beams, n_az, map_pts, voxel_size=voxel, max_range=40.0, sensor_height=1.2)
    omap = okicp.VoxelHashMap(cfg.voxel_size, cfg.max_range, cfg.max_points_per_voxel)
    syn.build_map_points(scene, cfg, omap.AddPoints, omap.num_points, rng, batch=4000)
    dirs = syn.beam_directions(cfg.n_beams, cfg.n_az, cfg.elev_deg)
    true_pose = syn.planar_pose(0.7, -0.4, 0.3)
    frame = syn.make_scan(scene, true_pose, dirs, cfg.sensor_height, rng)
    return cfg, omap, frame, true_pose


def main():
    out = {}
    for name, seed, guess_err in (("a", 101, (0.05, 0.2)), ("b", 202, (0.30, 1.5)), ("c", 303, (-0.2, -2.5))):
        cfg, omap, frame, true_pose = small_case(seed)
        guess = syn.pose_mul(true_pose, syn.planar_pose(guess_err[0], 0.0, np.deg2rad(guess_err[1])))
        rel = syn.planar_pose(0.4, 0.0, np.deg2rad(2.0))
        last = syn.pose_mul(guess, syn.pose_inverse(rel))
        tau = cfg.first_frame_tau()
        reg = okicp.KinematicRegistration()
        pose = reg.ComputeRobotMotion(frame, omap, last, rel, tau, count_work=True)
        st = reg.last_stats
        k = st.iterations
        sums0, _ = okicp.icp_pass(omap, frame, syn.pose_mul(last, rel), tau)
        out.update({
            name + "_map": omap.Pointcloud(), name + "_frame": frame, name + "_last": last, name + "_rel": rel,
            name + "_tau": np.array(tau), name + "_voxel": np.array(cfg.voxel_size), name + "_maxrange": np.array(cfg.max_range),
            name + "_pose": pose, name + "_iters": np.array(k), name + "_converged": np.array(st.converged),
            name + "_ncorr": np.array(st.n_corr[:k]), name + "_dx": np.array([list(st.dx[i]) for i in range(k)]),
            name + "_sums0": sums0, name + "_beta": np.array(st.beta)})
        print(name, "iters", k, "converged", st.converged, "ncorr", list(st.n_corr[:k]), "pose", pose)
    np.savez_compressed(os.path.join(HERE, "registration_small.npz"), **out)
    print("wrote", os.path.join(HERE, "registration_small.npz"))


if __name__ == "__main__" and "pipeline" not in sys.argv[1:]:
    main()


# ---- second fixture: the rows around the hot path ---------------------------------------------
def pipeline_fixture():
    rng = np.random.Generator(np.random.PCG64(404))
    scene = syn.make_scene(rng, half=14.0, height=4.0, n_boxes=5, box_xy=(2.0, 5.0), box_z=(1.5, 3.5), keep_clear=2.5)
    dirs = syn.beam_directions(6, 256, (-20.0, 8.0))
    ext = np.concatenate([[0, 0, np.sin(0.04), np.cos(0.04)], [0.25, 0.0, 0.8]])  # lidar_to_base
    voxel, max_range, min_range = 0.5, 25.0, 0.5
    out = {"ext": ext, "voxel": np.array(voxel), "max_range": np.array(max_range), "min_range": np.array(min_range)}
    # PointCloud2-like records: x y z f32 @0,4,8; intensity f32 @16; t u32 (ns since scan start) @20; 32-byte step
    dt = np.dtype({"names": ["x", "y", "z", "intensity", "t"], "formats": ["<f4", "<f4", "<f4", "<f4", "<u4"], "offsets": [0, 4, 8, 16, 20], "itemsize": 32})
    out["layout"] = np.array([32, 0, 4, 8, 6, 20])  # point_step, offsets x y z, stamp datatype (UINT32), stamp offset
    omap = okicp.VoxelHashMap(voxel, max_range, 20)
    thr = okicp.CorrespondenceThreshold(voxel / np.sqrt(20), max_range, True, 1.0)
    reg = okicp.KinematicRegistration()
    pose = syn.planar_pose(0.0, 0.0, 0.1)
    last = okicp.IDENTITY.copy()
    n_frames = 4
    for k in range(n_frames):
        delta_true = syn.planar_pose(0.3, 0.0, np.deg2rad(2.0 + k))
        pose = syn.pose_mul(pose, delta_true)
        wl = syn.pose_mul(pose, ext)
        t = scene.raycast(wl[4:], dirs @ syn.quat_to_matrix(wl[:4]).T) + rng.normal(0, 0.01, len(dirs))
        rec = np.zeros(len(dirs), dtype=dt)
        pts = dirs * t[:, None]
        rec["x"], rec["y"], rec["z"] = pts[:, 0], pts[:, 1], pts[:, 2]
        rec["intensity"] = rng.uniform(0, 1, len(dirs))
        rec["t"] = np.round(np.linspace(0.0, 0.1, len(dirs)) * 1e9).astype(np.uint32)
        raw = np.frombuffer(rec.tobytes(), dtype=np.uint8).copy()
        delta = syn.pose_mul(delta_true, syn.planar_pose(0.01 * (-1) ** k, 0.0, np.deg2rad(0.15)))  # noisy wheel odometry
        # the reference's RegisterFrame (pipeline/KinematicICP.cpp:48-85) with the oracle's pieces, deskew on
        xyz, stamps, mm = okicp.ingest(raw.tobytes(), len(rec), 32, 0, 4, 8, 6, 20)
        rel_lidar = okicp.se3_mul(okicp.se3_mul(okicp.se3_inverse(ext), delta), ext)
        in_base = okicp.se3_act(ext, okicp.preprocess(xyz, stamps, rel_lidar, max_range, min_range, True))
        down = okicp.voxel_downsample(in_base, voxel * 0.5)   # the reference's order: its hash table's iteration order
        source = okicp.voxel_downsample(down, voxel * 1.5)
        new = reg.ComputeRobotMotion(source, omap, last, delta, thr.ComputeThreshold())
        thr.UpdateOdometryError(okicp.se3_mul(okicp.se3_inverse(okicp.se3_mul(last, delta)), new))
        omap.Update(down, new)
        last = new
        out.update({"raw%d" % k: raw, "delta%d" % k: delta, "minmax%d" % k: np.array(mm), "down%d" % k: down, "source%d" % k: source,
                    "pose%d" % k: new, "n_in_base%d" % k: np.array(len(in_base)), "map_points%d" % k: np.array(omap.num_points()),
                    "map_voxels%d" % k: np.array(omap.num_voxels())})
        if k == 0:  # the decoded cloud and the preprocessed frame once (the later frames pin them through down / source)
            out.update({"xyz0": xyz, "stamps0": stamps, "in_base0": in_base})
        print("frame", k, "in", len(in_base), "down", len(down), "source", len(source), "map", omap.num_points(), "pose", new)
    pc = omap.Pointcloud()
    out["n_frames"] = np.array(n_frames)
    out["final_map_sorted"] = pc[np.lexsort((pc[:, 2], pc[:, 1], pc[:, 0]))]
    np.savez_compressed(os.path.join(HERE, "pipeline_small.npz"), **out)
    print("wrote", os.path.join(HERE, "pipeline_small.npz"))


if __name__ == "__main__" and "pipeline" in sys.argv[1:]:
    pipeline_fixture()
