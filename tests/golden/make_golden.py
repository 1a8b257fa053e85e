"""Generates tests/golden/*.npz.  The reference ships no fixtures and cannot be built or imported offline
(SURVEY.md F5-F7), so these are REGRESSION vectors: inputs from the seeded synthetic generator, outputs from the
CPU oracle (oracle/kicp_oracle.cpp) at the time of freezing, cross-checked against tests/ref_numpy.py by
tests/test_oracle.py.  Re-run only deliberately:  python tests/golden/make_golden.py
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


if __name__ == "__main__":
    main()
