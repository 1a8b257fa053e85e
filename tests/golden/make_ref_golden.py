"""Generates tests/golden/ref_outputs.npz: outputs of the REFERENCE'S OWN SOURCES (oracle/_ref/libkicp_ref.so =
Registration.cpp, CorrespondenceThreshold.cpp, KinematicICP.cpp compiled unmodified against oracle/ref_shim) on the
inputs already frozen in registration_small.npz / pipeline_small.npz.  /root/reference exists only in the build container,
so these vectors are what carries the reference's results to the GPU box:  python tests/golden/make_ref_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import okicp, rkicp  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
REG_VARIANTS = (("default", dict()), ("fixed0", dict(use_adaptive_odometry_regularization=False, fixed_regularization=0.0)),
                ("fixed5", dict(use_adaptive_odometry_regularization=False, fixed_regularization=5.0)),
                ("it3", dict(max_num_iteration=3)), ("loose", dict(convergence_criterion=1e-2)))


def main():
    assert rkicp.reference_present(), "needs /root/reference"
    out = {}
    g = np.load(os.path.join(HERE, "registration_small.npz"))
    for name in ("a", "b", "c"):
        m = rkicp.VoxelHashMap(float(g[name + "_voxel"]), float(g[name + "_maxrange"]), 20)
        m.AddPoints(g[name + "_map"])
        for vname, kw in REG_VARIANTS:
            for tau_scale in (1.0, 0.4):
                pose = rkicp.KinematicRegistration(**kw).ComputeRobotMotion(g[name + "_frame"], m, g[name + "_last"], g[name + "_rel"],
                                                                            float(g[name + "_tau"]) * tau_scale)
                out["reg_%s_%s_%g" % (name, vname, tau_scale)] = pose
        q = g[name + "_frame"][::7]
        nn, d = m.GetClosestNeighbor(rkicp.se3_act(rkicp.se3_mul(g[name + "_last"], g[name + "_rel"]), q))
        out["nn_%s" % name], out["nnd_%s" % name] = nn, d
    # CorrespondenceThreshold over a sequence of odometry errors
    rng = np.random.Generator(np.random.PCG64(11))
    errs, taus = [], []
    t = rkicp.CorrespondenceThreshold(1.0 / np.sqrt(20), 100.0, True, 1.0)
    taus.append(t.ComputeThreshold())
    for _ in range(12):
        q = rng.normal(size=4) * np.array([0.02, 0.02, 0.05, 1.0])
        q /= np.linalg.norm(q)
        e = np.concatenate([q, rng.normal(size=3) * 0.05])
        errs.append(e)
        t.UpdateOdometryError(e)
        taus.append(t.ComputeThreshold())
    out["thr_errs"], out["thr_taus"] = np.array(errs), np.array(taus)
    # the whole RegisterFrame pipeline on the frozen PointCloud2-style frames (decoded by the oracle's ingest: ROS-side glue)
    p = np.load(os.path.join(HERE, "pipeline_small.npz"))
    L = [int(v) for v in p["layout"]]
    for deskew in (0, 1):
        icp = rkicp.KinematicICP(max_range=float(p["max_range"]), min_range=float(p["min_range"]), voxel_size=float(p["voxel"]), deskew=deskew)
        for k in range(int(p["n_frames"])):
            raw = p["raw%d" % k]
            xyz, stamps, _ = okicp.ingest(raw.tobytes(), len(raw) // L[0], L[0], L[1], L[2], L[3], L[4], L[5])
            tau = icp.tau()
            frame, source = icp.RegisterFrame(xyz, stamps, p["ext"], p["delta%d" % k])
            tag = "pipe%d_%d" % (deskew, k)
            out[tag + "_tau"], out[tag + "_pose"] = np.array(tau), icp.pose()
            out[tag + "_nframe"], out[tag + "_source"] = np.array(len(frame)), source
            out[tag + "_nmap"] = np.array(len(icp.LocalMap()))
            print(tag, "tau %.6f" % tau, "frame", len(frame), "source", len(source), "map", len(icp.LocalMap()), icp.pose())
        pc = icp.LocalMap()
        out["pipe%d_final_map_sorted" % deskew] = pc[np.lexsort((pc[:, 2], pc[:, 1], pc[:, 0]))]
    np.savez_compressed(os.path.join(HERE, "ref_outputs.npz"), **out)
    print("wrote ref_outputs.npz with", len(out), "arrays")


if __name__ == "__main__":
    main()
