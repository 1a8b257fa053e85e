"""CPU-side model of how many bucket-visit rounds a 64-lane wave of the pass kernel runs under different visiting policies
(no GPU needed): the current one (every lane walks its own work list, re-culling after every visit: rounds = max over the
lanes) against pooling the wave's remaining visits after the first round and dealing them one per lane.

python tools/sim_rounds.py [--workload cfg2]"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle"))

SHIFTS = np.array([[0, 0, 0], [1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1], [1, 1, 0], [1, -1, 0], [-1, 1, 0], [-1, -1, 0],
                   [1, 0, 1], [1, 0, -1], [-1, 0, 1], [-1, 0, -1], [0, 1, 1], [0, 1, -1], [0, -1, 1], [0, -1, -1], [1, 1, 1], [1, 1, -1],
                   [1, -1, 1], [1, -1, -1], [-1, 1, 1], [-1, 1, -1], [-1, -1, 1], [-1, -1, -1]])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="cfg2")
    ap.add_argument("--wave", type=int, default=64)
    args = ap.parse_args()
    import okicp
    from kinematic_icp_amd import synthetic as syn

    cfg, scene, scans, rng = syn.make_case(args.workload, n_scans=1)
    omap = okicp.VoxelHashMap(cfg.voxel_size, cfg.max_range, cfg.max_points_per_voxel)
    syn.build_map_points(scene, cfg, omap.AddPoints, omap.num_points, rng)
    pts = omap.Pointcloud()
    vs = cfg.voxel_size
    tau = cfg.first_frame_tau()
    s = scans[0]
    q = syn.pose_act(syn.pose_mul(s["last_pose"], s["rel_odom"]), s["frame"])
    n = q.shape[0]
    print("map %d points, scan %d points, voxel %.2f tau %.4f" % (pts.shape[0], n, vs, tau))

    def pack(v):
        v = v.astype(np.int64) + (1 << 20)
        return (v[:, 0] << 42) | (v[:, 1] << 21) | v[:, 2]

    pk = pack(np.floor(pts / vs))
    order = np.argsort(pk, kind="stable")
    pk, pts = pk[order], pts[order]
    keys, start, count = np.unique(pk, return_index=True, return_counts=True)
    qv = np.floor(q / vs)
    bound = tau * tau
    # d2[i, s] = squared distance of query i to the nearest point of neighbour voxel s (inf if empty), box[i, s] = lower bound
    d2 = np.full((n, 27), np.inf)
    occ = np.zeros((n, 27), bool)
    cap = int(count.max())
    for si, sh in enumerate(SHIFTS):
        k = pack(qv + sh)
        j = np.searchsorted(keys, k)
        j[j >= keys.size] = 0
        hit = keys[j] == k
        occ[:, si] = hit
        idx = np.nonzero(hit)[0]
        st, ct = start[j[idx]], count[j[idx]]
        best = np.full(idx.size, np.inf)
        for u in range(cap):
            m = u < ct
            d = np.sum((pts[np.minimum(st + u, pts.shape[0] - 1)] - q[idx]) ** 2, axis=1)
            best = np.where(m, np.minimum(best, d), best)
        d2[idx, si] = best
    l = q - qv * vs
    lo, hi = l * l, (vs - l) ** 2
    comp = np.where(SHIFTS[None, :, :] < 0, lo[:, None, :], np.where(SHIFTS[None, :, :] > 0, hi[:, None, :], 0.0))
    box = comp.sum(axis=2)

    # ---- policy 0: the kernel's: per lane, repeatedly visit the first alive occupied voxel; cull against the running minimum
    best = np.full(n, bound)
    todo = occ.copy()
    visits = np.zeros(n, int)
    rounds_lane = np.zeros(n, int)
    first_best = None
    remaining_after_first = None
    while True:
        alive = todo & (box <= best[:, None])
        any_alive = alive.any(axis=1)
        if not any_alive.any():
            break
        sidx = np.argmax(alive, axis=1)
        rows = np.nonzero(any_alive)[0]
        best[rows] = np.minimum(best[rows], d2[rows, sidx[rows]])
        todo = alive
        todo[rows, sidx[rows]] = False
        visits[rows] += 1
        if first_best is None:
            first_best = best.copy()
            remaining_after_first = (todo & (box <= best[:, None])).sum(axis=1)
    W = args.wave
    nw = (n + W - 1) // W
    pad = nw * W - n
    v_w = np.pad(visits, (0, pad)).reshape(nw, W)
    print("policy 0 (current): visits/query %.3f, rounds per wave %.3f (max over lanes); histogram of per-lane visits %s" % (
        visits.mean(), v_w.max(axis=1).mean(), np.bincount(visits)[:8]))
    # ---- policy 1: round 1 as above, then ALL remaining alive voxels of the wave pooled and dealt one per lane
    r_w = np.pad(remaining_after_first, (0, pad)).reshape(nw, W)
    pooled_rounds = np.ceil(r_w.sum(axis=1) / W)
    print("policy 1 (pool everything after round 1): visits/query %.3f, rounds per wave %.3f; waves needing >1 pooled round: %.1f %%" % (
        (visits > 0).mean() + remaining_after_first.mean(), 1 + pooled_rounds.mean(), 100 * (pooled_rounds > 1).mean()))
    # ---- policy 2: rounds 1 and 2 per lane (sequential culling), then pool
    best = np.full(n, bound)
    todo = occ.copy()
    vis2 = np.zeros(n, int)
    for r in range(2):
        alive = todo & (box <= best[:, None])
        any_alive = alive.any(axis=1)
        sidx = np.argmax(alive, axis=1)
        rows = np.nonzero(any_alive)[0]
        best[rows] = np.minimum(best[rows], d2[rows, sidx[rows]])
        todo = alive
        todo[rows, sidx[rows]] = False
        vis2[rows] += 1
    rem2 = (todo & (box <= best[:, None])).sum(axis=1)
    r2_w = np.pad(rem2, (0, pad)).reshape(nw, W)
    any2 = np.pad(vis2, (0, pad)).reshape(nw, W).max(axis=1)
    print("policy 2 (two own rounds, then pool): visits/query %.3f, rounds per wave %.3f" % (
        vis2.mean() + rem2.mean(), (any2 + np.ceil(r2_w.sum(axis=1) / W)).mean()))
    # ---- policy 3: pool from the start (every alive voxel after culling against the acceptance bound only)
    alive0 = occ & (box <= bound)
    a_w = np.pad(alive0.sum(axis=1), (0, pad)).reshape(nw, W)
    print("policy 3 (pool everything, no own round): visits/query %.3f, rounds per wave %.3f" % (alive0.sum(axis=1).mean(), np.ceil(a_w.sum(axis=1) / W).mean()))


if __name__ == "__main__":
    main()
