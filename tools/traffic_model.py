"""Where the pass kernel's HBM traffic goes: a CPU count of the DISTINCT 128-byte lines one pass touches in each structure of
the device map, next to SURVEY.md section 8d's compulsory bytes (12 B per query + 16 B per touched slot +
12 B per touched bucket point).  The kernel's visiting policy is re-enacted (own voxel first, reference order, neighbours culled
against the running minimum), so "buckets read" are the buckets the kernel reads, and "winner points" are the fp64 points its
exact phase fetches.  A line is the unit of every fetch (L2 line = 128 B on gfx950): a 16-byte key costs a line, a 160-byte
mirror bucket two (it is not line aligned), a 24-byte fp64 point one or two.

    python tools/traffic_model.py [--workload cfg5]
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
# the 27 neighbour shifts in the reference's visiting order (Registration.cpp:40-57 / kicp_common.hpp)
SHIFTS = np.array([[0, 0, 0], [1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1], [1, 1, 0], [1, -1, 0], [-1, 1, 0], [-1, -1, 0],
                   [1, 0, 1], [1, 0, -1], [-1, 0, 1], [-1, 0, -1], [0, 1, 1], [0, 1, -1], [0, -1, 1], [0, -1, -1], [1, 1, 1], [1, 1, -1],
                   [1, -1, 1], [1, -1, -1], [-1, 1, 1], [-1, 1, -1], [-1, -1, 1], [-1, -1, -1]])


def lines(byte_lo, byte_hi):
    """distinct 128-byte lines covered by the byte ranges [lo, hi)"""
    a, b = byte_lo // 128, (byte_hi - 1) // 128
    out = [a]
    k = 1
    while (a + k <= b).any():
        out.append(np.where(a + k <= b, a + k, a))
        k += 1
    return np.unique(np.concatenate(out)).size


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="cfg5")
    args = ap.parse_args()
    import okicp
    from kinematic_icp_amd import synthetic as syn
    cfg, scene, scans, rng = syn.make_case(args.workload, n_scans=1)
    omap = okicp.VoxelHashMap(cfg.voxel_size, cfg.max_range, cfg.max_points_per_voxel)
    syn.build_map_points(scene, cfg, omap.AddPoints, omap.num_points, rng)
    pts = omap.Pointcloud()
    vs, tau, cap = cfg.voxel_size, cfg.first_frame_tau(), cfg.max_points_per_voxel
    s = scans[0]
    q = syn.pose_act(syn.pose_mul(s["last_pose"], s["rel_odom"]), s["frame"])
    n = q.shape[0]

    def pack(v):
        v = v.astype(np.int64) + (1 << 20)
        return (v[:, 0] << 42) | (v[:, 1] << 21) | v[:, 2]
    pk = pack(np.floor(pts / vs))
    order = np.argsort(pk, kind="stable")
    pk, pts = pk[order], pts[order]
    keys, start, count = np.unique(pk, return_index=True, return_counts=True)
    qv = np.floor(q / vs)
    bound = tau * tau
    d2 = np.full((n, 27), np.inf)
    arg = np.zeros((n, 27), np.int64)      # index (into the sorted points) of the nearest point of neighbour voxel s
    vox = np.full((n, 27), -1, np.int64)   # index of that voxel in `keys`
    for si, sh in enumerate(SHIFTS):
        k = pack(qv + sh)
        j = np.searchsorted(keys, k)
        j[j >= keys.size] = 0
        hit = keys[j] == k
        idx = np.nonzero(hit)[0]
        vox[idx, si] = j[idx]
        st, ct = start[j[idx]], count[j[idx]]
        best, barg = np.full(idx.size, np.inf), np.zeros(idx.size, np.int64)
        for u in range(int(count.max())):
            m = u < ct
            p = np.minimum(st + u, pts.shape[0] - 1)
            d = np.sum((pts[p] - q[idx]) ** 2, axis=1)
            better = m & (d < best)
            best, barg = np.where(better, d, best), np.where(better, p, barg)
        d2[idx, si], arg[idx, si] = best, barg
    l = q - qv * vs
    lo, hi = l * l, (vs - l) ** 2
    box = np.where(SHIFTS[None, :, :] < 0, lo[:, None, :], np.where(SHIFTS[None, :, :] > 0, hi[:, None, :], 0.0)).sum(axis=2)
    occ = vox >= 0
    best, todo = np.full(n, bound), occ.copy()
    win = np.full(n, -1, np.int64)
    visited = np.zeros((n, 27), bool)
    while True:
        alive = todo & (box <= best[:, None])
        if not alive.any():
            break
        sidx = np.argmax(alive, axis=1)
        rows = np.nonzero(alive.any(axis=1))[0]
        visited[rows, sidx[rows]] = True
        better = d2[rows, sidx[rows]] < best[rows]
        best[rows] = np.where(better, d2[rows, sidx[rows]], best[rows])
        win[rows] = np.where(better, arg[rows, sidx[rows]], win[rows])
        todo = alive
        todo[rows, sidx[rows]] = False
    own = np.unique(pack(qv))                       # one table slot (128 B = one line) per distinct own voxel with an entry
    has_entry = occ.any(axis=1)
    own_lines = np.unique(pack(qv[has_entry])).size
    b_read = np.unique(vox[visited])                # buckets read from the 16-bit mirror
    stride16 = -(-cap // 20) * 20 * 8               # bytes per mirror bucket
    mirror_lines = lines(b_read * stride16, b_read * stride16 + stride16)
    w = win[win >= 0]
    wb = np.searchsorted(start, w, side="right") - 1   # the winner's bucket, its position inside it
    off = wb * cap * 24 + (w - start[wb]) * 24
    pool_lines = lines(off, off + 24)
    src_lines = -(-n * 24 // 128)
    touched_slots = np.unique(np.concatenate([pack(qv + sh) for sh in SHIFTS])).size
    touched_points = int(count[np.unique(vox[occ])].sum())
    b_min = 12 * n + 16 * touched_slots + 12 * touched_points
    total = 128 * (src_lines + own_lines + mirror_lines + pool_lines)
    print("%s: %d queries, %d map points in %d voxels; visits per query %.2f, correspondences %d" % (args.workload, n, pts.shape[0], keys.size, visited.sum() / n, (win >= 0).sum()))
    print("distinct 128-byte lines of one pass, if every line were fetched exactly once:")
    for name, k, what in (("scan (fp64 xyz, 24 B per query)", src_lines, "SURVEY counts 12 B per query"),
                          ("table slots (one 128-B slot per distinct own voxel)", own_lines, "%d distinct own voxels; SURVEY counts 16 B for each of %d touched slots" % (own.size, touched_slots)),
                          ("16-bit mirror buckets read (%d B each, not line aligned)" % stride16, mirror_lines, "%d buckets; SURVEY counts 12 B for each of %d touched points" % (b_read.size, touched_points)),
                          ("fp64 points of the winners (exact phase)", pool_lines, "not in SURVEY's count: it prices a point once, at 12 B")):
        print("  %-62s %9d lines = %6.1f MB   (%s)" % (name, k, k * 128 / 1e6, what))
    print("  sum %.1f MB = %.2f x b_min (%.1f MB)" % (total / 1e6, total / b_min, b_min / 1e6))
    # The device has EIGHT L2s (one per XCD) and workgroup w (queries 256 w .. 256 w + 255) runs on XCD w mod 8: a line that queries of
    # several XCDs need is fetched from memory once per XCD.  The same count per (line, XCD):
    xcd = (np.arange(n) // 256) % 8
    per_xcd = src_lines
    for x in range(8):
        mine = xcd == x
        per_xcd += np.unique(pack(qv[has_entry & mine])).size
        bx = np.unique(vox[visited & mine[:, None]])
        if bx.size:
            per_xcd += lines(bx * stride16, bx * stride16 + stride16)
        wx = win[(win >= 0) & mine]
        if wx.size:
            wbx = np.searchsorted(start, wx, side="right") - 1
            offx = wbx * cap * 24 + (wx - start[wbx]) * 24
            per_xcd += lines(offx, offx + 24)
    predicted = 128 * per_xcd
    line = "  with a fetch per (line, XCD): %.1f MB = %.2f x b_min" % (predicted / 1e6, predicted / b_min)
    try:
        import json
        measured = json.load(open(os.path.join(ROOT, "profiles", "r06_counters_%s.json" % args.workload)))["hbm_bytes_per_launch"]
        line += "; measured (profiles/r06_counters_%s.json, FETCH_SIZE x 2 + WRITE_SIZE) %.1f MB: the model is %+.0f %% off" % (
            args.workload, measured / 1e6, 100.0 * (predicted - measured) / measured)
    except (OSError, KeyError, ValueError):
        pass
    print(line)


if __name__ == "__main__":
    main()
