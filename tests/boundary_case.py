"""A frame of points next to the pre-steps' decision boundaries (tests/test_gpu_presteps.py, tests/test_table_order.py)."""
import numpy as np

from kinematic_icp_amd import synthetic as syn
from oracle import rkicp

EXT = np.concatenate([[0.01, -0.02, np.sin(0.05), np.sqrt(1 - np.sin(0.05) ** 2 - 5e-4)], [0.3, 0.1, 0.9]])
EXT[:4] /= np.linalg.norm(EXT[:4])
REL = syn.pose_mul(syn.planar_pose(0.6, 0.05, 0.04), np.array([0.004, -0.003, 0, np.sqrt(1 - 25e-6), 0, 0, 0.01]))


def boundary_frame(rel, ext, max_range, min_range, v, deltas, seed=5):
    """A frame whose points, AFTER the reference build's own deskew (+ base transform), lie within `delta` of a decision
    boundary: the crop radii max_range / min_range (Preprocessing.cpp: strict comparisons of the deskewed norm) and the faces
    of the 0.5 * voxel_size grid the first VoxelDownsample floors into (VoxelUtils / PointToVoxel).  Device sin / cos differ
    from the host libm's in the last bit, so the device's deskewed points are not bit-equal to the reference build's
    (DESIGN.md section 2); what must be equal are the DECISIONS, down to the distance from a boundary this test states.
    Returns raw points (sensor frame), stamps, and per group the signed distance the reference build ended up at."""
    rng = np.random.default_rng(seed)
    far = lambda pts, ts: rkicp.preprocess(pts, ts, rel, 1e300, -1.0, True)  # deskew only: nothing is cropped
    stamps, want_sensor, want_base, kind = [], [], [], []
    for delta in deltas:
        for sign in (1.0, -1.0):
            for radius in (max_range, min_range):  # 12 points per (delta, side, radius)
                u = rng.normal(size=(12, 3))
                u /= np.linalg.norm(u, axis=1)[:, None]
                want_sensor.append((radius + sign * delta) * u), want_base.append(np.full((12, 3), np.nan))
                stamps.append(rng.uniform(0.0, 1.0, 12)), kind.append(np.full(12, 0))
    # voxel faces: a companion in the middle of the voxel on the + side comes first, so the boundary point survives the
    # downsample (first point of a voxel wins) exactly when it is on the - side
    cells = [(kx, ky, kz) for kx in range(-20, 21, 4) for ky in range(-20, 21, 4) for kz in (-2, 2) if 8.0 < np.hypot(kx, ky) * v < 0.9 * max_range]
    rng.shuffle(cells)
    comp, face = [], []
    c = 0
    for delta in deltas:
        for sign in (1.0, -1.0):
            for axis in range(3):
                for _ in range(6):
                    k = np.array(cells[c], dtype=np.float64)
                    c += 1
                    centre = (k + np.array([0.37, 0.41, 0.53])) * v
                    b = centre.copy()
                    b[axis] = k[axis] * v + sign * delta
                    other = centre.copy()
                    other[axis] = (k[axis] + 0.5) * v
                    comp.append(other), face.append(b)
    m = len(face)
    want_base.append(np.array(comp)), want_sensor.append(np.full((m, 3), np.nan)), stamps.append(rng.uniform(0.0, 1.0, m)), kind.append(np.full(m, 1))
    want_base.append(np.array(face)), want_sensor.append(np.full((m, 3), np.nan)), stamps.append(rng.uniform(0.0, 1.0, m)), kind.append(np.full(m, 2))
    ws, wb, ts, kind = np.concatenate(want_sensor), np.concatenate(want_base), np.concatenate(stamps), np.concatenate(kind)
    raw = np.where(kind[:, None] == 0, ws, rkicp.se3_act(rkicp.se3_inverse(ext), np.nan_to_num(wb)))
    for _ in range(40):  # the chain is rigid per point and nearly the identity: the plain residual iteration contracts
        d = far(raw, ts)
        err = np.where(kind[:, None] == 0, ws - d, rkicp.se3_act(rkicp.se3_inverse(ext), np.nan_to_num(wb)) - d)
        raw = raw + err
    return raw, ts, kind
