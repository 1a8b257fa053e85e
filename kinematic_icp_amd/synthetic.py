"""Seeded synthetic scenes / scans for the BASELINE.json configs.

Analytic, axis-aligned world: ground plane z=0, a closed outer box (walls + ceiling) so that every
ray returns (exact N), and K rectangular "building" boxes.  Scans are ray-cast from a sensor mounted
at (0,0,h) in the robot base frame and returned IN THE BASE FRAME - the frame
KinematicRegistration::ComputeRobotMotion expects its `frame` argument in
(/root/reference/cpp/kinematic_icp/pipeline/KinematicICP.cpp:59,68).  Map samples are area-uniform
draws of the same surfaces, to be inserted through VoxelHashMap::AddPoints by the caller.

Pure numpy; PRNG = numpy PCG64 with seed 0x4B494350 + cfg.  No I/O.
"""
from dataclasses import dataclass, field

import numpy as np

SEED_BASE = 0x4B494350


# --------------------------------------------------------------------------------------------
# small SE3 helpers on [qx,qy,qz,qw,tx,ty,tz] (Sophus parameter order); host-side plumbing only
# --------------------------------------------------------------------------------------------
def quat_mul(a, b):
    ax, ay, az, aw = a
    bx, by, bz, bw = b
    return np.array([aw * bx + ax * bw + ay * bz - az * by, aw * by + ay * bw + az * bx - ax * bz,
                     aw * bz + az * bw + ax * by - ay * bx, aw * bw - ax * bx - ay * by - az * bz])


def quat_to_matrix(q):
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def pose_mul(a, b):
    q = quat_mul(a[:4], b[:4])
    q /= np.linalg.norm(q)
    return np.concatenate([q, a[4:] + quat_to_matrix(a[:4]) @ b[4:]])


def pose_inverse(a):
    qi = np.array([-a[0], -a[1], -a[2], a[3]])
    return np.concatenate([qi, -(quat_to_matrix(qi) @ a[4:])])


def pose_act(a, pts):
    return pts @ quat_to_matrix(a[:4]).T + a[4:]


def planar_pose(x, y, yaw, z=0.0):
    return np.array([0.0, 0.0, np.sin(yaw / 2), np.cos(yaw / 2), x, y, z])


IDENTITY = planar_pose(0.0, 0.0, 0.0)


# --------------------------------------------------------------------------------------------
@dataclass
class Scene:
    half: float                      # outer box is [-half, half]^2 x [0, height]
    height: float
    boxes: np.ndarray                # (K, 6): xmin, ymin, zmin, xmax, ymax, zmax
    rects: list = field(default_factory=list)  # surface rectangles (origin, edge_u, edge_v)

    def __post_init__(self):
        r = []
        h, H = self.half, self.height
        r.append((np.array([-h, -h, 0.0]), np.array([2 * h, 0, 0.0]), np.array([0, 2 * h, 0.0])))      # ground
        r.append((np.array([-h, -h, H]), np.array([2 * h, 0, 0.0]), np.array([0, 2 * h, 0.0])))        # ceiling
        for s in (-1.0, 1.0):
            r.append((np.array([s * h, -h, 0.0]), np.array([0, 2 * h, 0.0]), np.array([0, 0, H])))      # x walls
            r.append((np.array([-h, s * h, 0.0]), np.array([2 * h, 0, 0.0]), np.array([0, 0, H])))      # y walls
        for b in self.boxes:
            x0, y0, z0, x1, y1, z1 = b
            r.append((np.array([x0, y0, z1]), np.array([x1 - x0, 0, 0.0]), np.array([0, y1 - y0, 0.0])))  # roof
            for xs in (x0, x1):
                r.append((np.array([xs, y0, z0]), np.array([0, y1 - y0, 0.0]), np.array([0, 0, z1 - z0])))
            for ys in (y0, y1):
                r.append((np.array([x0, ys, z0]), np.array([x1 - x0, 0, 0.0]), np.array([0, 0, z1 - z0])))
        self.rects = r
        self._areas = np.array([np.linalg.norm(np.cross(u, v)) for _, u, v in r])

    def surface_area(self):
        return float(self._areas.sum())

    def sample_surface(self, n, rng):
        """n area-uniform samples of all surfaces -> (n,3) float64."""
        idx = rng.choice(len(self.rects), size=n, p=self._areas / self._areas.sum())
        o = np.stack([self.rects[i][0] for i in range(len(self.rects))])[idx]
        u = np.stack([self.rects[i][1] for i in range(len(self.rects))])[idx]
        v = np.stack([self.rects[i][2] for i in range(len(self.rects))])[idx]
        a = rng.random((n, 1))
        b = rng.random((n, 1))
        return o + a * u + b * v

    def raycast(self, origin, dirs):
        """Range along each unit direction from `origin` (inside the outer box, outside all boxes)."""
        with np.errstate(divide="ignore", invalid="ignore"):
            inv = 1.0 / dirs
            lo = np.array([-self.half, -self.half, 0.0])
            hi = np.array([self.half, self.half, self.height])
            # exit distance of the enclosing box
            t_exit = np.where(dirs > 0, (hi - origin) * inv, np.where(dirs < 0, (lo - origin) * inv, np.inf)).min(axis=1)
            t = t_exit
            for b in self.boxes:
                t0 = (b[:3] - origin) * inv
                t1 = (b[3:] - origin) * inv
                tn = np.nanmax(np.minimum(t0, t1), axis=1)
                tf = np.nanmin(np.maximum(t0, t1), axis=1)
                hit = (tn <= tf) & (tn > 0)
                t = np.where(hit & (tn < t), tn, t)
        return t


def make_scene(rng, half=65.0, height=25.0, n_boxes=60, box_xy=(6.0, 18.0), box_z=(4.0, 14.0), keep_clear=6.0):
    boxes = []
    while len(boxes) < n_boxes:
        sx, sy = rng.uniform(*box_xy, size=2)
        sz = rng.uniform(*box_z)
        cx, cy = rng.uniform(-half + sx, half - sx), rng.uniform(-half + sy, half - sy)
        if abs(cx) < keep_clear + sx / 2 and abs(cy) < keep_clear + sy / 2:
            continue  # keep the robot's neighbourhood free
        boxes.append([cx - sx / 2, cy - sy / 2, 0.0, cx + sx / 2, cy + sy / 2, sz])
    return Scene(half, height, np.array(boxes))


def beam_directions(n_beams, n_az, elev_deg=(-24.8, 2.0), az_span_deg=360.0, order="ring"):
    """Unit directions in the sensor frame.  order='ring': beam-major (organised cloud rows);
    'azimuth': firing-major (all beams of one azimuth step consecutive)."""
    elev = np.deg2rad(np.linspace(elev_deg[0], elev_deg[1], n_beams)) if n_beams > 1 else np.array([0.0])
    az0 = -np.deg2rad(az_span_deg) / 2
    az = az0 + np.deg2rad(az_span_deg) * np.arange(n_az) / (n_az if az_span_deg >= 360.0 else max(n_az - 1, 1))
    if order == "ring":
        e, a = np.meshgrid(elev, az, indexing="ij")
    else:
        a, e = np.meshgrid(az, elev, indexing="ij")
    e, a = e.ravel(), a.ravel()
    return np.stack([np.cos(e) * np.cos(a), np.cos(e) * np.sin(a), np.sin(e)], axis=1)


def make_scan(scene, true_pose, dirs_sensor, sensor_height, rng, range_sigma=0.01):
    """Scan taken by a robot at `true_pose` (world<-base); returns points in the BASE frame."""
    R = quat_to_matrix(true_pose[:4])
    origin_w = true_pose[4:] + R @ np.array([0.0, 0.0, sensor_height])
    dirs_w = dirs_sensor @ R.T
    t = scene.raycast(origin_w, dirs_w)
    t = t + rng.normal(0.0, range_sigma, size=t.shape)
    return dirs_sensor * t[:, None] + np.array([0.0, 0.0, sensor_height])


@dataclass
class Config:
    name: str
    n_beams: int
    n_az: int
    map_points: int
    voxel_size: float = 1.0
    max_points_per_voxel: int = 20
    max_range: float = 100.0
    sensor_height: float = 1.7
    elev_deg: tuple = (-24.8, 2.0)
    az_span_deg: float = 360.0
    scene_kw: dict = field(default_factory=dict)
    seed: int = 0

    @property
    def n_points(self):
        return self.n_beams * self.n_az

    def map_resolution(self):  # pipeline/KinematicICP.hpp:46
        return self.voxel_size / np.sqrt(self.max_points_per_voxel)

    def first_frame_tau(self):  # CorrespondenceThreshold.cpp:52-54 with odom_sse_ = 0
        return 3.0 * self.map_resolution()


CONFIGS = {
    # cfg1: 16 beams x 1024 az = 16384 pts vs 100k-pt map (CPU plumbing case)
    "cfg1": Config("cfg1", 16, 1024, 100_000, scene_kw=dict(half=30.0, height=8.0, n_boxes=12, box_xy=(4.0, 10.0), box_z=(3.0, 7.0)),
                   seed=SEED_BASE + 1),
    # cfg2: 64 x 2048 = 131072 pts vs ~1M-pt map (the headline config)
    "cfg2": Config("cfg2", 64, 2048, 1_000_000, seed=SEED_BASE + 2),
    # cfg4: 2-D LaserScan, 1080 pts (270 deg @ 0.25 deg) vs 50k-pt map, voxel 0.2
    "cfg4": Config("cfg4", 1, 1080, 50_000, voxel_size=0.2, max_range=30.0, sensor_height=0.3, elev_deg=(0.0, 0.0),
                   az_span_deg=270.0, scene_kw=dict(half=20.0, height=1.2, n_boxes=25, box_xy=(2.0, 7.0), box_z=(1.2, 1.2),
                                                    keep_clear=2.5), seed=SEED_BASE + 4),
    # cfg5: dense 500k-pt scan vs 10M-pt map, voxel 0.1
    "cfg5": Config("cfg5", 125, 4000, 10_000_000, voxel_size=0.1, max_range=100.0,
                   scene_kw=dict(half=40.0, height=12.0, n_boxes=40, box_xy=(5.0, 14.0), box_z=(3.0, 10.0)), seed=SEED_BASE + 5),
}


def build_map_points(scene, cfg, add_points, map_size, rng, batch=None, tol=0.02, max_rounds=400):
    """Feed area-uniform surface samples to add_points(xyz) until map_size() reaches the target.

    Returns the number of samples drawn.  The last batch is sized from the observed acceptance rate
    so the final count lands within ~tol of the target."""
    target = cfg.map_points
    batch = batch or max(20_000, target // 4)
    drawn = 0
    prev = map_size()
    rate = 1.0
    for _ in range(max_rounds):
        cur = map_size()
        if cur >= target * (1.0 - tol):
            break
        need = target - cur
        k = int(min(batch, max(2_000, need / max(rate, 0.02) * 0.9)))
        pts = scene.sample_surface(k, rng)
        add_points(pts)
        drawn += k
        now = map_size()
        rate = max((now - cur) / k, 1e-3)
        prev = now
    return drawn


def make_case(cfg_name, n_scans=1, order="ring", scene=None, rng=None):
    """Scene + `n_scans` (scan, last_pose, rel_odom, true_pose) tuples for a BASELINE config.

    Initial guess = truth perturbed by a seeded (dd in +-0.10*voxel_size along body x, dtheta in +-0.2 deg);
    never exactly zero (reference quirk F9, Registration.cpp:163-165)."""
    cfg = CONFIGS[cfg_name]
    rng = rng or np.random.Generator(np.random.PCG64(cfg.seed))
    scene = scene or make_scene(rng, **cfg.scene_kw)
    dirs = beam_directions(cfg.n_beams, cfg.n_az, cfg.elev_deg, cfg.az_span_deg, order)
    scans = []
    for _ in range(n_scans):
        true_pose = planar_pose(rng.uniform(-2.0, 2.0), rng.uniform(-2.0, 2.0), rng.uniform(-np.pi, np.pi))
        pts = make_scan(scene, true_pose, dirs, cfg.sensor_height, rng)
        dd = rng.uniform(0.02, 0.10) * cfg.voxel_size * rng.choice([-1.0, 1.0])
        dth = np.deg2rad(rng.uniform(0.04, 0.2)) * rng.choice([-1.0, 1.0])
        guess = pose_mul(true_pose, planar_pose(dd, 0.0, dth))
        # split the guess into last_pose * relative_odometry with a generic odometry step
        rel = planar_pose(rng.uniform(0.2, 0.6), 0.0, np.deg2rad(rng.uniform(-3.0, 3.0)))
        last = pose_mul(guess, pose_inverse(rel))
        scans.append(dict(frame=np.ascontiguousarray(pts), last_pose=last, rel_odom=rel, true_pose=true_pose))
    return cfg, scene, scans, rng


def raycast_torch(scene, origin, dirs, device="cuda"):
    """Scene.raycast with the arithmetic done by torch on `device` (fp64, the same slab test box by box): what bench.py uses to
    produce dozens of distinct scans in seconds.  Not bit-compared with the numpy version anywhere: the scans it makes are
    inputs, checked against the oracle like any other."""
    import torch
    with torch.no_grad():
        d = torch.as_tensor(np.ascontiguousarray(dirs), dtype=torch.float64, device=device)
        o = torch.as_tensor(np.asarray(origin, dtype=np.float64), device=device)
        inv = 1.0 / d
        lo = torch.tensor([-scene.half, -scene.half, 0.0], dtype=torch.float64, device=device)
        hi = torch.tensor([scene.half, scene.half, scene.height], dtype=torch.float64, device=device)
        inf = torch.full_like(d, float("inf"))
        t = torch.where(d > 0, (hi - o) * inv, torch.where(d < 0, (lo - o) * inv, inf)).min(dim=1).values
        for b in scene.boxes:
            bl = torch.as_tensor(b[:3], dtype=torch.float64, device=device)
            bh = torch.as_tensor(b[3:], dtype=torch.float64, device=device)
            t0, t1 = (bl - o) * inv, (bh - o) * inv
            tn = torch.nan_to_num(torch.minimum(t0, t1), nan=-float("inf")).max(dim=1).values
            tf = torch.nan_to_num(torch.maximum(t0, t1), nan=float("inf")).min(dim=1).values
            hit = (tn <= tf) & (tn > 0) & (tn < t)
            t = torch.where(hit, tn, t)
        return t.cpu().numpy()


def extra_scans(cfg, scene, n, seed, order="ring", raycast=None):
    """`n` more scans of `scene` in the manner of make_case (same pose / initial-guess distributions), from a generator of their
    own - so that asking for more scans leaves the first ones and the map built after them unchanged.  `raycast(origin, dirs)`
    replaces Scene.raycast (e.g. raycast_torch)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    dirs = beam_directions(cfg.n_beams, cfg.n_az, cfg.elev_deg, cfg.az_span_deg, order)
    scans = []
    for _ in range(n):
        true_pose = planar_pose(rng.uniform(-2.0, 2.0), rng.uniform(-2.0, 2.0), rng.uniform(-np.pi, np.pi))
        R = quat_to_matrix(true_pose[:4])
        origin_w = true_pose[4:] + R @ np.array([0.0, 0.0, cfg.sensor_height])
        dirs_w = dirs @ R.T
        t = (raycast or scene.raycast)(origin_w, dirs_w)
        t = t + rng.normal(0.0, 0.01, size=t.shape)
        pts = dirs * t[:, None] + np.array([0.0, 0.0, cfg.sensor_height])
        dd = rng.uniform(0.02, 0.10) * cfg.voxel_size * rng.choice([-1.0, 1.0])
        dth = np.deg2rad(rng.uniform(0.04, 0.2)) * rng.choice([-1.0, 1.0])
        guess = pose_mul(true_pose, planar_pose(dd, 0.0, dth))
        rel = planar_pose(rng.uniform(0.2, 0.6), 0.0, np.deg2rad(rng.uniform(-3.0, 3.0)))
        last = pose_mul(guess, pose_inverse(rel))
        scans.append(dict(frame=np.ascontiguousarray(pts), last_pose=last, rel_odom=rel, true_pose=true_pose))
    return scans
