"""KAT-4 scenes (kiss-icp v1.2.0 VoxelHashMap::GetClosestNeighbor as called from
/root/reference/cpp/kinematic_icp/registration/Registration.cpp:74-77): hand-built maps in which a query has 2, 3 and >= 4
candidates that are

  (i)   exactly equidistant across two (and more) of the 27 shifts            -> the earlier shift keeps the tie
  (ii)  exactly equidistant inside one bucket                                  -> the earlier slot (insertion order) keeps it
  (iii) 1-2 ulp apart in SQUARED distance but equal after the rounding of sqrt -> the earlier one keeps it (the reference compares
        norms with strict '<'; a comparison of squared distances picks the other one)
  (iv)  within the 16-bit mirror's error margin of each other, ordered one way by the mirror and the other way in fp64
        -> the fp64 order decides (the mirror only pre-selects)
  (v)   exactly at tau                                                         -> rejected (strict '<', Registration.cpp:75)

Every case lives in a cell of its own (8 voxels apart, so cells cannot see each other), the tied targets are placed asymmetrically
(a wrong pick moves the query's residual by >= 0.3 m, i.e. JTr by >> 1e-10), and the scene is padded with queries that have one
candidate in their own voxel only - lanes that run out of work early, which is what lets the four-waves build lend them voxels of
the tie queries (gather32_pass).  `premises()` re-derives, in numpy and from the coordinates alone, that each case really is the
tie it claims to be; tests/test_ties.py holds the oracle and the reference build to the expected winners on the CPU,
tests/test_gpu_ties.py sends the scenes through every fused pass kernel."""
import numpy as np

VS = 1.0          # voxel size
CAP = 20          # max_points_per_voxel
TAU = 1.125       # max correspondence distance: beyond one voxel, so all 27 neighbours matter
U = VS / 65536.0  # one unit of the 16-bit mirror


def _d2(t, q):  # the reference's (and the kernels') expression, fp64, left to right
    d = np.asarray(t, dtype=np.float64) - np.asarray(q, dtype=np.float64)
    return d[0] * d[0] + d[1] * d[1] + d[2] * d[2]


def _norm(t, q):
    return np.sqrt(_d2(t, q))


def _mirror_d2(t, q):
    """squared distance in mirror units as the pre-selection sees it: the target quantised to 1 / 65536 of its voxel, the query exact"""
    t, q = np.asarray(t, dtype=np.float64), np.asarray(q, dtype=np.float64)
    v = np.floor(t / VS)
    m = np.clip(np.floor((t - v * VS) / U + 0.5), 0, 65535) * U + v * VS
    return float(np.sum(((m - q) / U) ** 2))


def _sqrt_equal_dz(rng, planar2, smaller_by=(1, 2)):
    """dz_a, dz_b (multiples of 2^-53 near 1 / 16) such that fl(planar2 + fl(dz_b^2)) lies 1-2 ulp BELOW fl(planar2 + fl(dz_a^2)) and
    both have the same correctly rounded square root"""
    step = 2.0 ** -53
    for _ in range(100000):
        dz_a = 0.0625 * (1.0 + rng.uniform(0.0, 0.5))
        dz_a = np.round(dz_a / step) * step
        d2_a = planar2 + dz_a * dz_a
        for k in range(1, 6):
            dz_b = dz_a - k * step
            d2_b = planar2 + dz_b * dz_b
            ulps = round((d2_a - d2_b) / np.spacing(d2_b))
            if ulps in smaller_by and np.sqrt(d2_b) == np.sqrt(d2_a):
                return dz_a, dz_b
    raise AssertionError("no sqrt-equal pair found")


# ---- the cases: f(base, rng) -> (map points in insertion order, query, expected target | None, kind) -----------------------------
def two_shifts(b, rng):
    q = b + [0.5, 0.5, 0.5]
    pts = [b + [-0.25, 0.5, 0.5], b + [1.25, 0.5, 0.5]]  # -x (shift 2) inserted first, +x (shift 1) visited first
    return pts, q, pts[1], "tie"


def own_voxel_against_a_neighbour(b, rng):
    q = b + [0.25, 0.5, 0.5]
    pts = [b + [-0.25, 0.5, 0.5], b + [0.75, 0.5, 0.5]]  # shift 2 | shift 0
    return pts, q, pts[1], "tie"


def face_against_edge(b, rng):
    q = b + [0.5, 0.5, 0.5]
    pts = [b + [1.0625, 1.25, 0.5], b + [1.4375, 0.5, 0.5]]  # edge ++0 (shift 7), (18, 24, 0) / 32 | face +x (shift 1), 30 / 32
    return pts, q, pts[1], "tie"


def edge_against_a_later_face(b, rng):
    q = b + [0.5, 0.5, 0.5]
    pts = [b + [1.0625, 1.25, 0.5], b + [0.5, 0.5, -0.4375]]  # shift 7 | face -z (shift 6)
    return pts, q, pts[1], "tie"


def corner_against_edge(b, rng):
    q = b + [0.5, 0.5, 0.5]
    pts = [b + [1.0, 1.0625, 1.25], b + [0.5, 1.0, 1.4375]]  # corner +++ (shift 19), (8, 9, 12) / 16 | edge 0++ (shift 15), (0, 8, 15) / 16: 17 / 16
    return pts, q, pts[1], "tie"


def bucket_two(b, rng):
    q = b + [0.5, 0.5, 0.5]
    pts = [b + [0.75, 0.5, 0.5], b + [0.25, 0.5, 0.5]]
    return pts, q, pts[0], "tie"


def bucket_five(b, rng):
    q = b + [0.5, 0.5, 0.5]
    pts = [b + [0.5, 0.75, 0.5], b + [0.25, 0.5, 0.5], b + [0.75, 0.5, 0.5], b + [0.5, 0.25, 0.5], b + [0.5, 0.5, 0.25]]
    return pts, q, pts[0], "tie"


def bucket_and_neighbour(b, rng):
    q = b + [0.25, 0.5, 0.5]
    pts = [b + [-0.25, 0.5, 0.5], b + [0.25, 0.5, 0.0], b + [0.75, 0.5, 0.5]]  # shift 2 | own voxel, first | own voxel, second
    return pts, q, pts[1], "tie"


def four_faces_behind_the_own_voxel(b, rng):
    q = b + [0.5, 0.5, 0.5]
    pts = [b + [0.5, -0.25, 0.5], b + [0.5, 1.25, 0.5], b + [-0.25, 0.5, 0.5], b + [1.25, 0.5, 0.5], b + [0.0625, 0.0625, 0.0625]]
    return pts, q, pts[3], "tie"  # the own voxel holds something farther (0.758): rounds 2.. visit the faces, +x first


def six_faces(b, rng):
    q = b + [0.5, 0.5, 0.5]
    pts = [b + [0.5, 0.5, -0.25], b + [0.5, 0.5, 1.25], b + [0.5, -0.25, 0.5], b + [0.5, 1.25, 0.5], b + [-0.25, 0.5, 0.5], b + [1.25, 0.5, 0.5]]
    return pts, q, pts[5], "tie"


def sqrt_equal_in_a_bucket(b, rng):
    q = b + [0.5, 0.5, 0.5]
    dz_a, dz_b = _sqrt_equal_dz(rng, 0.3125 * 0.3125)
    pts = [q + [-0.3125, 0.0, dz_a], q + [0.3125, 0.0, dz_b]]  # the later one is 1-2 ulp closer in d^2 - and no closer in norm
    return pts, q, pts[0], "sqrt"


def sqrt_equal_in_a_bucket_control(b, rng):
    q = b + [0.5, 0.5, 0.5]
    dz_a, dz_b = _sqrt_equal_dz(rng, 0.3125 * 0.3125)
    pts = [q + [0.3125, 0.0, dz_b], q + [-0.3125, 0.0, dz_a]]  # (the closer one first: it wins under either rule)
    return pts, q, pts[0], "sqrt"


def sqrt_equal_across_shifts(b, rng):
    q = b + [0.5, 0.5, 0.5]
    dz_a, dz_b = _sqrt_equal_dz(rng, 0.625 * 0.625)
    pts = [q + [-0.625, 0.0, dz_b], q + [0.625, 0.0, dz_a]]  # -x (shift 2), closer in d^2 | +x (shift 1), visited first
    return pts, q, pts[1], "sqrt"


def sqrt_equal_three(b, rng):
    q = b + [0.5, 0.5, 0.5]
    dz_a, dz_b = _sqrt_equal_dz(rng, 0.3125 * 0.3125)
    pts = [q + [-0.3125, 0.0, dz_a], q + [0.3125, 0.0, dz_b], q + [0.0, 0.3125, dz_b]]  # three within a hair: the exact search decides
    return pts, q, pts[0], "sqrt"


def margin_flip_in_a_bucket(b, rng):
    # query 0.3 units off the grid; A 0.25 - 0.1 u below it, B 0.25 + 0.1 u above it; the mirror rounds A away and B towards the query
    q = b + [0.5 + 0.3125 * U, 0.5, 0.5]
    pts = [b + [0.75 + 0.40625 * U, 0.5, 0.5], b + [0.25 + 0.40625 * U, 0.5, 0.5]]  # B first (the mirror's winner), A second (fp64's)
    return pts, q, pts[1], "margin"


def margin_flip_across_voxels(b, rng):
    q = b + [0.3125 * U, 0.5, 0.5]
    pts = [b + [0.375 + 0.40625 * U, 0.5, 0.5], b + [-0.375 + 0.40625 * U, 0.5, 0.5]]  # own voxel: the mirror's winner | -x voxel: fp64's
    return pts, q, pts[1], "margin"


def margin_three(b, rng):
    q = b + [0.5 + 0.3125 * U, 0.5 + 0.3125 * U, 0.5]
    pts = [b + [0.75 + 0.40625 * U, 0.5 + 0.3125 * U, 0.5], b + [0.5 + 0.3125 * U, 0.75 + 0.46875 * U, 0.5], b + [0.25 + 0.40625 * U, 0.5 + 0.3125 * U, 0.5]]
    return pts, q, pts[2], "margin"  # three inside the margin (exact search): the last one inserted is the closest


def exactly_at_tau(b, rng):
    q = b + [0.5, 0.5, 0.5]
    pts = [b + [0.5 + TAU, 0.5, 0.5]]
    return pts, q, None, "tau"


def two_exactly_at_tau(b, rng):
    q = b + [0.5, 0.5, 0.5]
    pts = [b + [0.5, 0.5 + TAU, 0.5], b + [0.5 - TAU, 0.5, 0.5]]
    return pts, q, None, "tau"


def just_inside_tau(b, rng):
    q = b + [0.5, 0.5, 0.5]
    t = b + [0.5 + TAU, 0.5, 0.5]
    t[0] = np.nextafter(t[0], -np.inf)
    return [t], q, t, "tau"


def at_tau_behind_a_tie_inside(b, rng):
    q = b + [0.5, 0.5, 0.5]
    pts = [b + [0.5 + TAU, 0.5, 0.5], b + [0.5, 0.5, 1.5], b + [0.5, 0.5, -0.5]]  # +x at tau exactly | +z and -z at 1.0: +z (shift 5) first
    return pts, q, pts[1], "tie"


CASES = [two_shifts, own_voxel_against_a_neighbour, face_against_edge, edge_against_a_later_face, corner_against_edge, bucket_two, bucket_five,
         bucket_and_neighbour, four_faces_behind_the_own_voxel, six_faces, sqrt_equal_in_a_bucket, sqrt_equal_in_a_bucket_control,
         sqrt_equal_across_shifts, sqrt_equal_three, margin_flip_in_a_bucket, margin_flip_across_voxels, margin_three, exactly_at_tau,
         two_exactly_at_tau, just_inside_tau, at_tau_behind_a_tie_inside]


class Scene:
    """map_points (insertion order), queries (world frame), expected (target per query, NaN rows = no correspondence), names, kinds,
    candidates (per query: the indices into map_points of its cell's points; empty for the padding)"""


def build(copies=1, fillers_per_case=3, seed=5):
    rng = np.random.default_rng(seed)
    pts, queries, expected, names, kinds, cand = [], [], [], [], [], []
    cell = 0

    def base():
        nonlocal cell
        i = cell
        cell += 1
        return np.array([8.0 * (i % 96) - 256.0, 8.0 * (i // 96) - 64.0, 0.0])  # integers: every sum below is exact

    for _ in range(copies):
        for f in CASES:
            b = base()
            p, q, want, kind = f(b, rng)
            first = len(pts)
            pts.extend(np.asarray(x, dtype=np.float64) for x in p)
            queries.append(np.asarray(q, dtype=np.float64))
            expected.append(np.full(3, np.nan) if want is None else np.asarray(want, dtype=np.float64))
            names.append(f.__name__), kinds.append(kind), cand.append(list(range(first, len(pts))))
            for _ in range(fillers_per_case):  # one candidate, own voxel only: a lane that is done after the first round
                b = base()
                t = b + rng.integers(4, 12, 3) / 16.0
                pts.append(t)
                queries.append(t + np.array([0.0625, -0.0625, 0.03125]))
                expected.append(t), names.append("filler"), kinds.append("filler"), cand.append([len(pts) - 1])
    order = rng.permutation(len(queries))  # tie queries and padding interleaved: both kinds in every wave
    s = Scene()
    s.map_points = np.array(pts)
    s.queries = np.array(queries)[order]
    s.expected = np.array(expected)[order]
    s.names = [names[i] for i in order]
    s.kinds = [kinds[i] for i in order]
    s.candidates = [cand[i] for i in order]
    return s


def premises(s):
    """each case is the tie it claims to be - from the coordinates alone"""
    for q, want, name, kind, cand in zip(s.queries, s.expected, s.names, s.kinds, s.candidates):
        c = s.map_points[cand]
        d2 = np.array([_d2(t, q) for t in c])
        nrm = np.sqrt(d2)
        if kind == "filler":
            assert len(c) == 1 and nrm[0] < 0.2
            continue
        if kind == "tau":
            if np.isnan(want[0]):
                assert (nrm == TAU).all(), name  # exactly at the threshold: `distance < tau` is false
            else:
                assert nrm[0] < TAU and nrm[0] > TAU * (1 - 1e-12), name
            continue
        w = int(np.flatnonzero((c == want).all(axis=1))[0])
        inside = nrm < TAU
        best = nrm[inside].min()
        tied = np.flatnonzero(inside & (nrm == best))
        assert w in tied, name
        if kind == "tie":
            assert len(tied) >= 2 and (d2[tied] == d2[w]).all(), name  # exactly equidistant, squared distances included
            others = [t for t in tied if t != w]
            assert min(np.linalg.norm(c[t] - c[w]) for t in others) >= 0.3, name  # a wrong pick is visible in the residual
        elif kind == "sqrt":
            assert len(tied) >= 2, name
            if name != "sqrt_equal_in_a_bucket_control":
                assert d2[tied].min() < d2[w], name  # somebody is closer in d^2 - by an ulp or two - and no closer in norm
            assert (d2[tied].max() - d2[tied].min()) <= 2.5 * np.spacing(d2[w]), name
        elif kind == "margin":
            md = np.array([_mirror_d2(t, q) for t in c])
            assert len(tied) == 1 and np.argmin(d2) == w and np.argmin(md) != w, name  # the mirror prefers somebody else
            assert nrm.max() - nrm.min() < 4 * U, name
