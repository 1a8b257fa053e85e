"""Hypothesis-driven fuzz of the fused pass: random scenes, voxel sizes, thresholds, bucket capacities and
poses through `pass_sums` (every build of the generic kernel in turn) and through `ComputeRobotMotion` on the kernel the library
picks, against the oracle - accepted counts exactly, sums to 1e-10 relative, poses to 1e-9, iteration counts equal.  200 cases.
Scenes are built to be awkward rather than pretty: points snapped to a coarse lattice (many exact ties and near-ties, cf.
tests/tie_cases.py), clusters that overflow buckets, queries on voxel faces, thresholds from a twentieth of a voxel to three voxels."""
import numpy as np
import pytest
from hypothesis import HealthCheck, given, seed, settings
from hypothesis import strategies as st

import kinematic_icp_amd as K
from checkers import okicp

pytestmark = pytest.mark.gpu
VARIANTS = [dict(), dict(small=0), dict(small=0, lanes_per_query=1), dict(small=0, lanes_per_query=1, latency_kernel=0), dict(small=0, lanes_per_query=4),
            dict(small=0, lanes_per_query=2), dict(small_wave=0)]
PASS_VARIANTS = [(None, None), (1, 2), (1, 0), (4, None), (2, None)]  # (sub-lanes per query, latency_kernel): tests/test_gpu_parity.py VARIANTS


@st.composite
def scenes(draw):
    rng = np.random.default_rng(draw(st.integers(0, 2**32 - 1)))
    vs = draw(st.sampled_from([0.1, 0.25, 0.5, 1.0, 1.0, 2.0, 0.3]))
    cap = draw(st.sampled_from([1, 3, 20, 20, 20, 40]))
    tau = vs * draw(st.sampled_from([0.05, 0.3, 0.67, 1.0, 1.5, 3.0]))
    n_map = draw(st.integers(50, 4000))
    kind = draw(st.sampled_from(["plane", "lattice", "clusters", "volume"]))
    extent = vs * draw(st.sampled_from([4.0, 12.0, 40.0]))
    centre = np.array([draw(st.sampled_from([0.0, 1234.5, -20000.25])), draw(st.sampled_from([0.0, -77.0])), 0.0])
    if kind == "plane":
        pts = np.concatenate([rng.uniform(-extent, extent, (n_map, 2)), rng.normal(0, 0.02 * vs, (n_map, 1))], 1)
    elif kind == "lattice":  # multiples of vs / 8: exact ties, points on voxel faces
        pts = np.round(rng.uniform(-extent, extent, (n_map, 3)) * 8 / vs) * vs / 8
    elif kind == "clusters":
        c = rng.uniform(-extent, extent, (max(1, n_map // 60), 3))
        pts = c[rng.integers(0, len(c), n_map)] + rng.normal(0, 0.3 * vs, (n_map, 3))
    else:
        pts = rng.uniform(-extent, extent, (n_map, 3)) * np.array([1.0, 1.0, 0.2])
    pts = pts + centre
    n_src = draw(st.integers(1, 3000))
    pick = pts[rng.integers(0, len(pts), n_src)]
    noise = draw(st.sampled_from([0.0, 0.02, 0.3, 1.0])) * vs
    world = pick + rng.normal(0, 1.0, (n_src, 3)) * noise
    if draw(st.booleans()):  # some queries nowhere near the map
        world[rng.integers(0, n_src, max(1, n_src // 10))] += 50.0 * vs
    pose = np.concatenate([[0, 0, np.sin(0.5 * (yaw := draw(st.sampled_from([0.0, 0.3, -2.0, np.pi])))), np.cos(0.5 * yaw)],
                           centre[:2] + rng.uniform(-1, 1, 2) * vs, [0.0]])
    src = okicp.se3_act(okicp.se3_inverse(pose), world)
    err = np.concatenate([[0, 0, np.sin(0.5 * (dy := draw(st.sampled_from([0.0, 0.002, 0.02])))), np.cos(0.5 * dy)], [draw(st.sampled_from([0.0, 0.01, 0.1])) * vs, 0.0, 0.0]])
    return dict(vs=vs, cap=cap, tau=tau, map=pts, src=src, pose=pose, rel=err, variant=draw(st.integers(0, 10**6)))


@seed(20260926)
@settings(max_examples=200, deadline=None, suppress_health_check=list(HealthCheck), derandomize=True, database=None)
@given(scenes())
def test_random_scenes_through_the_fused_pass(sc):
    from test_gpu_parity import _reg as _variant
    g = K.VoxelHashMap(sc["vs"], 1e6, sc["cap"])
    g.AddPoints(sc["map"])
    o = okicp.VoxelHashMap(sc["vs"], 1e6, sc["cap"])
    o.AddPoints(sc["map"])
    assert g.num_points() == o.num_points()
    # one pass at the fixed pose: a build of the generic kernel
    want, _ = okicp.icp_pass(o, sc["src"], sc["pose"], sc["tau"])
    got = _variant(*PASS_VARIANTS[sc["variant"] % len(PASS_VARIANTS)]).pass_sums(sc["src"], g, sc["pose"], sc["tau"])
    assert got[6] == want[6], (got[6], want[6])
    scale = np.maximum(np.abs(want[:6]), [1.0, 1.0, 1.0, 1e-3, 1e-3, 1e-6]) + sc["vs"] * np.abs(want[6]) * 1e-3
    assert np.all(np.abs(got[:6] - want[:6]) <= 1e-10 * scale + 1e-9), (got, want)
    # the whole registration on the kernel the library picks under one of the option sets
    opts = VARIANTS[(sc["variant"] // 7) % len(VARIANTS)]
    reg = K.KinematicRegistration()
    for k, v in opts.items():
        reg.set_option(k, v)
    oreg = okicp.KinematicRegistration()
    a = reg.ComputeRobotMotion(sc["src"], g, sc["pose"], sc["rel"], sc["tau"])
    b = oreg.ComputeRobotMotion(sc["src"], o, sc["pose"], sc["rel"], sc["tau"])
    if np.isnan(b).any():
        assert np.isnan(a).any()
    else:
        np.testing.assert_allclose(a, b, rtol=0, atol=1e-9 * max(1.0, float(np.abs(b[4:]).max())))
    assert reg.last_stats.iterations == oreg.last_stats.iterations
    k = min(oreg.last_stats.iterations, 32)
    np.testing.assert_array_equal(np.array(reg.last_stats.n_corr[:k]), np.array(oreg.last_stats.n_corr[:k]))
