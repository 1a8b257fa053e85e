"""KAT-4 on the CPU checkers: the tie scenes of tests/tie_cases.py are what they claim to be, the oracle
resolves every one of them to the expected neighbour, and the reference build (oracle/_ref: the reference's own Registration.cpp
over the kiss-icp stand-in) does the same, bit for bit.  tests/test_gpu_ties.py sends the same scenes through the HIP path."""
import numpy as np
import pytest

import tie_cases as tc
from checkers import okicp, ref_available, ref_map_like, rkicp

I = okicp.IDENTITY
POSES = {"identity": I, "translation": np.array([0, 0, 0, 1, 0, 64.0, 0]), "half_turn": np.array([0, 0, 1.0, 0, 0, 0, 0])}


def scan_for(scene, pose):
    """the scan that `pose` maps onto the scene's queries EXACTLY (the three poses are exact in fp64: no rotation arithmetic rounds)"""
    p = okicp.se3_act(okicp.se3_inverse(pose), scene.queries)
    assert np.array_equal(okicp.se3_act(pose, p), scene.queries)
    return p


@pytest.mark.parametrize("copies", [1, 7])
def test_tie_scenes_are_ties_and_the_checkers_agree_on_them(copies):
    s = tc.build(copies)
    tc.premises(s)
    o = okicp.VoxelHashMap(tc.VS, 100.0, tc.CAP)
    o.AddPoints(s.map_points)
    assert o.num_points() == len(s.map_points)  # nothing was dropped by the spacing rule: insertion order == slot order
    nn, d = o.GetClosestNeighbor(s.queries)
    accepted = ~np.isnan(s.expected[:, 0])
    assert np.array_equal(d < tc.TAU, accepted)
    assert np.array_equal(nn[accepted], s.expected[accepted])
    want_n = float(accepted.sum())
    for name, pose in POSES.items():
        scan = scan_for(s, pose)
        sums, _ = okicp.icp_pass(o, scan, pose, tc.TAU)
        assert sums[6] == want_n, name
        # the sums from the expected targets alone (App. B.1's closed form: J = [R ux | R (-sy, sx, 0)], r = T s - t)
        R = np.array([[1.0, 0, 0], [0, 1, 0], [0, 0, 1]]) if name != "half_turn" else np.array([[-1.0, 0, 0], [0, -1, 0], [0, 0, 1]])
        src, r = scan[accepted], s.queries[accepted] - s.expected[accepted]
        j0 = R @ np.array([1.0, 0, 0])
        j1 = (R @ np.stack([-src[:, 1], src[:, 0], np.zeros(len(src))])).T
        want = [want_n, float(np.sum(j1 @ j0)), float(np.sum(j1 * j1)), float(np.sum(r @ j0)), float(np.sum(j1 * r)), float(np.sum(r * r))]
        np.testing.assert_allclose(sums[:6], want, rtol=1e-12, atol=1e-9)
    if ref_available():
        r = ref_map_like(o)
        nn_r, d_r = r.GetClosestNeighbor(s.queries)
        assert np.array_equal(nn_r, nn) and np.array_equal(d_r, d)
        for name, pose in POSES.items():
            scan = scan_for(s, pose)
            for max_it in (1, 10):
                cfg = dict(max_num_iteration=max_it, convergence_criterion=1e-3, max_num_threads=1, use_adaptive_odometry_regularization=True, fixed_regularization=0.0)
                a = okicp.KinematicRegistration(**cfg).ComputeRobotMotion(scan, o, pose, I, tc.TAU)
                b = rkicp.KinematicRegistration(**cfg).ComputeRobotMotion(scan, r, pose, I, tc.TAU)
                assert np.array_equal(a, b), (name, max_it)
