"""KAT-4 THROUGH THE FUSED PASS KERNELS (Registration.cpp:74-77): the tie scenes of
tests/tie_cases.py - candidates exactly equidistant across shifts and inside a bucket, 1-2 ulp apart in squared distance but equal
in norm, inside the 16-bit mirror's margin but ordered the other way in fp64, exactly at tau; 2, 3 and >= 4 of them - go through
`ComputeRobotMotion` and `pass_sums` (NOT GetClosestNeighbor) on every pass kernel the library can pick and on the batch paths
(the kernel resident across the scans, several scans in flight on queues of their own), against the oracle AND the reference build.
A wrong pick moves a residual by >= 0.3 m; the sums are compared to 1e-12, the poses to 1e-9, the accepted counts exactly."""
import numpy as np
import pytest

import kinematic_icp_amd as K
import tie_cases as tc
from checkers import okicp, ref_available, ref_map_like, rkicp
from test_gpu_edge import KERNELS, _reg
from test_ties import POSES, scan_for

pytestmark = pytest.mark.gpu
I = okicp.IDENTITY
CFG1 = dict(max_num_iteration=1, convergence_criterion=1e-3, max_num_threads=1, use_adaptive_odometry_regularization=True, fixed_regularization=0.0)
CFG10 = dict(CFG1, max_num_iteration=10)
FIXED1 = dict(CFG1, use_adaptive_odometry_regularization=False, fixed_regularization=0.25)


class World:
    def __init__(self, copies):
        self.scene = tc.build(copies)
        tc.premises(self.scene)
        self.g = K.VoxelHashMap(tc.VS, 100.0, tc.CAP)
        self.g.AddPoints(self.scene.map_points)
        self.o = okicp.VoxelHashMap(tc.VS, 100.0, tc.CAP)
        self.o.AddPoints(self.scene.map_points)
        assert self.g.num_points() == self.o.num_points() == len(self.scene.map_points)
        self.r = ref_map_like(self.o) if ref_available() else None
        self.accepted = float((~np.isnan(self.scene.expected[:, 0])).sum())
        self.scans = {name: scan_for(self.scene, pose) for name, pose in POSES.items()}
        self._want = {}

    def want(self, name, cfg):
        """(oracle pose, oracle stats, oracle pass sums, reference build's pose | None) of scan `name` under `cfg`"""
        key = (name, tuple(sorted(cfg.items())))
        if key not in self._want:
            oreg = okicp.KinematicRegistration(**cfg)
            a = oreg.ComputeRobotMotion(self.scans[name], self.o, POSES[name], I, tc.TAU)
            sums, _ = okicp.icp_pass(self.o, self.scans[name], POSES[name], tc.TAU)
            assert sums[6] == self.accepted
            b = rkicp.KinematicRegistration(**cfg).ComputeRobotMotion(self.scans[name], self.r, POSES[name], I, tc.TAU) if self.r is not None else None
            self._want[key] = (a, oreg.last_stats, sums, b)
        return self._want[key]


@pytest.fixture(scope="module")
def small():
    return World(1)       # 84 queries: one wave per query / sub-lanes per query / the generic kernels on two waves


@pytest.fixture(scope="module")
def medium():
    return World(60)      # 5 040 queries: k_pass_small by default (two sub-lanes per query)


@pytest.fixture(scope="module")
def large():
    return World(110)     # 9 240 queries: beyond the small-scan kernels - the generic kernel, resident for a call's later passes


def _run(reg, w, name, cfg, via):
    scan, pose = w.scans[name], POSES[name]
    want, ost, osums, ref = w.want(name, cfg)
    if via == "device":
        got = reg.ComputeRobotMotion(K.DeviceFrame(scan, device=0), w.g, pose, I, tc.TAU)
    elif via == "batch":
        batch = reg.prepare_batch([K.DeviceFrame(scan, device=0)] * 9, [pose] * 9, [I] * 9)
        poses = reg.ComputeRobotMotionBatch(batch, w.g, tc.TAU)
        assert all(np.array_equal(poses[0], poses[k]) for k in range(1, 9))
        assert [int(x) for x in batch.iterations] == [ost.iterations] * 9
        got = poses[0].copy()
    else:
        got = reg.ComputeRobotMotion(scan, w.g, pose, I, tc.TAU)
    np.testing.assert_allclose(got, want, rtol=0, atol=1e-9)
    if ref is not None:
        np.testing.assert_allclose(got, ref, rtol=0, atol=1e-9)
    if via != "batch":
        st = reg.last_stats
        assert st.iterations == ost.iterations and st.converged == ost.converged
        k = min(ost.iterations, 32)
        np.testing.assert_array_equal(np.array(st.n_corr[:k]), np.array(ost.n_corr[:k]))
        assert st.n_corr[0] == w.accepted  # nothing at tau was accepted, everything inside it was
        # the first pass's sums, which a wrong pick among tied candidates would move by >= 0.3 in JTr
        np.testing.assert_allclose(np.array(st.sums[0][:6]), osums[:6], rtol=1e-12, atol=1e-9)
    return got


@pytest.mark.parametrize("name,options", KERNELS, ids=[k for k, _ in KERNELS])
@pytest.mark.parametrize("via", ["host", "device", "batch"])
def test_ties_through_every_pass_kernel(small, name, options, via):
    for pose in POSES:
        for cfg in (CFG1, FIXED1, CFG10):
            _run(_reg(options, **cfg), small, pose, cfg, via)


@pytest.mark.parametrize("options", [{}, {"small": 0}, {"small": 0, "lanes_per_query": 1}, {"small": 0, "lanes_per_query": 1, "latency_kernel": 0},
                                     {"small": 0, "lanes_per_query": 4}, {"small_resident": 2}, {"small": 0, "lanes_per_query": 2}],
                         ids=["default", "generic", "latency_build", "four_waves_lending", "four_lanes", "resident", "two_lanes"])
def test_ties_on_scans_of_many_workgroups(medium, large, options):
    """the same cells sixty and a hundred-and-ten times over: every wave holds ties AND padding (the four-waves build lends idle lanes
    the tie queries' voxels), several groups of workgroups, k_pass_small and the resident generic kernel on their own turf"""
    for w in (medium, large):
        for pose in POSES:
            for cfg in (CFG1, CFG10):
                reg = _reg(options, **cfg)
                _run(reg, w, pose, cfg, "device")
                if cfg is CFG10 and w is large and options.get("small_resident") == 2:
                    assert reg.get_option("resident_passes") >= 2  # (the resident generic kernel did serve the passes)


VARIANTS = [(None, None), (1, 0), (1, 2), (2, None), (4, None)]  # (sub-lanes per query, latency_kernel): tests/test_gpu_parity.py VARIANTS


@pytest.mark.parametrize("variant", VARIANTS)
def test_pass_sums_of_the_tie_scenes(small, large, variant):
    """kicp_pass_sums: ONE fused association + accumulation pass at a fixed pose, every build of the generic kernel; then every tie
    case on its own (a scan of its query and nothing else), so that two wrong picks cannot cancel"""
    from test_gpu_parity import _reg as _variant
    reg = _variant(*variant)
    for w in (small, large):
        for name, pose in POSES.items():
            got = reg.pass_sums(w.scans[name], w.g, pose, tc.TAU)
            want = w.want(name, CFG1)[2]
            assert got[6] == want[6]
            np.testing.assert_allclose(got[:6], want[:6], rtol=1e-12, atol=1e-9)
    s = small.scene
    for i, nm in enumerate(s.names):
        if nm == "filler":
            continue
        got = reg.pass_sums(s.queries[i:i + 1], small.g, I, tc.TAU)
        want, _ = okicp.icp_pass(small.o, s.queries[i:i + 1], I, tc.TAU)
        assert got[6] == want[6] == (0.0 if np.isnan(s.expected[i, 0]) else 1.0), nm
        np.testing.assert_allclose(got[:6], want[:6], rtol=1e-12, atol=1e-12, err_msg=nm)
        if got[6]:  # the residual IS the pick: r = q - target, JTr0 = r.x at the identity, ssq = |r|^2
            r = s.queries[i] - s.expected[i]
            np.testing.assert_allclose([got[3], got[5]], [r[0], r @ r], rtol=1e-12, atol=1e-12, err_msg=nm)


@pytest.mark.parametrize("mode", ["queues", "queues_headline_kernel", "resident_across_scans", "plain_loop"])
def test_ties_in_batches_with_several_scans_in_flight(small, large, mode):
    """kicp_register_device_batch on tie scans: four queues (the headline's path: large scans, the four-waves build with lent lanes),
    the kernel resident across the scans with three of them in flight, and the plain loop - all bit-equal to one call per scan, and
    equal to the oracle and the reference build"""
    options = {"queues": {"batch_queues": 4}, "queues_headline_kernel": {"batch_queues": 4, "small": 0, "lanes_per_query": 1},  # k_pass_gather32<256, 1, 4, false, false>
               "resident_across_scans": {"batch_queues": 0, "batch_depth": 3}, "plain_loop": {"batch_queues": 0, "batch_resident": 0}}[mode]
    for w in (large, small):
        names = list(POSES) * 4  # twelve scans, three poses
        dev = {n: K.DeviceFrame(w.scans[n], device=0) for n in POSES}
        for cfg in (CFG1, CFG10):
            reg = _reg(options, **cfg)
            batch = reg.prepare_batch([dev[n] for n in names], [POSES[n] for n in names], [I] * len(names))
            before = reg.get_option("batch_queue_passes"), reg.get_option("batch_resident_passes")
            got = reg.ComputeRobotMotionBatch(batch, w.g, tc.TAU).copy()
            single = _reg({"batch_queues": 0, "batch_resident": 0}, **cfg)
            for k, n in enumerate(names):
                want, ost, _, ref = w.want(n, cfg)
                np.testing.assert_allclose(got[k], want, rtol=0, atol=1e-9)
                if ref is not None:
                    np.testing.assert_allclose(got[k], ref, rtol=0, atol=1e-9)
                assert int(batch.iterations[k]) == ost.iterations
                if k < 3:
                    one = single.ComputeRobotMotion(dev[n], w.g, POSES[n], I, tc.TAU)
                    assert np.array_equal(got[k], one)
            if mode == "queues_headline_kernel" or (mode == "queues" and w is large):
                assert reg.get_option("batch_queue_passes") > before[0]
            if mode == "resident_across_scans":
                assert reg.get_option("batch_resident_passes") > before[1]
