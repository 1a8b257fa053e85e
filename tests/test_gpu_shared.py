"""The drop-in on a GPU it does not own: with the default options a registration of a small scan keeps its pass kernel
RESIDENT across the iterations of a call, polling for the host's next pose ("small_resident", kicp.h) - on a robot the same GPU also runs
perception.  A second PROCESS runs an unrelated kernel loop (torch: a bandwidth-bound and a compute-bound kernel) alone and then beside a
process that registers cfg4-sized scans (1 080 points, BASELINE.json configs[3]) back to back as fast as it can; the other process must
keep at least half of its solo rate, and the registrations their bits."""
import multiprocessing as mp
import os
import sys
import time

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _other_process(kind, seconds, start, q):
    import torch
    torch.cuda.set_device(0)
    if kind == "bandwidth":
        x = torch.zeros(64 << 20, dtype=torch.float32, device="cuda")  # 256 MB: every pass streams it through HBM
        step = lambda: x.add_(1.0)  # noqa: E731
    else:
        a = torch.randn(2048, 2048, dtype=torch.float16, device="cuda")
        b = torch.randn(2048, 2048, dtype=torch.float16, device="cuda")
        step = lambda: torch.matmul(a, b)  # noqa: E731
    for _ in range(20):
        step()
    torch.cuda.synchronize()
    q.put("ready")
    start.wait()
    rates = []
    for _ in range(2):  # [alone, beside the registrations] - the parent runs its loop during the second window
        n, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < seconds:
            for _ in range(10):
                step()
            torch.cuda.synchronize()
            n += 10
        rates.append(n / (time.perf_counter() - t0))
        q.put(("window", len(rates)))
        start.wait()
    q.put(("rates", rates))


@pytest.mark.parametrize("kind", ["bandwidth", "compute"])
def test_another_process_keeps_its_rate_beside_resident_registrations(kind):
    sys.path.insert(0, ROOT)
    import kinematic_icp_amd as K
    from kinematic_icp_amd import synthetic as syn
    cfg, scene, scans, rng = syn.make_case("cfg4", n_scans=4)
    gmap = K.VoxelHashMap(cfg.voxel_size, cfg.max_range, cfg.max_points_per_voxel)
    syn.build_map_points(scene, cfg, gmap.AddPoints, gmap.num_points, rng)
    tau = cfg.first_frame_tau()
    frames = [K.DeviceFrame(s["frame"]) for s in scans]
    rels = [syn.pose_mul(s["rel_odom"], syn.planar_pose(0.05, 0.0, np.deg2rad(0.8))) for s in scans]  # several iterations per call: the kernel stays
    reg = K.KinematicRegistration()  # default options: small_resident 1 (adaptive), resident_generic 1
    want = [reg.ComputeRobotMotion(f, gmap, s["last_pose"], r, tau).copy() for f, s, r in zip(frames, scans, rels)]
    assert reg.get_option("small_active") == 2.0 and reg.last_stats.iterations > 1
    seconds = 1.5
    ctx = mp.get_context("spawn")
    q, start = ctx.Queue(), ctx.Barrier(2)
    other = ctx.Process(target=_other_process, args=(kind, seconds, start, q))
    other.start()
    try:
        assert q.get(timeout=180) == "ready"
        start.wait()                                  # window 1: the other process alone
        assert q.get(timeout=60) == ("window", 1)
        start.wait()                                  # window 2: beside the registrations
        calls, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < seconds:
            for k in range(4):
                got = reg.ComputeRobotMotion(frames[k], gmap, scans[k]["last_pose"], rels[k], tau)
                assert np.array_equal(got, want[k])
            calls += 4
        mine = calls / (time.perf_counter() - t0)
        assert q.get(timeout=60) == ("window", 2)
        start.wait()
        tag, rates = q.get(timeout=60)
        assert tag == "rates"
    finally:
        other.join(timeout=60)
        if other.is_alive():
            other.kill()
    alone, beside = rates
    print("%s kernel loop: %.0f / s alone, %.0f / s beside %.0f registrations / s (%.0f %%)" % (kind, alone, beside, mine, 100.0 * beside / alone))
    assert mine > 1000.0
    assert beside >= 0.5 * alone, (alone, beside)


def test_device_locality_names_cpus_this_process_could_use():
    """kicp_device_locality: the GPU's NUMA node and CPU list from sysfs (what the library places its pinned buffers and helper threads
    by, and what a deployment binds its caller with): consistent with each other and with the process's own affinity mask"""
    import os
    import kinematic_icp_amd as K
    node, cpus = K.device_locality(0)
    assert node >= -1 and all(0 <= c < 4096 for c in cpus)
    if node >= 0 and os.path.isdir("/sys/devices/system/node/node%d" % node):
        assert cpus  # (a GPU with a node has that node's CPUs)
        with open("/sys/devices/system/node/node%d/cpulist" % node) as f:
            listed = set()
            for part in f.read().strip().split(","):
                lo, _, hi = part.partition("-")
                listed.update(range(int(lo), int(hi or lo) + 1))
        assert cpus <= listed
    near, near_l3 = K.cpus_near_gpu(0, one_l3_domain=False), K.cpus_near_gpu(0)
    assert near <= os.sched_getaffinity(0) and near_l3 <= (near or near_l3)
    with pytest.raises(K.KicpError):
        K.device_locality(4096)
