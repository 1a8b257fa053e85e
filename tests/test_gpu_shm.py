"""Multi-process mode on ONE GPU: two ranks (two processes sharing device 0) register their halves of a scan through
the node-wide shared segment (kicp_reg_shm_init) and must both return the bits of the single-process result."""
import multiprocessing as mp
import os
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu
GOLD = os.path.join(ROOT, "tests", "golden", "registration_small.npz")


def _worker(rank, world, name, barrier, q):
    sys.path.insert(0, ROOT)
    import kinematic_icp_amd as K
    from kinematic_icp_amd import sharding as sh
    g = np.load(GOLD)
    out = []
    try:
        reg = K.KinematicRegistration()
        if rank == 0:
            reg.shm_init(world, 0, name)
        barrier.wait()
        if rank != 0:
            reg.shm_init(world, rank, name)
        barrier.wait()
        for case in ("a", "b", "c"):
            m = K.VoxelHashMap(float(g[case + "_voxel"]), float(g[case + "_maxrange"]), 20)
            m.AddPoints(g[case + "_map"])
            frame = g[case + "_frame"]
            lo, hi = sh.shard_bounds(len(frame), world, rank)
            pose = reg.ComputeRobotMotion(frame[lo:hi], m, g[case + "_last"], g[case + "_rel"], float(g[case + "_tau"]))
            out.append((pose, reg.last_stats.iterations))
        barrier.wait()
        reg.shm_destroy()
        q.put((rank, out, None))
    except Exception as e:  # noqa: BLE001
        q.put((rank, out, repr(e)))
        try:
            barrier.abort()
        except Exception:  # noqa: BLE001
            pass


@pytest.mark.parametrize("world", [2, 3])
def test_two_processes_share_segment(world):
    import kinematic_icp_amd as K
    g = np.load(GOLD)
    ctx = mp.get_context("spawn")
    barrier, q = ctx.Barrier(world), ctx.Queue()
    name = "kicp_test_%d_%d" % (os.getpid(), world)
    procs = [ctx.Process(target=_worker, args=(r, world, name, barrier, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = {}
    for _ in range(world):
        rank, out, err = q.get(timeout=240)
        assert err is None, "rank %d: %s" % (rank, err)
        results[rank] = out
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    single = K.KinematicRegistration()
    for k, case in enumerate(("a", "b", "c")):
        m = K.VoxelHashMap(float(g[case + "_voxel"]), float(g[case + "_maxrange"]), 20)
        m.AddPoints(g[case + "_map"])
        ref = single.ComputeRobotMotion(g[case + "_frame"], m, g[case + "_last"], g[case + "_rel"], float(g[case + "_tau"]))
        for r in range(world):
            pose, iters = results[r][k]
            assert np.array_equal(pose, ref) and iters == single.last_stats.iterations == int(g[case + "_iters"])
