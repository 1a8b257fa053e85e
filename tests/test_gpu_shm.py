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


# ---- sharded BATCHES with several scans in flight (kicp_register_device_batch with the segment attached) ------
def _batch_case():
    """twelve registrations on one map: different starts (iteration counts 1 .. 10), different sizes, a two-point scan (some ranks'
    shards are empty), a scan without correspondences"""
    g = np.load(GOLD)
    frame = g["b_frame"]
    frames, lasts, rels = [], [], []
    for i in range(12):
        yaw = 0.004 * (i % 5)
        rel = g["b_rel"].copy()
        rel[4] += 0.03 * (i % 4)  # translation x of the odometry guess
        rel[2], rel[3] = np.sin(yaw / 2), np.cos(yaw / 2)
        f = frame[: len(frame) - 61 * i]
        if i == 4:
            f = frame[7:9]
        if i == 9:
            f = np.full((300, 3), 900.0)
        frames.append(np.ascontiguousarray(f)), lasts.append(g["b_last"]), rels.append(rel)
    return g, frames, lasts, rels


def _batch_worker(rank, world, name, queues, barrier, q):
    sys.path.insert(0, ROOT)
    import kinematic_icp_amd as K
    from kinematic_icp_amd import sharding as sh
    try:
        g, frames, lasts, rels = _batch_case()
        reg = K.KinematicRegistration()
        reg.set_option("batch_queues", queues)
        if rank == 0:
            reg.shm_init(world, 0, name)
        barrier.wait()
        if rank != 0:
            reg.shm_init(world, rank, name)
        barrier.wait()
        m = K.VoxelHashMap(float(g["b_voxel"]), float(g["b_maxrange"]), 20)
        m.AddPoints(g["b_map"])
        shards = []
        for f in frames:
            lo, hi = sh.shard_bounds(len(f), world, rank)
            shards.append(K.DeviceFrame(f[lo:hi] if hi > lo else np.zeros((0, 3)), device=0))
        batch = reg.prepare_batch(shards, lasts, rels)
        out = []
        for _ in range(3):  # (the lanes' hand-off counters go on from call to call)
            poses = reg.ComputeRobotMotionBatch(batch, m, float(g["b_tau"])).copy()
            out.append((poses, np.array(batch.iterations).copy(), reg.get_option("batch_queue_passes")))
        # a single call between batches uses the segment's own area and stays in step, too
        lo, hi = sh.shard_bounds(len(frames[0]), world, rank)
        one = reg.ComputeRobotMotion(frames[0][lo:hi], m, lasts[0], rels[0], float(g["b_tau"]))
        poses = reg.ComputeRobotMotionBatch(batch, m, float(g["b_tau"])).copy()
        out.append((poses, np.array(batch.iterations).copy(), reg.get_option("batch_queue_passes")))
        barrier.wait()
        reg.shm_destroy()
        q.put((rank, out, one, None))
    except Exception as e:  # noqa: BLE001
        q.put((rank, None, None, repr(e)))
        try:
            barrier.abort()
        except Exception:  # noqa: BLE001
            pass


@pytest.mark.parametrize("world,queues", [(2, 4), (2, 2), (3, 4), (2, 1)])
def test_sharded_batch_with_scans_in_flight(world, queues):
    """Two / three ranks sharing the box's GPU, each with ITS shards of twelve scans, `queues` of them in flight per rank, the
    lanes' exchanges interleaved through the shared segment: every rank returns, for every scan, the bits of the single-GPU batch
    on the whole scans - iteration counts included; with one queue the batch runs the plain loop (a scan at a time), same bits."""
    import kinematic_icp_amd as K
    ctx = mp.get_context("spawn")
    barrier, q = ctx.Barrier(world), ctx.Queue()
    name = "kicp_batch_%d_%d_%d" % (os.getpid(), world, queues)
    procs = [ctx.Process(target=_batch_worker, args=(r, world, name, queues, barrier, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = {}
    for _ in range(world):
        rank, out, one, err = q.get(timeout=300)
        assert err is None, "rank %d: %s" % (rank, err)
        results[rank] = (out, one)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    g, frames, lasts, rels = _batch_case()
    m = K.VoxelHashMap(float(g["b_voxel"]), float(g["b_maxrange"]), 20)
    m.AddPoints(g["b_map"])
    single = K.KinematicRegistration()
    whole = single.prepare_batch([K.DeviceFrame(f, device=0) for f in frames], lasts, rels)
    want = single.ComputeRobotMotionBatch(whole, m, float(g["b_tau"])).copy()
    want_it = np.array(whole.iterations).copy()
    assert want_it.min() == 1 and want_it.max() >= 3 and np.isnan(want[9]).any()
    for r in range(world):
        out, one = results[r]
        for poses, iters, served in out:
            assert np.array_equal(poses, want, equal_nan=True) and np.array_equal(iters, want_it)
        if queues >= 2:
            assert out[-1][2] >= 4 * want_it[want_it < 10].sum()  # (the passes went through the lanes; the NaN scan's later passes are accounted, not run)
        else:
            assert out[-1][2] == 0
        assert np.array_equal(one, want[0])


def test_a_segment_is_retired_before_it_is_replaced():
    """kicp_reg_shm_destroy by rank 0 marks the segment dead before unlinking it, kicp_reg_shm_init waits for the NEW one:
    rank 1 re-initialises while rank 0 has not recreated the segment yet, then both exchange on the new segment - twice over, so the
    second round runs against a name that has just been retired.  Two handles in one process, one thread per rank."""
    import threading
    import kinematic_icp_amd as K
    from kinematic_icp_amd import sharding as sh
    g = np.load(GOLD)
    m = K.VoxelHashMap(float(g["a_voxel"]), float(g["a_maxrange"]), 20)
    m.AddPoints(g["a_map"])
    frame = g["a_frame"]
    ref = K.KinematicRegistration().ComputeRobotMotion(frame, m, g["a_last"], g["a_rel"], float(g["a_tau"]))
    regs = [K.KinematicRegistration(), K.KinematicRegistration()]
    name = "kicp_test_retire_%d" % os.getpid()
    for _ in range(2):
        out, errs = {}, []

        def late_rank(rank=1):
            try:
                regs[rank].shm_init(2, rank, name)  # (starts before rank 0 has created the segment: waits for it)
                lo, hi = sh.shard_bounds(len(frame), 2, rank)
                out[rank] = regs[rank].ComputeRobotMotion(frame[lo:hi], m, g["a_last"], g["a_rel"], float(g["a_tau"]))
            except Exception as e:  # noqa: BLE001
                errs.append(repr(e))

        t = threading.Thread(target=late_rank)
        t.start()
        import time
        time.sleep(0.2)
        regs[0].shm_init(2, 0, name)
        lo, hi = sh.shard_bounds(len(frame), 2, 0)
        out[0] = regs[0].ComputeRobotMotion(frame[lo:hi], m, g["a_last"], g["a_rel"], float(g["a_tau"]))
        t.join(timeout=120)
        assert not t.is_alive() and not errs, errs
        assert np.array_equal(out[0], ref) and np.array_equal(out[1], ref)
        regs[1].shm_destroy()
        regs[0].shm_destroy()  # (rank 0: marks the segment dead, then unlinks it)
        assert not os.path.exists("/dev/shm/" + name)
