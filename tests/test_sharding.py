"""The N>1 path on CPU: world_size-2 processes over gloo run the sharded registration protocol (contiguous point
shards, replicated map, one int64 limb all-reduce per ICP iteration, identical solve on every rank) with the oracle
standing in for the per-shard GPU pass.  Checks: every rank ends with the bit-identical pose, equal to the unsharded
result, for G = 2; plus G in {1,2,4,8} emulated in-process."""
import os
import sys

import numpy as np
import pytest

from conftest import ROOT
from kinematic_icp_amd import sharding as sh
from kinematic_icp_amd import synthetic as syn
from oracle import okicp


def _world(seed=31):
    rng = np.random.Generator(np.random.PCG64(seed))
    scene = syn.make_scene(rng, half=14.0, height=4.0, n_boxes=5, box_xy=(2.0, 5.0), box_z=(1.5, 3.5), keep_clear=2.5)
    cfg = syn.Config("shard", 8, 256, 6000, max_range=40.0, sensor_height=1.2)
    omap = okicp.VoxelHashMap(cfg.voxel_size, cfg.max_range, cfg.max_points_per_voxel)
    syn.build_map_points(scene, cfg, omap.AddPoints, omap.num_points, rng, batch=4000)
    true_pose = syn.planar_pose(0.7, -0.4, 0.3)
    frame = syn.make_scan(scene, true_pose, syn.beam_directions(8, 256), cfg.sensor_height, rng)
    rel = syn.planar_pose(0.4, 0.0, np.deg2rad(2.0))
    last = syn.pose_mul(syn.pose_mul(true_pose, syn.planar_pose(0.2, 0.0, np.deg2rad(1.0))), syn.pose_inverse(rel))
    return cfg, omap, frame, last, rel


def shard_pass_fixed(omap, shard, T, tau):
    """What one rank's pass kernel produces for its shard: the seven sums as exact fixed-point integers."""
    tot = [0] * sh.NUM_SUMS
    if len(shard) == 0:
        return tot
    q = okicp.se3_act(T, shard)
    nn, d = omap.GetClosestNeighbor(q)
    keep = d < tau
    s, r = shard[keep], q[keep] - nn[keep]
    rot = lambda v: okicp.se3_act(np.concatenate([T[:4], np.zeros(3)]), v)  # noqa: E731
    j0 = rot(np.array([[1.0, 0.0, 0.0]]))[0]
    j1 = rot(np.stack([-s[:, 1], s[:, 0], np.zeros(len(s))], 1))
    terms = [np.full(len(s), j0 @ j0), j1 @ j0, np.sum(j1 * j1, 1), r @ j0, np.sum(j1 * r, 1), np.sum(r * r, 1), np.ones(len(s))]
    for i, t in enumerate(terms):
        tot[i] = int(sum(sh.quantize(x) for x in t))
    return tot


def sharded_registration(omap, frame, last, rel, tau, world, rank, allreduce, max_iter=10, conv=1e-3):
    lo, hi = sh.shard_bounds(len(frame), world, rank)
    shard = frame[lo:hi]
    T = okicp.se3_mul(last, rel)
    beta = None
    for it in range(max_iter):
        words = sh.pack(shard_pass_fixed(omap, shard, T, tau))
        words = allreduce(words)
        sums = sh.unpack(words)
        if it == 0:
            beta = 1.0 / (sums[5] / sums[6] + np.finfo(np.float64).tiny)
        dx = okicp.solve(sums[:5], sums[6], beta)
        T = okicp.se3_mul(T, okicp.motion_model(dx))
        if np.hypot(dx[0], dx[1]) < conv:
            return T, it + 1
    return T, max_iter


def test_limb_roundtrip():
    rng = np.random.default_rng(0)
    for t in [0, 1, -1, (1 << 100) + 12345, -(1 << 100) - 999, (1 << 40) - 1, -(1 << 40)] + [int(x) for x in rng.integers(-2**62, 2**62, 50)]:
        assert sh.from_limbs(sh.to_limbs(t)) == t
    parts = [int(x) << 30 for x in rng.integers(-2**50, 2**50, 64)]
    words = np.sum([sh.pack([p] * 7) for p in parts], axis=0)  # summing limb-wise == summing the integers
    assert sh.from_limbs(words[:3]) == sum(parts)
    assert sh.shard_bounds(10, 4, 0) == (0, 2) and sh.shard_bounds(10, 4, 3) == (7, 10)
    assert sum(b - a for a, b in (sh.shard_bounds(131072, 8, r) for r in range(8))) == 131072


def test_emulated_world_sizes_give_identical_bits():
    cfg, omap, frame, last, rel = _world()
    tau = cfg.first_frame_tau()
    ref = okicp.KinematicRegistration()
    expect = ref.ComputeRobotMotion(frame, omap, last, rel, tau)
    results = {}
    for g in (1, 2, 4, 8):
        # emulate the collective: every rank contributes its words, everybody sees the sum
        def run(g=g):
            # lock-step emulation
            T = [okicp.se3_mul(last, rel) for _ in range(g)]
            beta, its = None, 0
            for it in range(10):
                words = np.sum([sh.pack(shard_pass_fixed(omap, frame[slice(*sh.shard_bounds(len(frame), g, r))], T[r], tau)) for r in range(g)], 0)
                sums = sh.unpack(words)
                if it == 0:
                    beta = 1.0 / (sums[5] / sums[6] + np.finfo(np.float64).tiny)
                dx = okicp.solve(sums[:5], sums[6], beta)
                T = [okicp.se3_mul(t, okicp.motion_model(dx)) for t in T]
                its = it + 1
                if np.hypot(dx[0], dx[1]) < 1e-3:
                    break
            for t in T[1:]:
                assert np.array_equal(t, T[0])
            return T[0], its
        results[g] = run()
    for g in (2, 4, 8):
        assert np.array_equal(results[g][0], results[1][0]) and results[g][1] == results[1][1]  # exact, not approximately
    assert results[1][1] == ref.last_stats.iterations
    np.testing.assert_allclose(results[1][0], expect, atol=1e-9)


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg, omap, frame, last, rel = _world()

    def allreduce(words):
        t = torch.from_numpy(words.copy())
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return t.numpy()

    T, its = sharded_registration(omap, frame, last, rel, cfg.first_frame_tau(), world, rank, allreduce)
    np.save(os.path.join(out_dir, "pose_%d.npy" % rank), np.concatenate([T, [its]]))
    dist.barrier()
    dist.destroy_process_group()


def test_gloo_world_size_2(tmp_path):
    import torch.multiprocessing as mp
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    p0, p1 = np.load(tmp_path / "pose_0.npy"), np.load(tmp_path / "pose_1.npy")
    assert np.array_equal(p0, p1)  # every rank ran the identical solve on identical all-reduced integers
    cfg, omap, frame, last, rel = _world()
    ref = okicp.KinematicRegistration()
    expect = ref.ComputeRobotMotion(frame, omap, last, rel, cfg.first_frame_tau())
    assert int(p0[7]) == ref.last_stats.iterations
    np.testing.assert_allclose(p0[:7], expect, atol=1e-9)


def test_gloo_world_size_8(tmp_path):
    """The same worker with eight ranks (the node size the north star names): every rank ends with the single-process bits."""
    import torch.multiprocessing as mp
    port = 31500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(8, port, str(tmp_path)), nprocs=8, join=True)
    poses = [np.load(tmp_path / ("pose_%d.npy" % r)) for r in range(8)]
    assert all(np.array_equal(p, poses[0]) for p in poses[1:])
    cfg, omap, frame, last, rel = _world()
    ref = okicp.KinematicRegistration()
    expect = ref.ComputeRobotMotion(frame, omap, last, rel, cfg.first_frame_tau())
    assert int(poses[0][7]) == ref.last_stats.iterations
    np.testing.assert_allclose(poses[0][:7], expect, atol=1e-9)


def _lane_worker(rank, world, lanes, shm_name, out_dir, seed):
    """one rank of a sharded BATCH with `lanes` scans in flight: the library's loop (run_batch_queues, sharded) with the oracle
    standing in for the GPU's pass, non-blocking looks at the lanes in turn, random pauses so that the lanes' passes complete in a
    different order on every rank"""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import time
    from multiprocessing import shared_memory
    seg = shared_memory.SharedMemory(name=shm_name)
    cfg, omap, frame, last, rel = _world()
    tau = cfg.first_frame_tau()
    scans = _batch_of_scans(frame, last, rel)
    ex = sh.SegmentLanes(seg.buf, world, rank, lanes)
    rng = np.random.default_rng(seed + rank)
    state = [dict(todo=sh.lane_scans(j, lanes, len(scans)), k=None, waiting=None) for j in range(lanes)]
    poses = [None] * len(scans)
    deadline = time.time() + 120
    while any(st["todo"] or st["k"] is not None for st in state):
        assert time.time() < deadline, "a lane never got its peers' hand-off"
        for j, st in enumerate(state):
            if st["k"] is None:
                if not st["todo"]:
                    continue
                st["k"] = st["todo"].pop(0)
                fr, la, re = scans[st["k"]]
                lo, hi = sh.shard_bounds(len(fr), world, rank)
                st.update(shard=fr[lo:hi], T=okicp.se3_mul(la, re), it=0, beta=None)
            if st["waiting"] is None:
                if rng.random() < 0.3:
                    time.sleep(rng.random() * 2e-3)  # this rank's GPU is late with this lane's rows
                    continue
                st["waiting"] = ex.publish(j, sh.pack(shard_pass_fixed(omap, st["shard"], st["T"], tau)))
            words = ex.collect(j, st["waiting"])
            if words is None:
                continue
            st["waiting"] = None
            sums = sh.unpack(words)
            if st["it"] == 0:
                st["beta"] = 1.0 / (sums[5] / sums[6] + np.finfo(np.float64).tiny)
            dx = okicp.solve(sums[:5], sums[6], st["beta"])
            st["T"] = okicp.se3_mul(st["T"], okicp.motion_model(dx))
            st["it"] += 1
            if np.hypot(dx[0], dx[1]) < 1e-3 or st["it"] >= 10:
                poses[st["k"]] = np.concatenate([st["T"], [st["it"]]])
                st["k"] = None
    np.save(os.path.join(out_dir, "lanes_%d.npy" % rank), np.array(poses))
    del ex
    seg.close()


def _batch_of_scans(frame, last, rel):
    """eleven registrations that differ in size, start and iteration count (one of two points: some ranks' shards are empty)"""
    out = []
    for i in range(11):
        rel_i = syn.pose_mul(rel, syn.planar_pose(0.02 * i, 0.0, np.deg2rad(0.3 * i)))
        out.append((frame[: len(frame) - 97 * i] if i != 5 else frame[100:102], last, rel_i))
    return out


@pytest.mark.parametrize("world,lanes", [(8, 4), (3, 2)])
def test_sharded_batch_with_scans_in_flight_over_the_segment_lanes(tmp_path, world, lanes):
    """The interleaved protocol of sharded batches (kinematic_icp_amd/sharding.py::SegmentLanes = kicp_reg_queues.hip run_batch_queues,
    `sharded`): eight ranks (the node size the north star names), four scans in flight on each, the lanes' hand-offs completing in a
    different order on every rank - every rank ends with the same bits for every scan, equal to registering the scan sharded in
    lock step (one scan at a time), and to the unsharded oracle to 1e-9."""
    import multiprocessing as mp
    from multiprocessing import shared_memory
    seg = shared_memory.SharedMemory(create=True, size=sh.SegmentLanes.nbytes(world, lanes))
    try:
        np.ndarray((seg.size // 8,), dtype=np.int64, buffer=seg.buf)[:] = 0
        ctx = mp.get_context("spawn")
        procs = [ctx.Process(target=_lane_worker, args=(r, world, lanes, seg.name, str(tmp_path), 1234)) for r in range(world)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(timeout=300)
            assert p.exitcode == 0
    finally:
        seg.close()
        seg.unlink()
    got = [np.load(tmp_path / ("lanes_%d.npy" % r)) for r in range(world)]
    assert all(np.array_equal(g, got[0]) for g in got[1:])
    cfg, omap, frame, last, rel = _world()
    tau = cfg.first_frame_tau()
    for k, (fr, la, re) in enumerate(_batch_of_scans(frame, last, rel)):
        # lock step, one scan at a time (the emulated collective of test_emulated_world_sizes_give_identical_bits)
        T, beta, its = okicp.se3_mul(la, re), None, 0
        for it in range(10):
            words = np.sum([sh.pack(shard_pass_fixed(omap, fr[slice(*sh.shard_bounds(len(fr), world, r))], T, tau)) for r in range(world)], 0)
            sums = sh.unpack(words)
            if it == 0:
                beta = 1.0 / (sums[5] / sums[6] + np.finfo(np.float64).tiny)
            dx = okicp.solve(sums[:5], sums[6], beta)
            T, its = okicp.se3_mul(T, okicp.motion_model(dx)), it + 1
            if np.hypot(dx[0], dx[1]) < 1e-3:
                break
        assert np.array_equal(got[0][k][:7], T, equal_nan=True) and int(got[0][k][7]) == its, k
        if len(fr) > 100:
            ref = okicp.KinematicRegistration()
            np.testing.assert_allclose(T, ref.ComputeRobotMotion(fr, omap, la, re, tau), atol=1e-9)
            assert its == ref.last_stats.iterations
    assert sh.lane_scans(1, 4, 11) == [1, 5, 9] and sorted(sum((sh.lane_scans(j, lanes, 11) for j in range(lanes)), [])) == list(range(11))


def _group_lane_worker(rank, world, lanes, port, out_dir, seed):
    """one rank of a sharded batch whose lanes exchange through COLLECTIVES (kicp_reg_queues.hip run_batch_queues, `over_rccl`): lane j owns a
    sub-group of its own (ncclCommSplit there, dist.new_group here), its all-reduces go out asynchronously in the lane's own fixed order
    - scans j, j + lanes, ..., pass by pass - and are polled for, so the lanes' collectives interleave in a different order on every
    rank (random pauses) while every group sees the same sequence everywhere; the oracle stands in for the GPU's pass"""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import time
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    groups = [dist.new_group(list(range(world))) for _ in range(lanes)]  # (created in the same order on every rank, like the splits)
    cfg, omap, frame, last, rel = _world()
    tau = cfg.first_frame_tau()
    scans = _batch_of_scans(frame, last, rel)
    rng = np.random.default_rng(seed + rank)
    state = [dict(todo=sh.lane_scans(j, lanes, len(scans)), k=None, work=None, buf=None) for j in range(lanes)]
    poses = [None] * len(scans)
    deadline = time.time() + 200
    while any(st["todo"] or st["k"] is not None for st in state):
        assert time.time() < deadline, "a lane's collective never completed"
        for j, st in enumerate(state):
            if st["k"] is None:
                if not st["todo"]:
                    continue
                st["k"] = st["todo"].pop(0)
                fr, la, re = scans[st["k"]]
                lo, hi = sh.shard_bounds(len(fr), world, rank)
                st.update(shard=fr[lo:hi], T=okicp.se3_mul(la, re), it=0, beta=None)
            if st["work"] is None:
                if rng.random() < 0.3:
                    time.sleep(rng.random() * 2e-3)  # this rank's GPU is late with this lane's pass
                    continue
                st["buf"] = torch.from_numpy(sh.pack(shard_pass_fixed(omap, st["shard"], st["T"], tau)).copy())
                st["work"] = dist.all_reduce(st["buf"], op=dist.ReduceOp.SUM, group=groups[j], async_op=True)
            if not st["work"].is_completed():
                continue
            st["work"].wait()
            st["work"] = None
            sums = sh.unpack(st["buf"].numpy())
            if st["it"] == 0:
                st["beta"] = 1.0 / (sums[5] / sums[6] + np.finfo(np.float64).tiny)
            dx = okicp.solve(sums[:5], sums[6], st["beta"])
            st["T"] = okicp.se3_mul(st["T"], okicp.motion_model(dx))
            st["it"] += 1
            if np.hypot(dx[0], dx[1]) < 1e-3 or st["it"] >= 10:
                poses[st["k"]] = np.concatenate([st["T"], [st["it"]]])
                st["k"] = None
    np.save(os.path.join(out_dir, "glanes_%d.npy" % rank), np.array(poses))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,lanes", [(2, 4), (8, 4)])
def test_sharded_batch_with_a_collective_group_per_lane(tmp_path, world, lanes):
    """What --comm rccl runs since round 6: four sharded scans in flight per rank, lane j's all-reduces on a communicator of its own.
    World sizes 2 and 8 over gloo sub-groups: every rank ends with the same bits for every scan, equal to the lock-step sharded
    registration (the segment-lanes test above computes the same reference) and to the unsharded oracle."""
    import torch.multiprocessing as mp
    port = 33500 + (os.getpid() % 2000) + world
    mp.spawn(_group_lane_worker, args=(world, lanes, port, str(tmp_path), 4321), nprocs=world, join=True)
    got = [np.load(tmp_path / ("glanes_%d.npy" % r)) for r in range(world)]
    assert all(np.array_equal(g, got[0]) for g in got[1:])
    cfg, omap, frame, last, rel = _world()
    tau = cfg.first_frame_tau()
    for k, (fr, la, re) in enumerate(_batch_of_scans(frame, last, rel)):
        T, beta, its = okicp.se3_mul(la, re), None, 0
        for it in range(10):
            words = np.sum([sh.pack(shard_pass_fixed(omap, fr[slice(*sh.shard_bounds(len(fr), world, r))], T, tau)) for r in range(world)], 0)
            sums = sh.unpack(words)
            if it == 0:
                beta = 1.0 / (sums[5] / sums[6] + np.finfo(np.float64).tiny)
            dx = okicp.solve(sums[:5], sums[6], beta)
            T, its = okicp.se3_mul(T, okicp.motion_model(dx)), it + 1
            if np.hypot(dx[0], dx[1]) < 1e-3:
                break
        assert np.array_equal(got[0][k][:7], T, equal_nan=True) and int(got[0][k][7]) == its, k
        if len(fr) > 100:
            ref = okicp.KinematicRegistration()
            np.testing.assert_allclose(T, ref.ComputeRobotMotion(fr, omap, la, re, tau), atol=1e-9)


def test_small_scan_rows_add_up_exactly():
    """The small-scan kernels' row format (two 48-bit halves per 128-bit sum, kicp_small.hpp) against the limb payload: totals of
    either sign up to the accumulation range (|term| < 2^43, 1024 terms per workgroup), 272 rows, stale and marked rows."""
    from kinematic_icp_amd import sharding as sh
    rng = np.random.Generator(np.random.PCG64(9))
    tag = 0x1234
    per_row, rows = [], []
    for k in range(272):
        scale = [1.0, 1e3, 8.7e12][k % 3]
        totals = [int(sum(sh.quantize(x) for x in rng.uniform(-scale, scale, 64))) for _ in range(sh.NUM_SUMS)]
        per_row.append(totals)
        rows.append(sh.small_row(totals, tag))
    words, flags, ok = sh.add_small_rows(rows, tag)
    assert ok and flags == 0
    want = [sum(t[i] for t in per_row) for i in range(sh.NUM_SUMS)]
    assert [sh.from_limbs(words[3 * i:3 * i + 3]) for i in range(sh.NUM_SUMS)] == want
    assert any(w < 0 for w in want) and max(abs(w) for w in want) > 1 << 90
    np.testing.assert_array_equal(words, sh.pack(want))
    # a row of the previous pass, a torn row and the markers
    stale = list(rows)
    stale[5] = sh.small_row(per_row[5], tag - 1)
    assert not sh.add_small_rows(stale, tag)[2]
    torn = [r.copy() for r in rows]
    torn[7][3] = sh.small_row(per_row[7], tag - 1)[3]
    assert not sh.add_small_rows(torn, tag)[2]
    marked = list(rows)
    marked[100] = sh.small_row([0] * 7, tag, flags=2)   # "gave up": the host launches afresh
    marked[101] = sh.small_row(per_row[101], tag, flags=1)  # range error
    assert sh.add_small_rows(marked, tag)[1] == 3


def test_tagged_rows_carry_group_sums_exactly():
    """The hand-off's word format (value << 16 | tag): a workgroup's limbs, the sum of a group's 32 rows and the top limb's
    sign all survive the round trip; a row with a stale tag is recognised; the tag never collides with value bits."""
    from kinematic_icp_amd import sharding as sh
    rng = np.random.Generator(np.random.PCG64(3))
    tag = 0xBEEF
    rows = []
    for _ in range(sh.GROUP):
        # 512 per-lane terms of either sign, each below 2^43 in magnitude (the documented range), 7 sums
        totals = [int(sum(sh.quantize(x) for x in rng.uniform(-8.7e12, 8.7e12, 512))) for _ in range(sh.NUM_SUMS)]
        rows.append(sh.pack(totals))
    # worst case of the unsigned limbs: all ones
    rows[0][:2] = sh.LIMB_MASK
    group = np.sum(np.stack(rows), axis=0)
    assert int(group[:21].max()) < (1 << 47) and int(group[:21].min()) > -(1 << 47)
    for words in rows + [group]:
        back, ok = sh.untag_row(sh.tag_row(words, tag), tag)
        assert ok and np.array_equal(back, words)
    # stale or half-written rows are not accepted
    stale = sh.tag_row(rows[1], tag - 1)
    assert not sh.untag_row(stale, tag)[1]
    mixed = sh.tag_row(rows[1], tag)
    mixed[5] = stale[5]
    assert not sh.untag_row(mixed, tag)[1]
    # the host's sum of untagged group rows equals the plain sum of all workgroup rows (what the old device tree produced)
    groups = [np.sum(np.stack(rows[i:i + 8]), axis=0) for i in range(0, sh.GROUP, 8)]
    host_total = sum(sh.untag_row(sh.tag_row(g, tag), tag)[0] for g in groups)
    assert np.array_equal(host_total, group)
    np.testing.assert_array_equal(sh.unpack(host_total), sh.unpack(group))
