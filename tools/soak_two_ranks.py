"""Soak of the N-rank launch path of bench.py without paying a process start per repetition: under torch.distributed.run every rank
repeats, `--cycles` times, what one bench.py run does with an exchange - attach it (shared segment / peer mailboxes), settle for a
while (the ranks agree on every "another block?" like bench.py's loop), time a few batches between barriers, detach - with random
pauses on random ranks in between, so that the ranks arrive at every collective with changing skew.  Every cycle's poses are
compared across the ranks (bit-equal) and with the first cycle's.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 \
        tools/soak_two_ranks.py --cycles 200 [--device 0]      # --device: every rank on that GPU (1-GPU boxes)
Prints one JSON line on rank 0: cycles run, failures, seconds.
"""
import argparse
import json
import os
import random
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

ap = argparse.ArgumentParser()
ap.add_argument("--cycles", type=int, default=200)
ap.add_argument("--device", type=int, default=-1)
ap.add_argument("--workload", default="cfg1")
ap.add_argument("--settle-s", type=float, default=0.05)
args = ap.parse_args()

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import kinematic_icp_amd as K  # noqa: E402
from kinematic_icp_amd import synthetic as syn  # noqa: E402

world, rank = int(os.environ["WORLD_SIZE"]), int(os.environ["RANK"])
device = args.device if args.device >= 0 else int(os.environ.get("LOCAL_RANK", "0"))
torch.cuda.set_device(device)
dist.init_process_group(backend="gloo")
cfg, scene, scans, rng = syn.make_case(args.workload, n_scans=4)
gmap = K.VoxelHashMap(cfg.voxel_size, cfg.max_range, cfg.max_points_per_voxel)
ident = np.array([0, 0, 0, 1.0, 0, 0, 0])
syn.build_map_points(scene, cfg, lambda pts: gmap.UpdateDevice(K.DeviceFrame(pts, device=device), ident), gmap.num_points, rng)
gmap.sync(device)
tau = cfg.first_frame_tau()
n = scans[0]["frame"].shape[0]
lo, hi = n * rank // world, n * (rank + 1) // world
frames = [K.DeviceFrame(s["frame"][lo:hi], device=device) for s in scans]
extra = syn.planar_pose(0.05, 0.0, np.deg2rad(0.5))
rels = [syn.pose_mul(s["rel_odom"], extra) for s in scans]
jitter = random.Random(1234 + rank)


def pause():
    if jitter.random() < 0.3:
        time.sleep(jitter.random() * 0.004)


def attach(kind, cycle):
    reg = K.KinematicRegistration(device=device)
    if kind == "shm":
        name = "kicp_soak_%s_%d" % (os.environ.get("MASTER_PORT", "0"), cycle)
        if rank == 0:
            reg.shm_init(world, 0, name)
        dist.barrier()
        if rank != 0:
            reg.shm_init(world, rank, name)
        dist.barrier()
    else:
        mine = torch.frombuffer(bytearray(reg.p2p_export(world, rank)), dtype=torch.uint8)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        reg.p2p_connect([bytes(t.numpy().tobytes()) for t in every])
        dist.barrier()
    return reg


def detach(reg, kind):
    dist.barrier()
    reg.shm_destroy() if kind == "shm" else reg.p2p_destroy()


first, failures, t0 = {}, [], time.time()
for cycle in range(args.cycles):
    kind = "shm" if cycle % 2 == 0 else "p2p"
    try:
        pause()
        reg = attach(kind, cycle)
        pause()
        t_settle = time.perf_counter()
        while True:  # bench.py's settling loop: the ranks agree on the count
            for i in range(10):
                reg.ComputeRobotMotion(frames[i % 4], gmap, scans[i % 4]["last_pose"], rels[i % 4], tau)
            go = torch.tensor([1.0 if time.perf_counter() - t_settle < args.settle_s else 0.0], dtype=torch.float64)
            dist.all_reduce(go, op=dist.ReduceOp.MIN)
            if float(go.item()) <= 0.0:
                break
        batch = reg.prepare_batch([frames[i % 4] for i in range(8)], [scans[i % 4]["last_pose"] for i in range(8)], [rels[i % 4] for i in range(8)])
        dist.barrier()
        for _ in range(3):
            poses = reg.ComputeRobotMotionBatch(batch, gmap, tau).copy()
            pause()
        dist.barrier()
        mine = torch.from_numpy(poses.copy())
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        if any(not torch.equal(every[0], e) for e in every):
            failures.append((cycle, kind, "poses differ between ranks"))
        if kind not in first:
            first[kind] = poses
        elif not np.array_equal(first[kind], poses):
            failures.append((cycle, kind, "poses differ from the first cycle's"))
        detach(reg, kind)
        del reg
    except K.KicpError as e:
        failures.append((cycle, kind, str(e)[:200]))
        break
dist.barrier()
dist.destroy_process_group()
if rank == 0:
    print(json.dumps({"ranks": world, "cycles_requested": args.cycles, "cycles_run": cycle + 1, "failures": failures, "seconds": round(time.time() - t0, 1),
                      "workload": args.workload, "points_per_rank": hi - lo}))
