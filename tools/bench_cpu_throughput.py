"""cpu_baseline.throughput of bench.py: the reference's own Registration.cpp (oracle/_ref) registering T
INDEPENDENT scans at a time, one thread each - the CPU's twin of the GPU's scans-in-flight mode - next to the one-call-at-a-time
figures bench.py takes itself.  Runs in a process of its own (no OpenMP pinning inherited from the caller: every Python thread is
the master of its own one-thread team and must be free to sit on a core of its own).

    python tools/bench_cpu_throughput.py <npz with map, frames, last, rel, tau, voxel, max_range, cap> <threads> <seconds>
prints one line of JSON."""
import json
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import rkicp  # noqa: E402

d = np.load(sys.argv[1])
threads, budget = int(sys.argv[2]), float(sys.argv[3])
if not rkicp.available():
    print(json.dumps({"note": "oracle/_ref not present"}))
    raise SystemExit(0)
rmap = rkicp.VoxelHashMap(float(d["voxel"]), float(d["max_range"]), int(d["cap"]))
rmap.AddPoints(d["map"])
frames, last, rel, tau = d["frames"], d["last"], d["rel"], float(d["tau"])
regs = [rkicp.KinematicRegistration(max_num_threads=1) for _ in range(threads)]
done = [0] * threads
inside = [0.0] * threads
go = threading.Event()
stop_at = [0.0]


def work(t):
    go.wait()
    i = t
    while time.perf_counter() < stop_at[0]:
        k = i % len(frames)
        _, sec = regs[t].timed(frames[k], rmap, last[k], rel[k], tau, 1)  # (ctypes releases the GIL for the call)
        done[t] += 1
        inside[t] += sec
        i += threads


ths = [threading.Thread(target=work, args=(t,)) for t in range(threads)]
for th in ths:
    th.start()
t0 = time.perf_counter()
stop_at[0] = t0 + budget
go.set()
for th in ths:
    th.join()
wall = time.perf_counter() - t0
print(json.dumps({"threads": threads, "scans": int(sum(done)), "wall_s": round(wall, 3), "scans_per_s": round(sum(done) / wall, 3),
                  "mean_call_ms": round(1e3 * sum(inside) / max(1, sum(done)), 2)}))
