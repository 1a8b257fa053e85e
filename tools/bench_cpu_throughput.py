"""cpu_baseline.throughput of bench.py: the reference's own Registration.cpp (oracle/_ref) registering T
INDEPENDENT scans at a time, one thread each - the CPU's twin of the GPU's scans-in-flight mode - next to the one-call-at-a-time
figures bench.py takes itself.  Runs in a process of its own (no OpenMP pinning inherited from the caller: every Python thread is
the master of its own one-thread team and must be free to sit on a core of its own).

    python tools/bench_cpu_throughput.py <npz with map, frames, last, rel, tau, voxel, max_range, cap> <threads> <seconds>
prints one line of JSON."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import rkicp  # noqa: E402

try:  # (a caller whose main thread is bound to one core - OMP_PROC_BIND binds the initial thread of many a process - hands that mask down)
    os.sched_setaffinity(0, set(range(os.cpu_count() or 1)))
except (AttributeError, OSError):
    pass
d = np.load(sys.argv[1])
threads, budget = int(sys.argv[2]), float(sys.argv[3])
if not rkicp.available():
    print(json.dumps({"note": "oracle/_ref not present"}))
    raise SystemExit(0)
rmap = rkicp.VoxelHashMap(float(d["voxel"]), float(d["max_range"]), int(d["cap"]))
rmap.AddPoints(d["map"])
frames, last, rel, tau = d["frames"], d["last"], d["rel"], float(d["tau"])
# the loop itself runs inside the reference build (oracle/ref_capi.cpp::rkicp_register_throughput: std::threads, frames converted once): a
# Python thread per lane spends more time on the interpreter lock than in a 0.25 ms registration of a 1 080-point scan
scans, wall = rkicp.register_throughput(rmap, list(frames), list(last), list(rel), tau, threads, budget)
print(json.dumps({"threads": threads, "scans": int(scans), "wall_s": round(wall, 3), "scans_per_s": round(scans / wall, 3)}))
