"""profiles/r02_counters_<workload>.json from a set of rocprofv3 --pmc passes (one rocpd database each) of
tools/prof_target.py: per-dispatch means of every counter collected for the pass kernel, plus the derived figures
bench.py reports under roofline.counters.

usage: python tools/prof_counters_json.py <out.json> <kernel substring> <pass-kernel avg us> <db> [<db> ...]

Derived (each only when its inputs were collected):
  hbm_bytes_per_launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024     [KiB counters; gfx950 FETCH_SIZE x2 correction, MI355X_MICROARCH.md]
  l2_read_bytes        = TCP_TCC_READ_REQ_sum * 64                 [requests from the L1s to L2, tallied at 64 B like FETCH_SIZE]
  l2_hit_rate          = TCC_HIT_sum / (TCC_HIT_sum + TCC_MISS_sum)
  l1_accesses          = TCP_TOTAL_CACHE_ACCESSES_sum              [wave-level cache-line accesses]
  valu_busy / issue split from SQ_ACTIVE_INST_VALU, SQ_WAIT_INST_ANY, SQ_WAIT_ANY, SQ_ACTIVE_INST_ANY over SQ_WAVE_CYCLES
"""
import json
import sqlite3
import sys

import numpy as np

out_path, sub, avg_us = sys.argv[1], sys.argv[2], float(sys.argv[3])
vals = {}
for db in sys.argv[4:]:
    try:
        c = sqlite3.connect(db)
        cols = [r[1] for r in c.execute("pragma table_info('counters_collection')")]
        name_col = "kernel_name" if "kernel_name" in cols else "name"
        rows = c.execute("select counter_name, value from counters_collection where %s like ?" % name_col, ("%" + sub + "%",)).fetchall()
    except sqlite3.Error as e:
        print("skip", db, e, file=sys.stderr)
        continue
    acc = {}
    for cn, v in rows:
        acc.setdefault(cn, []).append(float(v))
    for cn, v in acc.items():
        vals[cn] = {"mean_per_dispatch": float(np.mean(v)), "dispatches": len(v)}

def m(name):
    return vals[name]["mean_per_dispatch"] if name in vals else None

import os
import subprocess
try:
    sha = subprocess.check_output(["git", "-C", os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "rev-parse", "--short", "HEAD"], text=True).strip()
except Exception:  # noqa: BLE001  (the GPU box has no .git: the caller passes KICP_GIT_SHA)
    sha = os.environ.get("KICP_GIT_SHA", "unknown")
out = {"kernel": sub, "pass_kernel_avg_us": avg_us, "git_sha": sha, "raw": vals}
if m("FETCH_SIZE") is not None and m("WRITE_SIZE") is not None:
    out["hbm_bytes_per_launch"] = int((2.0 * m("FETCH_SIZE") + m("WRITE_SIZE")) * 1024)
    out["hbm_GBps"] = round(out["hbm_bytes_per_launch"] / (avg_us * 1e-6) / 1e9, 1)
    out["hbm_frac_of_8TBps"] = round(out["hbm_GBps"] / 8000.0, 4)
if m("TCP_TCC_READ_REQ_sum") is not None:
    out["l2_read_bytes"] = int(m("TCP_TCC_READ_REQ_sum") * 64)
    out["l2_read_GBps"] = round(out["l2_read_bytes"] / (avg_us * 1e-6) / 1e9, 1)
    out["l2_frac_of_34.5TBps"] = round(out["l2_read_GBps"] / 34500.0, 4)
if m("TCC_HIT_sum") is not None and m("TCC_MISS_sum") is not None and m("TCC_HIT_sum") + m("TCC_MISS_sum") > 0:
    out["l2_hit_rate"] = round(m("TCC_HIT_sum") / (m("TCC_HIT_sum") + m("TCC_MISS_sum")), 4)
if m("TCP_TOTAL_CACHE_ACCESSES_sum") is not None:
    out["l1_accesses"] = int(m("TCP_TOTAL_CACHE_ACCESSES_sum"))
    if m("TCP_TCC_READ_REQ_sum") is not None and m("TCP_TOTAL_CACHE_ACCESSES_sum") > 0:
        out["l1_hit_rate_est"] = round(1.0 - m("TCP_TCC_READ_REQ_sum") / m("TCP_TOTAL_CACHE_ACCESSES_sum"), 4)
if m("SQ_WAVE_CYCLES"):
    wc = m("SQ_WAVE_CYCLES")
    for k, n in (("valu_active_frac_of_wave_cycles", "SQ_ACTIVE_INST_VALU"), ("issue_stall_frac", "SQ_WAIT_INST_ANY"), ("parked_frac", "SQ_WAIT_ANY"),
                 ("any_inst_active_frac", "SQ_ACTIVE_INST_ANY")):
        if m(n) is not None:
            out[k] = round(m(n) / wc, 4)
for k in ("VALUBusy", "MeanOccupancyPerCU", "OccupancyPercent", "MemUnitBusy", "MemUnitStalled"):
    if m(k) is not None:
        out[k] = round(m(k), 3)
out["method"] = ("rocprofv3 --pmc <counters> in separate passes over `python tools/prof_target.py --workload W --calls 300`; means per dispatch of the "
                 "pass kernel; pass_kernel_avg_us from the --kernel-trace pass of the same command")
json.dump(out, open(out_path, "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k != "raw"}))
