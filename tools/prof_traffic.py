"""profiles/pmc_traffic.json from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of bench.py.

usage: python tools/prof_traffic.py <fetch.db> <write.db> <kernel substring> <out.json>
HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: both counters are reported in KiB per dispatch and on
gfx950 FETCH_SIZE reads half the bytes of a wide coalesced stream (MI355X_MICROARCH.md, HBM section); the factor 2
is therefore an upper-bound correction for this gather-dominated kernel.  WRITE_SIZE is uncalibrated."""
import json
import sqlite3
import sys

import numpy as np


def per_dispatch(db, counter, sub):
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info('counters_collection')")]
    name_col = "kernel_name" if "kernel_name" in cols else "name"
    v = [float(x[0]) for x in c.execute("select value from counters_collection where counter_name=? and %s like ?" % name_col,
                                        (counter, "%" + sub + "%"))]
    return np.array(v)


fetch, write = per_dispatch(sys.argv[1], "FETCH_SIZE", sys.argv[3]), per_dispatch(sys.argv[2], "WRITE_SIZE", sys.argv[3])
out = {"kernel": sys.argv[3], "dispatches": int(len(fetch)), "FETCH_SIZE_KiB_mean": float(fetch.mean()), "WRITE_SIZE_KiB_mean": float(write.mean()),
       "hbm_bytes_per_launch": int((2.0 * fetch.mean() + write.mean()) * 1024),
       "method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes over `python bench.py --steps 40 --warmup 10 --no-cpu-baseline`; "
                 "bytes = (2*FETCH_SIZE + WRITE_SIZE) KiB per dispatch (gfx950 FETCH_SIZE x2 correction, MI355X_MICROARCH.md)"}
json.dump(out, open(sys.argv[4], "w"), indent=1)
print(json.dumps(out))
