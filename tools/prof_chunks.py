"""Per-kernel medians over consecutive chunks of calls (rocprofv3 rocpd db)."""
import sqlite3, sys
from collections import defaultdict
import numpy as np
c = sqlite3.connect(sys.argv[1])
chunk = int(sys.argv[2]) if len(sys.argv) > 2 else 12
dd = defaultdict(list)
for name, g, d in c.execute("select name, grid_x, duration from kernels order by start"):
    dd[(name.split("(")[0].replace("kicp::", "").replace("void ", ""), g)].append(d / 1000)
for k, v in dd.items():
    v = np.array(v)
    print(k, len(v), "med %.1f" % np.median(v), [round(float(np.median(v[i:i + chunk])), 1) for i in range(0, len(v), chunk)])
