"""Per-kernel PMC counter totals from a rocprofv3 rocpd database (one --pmc pass)."""
import sqlite3
import sys
from collections import defaultdict

import numpy as np

c = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
view = "counters_collection" if "counters_collection" in tabs else None
if view is None:
    raise SystemExit("no counters_collection view in " + sys.argv[1])
cols = [r[1] for r in c.execute("pragma table_info('%s')" % view)]
name_col = "kernel_name" if "kernel_name" in cols else "name"
rows = c.execute("select %s, counter_name, value from %s" % (name_col, view)).fetchall()
acc = defaultdict(list)
for k, cn, v in rows:
    acc[(k.split("(")[0].replace("kicp::", "").replace("void ", ""), cn)].append(float(v))
print("%-40s %-22s %8s %16s %16s" % ("kernel", "counter", "samples", "mean/dispatch", "max/dispatch"))
for (k, cn), v in sorted(acc.items()):
    v = np.array(v)
    print("%-40s %-22s %8d %16.1f %16.1f" % (k[:40], cn, len(v), v.mean(), v.max()))
