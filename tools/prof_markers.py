"""Summarise the roctx ranges of a rocprofv3 --marker-trace run (rocpd database): per range name -> calls, avg / median us.
The ranges are what KICP_ROCTX=1 makes the library open around its traced C-ABI calls and around every ICP pass (kicp_internal.hpp).
The rocpd schema differs between rocprofiler-sdk versions, so the table is looked for by its columns; when nothing fits, the
tables found are listed instead."""
import sqlite3
import sys

import numpy as np


def main(path):
    c = sqlite3.connect(path)
    tables = [r[0] for r in c.execute("select name from sqlite_master where type in ('table', 'view')")]
    for t in sorted(tables, key=lambda n: (0 if "region" in n.lower() else 1 if "marker" in n.lower() else 2, n)):
        cols = [r[1] for r in c.execute("pragma table_info('%s')" % t)]
        low = [x.lower() for x in cols]
        if t.lower() in ("kernels", "top_kernels", "kernel_dispatch") or "kernel" in t.lower():
            continue
        if "start" in low and "end" in low and ("name" in low or "name_id" in low):
            if "name" in low:
                rows = c.execute("select name, start, end from %s" % t).fetchall()
            else:
                st = [x for x in tables if "string" in x.lower()]
                if not st:
                    continue
                rows = c.execute("select s.string, r.start, r.end from %s r join %s s on r.name_id = s.id" % (t, st[0])).fetchall()
            groups = {}
            for name, a, b in rows:
                if name is None or b is None or a is None or b < a:
                    continue
                groups.setdefault(str(name), []).append((b - a) / 1000.0)
            groups = {k: v for k, v in groups.items() if k.startswith("kicp") or k.startswith("icp pass")}
            if not groups:
                continue
            print("roctx ranges (table %s)" % t)
            print("%-48s %7s %10s %10s" % ("range", "calls", "avg_us", "med_us"))
            for k, v in sorted(groups.items(), key=lambda kv: -sum(kv[1])):
                print("%-48s %7d %10.2f %10.2f" % (k[:48], len(v), float(np.mean(v)), float(np.median(v))))
            return
    print("no table with roctx ranges found; tables:")
    for t in tables:
        print("  %s: %s" % (t, ", ".join(r[1] for r in c.execute("pragma table_info('%s')" % t))))


if __name__ == "__main__":
    main(sys.argv[1])
