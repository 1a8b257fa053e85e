"""Summarise the roctx ranges of a rocprofv3 --marker-trace run (rocpd database): per range message -> calls, avg / median us.
The ranges are what KICP_ROCTX=1 makes the library open around its traced C-ABI calls and around every ICP pass
(kicp_internal.hpp).  rocprofv3 files a range under the name of the API call (roctxThreadRangeA, category MARKER_CORE_RANGE_API)
and keeps the message in the region's extdata ({"message": ...}; older schemas: an argument of the event); all places are looked at."""
import sqlite3
import sys

import numpy as np


def main(path):
    c = sqlite3.connect(path)
    tables = [r[0] for r in c.execute("select name from sqlite_master where type in ('table', 'view')")]
    rows = []
    if "regions" in tables:  # rocprofiler-sdk 1.x: the message travels in the region's extdata, {"message": "..."}
        import json
        try:
            for name, ext, a, b in c.execute("select name, extdata, start, end from regions where category like 'MARKER%'"):
                try:
                    msg = json.loads(ext or "{}").get("message")
                except ValueError:
                    msg = None
                rows.append((msg or name, a, b))
        except sqlite3.Error:
            rows = []
    if not rows and "regions" in tables:
        args = [t for t in tables if t.startswith("rocpd_arg")]
        for a in sorted(args, key=len):  # (the view without the guid suffix first)
            try:
                rows = c.execute("select a.value, r.start, r.end from regions r join %s a on a.event_id = r.event_id "
                                 "where r.category like 'MARKER%%' and a.name in ('message', 'msg', 'name')" % a).fetchall()
            except sqlite3.Error:
                rows = []
            if rows:
                break
        if not rows:
            rows = c.execute("select name, start, end from regions where category like 'MARKER%'").fetchall()
    groups = {}
    for name, a, b in rows:
        if name is None or a is None or b is None or b < a:
            continue
        groups.setdefault(str(name).strip('"'), []).append((b - a) / 1000.0)
    if not groups:
        print("no roctx ranges in this database; tables: " + ", ".join(tables))
        return
    print("%-52s %7s %10s %10s" % ("roctx range", "calls", "avg_us", "med_us"))
    for k, v in sorted(groups.items(), key=lambda kv: -sum(kv[1])):
        print("%-52s %7d %10.2f %10.2f" % (k[:52], len(v), float(np.mean(v)), float(np.median(v))))


if __name__ == "__main__":
    main(sys.argv[1])
