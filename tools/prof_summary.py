"""Summarise a rocprofv3 rocpd database (kernel-trace) as a text table: per kernel name/grid -> calls, avg/min/max us."""
import sqlite3
import sys

import numpy as np


def main(path, min_us=0.0):
    c = sqlite3.connect(path)
    rows = c.execute("select name, grid_x, workgroup_x, duration from kernels order by start").fetchall()
    groups = {}
    for name, g, w, d in rows:
        groups.setdefault((name, g, w), []).append(d / 1000.0)
    print("%-70s %9s %5s %6s %9s %9s %9s %9s" % ("kernel", "grid", "wg", "calls", "avg_us", "med_us", "min_us", "max_us"))
    for (name, g, w), d in sorted(groups.items(), key=lambda kv: -sum(kv[1])):
        d = np.array(d)
        if d.mean() < min_us:
            continue
        print("%-70s %9d %5d %6d %9.2f %9.2f %9.2f %9.2f" % (name[:70], g, w, len(d), d.mean(), np.median(d), d.min(), d.max()))


if __name__ == "__main__":
    main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 0.0)
