"""Summarise the HIP API regions of a rocprofv3 rocpd database (--hip-trace): per call name -> calls, total/avg/max ms."""
import sqlite3
import sys


def main(path, skip_ms=0.0):
    c = sqlite3.connect(path)
    cols = [r[1] for r in c.execute("pragma table_info(regions)").fetchall()]
    rows = c.execute("select name, start, end from regions order by start").fetchall()
    if not rows:
        print("no regions; columns:", cols)
        return
    t0 = rows[0][1]
    agg = {}
    for name, s, e in rows:
        if (s - t0) / 1e6 < skip_ms:
            continue
        a = agg.setdefault(name, [0, 0.0, 0.0])
        a[0] += 1
        a[1] += (e - s) / 1e6
        a[2] = max(a[2], (e - s) / 1e6)
    print("%-40s %7s %10s %9s %9s" % ("api", "calls", "total_ms", "avg_ms", "max_ms"))
    for name, (n, tot, mx) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:25]:
        print("%-40s %7d %10.3f %9.4f %9.3f" % (name[:40], n, tot, tot / n, mx))


if __name__ == "__main__":
    main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 0.0)
