"""Timeline of ONE frame of the drop-in pipeline from a rocprofv3 kernel trace (rocpd sqlite database): every kernel between the
`k_frame_pre` of frame K and the `k_frame_pre` of frame K + 1 with its start offset, duration and queue - which kernels overlap, where
the device idles between dependent launches, what the frame's critical path is.

    python tools/pipeline_timeline.py <trace.db> [--frame 30]
"""
import argparse
import sqlite3

ap = argparse.ArgumentParser()
ap.add_argument("db")
ap.add_argument("--frame", type=int, default=30)
a = ap.parse_args()
c = sqlite3.connect(a.db)
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
q = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else "0")
rows = c.execute("select name, start, end, %s from kernels order by start" % q).fetchall()
marks = [i for i, r in enumerate(rows) if "k_frame_pre" in r[0]]
if len(marks) <= a.frame + 1:
    raise SystemExit("only %d frames in the trace" % len(marks))
i0, i1 = marks[a.frame], marks[a.frame + 1]
# (kernels of the previous frame's map update may still be running: include what ends after this frame's start)
t0 = rows[i0][1]
shown = [r for r in rows[max(0, i0 - 12):i1] if r[2] >= t0 - 20000]
print("# frame %d of %s: %.1f us from its k_frame_pre to the next frame's" % (a.frame, a.db, (rows[i1][1] - t0) / 1e3))
print("%10s %9s %6s  %s" % ("start us", "dur us", "queue", "kernel"))
busy_until = None
for name, s, e, qid in shown:
    short = name.replace("kicp::", "").replace("host::", "").split("(")[0][:60]
    gap = "" if busy_until is None or s <= busy_until else "   <- device idle %.1f us" % ((s - busy_until) / 1e3)
    print("%10.1f %9.1f %6s  %s%s" % ((s - t0) / 1e3, (e - s) / 1e3, qid, short, gap))
    busy_until = e if busy_until is None else max(busy_until, e)
