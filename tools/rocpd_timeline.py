"""Prints the dispatch timeline (start/end relative to the first kernel of the last frame) from a rocprofv3 rocpd database.
usage: python tools/rocpd_timeline.py <results.db> [n_last_dispatches]"""
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in con.execute("pragma table_info(kernels)").fetchall()]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 60
qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
sel = "select name, start, end%s from kernels order by start" % ((", " + qcol) if qcol else "")
rows = con.execute(sel).fetchall()[-n:]
t0 = rows[0][1]
prev_end = t0
for r in rows:
    name, a, b = r[0], r[1], r[2]
    q = r[3] if qcol else "-"
    print("%9.1f %9.1f  dur %8.1f  gap %7.1f  q=%s  %s" % ((a - t0) / 1e3, (b - t0) / 1e3, (b - a) / 1e3, (a - prev_end) / 1e3, q, name[:40]))
    prev_end = max(prev_end, b)
