"""Turns a rocprofv3 (ROCm 7.2, rocpd sqlite output) kernel trace into the per-kernel stats table that
`rocprofv3 --kernel-trace --stats` describes: calls, total / average / min / max duration, share.
usage: python tools/rocpd_summary.py <results.db> [out.md]"""
import sqlite3
import sys


def summarize(db):
    con = sqlite3.connect(db)
    rows = con.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by name order by 3 desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    lines = ["| kernel | calls | total ms | avg us | min us | max us | % |", "|---|---|---|---|---|---|---|"]
    for name, calls, tot, avg, mn, mx in rows:
        lines.append("| `%s` | %d | %.3f | %.2f | %.2f | %.2f | %.1f |" % (name[:110], calls, tot / 1e6, avg / 1e3, mn / 1e3, mx / 1e3, 100.0 * tot / total))
    return "\n".join(lines)


if __name__ == "__main__":
    out = summarize(sys.argv[1])
    if len(sys.argv) > 2:
        open(sys.argv[2], "a").write(out + "\n")
    print(out)
