#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd sqlite database (``--kernel-trace --stats`` run) into the per-kernel
table that is committed under profiles/.   usage: rocprof_summary.py results.db [> summary.txt]"""
import sqlite3
import sys


def main(path):
    c = sqlite3.connect(path)
    rows = c.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                     "from kernels group by name order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows)
    print(f"# rocprofv3 --kernel-trace --stats : {path}")
    print(f"# total kernel time {tot / 1e6:.2f} ms over {sum(r[1] for r in rows)} dispatches")
    print(f"{'kernel':100s} {'calls':>7s} {'total_ms':>10s} {'pct':>6s} {'avg_us':>10s} {'min_us':>9s} {'max_us':>9s}")
    for name, n, t, avg, mn, mx in rows:
        print(f"{name[:100]:100s} {n:7d} {t / 1e6:10.3f} {100 * t / tot:6.2f} {avg / 1e3:10.2f} {mn / 1e3:9.2f} {mx / 1e3:9.2f}")


if __name__ == "__main__":
    main(sys.argv[1])
