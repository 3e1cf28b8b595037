#!/usr/bin/env python3
"""Idle time between consecutive kernels of the greedy-decode steps in a rocprofv3 kernel trace (rocpd sqlite):
usage: decode_gaps.py results.db   -- prints, for the decode region, kernel time, gap time and the gap histogram."""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name, start, end from kernels order by start").fetchall()
# decode region = from the first attn_decode kernel to the last one
idx = [i for i, r in enumerate(rows) if "attn_decode" in r[0]]
lo, hi = idx[0], idx[-1]
seg = rows[lo:hi + 1]
busy = sum(e - s for _, s, e in seg)
gaps = [max(0, seg[i + 1][1] - seg[i][2]) for i in range(len(seg) - 1)]
span = seg[-1][2] - seg[0][1]
big = [g for g in gaps if g > 20000]
print(f"decode region: {len(seg)} kernels, span {span / 1e6:.2f} ms, kernel time {busy / 1e6:.2f} ms ({100 * busy / span:.1f} %), "
      f"gaps {sum(gaps) / 1e6:.2f} ms; median gap {sorted(gaps)[len(gaps) // 2] / 1e3:.2f} us; gaps > 20 us: {len(big)} totalling {sum(big) / 1e6:.2f} ms")
