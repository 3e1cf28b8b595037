#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd sqlite database (``--kernel-trace --stats`` run) into the per-kernel table that is committed under
profiles/.

    rocprof_summary.py results.db                         every dispatch of the process
    rocprof_summary.py results.db --step-marker embed_assemble --skip 2
        the STEP ALONE: dispatches are cut at every launch of the marker kernel (one per training / forward step: the embedding
        gather that opens it); the first ``--skip`` steps (warm-up) and everything before the first / after the last marker (model
        set-up, the tail after the last step opens) are dropped; the table is per step (totals divided by the number of whole
        steps kept), so `frac` figures reproduce from it without subtracting set-up or warm-up work."""
import argparse
import sqlite3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--step-marker", default=None, help="substring of the kernel that opens a step")
    ap.add_argument("--skip", type=int, default=0, help="leading steps to drop (warm-up)")
    a = ap.parse_args()
    c = sqlite3.connect(a.db)
    print(f"# rocprofv3 --kernel-trace --stats : {a.db}")
    if a.step_marker is None:
        rows = c.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                         "from kernels group by name order by 3 desc").fetchall()
        tot = sum(r[2] for r in rows)
        print(f"# total kernel time {tot / 1e6:.2f} ms over {sum(r[1] for r in rows)} dispatches")
        print(f"{'kernel':100s} {'calls':>7s} {'total_ms':>10s} {'pct':>6s} {'avg_us':>10s} {'min_us':>9s} {'max_us':>9s}")
        for name, n, t, avg, mn, mx in rows:
            print(f"{name[:100]:100s} {n:7d} {t / 1e6:10.3f} {100 * t / tot:6.2f} {avg / 1e3:10.2f} {mn / 1e3:9.2f} {mx / 1e3:9.2f}")
        return
    ev = c.execute("select name, start, end from kernels order by start").fetchall()
    marks = [i for i, (n, _, _) in enumerate(ev) if a.step_marker in n]
    if len(marks) < a.skip + 2:
        raise SystemExit(f"only {len(marks)} launches of a kernel matching {a.step_marker!r}: need skip + 2")
    lo, hi = marks[a.skip], marks[-1]
    nsteps = len(marks) - 1 - a.skip
    agg = {}
    for n, s, e in ev[lo:hi]:
        d = agg.setdefault(n, [0, 0, 1 << 62, 0])
        d[0] += 1
        d[1] += e - s
        d[2] = min(d[2], e - s)
        d[3] = max(d[3], e - s)
    tot = sum(d[1] for d in agg.values())
    wall = ev[hi][1] - ev[lo][1]
    print(f"# STEP ALONE: {nsteps} whole steps between launches of '{a.step_marker}' (first {a.skip} dropped); per step: "
          f"{tot / nsteps / 1e6:.2f} ms of kernel time in {sum(d[0] for d in agg.values()) / nsteps:.0f} dispatches, {wall / nsteps / 1e6:.2f} ms wall")
    print(f"{'kernel':100s} {'calls/step':>10s} {'ms/step':>10s} {'pct':>6s} {'avg_us':>10s} {'min_us':>9s} {'max_us':>9s}")
    for name, d in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{name[:100]:100s} {d[0] / nsteps:10.1f} {d[1] / nsteps / 1e6:10.3f} {100 * d[1] / tot:6.2f} {d[1] / d[0] / 1e3:10.2f} "
              f"{d[2] / 1e3:9.2f} {d[3] / 1e3:9.2f}")


if __name__ == "__main__":
    main()
