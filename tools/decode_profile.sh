#!/bin/bash
# per-kernel (and per-GEMV-grid) times of the decode loop: rocprofv3 kernel trace of a short bench run
root=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_dec
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_dec -o dec -- python $root/bench.py --no-cpu-baseline --no-train --steps 2 --warmup 1 > /dev/null 2>&1
cd $root
python tools/rocprof_summary.py /tmp/prof_dec/dec_results.db | grep -i "gemv\|skinny\|attn_decode\|rmsnorm\|rope_kv_small\|combine\|argmax"
python - <<PY
import sqlite3, collections
c = sqlite3.connect("/tmp/prof_dec/dec_results.db")
rows = c.execute("select name, grid_x, (end-start) from kernels where name like '%gemv%' order by start").fetchall()
d = collections.defaultdict(list)
for n, g, t in rows: d[g].append(t / 1e3)
for g, v in sorted(d.items()): print("gemv grid", g, "blocks", g // 256, "n", len(v), "avg us %.1f min %.1f" % (sum(v) / len(v), min(v)))
PY
