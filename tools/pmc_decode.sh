#!/bin/bash
# PMC pass (FETCH_SIZE / WRITE_SIZE, separate runs, --kernel-trace only) over a short greedy decode: per-kernel mean bytes per launch.
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/pmc_decode
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $out/$c
  timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/$c -o p -- python $root/bench.py --no-cpu-baseline --no-train --steps 1 --warmup 1 --decode-steps 8 > $out/$c.log 2>&1
done
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$out/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "gemv" in n or "attn_decode" in n:
            acc[n[:90]][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("# rocprofv3 --kernel-trace --pmc {FETCH_SIZE | WRITE_SIZE} (separate passes): mean per launch, KB as reported (x2 gfx950 correction for 16-B/lane streaming reads NOT applied)")
for k, d in sorted(acc.items()):
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:12s} n={len(v):5d} mean={sum(v)/len(v):12.1f} KB   min={min(v):10.1f} max={max(v):10.1f}")
PY
