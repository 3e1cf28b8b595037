#!/bin/bash
# PMC passes (counters only, with --kernel-trace) for one adapter-sized NT product; summary into gpurun_out/pmc_skinny_<tag>/summary.txt
# usage: tools/pmc_skinny.sh tag M N K S
set -u
tag=$1; M=$2; N=$3; K=$4; S=$5
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/pmc_skinny_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for ctrs in "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $out/p$i -o p$i -- python $root/tools/skinny_one.py $M $N $K $S 4 > $out/p$i.log 2>&1
done
python - <<PY | tee $out/summary.txt
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$out/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "skinny" in r["Kernel_Name"] or "splitk_reduce" in r["Kernel_Name"]:
            acc[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
M, N, K, S = $M, $N, $K, $S
print(f"a3v_gemm_nt_splitk {M} x {N} x {K}, S = {S}: algorithmic bytes of the product = streamed operand {2*M*K/1e6:.1f} MB + second operand {2*N*K/1e6:.1f} MB; planes {4*S*M*N/1e6:.1f} MB written, read once by the reduce")
for k, d in acc.items():
    print(k)
    m = {c: sum(v) / len(v) for c, v in d.items()}
    for c, v in sorted(m.items()):
        print(f"   {c:28s} {v:.5g}")
    if "FETCH_SIZE" in m:
        print(f"   -> HBM read {2 * m['FETCH_SIZE'] * 1024 / 1e6:.1f} MB per launch (FETCH_SIZE KB x 2: the gfx950 correction for 16-B/lane streaming reads, MI355X_MICROARCH.md), write {m.get('WRITE_SIZE', 0) * 1024 / 1e6:.1f} MB")
    if "TCC_HIT_sum" in m:
        print(f"   -> L2 hit rate {m['TCC_HIT_sum'] / max(1.0, m['TCC_HIT_sum'] + m['TCC_MISS_sum']):.3f}")
PY
