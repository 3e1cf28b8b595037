#!/bin/bash
# PMC passes (counters only, with --kernel-trace) for one GEMM config; CSV into gpurun_out/pmc_<tag>/
# usage: tools/pmc_gemm.sh tag M N K cfg
set -u
tag=$1; M=$2; N=$3; K=$4; cfg=$5
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/pmc_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for ctrs in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES" \
            "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_LDS SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE" \
            "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $out/p$i -o p$i -- python $root/tools/gemm_one.py $M $N $K $cfg 3 > $out/p$i.log 2>&1
done
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$out/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gemm" in r["Kernel_Name"]:
            acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
import json, os
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:34s} n={len(v)} mean={sum(v)/len(v):.4g}")
    m = {c: sum(v) / len(v) for c, v in d.items()}
    if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
        M, N, K = $M, $N, $K
        alg = 2 * (M * K + N * K + M * N)
        hit = m.get("TCC_HIT_sum", 0) / max(1.0, m.get("TCC_HIT_sum", 0) + m.get("TCC_MISS_sum", 0))
        out = {"kernel": k.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0].strip(), "shape": [M, N, K], "fetch_size_kb": m["FETCH_SIZE"], "write_size_kb": m["WRITE_SIZE"],
               "correction": "FETCH_SIZE x2 on gfx950 for 16-B/lane streaming reads (MI355X_MICROARCH.md HBM section); WRITE_SIZE as reported",
               "traffic_bytes_per_launch": int((2 * m["FETCH_SIZE"] + m["WRITE_SIZE"]) * 1024), "algorithmic_bytes_per_launch": alg,
               "l2_hit_rate": round(hit, 4), "mfma_busy_frac_of_gui_active": round(m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / 4 / 256 / max(1.0, m.get("GRBM_GUI_ACTIVE", 1) / 8), 4) if "GRBM_GUI_ACTIVE" in m else None,
               "source": "tools/pmc_gemm.sh $tag $M $N $K $cfg (rocprofv3 --pmc, separate passes with --kernel-trace only, 3 launches averaged)"}
        open(os.path.join("$out", "traffic.json"), "w").write(json.dumps(out, indent=1))
PY
