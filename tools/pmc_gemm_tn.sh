#!/bin/bash
# LDS bank-conflict counters of the TN / NN ping-pong GEMMs (transposing LDS reads) on the 7B weight-gradient / input-gradient shapes.
# usage (repo root on the GPU box): tools/pmc_gemm_tn.sh <tag>  -> gpurun_out/pmc_gemm_tn_<tag>.txt
tag=${1:-x}
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/pmc_gemm_tn_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for b in gemm_tn_bench gemm_nn_bench; do
  PYTHONPATH=$root timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS --output-format csv -d $out/$b -o $b -- python $root/tools/$b.py > $out/$b.log 2>&1
done
python - <<PY > $root/gpurun_out/pmc_gemm_tn_$tag.txt
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$out/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "gemm_" in k:
            acc[k[:80]][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("# rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS ; per-dispatch means over tools/gemm_tn_bench.py + gemm_nn_bench.py")
for k, d in sorted(acc.items()):
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:28s} n={len(v):4d} mean={sum(v)/len(v):.5g}")
PY
cat $root/gpurun_out/pmc_gemm_tn_$tag.txt
