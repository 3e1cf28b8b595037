#!/bin/bash
# PMC passes (counters only, with --kernel-trace) over tools/attn_bench.py: MFMA busy / wait cycles, LDS conflicts and L2 traffic of the
# prefill and backward attention kernels.  usage (repo root on the GPU box): tools/pmc_attn.sh <tag>   -> gpurun_out/pmc_attn_<tag>.txt
set -u
tag=${1:-x}
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/pmc_attn_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for ctrs in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES" \
            "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" \
            "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $out/p$i -o p$i -- python $root/tools/attn_bench.py > $out/p$i.log 2>&1
done
python - <<PY > $root/gpurun_out/pmc_attn_$tag.txt
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$out/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "attn_" in k:
            acc[k[:90]][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("# rocprofv3 --kernel-trace --pmc <group> (one group per pass) -- python tools/attn_bench.py ; per-dispatch means")
print("# shapes: prefill B=8 S=1091 H=32 hd=128 causal | ViT B=8 S=577 H=16 hd=64 | prefill S=2048 | backward B=8 S=1091 H=32 hd=128")
for k, d in sorted(acc.items()):
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:34s} n={len(v):3d} mean={sum(v)/len(v):.5g}")
    if "SQ_VALU_MFMA_BUSY_CYCLES" in d and "SQ_BUSY_CYCLES" in d:
        mb = sum(d["SQ_VALU_MFMA_BUSY_CYCLES"]) / len(d["SQ_VALU_MFMA_BUSY_CYCLES"])
        bc = sum(d["SQ_BUSY_CYCLES"]) / len(d["SQ_BUSY_CYCLES"])
        print(f"   -> MFMA busy / SQ busy cycles = {mb / bc:.3f}")
PY
tail -5 $root/gpurun_out/pmc_attn_$tag.txt
