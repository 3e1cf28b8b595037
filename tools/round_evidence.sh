#!/bin/bash
# Evidence of a round at the current commit (repo root on the GPU box): the full GPU suite, the bench line, and rocprofv3
# --kernel-trace --stats tables OF THE STEP ALONE (tools/rocprof_summary.py --step-marker: dispatches between launches of the kernel that
# opens a step, warm-up steps dropped, per-step averages) for the LoRA step (headline), the full fine-tune step, the inference forward
# and the greedy decode step.   usage: tools/round_evidence.sh r03d
tag=${1:-r03x}
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/$tag
mkdir -p $out
cd $root
timeout 2400 python -m pytest tests -q -m gpu > $out/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; grep -n "passed\|failed" $out/pytest_gpu.txt | tail -2
timeout 1500 python bench.py > $out/bench_7b.json 2> $out/bench.err; echo "bench rc=$?"
cd /tmp && export TMPDIR=/tmp
prof() {   # name, marker, skip, bench args...
  name=$1; marker=$2; skip=$3; shift 3
  rm -rf /tmp/pe_$name
  timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/pe_$name -o t -- python $root/bench.py "$@" --no-roofline --no-cpu-baseline 2> $out/prof_$name.err | tail -1 > $out/bench_${name}_profiled.json
  python $root/tools/rocprof_summary.py /tmp/pe_$name/t_results.db --step-marker $marker --skip $skip > $out/kernel_stats_$name.txt
  head -4 $out/kernel_stats_$name.txt | cut -c1-200
}
prof lora_step embed_assemble 3 --legs lora --steps 6 --warmup 2
prof train_step embed_assemble 3 --legs train --steps 6 --warmup 2
prof forward_step embed_assemble 3 --legs forward --steps 8 --warmup 2
prof decode_step rows_ssq 8 --legs decode --decode-steps 32
cd $root
python - <<PY
import json
d=json.loads(open('$out/bench_7b.json').read().strip().splitlines()[-1])
print('headline', d['value'], d['ms_per_step'], 'frac', d['roofline']['frac'], d['roofline']['frac_full_fine_tune_mix'])
for leg in ('forward','decode','generate','decode_fp8','train_lora','train'):
    x=d[leg]; print(leg, {k:v for k,v in x.items() if k in ('samples_s','ms_per_step','tok_s','tok_s_end_to_end','mfma_frac','hbm_frac','hbm_gib')})
print('m13b', d['m13b'].get('train_replica'))
PY
