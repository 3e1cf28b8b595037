cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r7
timeout 2400 python -m pytest tests -q -m gpu > gpurun_out/r7/pytest_gpu.txt 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/r7/pytest_gpu.txt
timeout 1500 python bench.py > gpurun_out/r7/bench_7b.json 2> gpurun_out/r7/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r7/bench_7b.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['config']['workload'][:60])
r=d['roofline']; print(r['frac'], r['frac_full_fine_tune_mix'], r['families'])
for leg in ('forward','decode','generate','decode_fp8','train_lora','train'):
    x=d[leg]; print(leg, {k:v for k,v in x.items() if k in ('samples_s','ms_per_step','tok_s','tok_s_end_to_end','mfma_frac','hbm_frac')})
print(d['m13b'])
PY
