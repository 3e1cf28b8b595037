cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r12
timeout 1500 python bench.py > gpurun_out/r12/bench_7b.json 2> gpurun_out/r12/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r12/bench_7b.json').read().strip().splitlines()[-1])
print('headline', d['value'], d['ms_per_step'])
r=d['roofline']; print(r['frac'], r['frac_full_fine_tune_mix'], r['families_of_the_headline_step'], r['families'])
for s in r['shapes']:
    if s['kind'] in ('nt_dgrad','nn'): print(s)
for leg in ('forward','decode','train_lora','train'):
    x=d[leg]; print(leg, {k:v for k,v in x.items() if k in ('samples_s','ms_per_step','tok_s','mfma_frac','hbm_gib')})
PY
