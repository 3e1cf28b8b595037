cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r6
timeout 600 python tools/decode_graph_probe.py 2>&1 | tail -3 | tee gpurun_out/r6/decode_graph_probe.txt
B="python bench.py --legs lora --no-roofline --no-cpu-baseline --steps 8 --warmup 3"
for cfg in "A3V_SKINNY_NARROW=1" "A3V_SKINNY_NARROW=3"; do
  env $cfg $B 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg', d['ms_per_step'])"
done | tee gpurun_out/r6/lora_narrow.txt
