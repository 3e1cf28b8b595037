cd ${GRAFT_REPO_ROOT:-/root/repo}
B="python bench.py --legs lora --no-roofline --no-cpu-baseline --steps 8 --warmup 3"
for cfg in "A3V_STRIP_WGRAD=1" "A3V_STRIP_WGRAD=0" "A3V_STRIP_WGRAD=1"; do
  env $cfg $B 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg', d['ms_per_step'], d['train_lora']['loss'])"
done
