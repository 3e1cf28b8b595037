for v in 1 0 1 0; do
  echo "== OVERLAP=$v"
  A3V_ADAMW_OVERLAP=$v python bench.py --legs lora,train 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('train', d['train']['ms_per_step'], 'lora', d['train_lora']['ms_per_step'])"
done
