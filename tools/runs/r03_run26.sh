cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -k "runs_on_across or ragged_rows or fast_epilogue" 2>&1 | tail -3
timeout 900 python bench.py --legs lora,forward --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('lora', d['train_lora']['ms_per_step'], 'forward', d['forward']['ms_per_step'], 'frac', d['roofline']['frac'], {k: v['tflops'] for k, v in d['roofline']['families'].items()})"
