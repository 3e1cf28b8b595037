cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -k "swiglu" 2>&1 | tail -3
timeout 1500 python -m pytest tests/test_gpu_train.py tests/test_gpu_trainer.py tests/test_gpu_dp.py -q -x 2>&1 | tail -3
for v in 1 0 1 0; do A3V_FUSE_SWIGLU_BWD=$v timeout 900 python bench.py --legs train --no-roofline --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('fuse=$v train', d['train']['ms_per_step'], d['train']['loss'])"; done
