cd ${GRAFT_REPO_ROOT:-/root/repo}
for st in 2 3; do echo "== stages $st"; A3V_STRIP_STAGES=$st timeout 300 python tools/lora_skinny_bench.py 2>&1 | grep "strip"; done
A3V_STRIP_STAGES=3 timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "tn_strip" 2>&1 | tail -3
