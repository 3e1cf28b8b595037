set -x
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r3
timeout 2400 python -m pytest tests -q -m gpu > gpurun_out/r3/tests.log 2>&1; echo "tests rc=$?"
tail -25 gpurun_out/r3/tests.log
