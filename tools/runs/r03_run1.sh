set -x
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r1
timeout 1500 python -m pytest tests/test_gpu_sampling.py tests/test_gpu_13b_shapes.py tests/test_gpu_mid.py tests/test_gpu_two_image.py -x -q -m gpu -s > gpurun_out/r1/new_tests.log 2>&1; echo "new tests rc=$?"
tail -5 gpurun_out/r1/new_tests.log
timeout 600 python tools/hipblaslt_calib.py > gpurun_out/r1/calib.log 2>&1; echo "calib rc=$?"
cat gpurun_out/r1/calib.log
timeout 2400 python -m pytest tests -q -m gpu -x --deselect tests/test_gpu_sampling.py --deselect tests/test_gpu_13b_shapes.py > gpurun_out/r1/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -5 gpurun_out/r1/pytest_gpu.log
timeout 1500 python bench.py > gpurun_out/r1/bench.json 2> gpurun_out/r1/bench.err; echo "bench rc=$?"
tail -c 1500 gpurun_out/r1/bench.json
