cd ${GRAFT_REPO_ROOT:-/root/repo}
bash tools/pmc_gemm.sh r03_ring_w13 8728 22016 4096 ring > gpurun_out/r03_pmc_gemm_ring_w13.txt 2>&1
bash tools/pmc_gemm.sh r03_tn_w13 22016 4096 8728 tn > gpurun_out/r03_pmc_gemm_tn_w13.txt 2>&1
bash tools/pmc_gemm.sh r03_ring_w13dgrad 8728 4096 22080 auto > gpurun_out/r03_pmc_gemm_ring_w13_dgrad.txt 2>&1
tail -30 gpurun_out/r03_pmc_gemm_ring_w13.txt
cat gpurun_out/pmc_r03_ring_w13/traffic.json gpurun_out/pmc_r03_tn_w13/traffic.json gpurun_out/pmc_r03_ring_w13dgrad/traffic.json
timeout 300 python -m pytest tests/test_gpu_dp.py -q -m gpu 2>&1 | tail -3
