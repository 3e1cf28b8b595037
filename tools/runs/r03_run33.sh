cd ${GRAFT_REPO_ROOT:-/root/repo}
bash tools/pmc_gemm.sh r03f_ring_w13 8728 22016 4096 ring > gpurun_out/pmc_r03f_ring_w13.txt 2>&1
bash tools/pmc_gemm.sh r03f_ring_dgrad_qkv 8728 4096 12352 ring > gpurun_out/pmc_r03f_ring_dgrad_qkv.txt 2>&1
tail -30 gpurun_out/pmc_r03f_ring_w13.txt
cat gpurun_out/pmc_r03f_ring_w13/traffic.json gpurun_out/pmc_r03f_ring_dgrad_qkv/traffic.json
