#!/bin/bash
# timing experiments on the attention backward kernels (see variant_attn.sh); per-kernel times from a rocprofv3 trace
cd ${GRAFT_REPO_ROOT:-/root/repo}
root=$PWD
for v in "$@"; do
  echo "=== variant: '$v'"
  touch a3vlm_amd/csrc/a3v_attn_bwd.hip
  make -C a3vlm_amd/csrc EXTRA="$v" 2>&1 | grep -E " error" | head -3
  (cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pab && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pab -o ab -- python $root/tools/attn_bench.py > /dev/null 2>&1)
  python tools/rocprof_summary.py /tmp/pab/ab_results.db | grep -E "attn_bwd|rowdot|transpose" | cut -c1-60,100-150
done
