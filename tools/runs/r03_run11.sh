cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 300 python tools/strip_nn_vs_tn.py 2>&1 | tail -6
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -2
