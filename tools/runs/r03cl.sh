cd $GRAFT_REPO_ROOT
timeout -k 10 300 python tools/sweep_soak.py 20000 120 2>&1 | tail -4
timeout -k 10 300 python tools/sweep_soak.py 40000 40 2>&1 | tail -4
timeout -k 10 300 python tools/sweep_soak.py 3000 400 2>&1 | tail -4
