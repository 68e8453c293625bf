cd $GRAFT_REPO_ROOT
timeout -k 10 900 python tools/fuzz_sizes.py 26 2>&1 | tail -30
