cd $GRAFT_REPO_ROOT
timeout -k 10 200 python tools/leaf_phases.py 2>&1 | tail -40
