cd $GRAFT_REPO_ROOT
timeout -k 10 300 python tools/facade_overhead.py 2>&1 | grep "^n="
