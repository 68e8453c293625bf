cd $GRAFT_REPO_ROOT
timeout -k 10 300 python tools/macro_tile_probe.py 2>&1 | tail -6
