cd $GRAFT_REPO_ROOT
O=gpurun_out/r03au; mkdir -p $O
FVGP_LEAF_TILES_ROWS=100000 FVGP_SMALL_THRESHOLD=0 timeout -k 10 200 python tools/chain_stamps.py 20000 > $O/cs.log 2>&1; head -60 $O/cs.log
