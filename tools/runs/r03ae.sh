cd $GRAFT_REPO_ROOT
O=gpurun_out/r03ae; mkdir -p $O
timeout -k 10 200 python tools/chain_stamps.py 20000 > $O/cs.log 2>&1; cat $O/cs.log
