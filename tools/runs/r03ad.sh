cd $GRAFT_REPO_ROOT
O=gpurun_out/r03ad; mkdir -p $O
timeout -k 10 200 python tools/chain_interference.py > $O/ci.log 2>&1; cat $O/ci.log
