cd $GRAFT_REPO_ROOT
O=gpurun_out/r03cr; mkdir -p $O
timeout -k 10 600 python tools/option_ab.py lookahead_min 6144,3072 3500,4000,4800,5600,6000 6 > $O/ab.log 2>&1; cat $O/ab.log
