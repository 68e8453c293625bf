cd $GRAFT_REPO_ROOT
O=gpurun_out/r03bl; mkdir -p $O
timeout -k 10 600 python tools/option_ab.py bwd_sweep 0,1 4000,8000,12000,20000,50000 5 > $O/ab.log 2>&1; cat $O/ab.log
