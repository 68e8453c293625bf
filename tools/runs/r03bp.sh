cd $GRAFT_REPO_ROOT
O=gpurun_out/r03bp; mkdir -p $O
timeout -k 10 800 python tools/option_ab.py panel_fit 0,1 12000,16000,20000,30000,50000 4 > $O/ab.log 2>&1; cat $O/ab.log
