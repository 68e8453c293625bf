cd $GRAFT_REPO_ROOT
O=gpurun_out/r03bo; mkdir -p $O
timeout -k 10 600 python tools/option_ab.py lookahead 1,0 6000,8000,10000,12000,16000 5 > $O/ab.log 2>&1; cat $O/ab.log
