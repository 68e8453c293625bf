cd $GRAFT_REPO_ROOT
O=gpurun_out/r03cs; mkdir -p $O
timeout -k 10 800 python tools/option_ab.py big_threshold 24576,16384,20480,32768 26000,30000,36000 4 > $O/ab.log 2>&1; cat $O/ab.log
