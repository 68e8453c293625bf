cd $GRAFT_REPO_ROOT
O=gpurun_out/r03aj; mkdir -p $O
timeout -k 10 500 python tools/option_ab.py small_threshold 0,8192,12288,16384 8000,12000,20000,50000 5 > $O/ab.log 2>&1; cat $O/ab.log
