cd $GRAFT_REPO_ROOT
O=gpurun_out/r03at; mkdir -p $O
timeout -k 10 600 python tools/option_ab.py update_stagger 0,2,4,8 8000,12000,20000,50000 4 > $O/ab.log 2>&1; cat $O/ab.log
