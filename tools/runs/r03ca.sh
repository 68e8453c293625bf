cd $GRAFT_REPO_ROOT
O=gpurun_out/r03ca; mkdir -p $O
timeout -k 10 600 python tools/option_ab.py update_low 0,1 8000,12000,20000,50000 4 > $O/ab.log 2>&1; cat $O/ab.log
