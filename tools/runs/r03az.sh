cd $GRAFT_REPO_ROOT
O=gpurun_out/r03az; mkdir -p $O
timeout -k 10 800 python tools/option_ab.py update_atomic_k 0,512,1024,2048 8000,12000,20000,50000 4 > $O/ab.log 2>&1; cat $O/ab.log
