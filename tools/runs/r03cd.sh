cd $GRAFT_REPO_ROOT
O=gpurun_out/r03cd; mkdir -p $O
timeout -k 10 800 python tools/option_ab.py first_panel 0,512,1024,256 20000,30000,50000 4 > $O/ab.log 2>&1; cat $O/ab.log
