cd $GRAFT_REPO_ROOT
O=gpurun_out/r03bt; mkdir -p $O
timeout -k 10 800 python tools/option_ab.py small_tile_max 160,96,256,400 8000,12000,20000 4 > $O/ab1.log 2>&1; cat $O/ab1.log
timeout -k 10 800 python tools/option_ab.py small_tile_max_update 512,256,768,1024 8000,12000,20000 4 > $O/ab2.log 2>&1; cat $O/ab2.log
