cd $GRAFT_REPO_ROOT
O=gpurun_out/r03ai; mkdir -p $O
timeout -k 10 400 python tools/option_ab.py outer_block 1024,512,768,1536 8000,12000,20000 5 > $O/ab.log 2>&1; cat $O/ab.log
