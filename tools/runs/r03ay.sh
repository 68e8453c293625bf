cd $GRAFT_REPO_ROOT
O=gpurun_out/r03ay; mkdir -p $O
timeout -k 10 800 python tools/option_ab.py chain_loop=0,leaf_tiles_rows=8192/chain_loop=0,leaf_tiles_rows=100000/chain_loop=60,leaf_tiles_rows=100000/chain_loop=60,leaf_tiles_rows=24576 - 12000,20000,50000 4 > $O/ab.log 2>&1; cat $O/ab.log
