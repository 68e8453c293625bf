cd $GRAFT_REPO_ROOT
O=gpurun_out/r03bj; mkdir -p $O
timeout -k 10 800 python tools/option_ab.py big_threshold 24576,16384,32768 50000 4 > $O/ab.log 2>&1; cat $O/ab.log
timeout -k 10 800 python tools/option_ab.py small_threshold=12288,outer_block_small=512/small_threshold=16384,outer_block_small=512/small_threshold=12288,outer_block_small=256/small_threshold=6144,outer_block_small=512/small_threshold=8192,outer_block_small=384 - 8000,12000,20000 5 > $O/ab2.log 2>&1; cat $O/ab2.log
