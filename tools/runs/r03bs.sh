cd $GRAFT_REPO_ROOT
O=gpurun_out/r03bs; mkdir -p $O
timeout -k 10 800 python tools/option_ab.py leaf_tiles=1,leaf_tiles_rows=8192/leaf_tiles=0,leaf_tiles_rows=8192/leaf_tiles=1,leaf_tiles_rows=4096/leaf_tiles=1,leaf_tiles_rows=2048 - 4000,8000,12000,20000,50000 4 > $O/ab.log 2>&1; cat $O/ab.log
for v in 1 0; do echo "leaf_tiles $v"; FVGP_LEAF_TILES=$v timeout -k 10 300 python tools/shard_emulate.py --world 4 --n 50000 2>&1 | grep "^world"; done
for v in 1 0; do echo "leaf_tiles $v"; FVGP_LEAF_TILES=$v timeout -k 10 300 python tools/shard_emulate.py --world 8 --n 100000 2>&1 | grep "^world"; done
