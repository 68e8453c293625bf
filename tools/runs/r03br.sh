cd $GRAFT_REPO_ROOT
for v in 1 0; do echo "leaf_tiles $v"; FVGP_LEAF_TILES=$v timeout -k 10 300 python tools/shard_emulate.py --world 8 --n 50000 2>&1 | grep "^world"; done
for v in 2 4; do echo "stagger $v"; FVGP_UPDATE_STAGGER=$v timeout -k 10 300 python tools/shard_emulate.py --world 8 --n 50000 2>&1 | grep "^world"; done
for v in 512 2048; do echo "panel $v"; timeout -k 10 300 python tools/shard_emulate.py --world 8 --n 50000 --panel $v 2>&1 | grep "^world"; done
