cd $GRAFT_REPO_ROOT
for r in 0 4 8 16 32; do echo "reserve $r"; FVGP_UPDATE_RESERVE=$r FVGP_RESERVE_ROWS=-1 timeout -k 10 300 python tools/shard_emulate.py --world 8 --n 50000 2>&1 | grep "^world"; done
for r in 0 8 16; do echo "reserve $r world 4"; FVGP_UPDATE_RESERVE=$r FVGP_RESERVE_ROWS=-1 timeout -k 10 300 python tools/shard_emulate.py --world 4 --n 50000 2>&1 | grep "^world"; done
