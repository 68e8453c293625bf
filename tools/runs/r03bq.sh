cd $GRAFT_REPO_ROOT
timeout -k 10 300 python tools/shard_emulate.py --world 8 --n 50000 --steps 1 --stamps 2>&1 | tail -56
