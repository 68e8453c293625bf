cd $GRAFT_REPO_ROOT
for c in 0 60 240 1000; do echo "chain_loop $c"; FVGP_CHAIN_LOOP=$c timeout -k 10 300 python tools/shard_emulate.py --world 8 --n 50000 2>&1 | grep "^world"; done
for c in 0 240; do echo "chain_loop $c"; FVGP_CHAIN_LOOP=$c timeout -k 10 300 python tools/shard_emulate.py --world 4 --n 50000 2>&1 | grep "^world"; done
