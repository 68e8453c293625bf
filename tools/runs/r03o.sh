cd $GRAFT_REPO_ROOT
O=gpurun_out/r03o; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_dist.py -x -q -m gpu 2>&1 | tail -25 > $O/dist.log; cat $O/dist.log
timeout -k 10 600 python -m pytest tests/test_gpu_facade.py -x -q -m gpu 2>&1 | tail -8
for w in 8 4 2; do timeout -k 10 300 python tools/shard_emulate.py --world $w --n 50000 2>&1 | grep "^world" >> $O/emul.log; done
cat $O/emul.log
