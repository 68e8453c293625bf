cd $GRAFT_REPO_ROOT
O=gpurun_out/r03ac; mkdir -p $O
for p in 512 1024 2048; do timeout -k 10 200 python tools/shard_emulate.py --world 8 --n 50000 --panel $p 2>&1 | grep "^world" >> $O/em.log; done
timeout -k 10 300 python -m pytest tests/test_gpu_facade.py tests/test_gpu_dist.py -x -q -m gpu 2>&1 | tail -3 >> $O/em.log
cat $O/em.log
