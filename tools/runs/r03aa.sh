cd $GRAFT_REPO_ROOT
O=gpurun_out/r03aa; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_facade.py tests/test_gpu_dist.py -x -q -m gpu 2>&1 | tail -25 > $O/t.log; cat $O/t.log
