cd $GRAFT_REPO_ROOT
O=gpurun_out/r03v; mkdir -p $O
timeout -k 10 1100 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -15 > $O/pytest_gpu.log; cat $O/pytest_gpu.log
