cd $GRAFT_REPO_ROOT
O=gpurun_out/r03bk; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_primitives.py -x -q -m gpu -k "backward_sweep or potrf_solve" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -5 $O/tests.log
timeout -k 10 600 python tools/option_ab.py bwd_sweep 0,1 4000,8000,12000,20000,50000 5 > $O/ab.log 2>&1; cat $O/ab.log
