cd $GRAFT_REPO_ROOT
O=gpurun_out/r03bu; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_primitives.py -x -q -m gpu -k "macro_tile or backward_sweep" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -5 $O/tests.log
