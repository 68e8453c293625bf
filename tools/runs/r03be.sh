cd $GRAFT_REPO_ROOT
O=gpurun_out/r03be; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_facade.py -x -q -m gpu -k "posterior" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
for v in 1 1 0; do FVGP_POSTERIOR_HALVES=$v python tools/eval_trace.py runpost 20000 1000 2>&1 | grep "^N"; done
