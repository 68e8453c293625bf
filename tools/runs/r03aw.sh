cd $GRAFT_REPO_ROOT
O=gpurun_out/r03aw; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_primitives.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
timeout -k 10 600 python tools/option_ab.py chain_loop 0,30,60,120,240 12000,20000,50000 4 > $O/ab.log 2>&1; cat $O/ab.log
FVGP_LEAF_TILES_ROWS=100000 FVGP_SMALL_THRESHOLD=0 timeout -k 10 200 python tools/chain_stamps.py 20000 > $O/cs.log 2>&1; sed -n 8,22p $O/cs.log
