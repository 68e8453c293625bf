cd $GRAFT_REPO_ROOT
O=gpurun_out/r03bf; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_primitives.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
for w in 8 4; do timeout -k 10 300 python tools/shard_emulate.py --world $w --n 50000 2>&1 | grep "^world"; done
