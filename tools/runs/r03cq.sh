cd $GRAFT_REPO_ROOT
O=gpurun_out/r03cq; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_primitives.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
timeout -k 10 200 python tools/leaf_phases.py 2>&1 | grep -E "TRSM|total"
for n in 4000 8000 12000; do python tools/eval_trace.py run $n 2>&1 | grep "^N" | awk '{print $1,$2,$3,$4,$(NF-2),$(NF-1),$NF}'; done
