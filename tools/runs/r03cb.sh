cd $GRAFT_REPO_ROOT
for sk in 0 1 2 3 5; do echo "side_skip $sk"; for n in 12000 20000; do FVGP_SIDE_SKIP=$sk python tools/eval_trace.py run $n 2>&1 | grep "^N" | awk '{print $1,$2,$(NF-2),$(NF-1),$NF}'; done; done
for q in 1 2 8; do echo "GPU_MAX_HW_QUEUES $q"; for n in 12000 20000; do GPU_MAX_HW_QUEUES=$q python tools/eval_trace.py run $n 2>&1 | grep "^N" | awk '{print $1,$2,$(NF-2),$(NF-1),$NF}'; done; done
