cd $GRAFT_REPO_ROOT
O=gpurun_out/r03ba; mkdir -p $O
timeout -k 10 300 python tools/gemm_bench.py > $O/gb.log 2>&1; cat $O/gb.log
