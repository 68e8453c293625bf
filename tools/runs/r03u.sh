cd $GRAFT_REPO_ROOT
O=gpurun_out/r03u; mkdir -p $O
for rep in 1 2; do for n in 8000 20000 50000; do for v in 0 1; do echo "n=$n overlap_cols=$v" >> $O/ab.log; FVGP_OVERLAP_COLS=$v timeout -k 5 200 python tools/eval_trace.py run $n 2>&1 | grep "^N" >> $O/ab.log; done; done; done
cat $O/ab.log
