cd $GRAFT_REPO_ROOT
O=gpurun_out/r03ab; mkdir -p $O
for n in 20000 50000; do for t in 24576 16384 12288 8192; do echo "n=$n big_threshold=$t" >> $O/bt.log; FVGP_BIG_THRESHOLD=$t timeout -k 5 200 python tools/eval_trace.py run $n 2>&1 | grep "^N" >> $O/bt.log; done; done
cat $O/bt.log
