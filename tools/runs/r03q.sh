cd $GRAFT_REPO_ROOT
O=gpurun_out/r03q; mkdir -p $O
for n in 8000 20000; do
for r in 0 8 16 32 64; do echo "n=$n reserve=$r rows=-1" >> $O/res.log; FVGP_UPDATE_RESERVE=$r FVGP_RESERVE_ROWS=-1 timeout -k 5 120 python tools/eval_trace.py run $n 2>&1 | grep "^N" >> $O/res.log; done
done
for r in 0 8 16; do for rows in 16384 -1; do echo "n=50000 reserve=$r rows=$rows" >> $O/res.log; FVGP_UPDATE_RESERVE=$r FVGP_RESERVE_ROWS=$rows timeout -k 5 200 python tools/eval_trace.py run 50000 2>&1 | grep "^N" >> $O/res.log; done; done
cat $O/res.log
