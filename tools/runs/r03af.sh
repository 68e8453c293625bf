cd $GRAFT_REPO_ROOT
O=gpurun_out/r03af; mkdir -p $O
for n in 8000 20000; do
echo "n=$n base" >> $O/h.log; timeout -k 5 120 python tools/eval_trace.py run $n 2>&1 | grep "^N" >> $O/h.log
for rows in -1 16384 8192; do echo "n=$n half rows=$rows" >> $O/h.log; FVGP_UPDATE_RESERVE=-1 FVGP_RESERVE_ROWS=$rows timeout -k 5 120 python tools/eval_trace.py run $n 2>&1 | grep "^N" >> $O/h.log; done
done
echo "n=50000 base" >> $O/h.log; timeout -k 5 200 python tools/eval_trace.py run 50000 2>&1 | grep "^N" >> $O/h.log
for rows in 24576 16384 8192; do echo "n=50000 half rows=$rows" >> $O/h.log; FVGP_UPDATE_RESERVE=-1 FVGP_RESERVE_ROWS=$rows timeout -k 5 200 python tools/eval_trace.py run 50000 2>&1 | grep "^N" >> $O/h.log; done
for r in 0 -1; do echo "world8 reserve=$r" >> $O/h.log; FVGP_UPDATE_RESERVE=$r FVGP_RESERVE_ROWS=-1 timeout -k 10 200 python tools/shard_emulate.py --world 8 --n 50000 2>&1 | grep "^world" >> $O/h.log; done
cat $O/h.log
