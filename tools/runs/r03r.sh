cd $GRAFT_REPO_ROOT
O=gpurun_out/r03r; mkdir -p $O
V=$GRAFT_REPO_ROOT/fvgp_amd/csrc/variants
for rep in 1 2; do
echo "== head" >> $O/ab.log; FVGP_HIP_LIB=$V/head/libfvgp_hip.so python tools/eval_trace.py run 50000 2>&1 | grep "^N" >> $O/ab.log
echo "== now" >> $O/ab.log; python tools/eval_trace.py run 50000 2>&1 | grep "^N" >> $O/ab.log
done
cat $O/ab.log
