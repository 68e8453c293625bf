set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03l; mkdir -p $O
V=$GRAFT_REPO_ROOT/fvgp_amd/csrc/variants
for rep in 1 2; do
for n in 20000 50000; do
  for v in ls lsatom atom; do
    echo "== $v" >> $O/eval.log
    FVGP_HIP_LIB=$V/$v/libfvgp_hip.so python tools/eval_trace.py run $n 2>&1 | grep "^N" >> $O/eval.log
  done
  echo "== dma(default)" >> $O/eval.log
  python tools/eval_trace.py run $n 2>&1 | grep "^N" >> $O/eval.log
done
done
cat $O/eval.log
