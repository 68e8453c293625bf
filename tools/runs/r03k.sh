set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03k; mkdir -p $O
python tools/gemm_ab.py 2816 6912 > $O/gemm_ab.log 2>&1
cat $O/gemm_ab.log
for n in 8000 20000 50000; do
  python tools/eval_trace.py run $n >> $O/eval_dma.log 2>&1
  FVGP_HIP_LIB=$GRAFT_REPO_ROOT/fvgp_amd/csrc/variants/atom/libfvgp_hip.so python tools/eval_trace.py run $n >> $O/eval_atom.log 2>&1
done
cat $O/eval_dma.log $O/eval_atom.log
