set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03f; mkdir -p $O
python tools/gemm_ab.py 0 512 > $O/gemm_ab.log 2>&1
for n in 8000 20000 50000; do
  FVGP_HIP_LIB=$GRAFT_REPO_ROOT/fvgp_amd/csrc/variants/lean/libfvgp_hip.so python tools/eval_trace.py run $n >> $O/eval_lean.log 2>&1
done
cat $O/gemm_ab.log $O/eval_lean.log
