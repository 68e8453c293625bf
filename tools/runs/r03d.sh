set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03d; mkdir -p $O
python tools/gemm_ab.py 0 256 > $O/gemm_ab.log 2>&1
for n in 8000 20000 50000; do
  FVGP_HIP_LIB=$GRAFT_REPO_ROOT/fvgp_amd/csrc/variants/swz/libfvgp_hip.so python tools/eval_trace.py run $n >> $O/eval_swz.log 2>&1
done
cat $O/gemm_ab.log $O/eval_swz.log
