set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03a
python tools/gemm_ab.py 0 128 > gpurun_out/r03a/gemm_ab.log 2>&1
for n in 8000 20000 50000; do
  python tools/eval_trace.py run $n >> gpurun_out/r03a/eval_base.log 2>&1
  FVGP_HIP_LIB=$GRAFT_REPO_ROOT/fvgp_amd/csrc/variants/pipe/libfvgp_hip.so python tools/eval_trace.py run $n >> gpurun_out/r03a/eval_pipe.log 2>&1
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r03a/tr20k -o trace -- python3 $GRAFT_REPO_ROOT/tools/eval_trace.py run 20000 > $GRAFT_REPO_ROOT/gpurun_out/r03a/tr20k.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r03a/tr8k -o trace -- python3 $GRAFT_REPO_ROOT/tools/eval_trace.py run 8000 > $GRAFT_REPO_ROOT/gpurun_out/r03a/tr8k.log 2>&1
cd $GRAFT_REPO_ROOT
cat gpurun_out/r03a/gemm_ab.log gpurun_out/r03a/eval_base.log gpurun_out/r03a/eval_pipe.log
