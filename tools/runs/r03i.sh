set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03i; mkdir -p $O
for n in 8000 20000 50000; do
  python tools/eval_trace.py run $n >> $O/eval_new.log 2>&1
  FVGP_HIP_LIB=$GRAFT_REPO_ROOT/fvgp_amd/csrc/variants/base/libfvgp_hip.so python tools/eval_trace.py run $n >> $O/eval_base.log 2>&1
done
cat $O/eval_new.log $O/eval_base.log
python -m pytest tests/test_gpu_primitives.py -x -q -m gpu 2>&1 | tail -5
