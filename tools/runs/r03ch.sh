cd $GRAFT_REPO_ROOT
O=gpurun_out/r03ch; mkdir -p $O
V=$GRAFT_REPO_ROOT/fvgp_amd/csrc/variants/onepercu/libfvgp_hip.so
C=$GRAFT_REPO_ROOT/fvgp_amd/csrc/libfvgp_hip.so
cd /tmp && export TMPDIR=/tmp
for t in cur one; do
lib=$C; [ $t = one ] && lib=$V
FVGP_HIP_LIB=$lib rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/tr$t -o trace -- python3 $GRAFT_REPO_ROOT/tools/shard_emulate.py --world 8 --n 50000 --steps 2 > $GRAFT_REPO_ROOT/$O/run$t.log 2>&1
done
cd $GRAFT_REPO_ROOT
for t in cur one; do echo "== $t"; tail -1 $O/run$t.log; python tools/trace_busy.py $O/tr$t/trace_kernel_trace.csv 110 | head -9; done
rm -rf $O/trcur $O/trone
