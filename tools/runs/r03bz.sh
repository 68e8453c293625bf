cd $GRAFT_REPO_ROOT
O=gpurun_out/r03bz; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for st in 0 30; do
FVGP_CHAIN_LOOP=$st rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/tr$st -o trace -- python3 $GRAFT_REPO_ROOT/tools/shard_emulate.py --world 8 --n 50000 --steps 2 > $GRAFT_REPO_ROOT/$O/run$st.log 2>&1
done
cd $GRAFT_REPO_ROOT
for st in 0 30; do echo "chain_loop $st"; tail -1 $O/run$st.log; python tools/trace_busy.py $O/tr$st/trace_kernel_trace.csv 100 | head -12; done
rm -rf $O/tr0 $O/tr30
