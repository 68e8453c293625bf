cd $GRAFT_REPO_ROOT
O=gpurun_out/r03ci; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for r in 16 32; do
FVGP_UPDATE_RESERVE=$r FVGP_RESERVE_ROWS=-1 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/tr$r -o trace -- python3 $GRAFT_REPO_ROOT/tools/shard_emulate.py --world 8 --n 50000 --steps 2 > $GRAFT_REPO_ROOT/$O/run$r.log 2>&1
done
cd $GRAFT_REPO_ROOT
for r in 16 32; do echo "== reserve $r"; tail -1 $O/run$r.log; python tools/trace_busy.py $O/tr$r/trace_kernel_trace.csv 112 | head -10; done
rm -rf $O/tr16 $O/tr32
