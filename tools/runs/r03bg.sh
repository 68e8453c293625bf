cd $GRAFT_REPO_ROOT
O=gpurun_out/r03bi; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
FVGP_CHAIN_LOOP=240 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/tr -o trace -- python3 $GRAFT_REPO_ROOT/tools/shard_emulate.py --world 8 --n 50000 --steps 2 > $GRAFT_REPO_ROOT/$O/run.log 2>&1
cd $GRAFT_REPO_ROOT
tail -1 $O/run.log
python tools/trace_busy.py $O/tr/trace_kernel_trace.csv 105 > $O/busy.txt 2>&1; cat $O/busy.txt
rm -rf $O/tr
