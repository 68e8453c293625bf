cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03s; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
FVGP_UPDATE_RESERVE=32 FVGP_RESERVE_ROWS=-1 rocprofv3 --kernel-trace --output-format csv -d $O/tr8k -o trace -- python3 $GRAFT_REPO_ROOT/tools/eval_trace.py run 8000 > $O/tr8k.log 2>&1
grep "^N" $O/tr8k.log
