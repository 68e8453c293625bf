set -e
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03m; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/tr50k -o trace -- python3 $GRAFT_REPO_ROOT/tools/eval_trace.py run 50000 > $O/tr50k.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $O/tr20k -o trace -- python3 $GRAFT_REPO_ROOT/tools/eval_trace.py run 20000 > $O/tr20k.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $O/tr8k -o trace -- python3 $GRAFT_REPO_ROOT/tools/eval_trace.py run 8000 > $O/tr8k.log 2>&1
cd $GRAFT_REPO_ROOT
cat $O/tr50k.log $O/tr20k.log $O/tr8k.log | grep "^N"
