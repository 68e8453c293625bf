cd $GRAFT_REPO_ROOT
O=gpurun_out/r03bd; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/tr -o trace -- python3 $GRAFT_REPO_ROOT/tools/eval_trace.py runpost 20000 1000 > $GRAFT_REPO_ROOT/$O/run.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/eval_trace.py show $O/tr/trace_kernel_trace.csv --seq --back 2 > $O/seq.txt 2>&1
cat $O/run.log | tail -2; cat $O/seq.txt | head -70
rm -rf $O/tr
