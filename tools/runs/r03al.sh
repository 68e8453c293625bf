cd $GRAFT_REPO_ROOT
O=gpurun_out/r03bm; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/tr20k -o trace -- python3 $GRAFT_REPO_ROOT/tools/eval_trace.py run 20000 > $GRAFT_REPO_ROOT/$O/run.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/trace_timeline.py $O/tr20k/trace_kernel_trace.csv > $O/timeline_n20000.txt 2>&1
python tools/eval_trace.py show $O/tr20k/trace_kernel_trace.csv --seq > $O/seq.txt 2>&1
head -50 $O/timeline_n20000.txt
rm -rf $O/tr20k
