cd $GRAFT_REPO_ROOT
O=gpurun_out/r03aq; mkdir -p $O
export FVGP_PANEL_SQUARE=$1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/tr -o trace -- python3 $GRAFT_REPO_ROOT/tools/eval_trace.py run 50000 > $GRAFT_REPO_ROOT/$O/run.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/trace_timeline.py $O/tr/trace_kernel_trace.csv > $O/timeline_n50000_sq$1.txt 2>&1
python tools/eval_trace.py show $O/tr/trace_kernel_trace.csv --seq > $O/seq_sq$1.txt 2>&1
head -48 $O/timeline_n50000_sq$1.txt
rm -rf $O/tr
