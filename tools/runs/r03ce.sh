cd $GRAFT_REPO_ROOT
O=gpurun_out/r03ce; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/tr -o trace -- python3 $GRAFT_REPO_ROOT/tools/eval_trace.py run 8000 > $GRAFT_REPO_ROOT/$O/run.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/eval_trace.py show $O/tr/trace_kernel_trace.csv --seq > $O/seq.txt 2>&1
head -24 $O/seq.txt
rm -rf $O/tr
