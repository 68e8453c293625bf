cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03t; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/post -o trace -- python3 $GRAFT_REPO_ROOT/tools/eval_trace.py runpost 20000 1000 > $O/post.log 2>&1
grep "^N" $O/post.log
cd $GRAFT_REPO_ROOT
python tools/eval_trace.py show $O/post/trace_kernel_trace.csv --seq --back 2 > $O/post_seq.txt
head -40 $O/post_seq.txt
