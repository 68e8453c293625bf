cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03y; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/em8 -o trace -- python3 $GRAFT_REPO_ROOT/tools/shard_emulate.py --world 8 --n 50000 --steps 2 > $O/em8.log 2>&1
grep "^world" $O/em8.log
ls $O/em8
