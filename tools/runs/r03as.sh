cd $GRAFT_REPO_ROOT
O=gpurun_out/r03as; mkdir -p $O
timeout -k 10 1000 python -m pytest tests/ -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?"; tail -4 $O/tests.log
timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; python tools/pick_bench_fields.py < $O/bench.json
