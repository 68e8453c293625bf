cd $GRAFT_REPO_ROOT
O=gpurun_out/r03x; mkdir -p $O
timeout -k 10 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-configs > $O/b0.json 2>/dev/null
FVGP_OVERLAP_COLS=1 timeout -k 10 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-configs > $O/b1.json 2>/dev/null
python tools/pick_bench_fields.py $O/b0.json; python tools/pick_bench_fields.py $O/b1.json
