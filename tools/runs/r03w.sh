cd $GRAFT_REPO_ROOT
O=gpurun_out/r03w; mkdir -p $O
timeout -k 10 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "rc=$?"
tail -c 6000 $O/bench_default.json
tail -5 $O/bench_default.err
timeout -k 10 300 python bench.py --gpus 1 --mode sharded --backend nccl --steps 2 --warmup 1 > $O/bench_sharded1.json 2> $O/bench_sharded1.err; echo "rc=$?"
tail -c 3000 $O/bench_sharded1.json; tail -3 $O/bench_sharded1.err
