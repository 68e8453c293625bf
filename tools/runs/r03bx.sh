cd $GRAFT_REPO_ROOT
O=gpurun_out/r03bx; mkdir -p $O
timeout -k 10 400 python bench.py --gpus 1 --mode sharded --backend nccl --steps 2 --warmup 1 --no-cpu-baseline --no-configs > $O/sharded.json 2> $O/sharded.err; echo "rc=$?"; tail -2 $O/sharded.err; python - <<'PY'
import json
l=[x for x in open('gpurun_out/r03bx/sharded.json') if x.startswith('{')][-1]
d=json.loads(l)
print({k:d.get(k) for k in ['value','ms_per_step','scaling','rel_diff_vs_single_gpu','replicas','collectives_via']})
PY
