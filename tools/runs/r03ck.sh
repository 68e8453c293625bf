cd $GRAFT_REPO_ROOT
O=gpurun_out/r03ck; mkdir -p $O
timeout -k 10 1000 python -m pytest tests/ -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
s=$(date +%s); timeout -k 10 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$? in $(( $(date +%s) - s )) s"; python tools/pick_bench_fields.py < $O/bench_default.json; python - <<'PY'
import json
l=[x for x in open('gpurun_out/r03ck/bench_default.json') if x.startswith('{')]
print(len(l),'json lines'); d=json.loads(l[-1]); print(d['value'], d['cpu_baseline'], list(d.get('configs',{}).keys()) if isinstance(d.get('configs'),dict) else d.keys())
PY
