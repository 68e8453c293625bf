import sys, json
for line in sys.stdin:
    if line.startswith("{"):
        o = json.loads(line)
        print({k: round(o[k], 3) if isinstance(o[k], float) else o[k] for k in ("ms_per_step", "cholesky_tflops")}, "update TF/s", round(o["roofline"]["achieved"], 2), "avg launch ms", round(o["roofline"]["avg_launch_ms"], 3))
