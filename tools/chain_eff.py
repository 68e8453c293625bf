"""Efficiency of the resident panel kernel at one size: per launch of chain_kernel in a `rocprofv3 --kernel-trace` csv, its duration against the
MFMA time of its flops (products K = 128 at 13.65 us per 128^3 on one of 256 compute units, triangular solves at half that).
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/ce -o trace -- python3 tools/eval_trace.py run 50000
  python tools/chain_eff.py gpurun_out/ce/trace_kernel_trace.csv 50000"""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "chain_kernel" in r["Kernel_Name"]]
n = int(sys.argv[2]); nb = -(-n // 128)
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last evaluation's launches: panels of 32 blocks
npan = -(-nb // 32)
rows = rows[-npan:]
tot = 0.0; tot_ideal = 0.0
for p, r in enumerate(rows):
    dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    R = nb - 32 * p; w = min(32, R)
    prods = sum((R - k - 1) * k for k in range(w))            # blocks below the diagonal in column k, k products each
    solves = sum(R - k - 1 for k in range(w))
    syrk = sum(k for k in range(w))
    ideal = (prods * 13.65 + (solves + syrk) * 6.8 + w * 23.4) / 256.0
    crit = w * 32.0
    tot += dur; tot_ideal += ideal
    print(f"panel {p}: rows {R} width {w}: {dur:8.1f} us, flop time on 256 CUs {ideal:8.1f} us ({ideal / dur:.2f}), leaf chain at 32 us per step {crit:6.0f} us, grid {r['Grid_Size_X'] if 'Grid_Size_X' in r else ''}")
print(f"all panels {tot / 1e3:.2f} ms, flop time {tot_ideal / 1e3:.2f} ms ({tot_ideal / tot:.2f})")
