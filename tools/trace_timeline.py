"""Post-process a `rocprofv3 --kernel-trace --output-format csv` trace of bench.py: per evaluation, where the time of
the trailing-update stream goes (update launches, gaps between them) and what the panel chain does underneath.
Usage: python tools/trace_timeline.py gpurun_out/<dir>/trace_kernel_trace.csv [eval_index]"""
import csv
import sys
from collections import defaultdict


def short(name):
    for key in ("kmat_kernel", "leaf_kernel", "fwd_step", "bwd_step", "rhs_rows", "rowsumsq", "rows_to_vec", "pad_identity",
                "sum_kernel", "copy_cols", "grad_trace", "copy_lower", "kt_alpha", "colsumsq", "panel_kernel", "bwd_sweep", "chain_kernel"):
        if key in name:
            return key
    if "gemm_f64_kernel" in name:
        return "gemm" + name[name.index("<"):name.index(">") + 1].replace(" ", "")
    return name[:40]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    for r in rows:
        r["s"], r["e"], r["k"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])
    rows.sort(key=lambda r: r["s"])
    starts = [i for i, r in enumerate(rows) if r["k"] == "kmat_kernel" and int(r["Grid_Size_Y"]) > 8]
    which = int(sys.argv[2]) if len(sys.argv) > 2 else len(starts) - 1
    lo = starts[which]
    hi = starts[which + 1] if which + 1 < len(starts) else len(rows)
    ev = rows[lo:hi]
    t0, t1 = ev[0]["s"], max(r["e"] for r in ev)
    print(f"evaluation {which}: {len(ev)} kernels, span {(t1 - t0) / 1e6:.2f} ms")
    agg = defaultdict(lambda: [0, 0.0])
    for r in ev:
        a = agg[(r["k"], r["Queue_Id"])]
        a[0] += 1
        a[1] += (r["e"] - r["s"]) / 1e6
    for (k, q), (c, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"  {k:28s} queue {q:>3s}  calls {c:5d}  sum {ms:9.2f} ms  avg {1e3 * ms / c:9.1f} us")
    upd = [r for r in ev if r["k"].startswith("gemm<0,0,1")]
    busy = sum(r["e"] - r["s"] for r in upd) / 1e6
    gaps = [(b["s"] - a["e"]) / 1e6 for a, b in zip(upd, upd[1:])]
    print(f"trailing updates: {len(upd)} launches, busy {busy:.2f} ms; first starts {(upd[0]['s'] - t0) / 1e6:.2f} ms in, last ends "
          f"{(t1 - upd[-1]['e']) / 1e6:.2f} ms before the end; gaps between them: sum {sum(g for g in gaps if g > 0):.2f} ms")
    print("  idx   start_ms   dur_ms    WGs   us/round(512)  gap_before_ms  chain kernels overlapping (count, ms)")
    for i, r in enumerate(upd):
        wgs = int(r["Grid_Size_X"]) // 256
        ov = [c for c in ev if c["Queue_Id"] != r["Queue_Id"] and c["s"] < r["e"] and c["e"] > r["s"]]
        ovms = sum(min(c["e"], r["e"]) - max(c["s"], r["s"]) for c in ov) / 1e6
        dur = (r["e"] - r["s"]) / 1e6
        gap = gaps[i - 1] if i else 0.0
        print(f"  {i:3d} {(r['s'] - t0) / 1e6:10.2f} {dur:8.3f} {wgs:6d} {1e3 * dur / max(wgs / 512.0, 1e-9):12.1f} {gap:12.3f}   {len(ov):4d} {ovms:8.2f}")


if __name__ == "__main__":
    main()
