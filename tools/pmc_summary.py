"""Per-kernel sums of a rocprofv3 --pmc counter csv:  python tools/pmc_summary.py <counter_collection.csv> [...]"""
import csv, sys
from collections import defaultdict
for path in sys.argv[1:]:
    agg = defaultdict(lambda: defaultdict(float)); cnt = defaultdict(set)
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "")[:70]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k].add(r["Dispatch_Id"])
    print(path)
    for k, d in agg.items():
        print(f"  {k}  dispatches {len(cnt[k])}")
        for c, v in sorted(d.items()):
            print(f"      {c:32s} {v / len(cnt[k]):.4g} per dispatch")
