set -e
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03e; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_DATA_FIFO_FULL SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS --output-format csv -d $O/p1 -o p1 -- python3 $GRAFT_REPO_ROOT/tools/pmc_probe.py 0 256 128 64 > $O/p1.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_CMD_FIFO_FULL SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_VALU --output-format csv -d $O/p2 -o p2 -- python3 $GRAFT_REPO_ROOT/tools/pmc_probe.py 0 256 128 64 > $O/p2.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv,glob
from collections import defaultdict
for p in sorted(glob.glob("gpurun_out/r03e/p*/*counter_collection.csv")):
    d=defaultdict(dict)
    for r in csv.DictReader(open(p)):
        if "gemm_f64" in r["Kernel_Name"]:
            d[(int(r["Dispatch_Id"]), r["Kernel_Name"][-40:])][r["Counter_Name"]]=float(r["Counter_Value"])
    print(p)
    for k in sorted(d):
        print(" ",k, {c:f"{v:.4g}" for c,v in sorted(d[k].items())})
PY
