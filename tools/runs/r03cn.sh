cd $GRAFT_REPO_ROOT
O=gpurun_out/r03cn; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $GRAFT_REPO_ROOT/$O/tcc -o t -- python3 $GRAFT_REPO_ROOT/tools/pmc_probe.py 0 9000 > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $GRAFT_REPO_ROOT/$O/fetch -o f -- python3 $GRAFT_REPO_ROOT/tools/pmc_probe.py 0 9000 > /dev/null 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_MFMA GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT --output-format csv -d $GRAFT_REPO_ROOT/$O/sq -o s -- python3 $GRAFT_REPO_ROOT/tools/pmc_probe.py 0 9000 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python tools/pmc_summary.py $O/tcc/t_counter_collection.csv $O/fetch/f_counter_collection.csv $O/sq/s_counter_collection.csv 2>&1 | grep -A8 "gemm_f64" | grep -v "^--"
timeout -k 10 300 python tools/macro_tile_probe.py 2>&1 | tail -4 > $O/timing.txt; cat $O/timing.txt
