set -e
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03c; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $O/p1 -o p1 -- python3 $GRAFT_REPO_ROOT/tools/pmc_kloop.py > $O/p1.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INST_LEVEL_LDS --output-format csv -d $O/p2 -o p2 -- python3 $GRAFT_REPO_ROOT/tools/pmc_kloop.py > $O/p2.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_INSTS_SALU SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_VMEM SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_VALU_MFMA_COEXEC_CYCLES --output-format csv -d $O/p3 -o p3 -- python3 $GRAFT_REPO_ROOT/tools/pmc_kloop.py > $O/p3.log 2>&1
cd $GRAFT_REPO_ROOT
find $O -name "*counter_collection.csv" | xargs python tools/pmc_summary.py
