cd $GRAFT_REPO_ROOT
export TAG=r03
bash tools/collect_profiles.sh > gpurun_out/r03_collect.log 2>&1; echo "collect rc=$?"
O=gpurun_out/r03_profiles
# the sharded schedule of one rank of 8 / 4 / 2 (collectives replaced by local copies of the same size)
for w in 8 4 2; do timeout -k 10 300 python tools/shard_emulate.py --world $w --n 50000 2>&1 | grep "^world" >> $O/shard_emulation.txt; done
timeout -k 10 300 python tools/shard_emulate.py --world 8 --n 100000 2>&1 | grep "^world" >> $O/shard_emulation.txt
# K-loop evidence: variants (0 = shipped loop; 64 = the round-2 loop: padded images, 8-byte reads, vector-ALU address bumps, register staging)
python tools/gemm_ab.py 0 64 > $O/gemm_ab_new_vs_r02.txt 2>&1
# small sizes
for n in 4000 8000 12000 20000; do python tools/eval_trace.py run $n 2>&1 | grep "^N" >> $O/small_sizes.txt; done
# round-3 mechanisms, same-process A/B
python tools/option_ab.py leaf_yield=0,chain_yield=0/leaf_yield=1,chain_yield=0/leaf_yield=1,chain_yield=1 - 8000,12000,20000,50000 4 > $O/yield_ab.txt 2>&1
python tools/option_ab.py bwd_sweep 0,1 4000,8000,20000,50000 4 > $O/bwd_sweep_ab.txt 2>&1
python tools/option_ab.py small_threshold 0,12288 8000,12000,20000,50000 4 > $O/small_panels_ab.txt 2>&1
for v in 0 1; do FVGP_POSTERIOR_HALVES=$v python tools/eval_trace.py runpost 20000 1000 2>&1 | grep "^N" | sed "s/^/posterior_halves=$v /" >> $O/posterior_ab.txt; done
FVGP_LEAF_TILES_ROWS=100000 FVGP_SMALL_THRESHOLD=0 python tools/chain_stamps.py 20000 2>&1 | sed -n 1,40p > $O/chain_stamps_n20000.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_kloop -o k -- python3 $GRAFT_REPO_ROOT/tools/pmc_probe.py 0 64 > /dev/null 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_tcc -o t -- python3 $GRAFT_REPO_ROOT/tools/pmc_probe.py 0 64 > /dev/null 2>&1
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/tr20k -o trace -- python3 $GRAFT_REPO_ROOT/tools/eval_trace.py run 20000 > /dev/null 2>&1
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/tr50k -o trace -- python3 $GRAFT_REPO_ROOT/tools/eval_trace.py run 50000 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
ls $O
cat $O/shard_emulation.txt $O/small_sizes.txt
tail -3 gpurun_out/r03_collect.log
