# Text evidence of one round (run on the GPU box from the repo root): TAG=r05 bash tools/collect_text_profiles.sh
# Kernel sequences of one evaluation at four sizes, the resident panel kernel's timeline, the small sizes, the schedule A/B
# (look-ahead on two streams, the default until round 5, against 4096-wide panels alone on the chip), the 8-rank emulation.
set -o pipefail
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
T=${TAG:-r05}
for n in 500 4000 20000 50000; do
  rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trx_$n -o trace -- python3 $R/tools/eval_trace.py run $n > /dev/null 2>&1
  f=$(find $R/gpurun_out/trx_$n -name "*kernel_trace.csv" | head -1)
  python3 $R/tools/eval_trace.py show $f --seq > $R/gpurun_out/${T}_kernel_sequence_n$n.txt
  rm -rf $R/gpurun_out/trx_$n
done
cd $R
python3 tools/chain_timeline.py 4096 0 2>&1 | grep -v amdgpu > gpurun_out/${T}_chain_timeline_n4096.txt
python3 tools/small_n_timing.py 2>&1 | grep -v amdgpu > gpurun_out/${T}_small_sizes.txt
python3 tools/leaf_phases.py 1 1 2>&1 | grep -v amdgpu > gpurun_out/${T}_leaf_phases_in_chain.txt
python3 tools/padding_cliff.py 2>&1 | grep -v amdgpu > gpurun_out/${T}_padding_sizes.txt
(for o in panel_chain=0 panel_chain=1 panel_chain=2; do for w in 8 4 2; do python3 tools/shard_emulate.py --world $w --n 50000 --panel 1024 --opt $o 2>&1 | tail -1 | sed "s/^/$o: /"; done; done; python3 tools/shard_emulate.py --world 8 --n 100000 --panel 2048 2>&1 | tail -1) > gpurun_out/${T}_shard_emulation.txt
python3 tools/option_ab.py "chain_wide=0,lookahead_min=4608/chain_wide=1,lookahead_min=1099511627776" - 500,1000,2000,4000,6000,8000,12000,20000,30000,50000 3 2>&1 | grep -v amdgpu > gpurun_out/${T}_schedule_ab.txt
python3 tools/update_in_situ.py 50000 4096 2>&1 | grep -v amdgpu > gpurun_out/${T}_update_in_situ.txt
tail -3 gpurun_out/${T}_schedule_ab.txt
