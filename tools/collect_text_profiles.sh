set -o pipefail
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for n in 20000 50000; do
  rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trx_$n -o trace -- python3 $R/tools/eval_trace.py run $n > /dev/null 2>&1
  f=$(find $R/gpurun_out/trx_$n -name "*kernel_trace.csv" | head -1)
  python3 $R/tools/trace_timeline.py $f > $R/gpurun_out/${TAG:-r05}_timeline_n$n.txt
  rm -rf $R/gpurun_out/trx_$n
done
cd $R
python3 tools/chain_timeline.py 4096 0 2>&1 | grep -v amdgpu > gpurun_out/${TAG:-r05}_chain_timeline_n4096.txt
python3 tools/chain_timeline.py 20000 -1 2>&1 | grep -v amdgpu > gpurun_out/${TAG:-r05}_chain_launches_n20000.txt
python3 tools/small_n_timing.py 2>&1 | grep -v amdgpu > gpurun_out/${TAG:-r05}_small_sizes.txt
(for o in panel_chain=1 panel_chain=2; do for w in 8 4 2; do python3 tools/shard_emulate.py --world $w --n 50000 --panel 1024 --opt $o 2>&1 | tail -1 | sed "s/^/$o: /"; done; done) > gpurun_out/${TAG:-r05}_shard_emulation.txt
python3 tools/option_ab.py "panel_chain=0,cols_split=0/panel_chain=1,cols_split=0/panel_chain=1,cols_split=1" - 8000,12000,20000,32000,50000 3 2>&1 | grep -v amdgpu > gpurun_out/${TAG:-r05}_panel_chain_ab.txt
python3 tools/update_in_situ.py 50000 2048 2>&1 | grep -v amdgpu > gpurun_out/${TAG:-r05}_update_in_situ.txt
python3 tools/update_in_situ.py 20000 1024 2>&1 | grep -v amdgpu >> gpurun_out/${TAG:-r05}_update_in_situ.txt
tail -3 gpurun_out/${TAG:-r05}_panel_chain_ab.txt
