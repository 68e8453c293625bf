#!/bin/bash
# Size fuzz + bit-identity soak of the final binary (GPU box, from the repo root): writes gpurun_out/r04_fuzz_sizes.txt, r04_chain_soak.txt
set -o pipefail
timeout -k 10 900 python3 tools/fuzz_sizes.py 26 2>&1 | grep -v amdgpu > gpurun_out/r04_fuzz_sizes.txt
tail -3 gpurun_out/r04_fuzz_sizes.txt
{
  echo "# python tools/chain_soak.py N reps: the resident panel kernel's hand-offs under look-ahead (and, at N = 4096, both one-launch sweeps), every bit of L and alpha compared with the first run"
  for c in "12000 60" "8000 100" "4096 100" "20000 12"; do
    set -- $c
    timeout -k 10 300 python3 tools/chain_soak.py $1 $2 2>&1 | grep "^N"
  done
} > gpurun_out/r04_chain_soak.txt
cat gpurun_out/r04_chain_soak.txt
