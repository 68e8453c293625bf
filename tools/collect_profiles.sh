#!/bin/bash
# Collect the rocprofv3 evidence for one round (run on the GPU box from the repo root): TAG=r02 tools/collect_profiles.sh
# Kernel trace + stats of the headline command, the two PMC passes for the trailing update's traffic and for what the covariance
# assembly (kmat_kernel) wrote (tools/summarize_profiles.py reads both kernels out of the same passes; separate runs:
# FETCH_SIZE and WRITE_SIZE do not fit one pass; no tracing next to --pmc), and kernel stats of the C2 / C3 / C5 paths.
set -o pipefail
TAG=${TAG:-r06}
OUT=gpurun_out/${TAG}_profiles
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-configs > $OUT/bench_n50k_steps5.json 2> $OUT/bench.err || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-configs > $OUT/bench_n50k_steps5_under_rocprof.json 2>> $OUT/bench.err || exit 1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o fetch -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-configs > /dev/null 2>> $OUT/bench.err || exit 1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o write -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-configs > /dev/null 2>> $OUT/bench.err || exit 1
# L2 hit rate of the trailing update (a third pass of its own: TCC slots): TCC_HIT_sum / (TCC_HIT_sum + TCC_MISS_sum)
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_l2 -o l2 -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-configs > /dev/null 2>> $OUT/bench.err || echo "L2 counters not collected" >> $OUT/bench.err
for c in C2 C3 C5; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$c -o $c -- python3 tools/config_profile.py $c > /dev/null 2>> $OUT/bench.err || exit 1
done
ls -R $OUT | head -50
