set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03b
python tools/peak_bench.py > gpurun_out/r03b/peak.log 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $GRAFT_REPO_ROOT/gpurun_out/r03b/counters.txt 2>&1 || true
cd $GRAFT_REPO_ROOT
cat gpurun_out/r03b/peak.log
grep -c . gpurun_out/r03b/counters.txt
