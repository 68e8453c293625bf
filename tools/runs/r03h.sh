set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03h; mkdir -p $O
python tools/gemm_ab.py 0 512 768 > $O/gemm_ab.log 2>&1
cat $O/gemm_ab.log
