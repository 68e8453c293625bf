set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03j; mkdir -p $O
python tools/gemm_ab.py 768 2816 > $O/gemm_ab.log 2>&1
cat $O/gemm_ab.log
