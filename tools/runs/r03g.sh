set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03g; mkdir -p $O
python tools/gemm_probe2.py 512 520 522 528 530 514 512 > $O/probe3.log 2>&1
cat $O/probe3.log
