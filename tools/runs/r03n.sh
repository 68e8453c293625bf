set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03n; mkdir -p $O
python -m pytest tests/test_gpu_primitives.py -x -q -m gpu -k "potri or gemm" 2>&1 | tail -5
python -m pytest tests/test_gpu_facade.py -x -q -m gpu 2>&1 | tail -5
echo "== new potri" > $O/c3.log; python tools/c3_grad_timing.py 50000 >> $O/c3.log 2>&1
echo "== old potri" >> $O/c3.log; FVGP_POTRI_KMINOR=0 python tools/c3_grad_timing.py 50000 >> $O/c3.log 2>&1
cat $O/c3.log
