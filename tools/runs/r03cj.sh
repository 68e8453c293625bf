cd $GRAFT_REPO_ROOT
O=gpurun_out/r03cj; mkdir -p $O
FVGP_PANEL_SQUARE=2 FVGP_UPDATE_RESERVE=4 FVGP_RESERVE_ROWS=-1 timeout -k 10 300 python -m pytest tests/test_gpu_primitives.py -x -q -m gpu -k "potrf or loglik" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
timeout -k 10 800 python tools/option_ab.py panel_square=0,update_reserve=0,reserve_rows=0/panel_square=2,update_reserve=4,reserve_rows=-1/panel_square=2,update_reserve=8,reserve_rows=-1/panel_square=2,update_reserve=0,reserve_rows=0/panel_square=0,update_reserve=4,reserve_rows=-1 - 12000,20000,50000 4 > $O/ab.log 2>&1; cat $O/ab.log
