cd $GRAFT_REPO_ROOT
O=gpurun_out/r03ar; mkdir -p $O
timeout -k 10 600 python tools/option_ab.py panel_square=0/panel_square=1,panel_square_rows=12288/panel_square=1,panel_square_rows=20480/panel_square=1,panel_square_rows=28672/panel_square=1,panel_square_rows=36864 - 50000 4 > $O/ab.log 2>&1; cat $O/ab.log
