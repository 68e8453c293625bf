cd $GRAFT_REPO_ROOT
O=gpurun_out/r03am; mkdir -p $O
timeout -k 10 500 python tools/option_ab.py panel_square=0,update_reserve=0/panel_square=1,update_reserve=0/panel_square=1,update_reserve=8/panel_square=1,update_reserve=16/panel_square=1,update_reserve=32/panel_square=0,update_reserve=16 - 8000,12000,20000,50000 4 > $O/ab.log 2>&1; cat $O/ab.log
