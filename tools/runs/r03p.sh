cd $GRAFT_REPO_ROOT
O=gpurun_out/r03p; mkdir -p $O
for v in 512 1200 2500; do echo "small_tile_max_update=$v" >> $O/n8k.log; FVGP_SMALL_TILE_MAX_UPDATE=$v python tools/eval_trace.py run 8000 2>&1 | grep "^N" >> $O/n8k.log; done
for v in 512 2500; do echo "small_tile_max_update=$v" >> $O/n8k.log; FVGP_SMALL_TILE_MAX_UPDATE=$v python tools/eval_trace.py run 20000 2>&1 | grep "^N" >> $O/n8k.log; done
cat $O/n8k.log
