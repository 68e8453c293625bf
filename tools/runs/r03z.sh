cd $GRAFT_REPO_ROOT
O=gpurun_out/r03z; mkdir -p $O
for r in 0 16 32 48; do echo "reserve=$r" >> $O/em.log; FVGP_UPDATE_RESERVE=$r FVGP_RESERVE_ROWS=-1 timeout -k 10 200 python tools/shard_emulate.py --world 8 --n 50000 2>&1 | grep "^world" >> $O/em.log; done
for r in 0 32; do echo "world4 reserve=$r" >> $O/em.log; FVGP_UPDATE_RESERVE=$r FVGP_RESERVE_ROWS=-1 timeout -k 10 200 python tools/shard_emulate.py --world 4 --n 50000 2>&1 | grep "^world" >> $O/em.log; done
cat $O/em.log
