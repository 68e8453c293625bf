cd $GRAFT_REPO_ROOT
V=$GRAFT_REPO_ROOT/fvgp_amd/csrc/variants/c4d2f/libfvgp_hip.so
C=$GRAFT_REPO_ROOT/fvgp_amd/csrc/libfvgp_hip.so
for i in 1 2; do for lib in $C $V; do echo "lib=$lib"; FVGP_HIP_LIB=$lib python tools/eval_trace.py run 50000 2>&1 | grep "^N" | awk '{print $1,$2,$(NF-2),$(NF-1),$NF}'; done; done
