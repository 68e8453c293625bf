#!/bin/bash
# Build a variant of libfvgp_hip.so with extra compiler flags into fvgp_amd/csrc/variants/<name>/ (A/B runs: FVGP_HIP_LIB=<path>).
#   tools/build_variant.sh pipe -DFVGP_GEMM_PIPE_DEFAULT=1
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/fvgp_amd/csrc/variants/$name
mkdir -p "$out"
for f in gemm kmat leaf chain solve api dist ipc; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function "$@" -c "$root/fvgp_amd/csrc/$f.hip" -o "$out/$f.o" &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$out/libfvgp_hip.so" "$out"/*.o -ldl
echo "$out/libfvgp_hip.so"
