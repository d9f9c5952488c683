#!/bin/bash
# Developer tool: link a variant of libgnf_hip.so that differs only in gnf_fused_big.hip's compile flags
# (the other objects come from the regular build's csrc/obj):  tools/build_big_variant.sh name "-DFLAGS"
set -e
cd "$(dirname "$0")/.."
PKG="graph-normalizing-flows_amd"
mkdir -p "$PKG/variants"
name=$1; flags=$2
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I include -I "$PKG/csrc" $flags -c "$PKG/csrc/gnf_fused_big.hip" -o "$PKG/variants/big_$name.o"
objs=$(ls "$PKG"/csrc/obj/*.o | grep -v gnf_fused_big.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs "$PKG/variants/big_$name.o" -o "$PKG/variants/libgnf_$name.so"
ls -la "$PKG/variants/libgnf_$name.so"
