#!/bin/bash
# Build A/B variants of libgnf_hip.so (developer tool): tools/build_variants.sh name "-DFLAGS" ...
set -e; fail=0
cd "$(dirname "$0")/.."
PKG="graph-normalizing-flows_amd"
mkdir -p "$PKG/variants"
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -I include -I "$PKG/csrc" $flags \
     "$PKG"/csrc/*.hip -o "$PKG/variants/libgnf_$name.so" 2>&1 | grep -E "error" | head -3 &
done
wait
ls -la "$PKG/variants"
