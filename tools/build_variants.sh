#!/bin/bash
# developer aid: variant builds of skg_gemm.hip -> build/variants/lib_<name>.so (SKG_LIB selects one at run time)
# usage: tools/build_variants.sh name="-DSKG_XABL=1 -DSKG_XNST=3" ...
set -e
cd "$(dirname "$0")/../skghoi_amd/csrc"
mkdir -p ../../build/variants
for spec in "$@"; do
  name=${spec%%=*}; flags=${spec#*=}
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -I../../include $flags -c skg_gemm.hip -o /tmp/skg_gemm_$name.o
  hipcc --offload-arch=gfx950 -shared -fPIC -pthread -o ../../build/variants/lib_$name.so /tmp/skg_gemm_$name.o $(ls *.o | grep -v '^skg_gemm\.o$')
done
