#!/bin/bash
# developer aid: variant builds of skg_gemm_x.hip -> build/variants/lib_<name>.so (SKG_LIB selects one at run time)
# usage: tools/build_gemmx_variants.sh name="-DSKG_XPROBE_NOLOOP" name2="" ...      (run `make` in csrc first: the other objects are reused)
set -e
cd "$(dirname "$0")/../skghoi_amd/csrc"
mkdir -p ../../build/variants
others=$(ls *.o | grep -v skg_gemm_x.o)
for spec in "$@"; do
  name=${spec%%=*}; flags=${spec#*=}
  src=${SKG_XSRC:-skg_gemm_x.hip}
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -I../../include -I. $flags -c $src -o /tmp/skg_gemm_x_$name.o
  hipcc --offload-arch=gfx950 -shared -fPIC -o ../../build/variants/lib_$name.so /tmp/skg_gemm_x_$name.o $others
done
