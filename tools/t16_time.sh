#!/bin/bash
# developer aid (GPU box): the gemmx time table with the default library and with variant builds (build/variants/lib_<name>.so)
out=${1:-gpurun_out/t16}; shift
mkdir -p $out
timeout -k 10 200 python tools/gemmx_gpu_time.py bf16 "$@" > $out/default.txt 2>&1
for v in build/variants/lib_*.so; do
  n=$(basename $v .so)
  SKG_LIB=$PWD/$v timeout -k 10 200 python tools/gemmx_gpu_time.py bf16 "$@" > $out/$n.txt 2>&1
done
