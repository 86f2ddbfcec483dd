#!/bin/bash
# developer aid (GPU box): socket power and clocks while the split GEMM runs back to back
python tools/gemm_split_one.py 51200 1024 1024 1 6000 > gpurun_out/power_gemm.txt 2>&1 &
pid=$!
sleep 4
for i in 1 2 3; do rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk|mclk"; sleep 0.5; done
wait $pid
cat gpurun_out/power_gemm.txt
