#!/bin/bash
# developer aid: batch-size sweep of the inference bench
for b in 1 4 16 64 256 512; do
  python bench.py --batch $b --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("batch", d["config"]["batch_per_gpu"], "img/s", d["value"], "ms/step", d["ms_per_step"], "gemm TF", d["roofline"]["all_gemm_tflops"])'
done
