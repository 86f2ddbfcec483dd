#!/bin/bash
# developer aid: small-batch bench sweep with/without per-launch event timing
for b in 16 64 256; do for t in "" "--no-gemm-timer"; do
  echo -n "timer=[$t] "
  python bench.py --batch $b --steps 20 --warmup 5 --no-cpu-baseline $t 2>&1 | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["config"]["batch_per_gpu"], d["value"], d["ms_per_step"])'
done; done
