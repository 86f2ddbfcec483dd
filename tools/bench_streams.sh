#!/bin/bash
# developer aid: chunk / stream sweep of the inference bench
for cfg in "128 1" "128 2" "64 2" "64 4" "32 4"; do set -- $cfg
  echo -n "chunk=$1 streams=$2  "
  python bench.py --steps 8 --warmup 2 --no-cpu-baseline --chunk $1 --streams $2 2>&1 | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("img/s", d["value"], "ms/step", d["ms_per_step"])'
done
