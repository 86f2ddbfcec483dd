#!/bin/bash
# GPU box: the profiles the bench line is judged against -> gpurun_out/prof_<tag>/ (copy the summaries into profiles/).
#   1. rocprofv3 --kernel-trace --stats of the default bench command (per-kernel average durations)
#   2. two separate --pmc passes (FETCH_SIZE, WRITE_SIZE) of a short bench run: HBM bytes per launch
tag=${1:-r02}
prec=${2:-fp32}
R=$GRAFT_REPO_ROOT
export PYTHONPATH=$R
OUT=$R/gpurun_out/prof_$tag
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/stats --output-format csv -- python3 $R/bench.py --precision $prec --steps 5 --warmup 2 --no-cpu-baseline --no-legs > $OUT/bench_under_rocprof.json 2> $OUT/stats.log || echo "stats pass failed"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c -d $OUT/pmc_$c --output-format csv -- python3 $R/bench.py --precision $prec --steps 2 --warmup 1 --no-cpu-baseline --no-legs --no-gemm-timer > $OUT/pmc_$c.log 2>&1 || echo "$c pass failed"
done
python3 $R/tools/traffic_summary.py $OUT $tag
