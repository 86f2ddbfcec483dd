#!/bin/bash
# GPU box: per-kernel statistics of the TRAINING step -> gpurun_out/prof_train_<tag>/  (usage: profile_train.sh tag precision batch)
tag=${1:-r02}
prec=${2:-fp32}
batch=${3:-4}
R=$GRAFT_REPO_ROOT
export PYTHONPATH=$R
OUT=$R/gpurun_out/prof_train_${tag}_${prec}_b${batch}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/stats --output-format csv -- python3 $R/bench.py --mode train --precision $prec --batch $batch --steps 20 --warmup 5 > $OUT/bench_under_rocprof.json 2> $OUT/stats.log || echo "stats pass failed"
f=$(find $OUT/stats -name "*kernel_stats.csv" | head -1)
cp $f $OUT/kernel_stats.csv
