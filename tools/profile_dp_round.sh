#!/bin/bash
# GPU box: kernel timelines of the bf16 batch-4 training step, single process and on the data-parallel route at world size 1
# over RCCL -> gpurun_out/dp_<tag>/{single,dp}_timeline.txt (+ kernel stats)      usage: profile_dp_round.sh tag
tag=${1:-r04}
R=$GRAFT_REPO_ROOT
export PYTHONPATH=$R
OUT=$R/gpurun_out/dp_$tag
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for m in single dp; do
  extra=""; [ $m = dp ] && extra="--dp-world1"
  rocprofv3 --kernel-trace --stats -d $OUT/trace_$m --output-format csv -- python3 $R/bench.py --mode train --precision bf16 --steps 40 --warmup 8 $extra > $OUT/${m}_bench.txt 2> $OUT/${m}_log.txt || echo "$m trace failed"
  cp $(find $OUT/trace_$m -name "*kernel_stats.csv" | head -1) $OUT/${m}_kernel_stats.csv
  python3 $R/tools/step_timeline.py $OUT/trace_$m > $OUT/${m}_timeline.txt
  rm -rf $OUT/trace_$m
done
