#!/bin/bash
# GPU box: per-launch timelines of the bf16 training step with the split-K reduction inside the product launches (1) and in a
# second launch (0) -> gpurun_out/<tag>/inl{0,1}_timeline.txt   (usage: profile_inlaunch_ab.sh tag [precisions])
tag=${1:-r5ab}
precs=${2:-bf16}
R=$GRAFT_REPO_ROOT
export PYTHONPATH=$R
OUT=$R/gpurun_out/$tag
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for p in $precs; do
for v in 1 0; do
  export SKG_INLAUNCH_REDUCE=$v
  rocprofv3 --kernel-trace --stats -d $OUT/trace_$p$v --output-format csv -- python3 $R/tools/train_loop.py $p 1 30 > $OUT/${p}_inl${v}_loop.txt 2> $OUT/${p}_inl${v}_log.txt || echo "$p $v trace failed"
  cp $(find $OUT/trace_$p$v -name "*kernel_stats.csv" | head -1) $OUT/${p}_inl${v}_kernel_stats.csv
  python3 $R/tools/step_timeline.py $OUT/trace_$p$v > $OUT/${p}_inl${v}_timeline.txt
  rm -rf $OUT/trace_$p$v
done
done
