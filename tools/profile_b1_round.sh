#!/bin/bash
# GPU box: the single-image eval forward's profiles -> gpurun_out/b1_<tag>/  (usage: profile_b1_round.sh tag [B=1])
#   kernel_stats.csv   rocprofv3 --kernel-trace --stats of tools/small_batch_loop.py B 200
#   timeline.txt       per-launch timeline of the last forward (tools/replay_timeline.py)
#   host.txt           host cProfile + wall time (tools/b1_profile.py)
tag=${1:-r04}; B=${2:-1}
R=$GRAFT_REPO_ROOT
export PYTHONPATH=$R
OUT=$R/gpurun_out/b${B}_$tag
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/trace --output-format csv -- python3 $R/tools/small_batch_loop.py $B 200 > $OUT/loop.txt 2> $OUT/log.txt || echo "trace failed"
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
python3 $R/tools/replay_timeline.py $OUT/trace > $OUT/timeline.txt
rm -rf $OUT/trace
cd $R
python3 tools/b1_profile.py > $OUT/host.txt 2>&1
