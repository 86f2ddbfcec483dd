#!/bin/bash
# GPU box: the training step's profiles for DESIGN 9 -> gpurun_out/train_<tag>/  (usage: profile_train_round.sh tag)
#   <prec>_kernel_stats.csv  rocprofv3 --kernel-trace --stats of tools/train_loop.py <prec> 1 30 (38 steps with warm-up)
#   <prec>_timeline.txt      per-launch timeline of the last step (tools/step_timeline.py)
#   <prec>_bench.json        bench.py --mode train (not profiled)        cpu_probe.txt  wall vs host-thread CPU time per step
tag=${1:-r03}
R=$GRAFT_REPO_ROOT
export PYTHONPATH=$R
OUT=$R/gpurun_out/train_$tag
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for p in bf16 fp32; do
  rocprofv3 --kernel-trace --stats -d $OUT/trace_$p --output-format csv -- python3 $R/tools/train_loop.py $p 1 30 > $OUT/${p}_loop.txt 2> $OUT/${p}_log.txt || echo "$p trace failed"
  cp $(find $OUT/trace_$p -name "*kernel_stats.csv" | head -1) $OUT/${p}_kernel_stats.csv
  python3 $R/tools/step_timeline.py $OUT/trace_$p > $OUT/${p}_timeline.txt
  rm -rf $OUT/trace_$p
done
cd $R
for p in bf16 fp32; do
  python3 bench.py --mode train --precision $p --batch 4 --steps 100 --warmup 10 2>/dev/null > $OUT/${p}_bench.json
done
python3 tools/train_cpu_probe.py bf16 4 2>&1 | grep -v amdgpu.ids > $OUT/cpu_probe.txt
python3 tools/train_cpu_probe.py fp32 4 2>&1 | grep -v amdgpu.ids >> $OUT/cpu_probe.txt
