#!/bin/bash
# GPU box: SQ counters of the two main GEMM loops (exact fp32 MFMA, fp16x2 split operands) at the bench's dominant shape
# (M=102400, N=K=1024, BIAS_RELU) -> gpurun_out/pmc_<tag>/summary.txt (copy into profiles/).  Counter passes only
# (--pmc without any trace domain), one counter set per pass.
tag=${1:-r02}
R=$GRAFT_REPO_ROOT
export PYTHONPATH=$R
OUT=$R/gpurun_out/pmc_$tag
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for mode in exact fp16x2; do
  i=0
  for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES" \
             "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
             "SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
    i=$((i+1))
    if [ $mode = exact ]; then export EXACT=1; else unset EXACT; fi
    rocprofv3 --pmc $set -d $OUT/$mode/p$i --output-format csv -- python3 $R/tools/gemm_split_one.py 102400 1024 1024 1 12 > $OUT/${mode}_log$i.txt 2>&1 || echo "$mode pass $i failed"
  done
done
python3 - "$OUT" <<'PY' > $OUT/summary.txt
import csv, glob, sys, collections
out = sys.argv[1]
print("# rocprofv3 --pmc passes (counters only) of tools/gemm_split_one.py 102400 1024 1024 1 12: the bench's dominant GEMM shape")
print("# MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs); clock = GRBM_GUI_ACTIVE / 8 / duration")
for mode in ("exact", "fp16x2"):
    d = collections.defaultdict(lambda: [0.0, 0, 0.0])
    for f in glob.glob("%s/%s/**/*counter_collection.csv" % (out, mode), recursive=True):
        for r in csv.DictReader(open(f)):
            if "skg_gemm_kernel" not in r["Kernel_Name"]:
                continue
            e = d[r["Counter_Name"]]
            e[0] += float(r["Counter_Value"]); e[1] += 1; e[2] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    if not d:
        print(mode, "no data"); continue
    g = d["GRBM_GUI_ACTIVE"]; cyc = g[0] / g[1] / 8; us = g[2] / g[1] / 1e3
    print("\n== %s loop: %.1f us per launch, shader clock %.2f GHz" % (mode, us, cyc / us / 1e3))
    for k in sorted(d):
        v = d[k]
        print("   %-28s %16.0f per launch   (%6.2f %% of SIMD-cycles)" % (k, v[0] / v[1], 100 * (v[0] / v[1]) / (cyc * 1024)))
PY
cat $OUT/summary.txt
