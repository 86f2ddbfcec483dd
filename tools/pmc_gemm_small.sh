#!/bin/bash
# GPU box: SQ counters of the exact fp32 GEMM on a SMALL shape (the latency loop).  usage: pmc_gemm_small.sh tag M N K
tag=$1; M=$2; N=$3; K=$4
R=$GRAFT_REPO_ROOT
export PYTHONPATH=$R
OUT=$R/gpurun_out/pmc_gs_$tag
mkdir -p $OUT
cd $R && export TMPDIR=/tmp
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES" \
           "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD"; do
  i=$((i+1))
  EXACT=1 rocprofv3 --pmc $set -d $OUT/p$i --output-format csv -- python3 $R/tools/gemm_split_one.py $M $N $K 1 40 > $OUT/log$i.txt 2>&1 || echo "pass $i failed"
done
python3 - "$OUT" "$tag M=$M N=$N K=$K" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
d = collections.defaultdict(lambda: [0.0, 0, 0.0])
for f in glob.glob("%s/**/*counter_collection.csv" % out, recursive=True):
    for r in csv.DictReader(open(f)):
        if "skg_gemm_kernel" not in r["Kernel_Name"]:
            continue
        e = d[r["Counter_Name"]]
        e[0] += float(r["Counter_Value"]); e[1] += 1; e[2] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
g = d["GRBM_GUI_ACTIVE"]; cyc = g[0] / g[1] / 8; us = g[2] / g[1] / 1e3
print("== %s: %.1f us per launch, GUI_ACTIVE/8 = %.0f cycles (%.2f GHz if the launch spans it)" % (sys.argv[2], us, cyc, cyc / us / 1e3))
for k in sorted(d):
    v = d[k]
    print("   %-28s %16.0f per launch   (%6.2f %% of SIMD-cycles)" % (k, v[0] / v[1], 100 * (v[0] / v[1]) / (cyc * 1024)))
PY
