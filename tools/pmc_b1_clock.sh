#!/bin/bash
# developer aid (GPU box): shader clock and MFMA busy share per kernel of the single-image eval forward (PMC pass; kernels
# run one at a time under the counter collection)            usage: pmc_b1_clock.sh [B=1]
B=${1:-1}
R=$GRAFT_REPO_ROOT
export PYTHONPATH=$R
OUT=$R/gpurun_out/pmc_b${B}_clock
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $OUT/run --output-format csv -- python3 $R/tools/small_batch_loop.py $B 60 > $OUT/log.txt 2>&1 || echo "pmc run failed"
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
d = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0, 0.0]))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = (r["Kernel_Name"].split("(")[0][-40:], r.get("Grid_Size", ""))
        e = d[k][r["Counter_Name"]]; e[0] += float(r["Counter_Value"]); e[1] += 1; e[2] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
for k, c in sorted(d.items(), key=lambda kv: -kv[1]["GRBM_GUI_ACTIVE"][2]):
    g = c["GRBM_GUI_ACTIVE"]; m = c["SQ_VALU_MFMA_BUSY_CYCLES"]
    if not g[1]: continue
    cyc = g[0] / g[1] / 8; us = g[2] / g[1] / 1e3
    print("%-42s grid %-8s n %4d  %7.1f us  clock %.2f GHz  MFMA busy %5.1f %% of CU cycles" % (k[0], k[1], g[1], us, cyc / us / 1e3, 100 * (m[0] / max(m[1], 1)) / (cyc * 1024) if cyc else 0))
PY
rm -rf $OUT/run
