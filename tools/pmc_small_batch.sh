#!/bin/bash
# GPU box: shader clock and MFMA-busy of the kernels of a B-image graph replay (usage: pmc_small_batch.sh B)
B=${1:-1}
R=$GRAFT_REPO_ROOT
export PYTHONPATH=$R
OUT=$R/gpurun_out/pmc_small_b$B
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES -d $OUT/p1 --output-format csv -- python3 $R/tools/small_batch_loop.py $B 60 > $OUT/log1.txt 2>&1 || echo "pass failed"
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
d = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0, 0.0]))
for f in glob.glob("%s/**/*counter_collection.csv" % out, recursive=True):
    for r in csv.DictReader(open(f)):
        k = (r["Kernel_Name"].split("(")[0][-42:], r["Grid_Size"])
        e = d[k][r["Counter_Name"]]
        e[0] += float(r["Counter_Value"]); e[1] += 1; e[2] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
print("%-44s %9s %6s %8s %9s %8s" % ("kernel", "grid", "n", "us", "clk GHz", "MFMA %"))
for k, c in sorted(d.items(), key=lambda kv: -kv[1]["GRBM_GUI_ACTIVE"][2]):
    g = c["GRBM_GUI_ACTIVE"]
    if g[1] == 0: continue
    cyc = g[0] / g[1] / 8; us = g[2] / g[1] / 1e3
    m = c["SQ_VALU_MFMA_BUSY_CYCLES"]
    mf = 100 * (m[0] / max(m[1], 1)) / (cyc * 1024) if cyc else 0
    print("%-44s %9s %6d %8.1f %9.2f %8.1f" % (k[0], k[1], g[1], us, cyc / us / 1e3, mf))
PY
