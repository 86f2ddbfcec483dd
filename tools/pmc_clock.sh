#!/bin/bash
# developer aid (GPU box): shader clock + MFMA busy of the split GEMM loop for a list of variant libraries
R=$GRAFT_REPO_ROOT
export PYTHONPATH=$R
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  lib=$R/build/variants/lib_$v.so; [ "$v" = base ] && lib=$R/skghoi_amd/csrc/libskghoi_hip.so
  mkdir -p $R/gpurun_out/pmc_clk_$v
  SKG_LIB=$lib rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES -d $R/gpurun_out/pmc_clk_$v --output-format csv -- python3 $R/tools/gemm_split_one.py 51200 1024 1024 1 10 > $R/gpurun_out/pmc_clk_$v/log.txt 2>&1 || echo "$v failed"
  python3 - "$R/gpurun_out/pmc_clk_$v" "$v" <<'PY'
import csv, glob, sys, collections
d = collections.defaultdict(lambda: [0.0, 0, 0.0])
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "skg_gemm_kernel" not in r["Kernel_Name"]: continue
        e = d[r["Counter_Name"]]; e[0] += float(r["Counter_Value"]); e[1] += 1; e[2] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
g = d["GRBM_GUI_ACTIVE"]; m = d["SQ_VALU_MFMA_BUSY_CYCLES"]
if g[1]:
    cyc = g[0] / g[1] / 8; us = g[2] / g[1] / 1e3
    print("%-8s %.1f us  clock %.2f GHz  MFMA busy %.1f %%" % (sys.argv[2], us, cyc / us / 1e3, 100 * (m[0] / m[1]) / (cyc * 1024)))
PY
done
