"""Developer aid (GPU box): max |logit - reference golden| of the HIP head per case, exact fp32 path vs fp16x2 path."""
import sys, os
sys.path.insert(0, "."); sys.path.insert(0, "tests"); sys.dont_write_bytecode = True
import numpy as np
import cases, gpu_run, helpers
print("%-10s %-7s %12s %12s %12s %12s" % ("case", "path", "max|logit|", "err logits_p", "err logits_s", "err scores"))
for name in [c for c in cases.EVAL_CASES if c != "nanbox"]:
    case = cases.build_case(name)
    want = helpers.load_golden(name)
    if "logits_p" not in want:
        continue
    for prec in ("fp32", "fp16x2"):
        gpu_run.PRECISION = prec
        got = gpu_run.run_head(case)
        es = [np.abs(got[k] - want[k]).max() for k in want if k.endswith(".scores") and k.startswith("res") and want[k].size and k in got]
        print("%-10s %-7s %12.3g %12.2e %12.2e %12.2e" % (name, prec, np.abs(want["logits_p"]).max(),
              np.abs(got["logits_p"] - want["logits_p"]).max(), np.abs(got["logits_s"] - want["logits_s"]).max(), max(es) if es else 0.0))
