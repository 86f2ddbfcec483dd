"""Developer aid: per-key max abs error of the HIP head vs golden for one case (run on the GPU box)."""
import sys
import numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import cases, gpu_run, helpers

name = sys.argv[1]
case = cases.build_case(name)
got = gpu_run.run_head(case, reference_quirks=(len(sys.argv) < 3))
want = helpers.load_golden(name)
for k, w in want.items():
    if k not in got:
        print("%-28s missing" % k); continue
    g = got[k]
    if g.size != w.size:
        print("%-28s shape %s vs %s" % (k, g.shape, w.shape)); continue
    g = g.reshape(w.shape)
    if w.dtype.kind in "iub":
        print("%-28s int equal=%s" % (k, np.array_equal(g, w)))
    elif w.size:
        w2 = np.nan_to_num(w)
        print("%-28s max|err| %.3e  max|want| %.3e" % (k, np.abs(g - w2).max(), np.abs(w2).max()))
