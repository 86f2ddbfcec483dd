"""Developer aid (GPU box): per-parameter gradient errors of the fused fp32 step on a full-width training case, against the
oracle's autograd computed on THIS box and against the reference's gradient samples in the fixture, plus oracle-vs-fixture.
usage: debug_grad_case.py <case> [precision]"""
import os, sys
import numpy as np
import torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import cases, helpers, gpu_run
from collections import OrderedDict
prec = "fp32"
for name in sys.argv[1:]:
  try:
    case = cases.build_case(name)
    ograds, olosses = helpers.oracle_train_grads(case)
    head = gpu_run.build_head(case); head.fused_training = True; head.precision = prec
    det = gpu_run.to_cuda(case["detections"]); tg = gpu_run.to_cuda(case["targets"])
    feats = OrderedDict((k, case["feat3"].cuda()) for k in "0123")
    flat, grads = gpu_run._run_train(case, head, det, tg, feats, backward=True)
    want = helpers.load_golden(name)
    print("losses", {k: (float(flat[k]), float(want[k]), olosses[k]) for k in ("hoi_loss", "interactiveness_loss", "transH_loss")})
    rows = []
    for k, w in ograds.items():
        g = grads[k]
        scale = max(np.abs(w).max(), 1e-9)
        e_o = np.abs(g - w).max() / scale
        smp = want["grad.%s.sample" % k]; amax = max(float(want["grad.%s.absmax" % k]), 1e-9)
        e_r = np.abs(cases.grad_sample(torch.from_numpy(g).reshape(-1)).numpy() - smp).max() / amax
        e_or = np.abs(cases.grad_sample(torch.from_numpy(w).reshape(-1)).numpy() - smp).max() / amax
        rows.append((max(e_o, e_r), k, e_o, e_r, e_or, scale))
    rows = [r for r in rows if r[1] != "box_pair_head.adjacency.bias"]      # (exactly zero: rounding noise on every side)
    rows.sort(reverse=True)
    print("SUMMARY %-28s worst hip-oracle %.2e  hip-ref %.2e  oracle-ref %.2e" % (
        name, max(r[2] for r in rows), max(r[3] for r in rows), max(r[4] for r in rows)))
    for r in rows[:4]:
        print("%-60s hip-oracle %.2e  hip-ref %.2e  oracle-ref %.2e  scale %.2e" % (r[1], r[2], r[3], r[4], r[5]))
    k = "box_pair_head.spatial_head.0.weight"
    g, w = grads[k], ograds[k]
    d = np.abs(g - w) / max(np.abs(w).max(), 1e-9)
    print("per-column max err (hip-oracle) of", k, np.round(d.max(0) * 1e6, 1).tolist())
    for i in range(int(want["n_tables"])):
        key = "timg%d.spatial46" % i
        if key in flat and key in want:
            print(key, np.abs(flat[key] - want[key]).max())

  except Exception as e:                                  # noqa: BLE001
    print(name, "FAILED", type(e).__name__, e)
