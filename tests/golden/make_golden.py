"""BUILD-CONTAINER-ONLY generator of tests/golden/<case>.npz.

Runs the reference's own, unmodified interaction head (imported from /root/reference through oracle/ref_import.py)
on the seeded cases of tests/cases.py and stores its OUTPUTS (and the TransH tables it drew, so fixtures do not depend
on the torch RNG of the box that replays them).  Inputs and weights are regenerated from seeds (skghoi_amd/synth.py),
never stored.  Usage:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py [case ...]
"""
import os
import sys
from collections import OrderedDict

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np
import torch

import cases
from oracle import ref_import
from skghoi_amd import synth

OUT = os.path.dirname(os.path.abspath(__file__))


class StableTies:
    """The reference orders the kept detections with `torch.argsort(scores, descending=True)` (HEAD:131), which is not
    a stable sort: with more than a handful of elements the order of EQUAL scores is whatever the installed torch build
    and device happen to produce (the ground-truth boxes appended in training all score 1.0, HEAD:113 -- every training
    batch has ties).  The fixtures pin the one order that is defined -- ties in ascending input index -- by making that
    call stable while the reference runs; nothing in the reference's files is touched.  (The small fixtures are
    bit-identical with and without this: torch's sort is stable at their sizes.)"""

    def __enter__(self):
        self._orig = torch.argsort

        def argsort(x, dim=-1, descending=False, stable=False):
            return self._orig(x, dim=dim, descending=descending, stable=True)
        torch.argsort = argsort
        return self

    def __exit__(self, *exc):
        torch.argsort = self._orig
        return False


def run_reference(case):
    """Returns a flat {key: ndarray} of the reference's outputs for `case`."""
    ref = ref_import.load_reference()
    cfg = case["cfg"]; K = cfg["K"]
    head = ref_import.build_reference_head(K, cfg["human_idx"], case["o2v"], case["C"], case["p"], case["max_human"],
                                           case["max_object"], num_iter=case["num_iter"],
                                           box_nms_thresh=case["box_nms_thresh"],
                                           box_score_thresh=case["box_score_thresh"])
    head.load_state_dict(synth.make_state_dict(K, case["C"], case["p"], seed=case["weight_seed"]))
    head.train(case["training"])
    gh = head.box_pair_head
    out = {}
    cap = dict(spatial=[], adjacency=[], norm_h=[], norm_o=[])

    orig_sp = ref.compute_spatial_ratio_encodings

    def sp_wrap(*a, **k):
        r = orig_sp(*a, **k); cap["spatial"].append(r.detach().clone()); return r

    ref.compute_spatial_ratio_encodings = sp_wrap
    hooks = [gh.adjacency.register_forward_hook(lambda m, i, o: cap["adjacency"].append(o.detach().clone())),
             gh.norm_h.register_forward_hook(lambda m, i, o: cap["norm_h"].append(o.detach().clone())),
             gh.norm_o.register_forward_hook(lambda m, i, o: cap["norm_o"].append(o.detach().clone()))]
    want_grads = case["name"].partition("@")[0] in cases.FULL_TRAIN_CASES    # gradient samples from the reference's own autograd
    try:
        with torch.set_grad_enabled(want_grads), StableTies():
            det = head.preprocess(case["detections"], case["targets"], append_gt=case["training"])
            n_rows = sum(len(d["boxes"]) for d in det)
            pooled = cases.pooled_for(case, n_rows)
            head.box_roi_pool.pooled = pooled
            feats = OrderedDict((k, case["feat3"]) for k in "0123")
            torch.manual_seed(case["rng_seed"])
            with ref_import.TransHCapture() as tcap:
                if not case["training"]:
                    logits = {}
                    h1 = head.box_pair_predictor.register_forward_hook(lambda m, i, o: logits.__setitem__("p", o))
                    h2 = head.box_pair_suppressor.register_forward_hook(lambda m, i, o: logits.__setitem__("s", o))
                    h3 = head.box_pair_predictor.register_forward_pre_hook(lambda m, i: logits.__setitem__("pf", i[0]))
                    results = head(feats, case["detections"], case["shapes"], case["targets"])
                    h1.remove(); h2.remove(); h3.remove()
                    out["logits_p"] = logits["p"]; out["logits_s"] = logits["s"]; out["pair_features"] = logits["pf"]
                else:
                    # InteractionHead.forward cannot run in training as committed: drive its pieces.
                    coords = [d["boxes"] for d in det]; labels = [d["labels"] for d in det]
                    scores = [d["scores"] for d in det]
                    (pf, bh, bo, oc, lab, prior, pos, neg, he, te, re, rne) = gh(
                        feats, case["shapes"], pooled, coords, labels, scores, case["targets"])
                    pf = torch.cat(pf)
                    lp = head.box_pair_predictor(pf); ls = head.box_pair_suppressor(pf)
                    results = head.postprocess(lp, ls, prior, bh, bo, oc, lab)
                    out["logits_p"] = lp; out["logits_s"] = ls; out["pair_features"] = pf
                    out["hoi_loss"] = head.compute_interaction_classification_loss(results)
                    out["interactiveness_loss"] = head.compute_interactiveness_loss(results)
                    # TransH term, intended semantics: heads/NegativeSampling one-arg forward + upstream MarginLoss
                    from NegativeSampling import NegativeSampling
                    from OpenKE.openke.module.loss.MarginLoss import MarginLoss as UpstreamMarginLoss
                    ul = torch.cat([r["unary_labels"] for r in results])
                    n_p = len(torch.nonzero(ul))
                    ns = NegativeSampling(loss=UpstreamMarginLoss(margin=1), batch_size=256)
                    out["transH_loss"] = ns(torch.cat([torch.cat(pos), torch.cat(neg)])) / n_p
                    for i in range(len(pos)):
                        out["timg%d.pos_scores" % i] = pos[i]; out["timg%d.neg_scores" % i] = neg[i]
                        out["timg%d.head_ent" % i] = he[i]; out["timg%d.tail_ent" % i] = te[i]
                        out["timg%d.rel" % i] = re[i]; out["timg%d.rel_norm" % i] = rne[i]
                    if want_grads:
                        head.zero_grad()
                        (out["hoi_loss"] + out["interactiveness_loss"] + out["transH_loss"]).backward()
                        for k, prm in head.named_parameters():
                            if prm.grad is None:
                                continue
                            g = prm.grad.detach().reshape(-1)
                            out["grad." + k + ".sample"] = cases.grad_sample(g).clone()
                            out["grad." + k + ".absmax"] = g.abs().max()
                            out["grad." + k + ".norm"] = g.double().norm()
    finally:
        ref.compute_spatial_ratio_encodings = orig_sp
        for h in hooks:
            h.remove()
    for b, d in enumerate(det):
        out["pre%d.boxes" % b] = d["boxes"]; out["pre%d.labels" % b] = d["labels"]; out["pre%d.scores" % b] = d["scores"]
    for b, r in enumerate(results):
        for k, v in r.items():
            out["res%d.%s" % (b, k)] = v
    out["n_results"] = torch.tensor(len(results))
    # per processed (non-skipped) image intermediates
    n_iter = max(case["num_iter"], 1)
    for i, (e, r, n) in enumerate(tcap.tables):
        out["timg%d.ent" % i] = e; out["timg%d.rel_table" % i] = r; out["timg%d.norm_table" % i] = n
        out["timg%d.spatial46" % i] = cap["spatial"][i]
        if case["num_iter"] > 0:
            out["timg%d.adjacency" % i] = cap["adjacency"][(i + 1) * n_iter - 1]
            out["timg%d.h_node" % i] = cap["norm_h"][(i + 1) * n_iter - 1]
            out["timg%d.node" % i] = cap["norm_o"][(i + 1) * n_iter - 1]
    out["n_tables"] = torch.tensor(len(tcap.tables))
    return {k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in out.items()}


def main(names):
    for name in names:
        case = cases.build_case(name)
        flat = run_reference(case)
        if name.partition("@")[0] in cases.OUTPUT_ONLY:
            # output-only fixture: drop the bulky intermediates, keep what pins the result
            keep = ("logits_p", "logits_s", "n_results", "n_tables")
            keep += ("hoi_loss", "interactiveness_loss", "transH_loss")
            flat = {k: v for k, v in flat.items()
                    if k in keep or k.endswith((".ent", ".scores", ".index", ".prediction", ".labels", ".adjacency",
                                                ".weights", ".pos_scores", ".neg_scores", ".unary_labels"))
                    or k.startswith(("pre", "grad."))}
        path = os.path.join(OUT, name + ".npz")
        np.savez_compressed(path, **flat)
        print("%-12s %4d arrays  %8.1f KB" % (name, len(flat), os.path.getsize(path) / 1024))


if __name__ == "__main__":
    main(sys.argv[1:] or cases.ALL_CASES)
