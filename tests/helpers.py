"""Shared helpers for the oracle / parity tests."""
import os

import numpy as np
import torch

import cases
from oracle import skg_oracle as O
from skghoi_amd import synth

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN_DIR, name + ".npz")))


def golden_tables(g):
    n = int(g["n_tables"])
    out = []
    for i in range(n):
        ent = torch.from_numpy(g["timg%d.ent" % i])
        rel = torch.from_numpy(g["timg%d.rel_table" % i]) if "timg%d.rel_table" % i in g else None
        nrm = torch.from_numpy(g["timg%d.norm_table" % i]) if "timg%d.norm_table" % i in g else None
        out.append((ent, rel, nrm))
    return out


def run_oracle(case, tables=None, row_loop=False):
    """Oracle forward for a tests/cases.py case.  tables=None -> drawn from the torch CPU RNG under the case's seed
    (the reference's own behaviour)."""
    cfg = case["cfg"]
    sd = synth.make_state_dict(cfg["K"], case["C"], case["p"], seed=case["weight_seed"])
    cap = {}
    torch.manual_seed(case["rng_seed"])
    with torch.no_grad():
        results, extras = O.interaction_head_forward(
            sd, case["feat3"], case["detections"], case["shapes"],
            lambda coords: cases.pooled_for(case, sum(len(c) for c in coords)),
            cfg["K"], cfg["human_idx"], case["o2v"], targets=case["targets"], training=case["training"],
            max_human=case["max_human"], max_object=case["max_object"], box_nms_thresh=case["box_nms_thresh"],
            box_score_thresh=case["box_score_thresh"], num_iter=case["num_iter"], tables=tables, row_loop=row_loop,
            capture=cap)
    return results, extras, cap


def flatten_oracle(case, results, extras, cap):
    out = {"logits_p": extras["logits_p"], "logits_s": extras["logits_s"], "pair_features": extras["pair_features"],
           "n_results": torch.tensor(len(results))}
    for b, d in enumerate(extras["preprocessed"]):
        out["pre%d.boxes" % b] = d["boxes"]; out["pre%d.labels" % b] = d["labels"]; out["pre%d.scores" % b] = d["scores"]
    for b, r in enumerate(results):
        for k, v in r.items():
            out["res%d.%s" % (b, k)] = v
    for i, (e, r, n) in enumerate(cap.get("tables", [])):
        out["timg%d.ent" % i] = e; out["timg%d.rel_table" % i] = r; out["timg%d.norm_table" % i] = n
        out["timg%d.spatial46" % i] = cap["spatial46"][i]
        if case["num_iter"] > 0:
            out["timg%d.adjacency" % i] = cap["adjacency"][i].reshape(-1, 1)
            out["timg%d.h_node" % i] = cap["h_node"][i]; out["timg%d.node" % i] = cap["node"][i]
    out["n_tables"] = torch.tensor(len(cap.get("tables", [])))
    if case["training"]:
        g = extras["graph"]
        for i in range(len(g["pos_scores"])):
            out["timg%d.pos_scores" % i] = g["pos_scores"][i]; out["timg%d.neg_scores" % i] = g["neg_scores"][i]
            out["timg%d.head_ent" % i] = g["head_ent"][i]; out["timg%d.tail_ent" % i] = g["tail_ent"][i]
            out["timg%d.rel" % i] = g["rel"][i]; out["timg%d.rel_norm" % i] = g["rel_norm"][i]
        for k, v in extras["losses"].items():
            out[k] = v
    return {k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in out.items()}


INT_KINDS = ("i", "u", "b")


def compare_flat(got, want, atol=1e-6, rtol=1e-5, skip=(), only_common=False):
    """Integer arrays bit-exact; float arrays within atol + rtol*|want|.  Returns max abs float error."""
    worst = 0.0
    for k, w in want.items():
        if any(k.endswith(s) for s in skip):
            continue
        if k not in got:
            if only_common:
                continue
            raise AssertionError("missing key %s" % k)
        g = got[k]
        assert g.shape == w.shape or g.size == w.size == 0 or g.reshape(w.shape).shape == w.shape, \
            "%s: shape %s vs %s" % (k, g.shape, w.shape)
        g = g.reshape(w.shape)
        if w.dtype.kind in INT_KINDS:
            assert np.array_equal(g, w), "%s: integer mismatch" % k
        else:
            err = np.abs(g.astype(np.float64) - w.astype(np.float64))
            tol = atol + rtol * np.abs(w.astype(np.float64))
            bad = ~(err <= tol) & ~(np.isnan(g) & np.isnan(w)) & ~(g == w)
            assert not bad.any(), "%s: max err %.3e (tol %.1e) at %d/%d entries" % (
                k, float(np.nanmax(err)), atol, int(bad.sum()), bad.size)
            fin = err[np.isfinite(err)]
            if fin.size:
                worst = max(worst, float(fin.max()))
    return worst


def oracle_train_grads(case, with_flat=False):
    """Oracle training forward + backward of (hoi + interactiveness + transH) on CPU: {param name: grad}, losses
    (+ the flattened forward outputs when with_flat)."""
    cfg = case["cfg"]
    sd = synth.make_state_dict(cfg["K"], case["C"], case["p"], seed=case["weight_seed"])
    sd = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    cap = {}
    torch.manual_seed(case["rng_seed"])
    results, extras = O.interaction_head_forward(
        sd, case["feat3"], case["detections"], case["shapes"],
        lambda coords: cases.pooled_for(case, sum(len(c) for c in coords)),
        cfg["K"], cfg["human_idx"], case["o2v"], targets=case["targets"], training=True,
        max_human=case["max_human"], max_object=case["max_object"], box_nms_thresh=case["box_nms_thresh"],
        box_score_thresh=case["box_score_thresh"], num_iter=case["num_iter"], capture=cap)
    total = sum(extras["losses"].values())
    total.backward()
    grads = {k: v.grad.numpy() for k, v in sd.items() if v.grad is not None}
    losses = {k: float(v) for k, v in extras["losses"].items()}
    if with_flat:
        return grads, losses, flatten_oracle(case, results, extras, cap)
    return grads, losses
