"""Runs the product (HIP) interaction head on cuda:0 for a tests/cases.py case and flattens its outputs with the same
keys as the golden fixtures / helpers.flatten_oracle."""
from collections import OrderedDict

import numpy as np
import torch

import cases
from skghoi_amd import GraphHead, InteractionHead, synth


class CachedPool(torch.nn.Module):
    """Stands in for MultiScaleRoIAlign: returns the cached pooled features for however many boxes were kept."""

    def __init__(self, case):
        super().__init__()
        self.case = case

    def forward(self, features, boxes, image_shapes):
        return cases.pooled_for(self.case, sum(len(b) for b in boxes)).cuda()


PRECISION = "fp32"      # inference GEMM path of the heads built here ("fp32" exact | "fp16x2"); set by the precision fixture


def build_head(case, reference_quirks=True):
    cfg = case["cfg"]
    gh = GraphHead(case["C"], case["p"], 1024, 1024, cfg["K"], cfg["human_idx"], case["o2v"],
                   num_iter=case["num_iter"])
    head = InteractionHead(CachedPool(case), gh, torch.nn.Linear(2048, 1), torch.nn.Linear(2048, cfg["K"]),
                           human_idx=cfg["human_idx"], num_classes=cfg["K"], box_nms_thresh=case["box_nms_thresh"],
                           box_score_thresh=case["box_score_thresh"], max_human=case["max_human"],
                           max_object=case["max_object"], reference_quirks=reference_quirks, precision=PRECISION)
    head.load_state_dict(synth.make_state_dict(cfg["K"], case["C"], case["p"], seed=case["weight_seed"]))
    return head.cuda().train(case["training"])


def to_cuda(x):
    if torch.is_tensor(x):
        return x.cuda()
    if isinstance(x, dict):
        return {k: to_cuda(v) for k, v in x.items()}
    if isinstance(x, list):
        return [to_cuda(v) for v in x]
    return x


def run_head(case, head=None, reference_quirks=True):
    head = head or build_head(case, reference_quirks)
    K = case["cfg"]["K"]
    det = to_cuda(case["detections"]); tg = to_cuda(case["targets"])
    feats = OrderedDict((k, case["feat3"].cuda()) for k in "0123")
    out = {}
    head.engine().debug = True
    if "chunk_images" in case:
        head.engine().chunk_images = case["chunk_images"]
    if "n_streams" in case:
        head.engine().n_streams = case["n_streams"]
    if case["training"]:
        return _run_train(case, head, det, tg, feats)
    if tg is not None:                       # eval with targets: only the result dicts are comparable
        out = {}
        with torch.no_grad():
            for b, d in enumerate(head.preprocess(det, tg)):
                out["pre%d.boxes" % b] = d["boxes"]; out["pre%d.labels" % b] = d["labels"]; out["pre%d.scores" % b] = d["scores"]
            torch.manual_seed(case["rng_seed"])
            results = head(feats, det, case["shapes"], tg)
            after = torch.empty(4).uniform_()          # position of the host RNG after the call
        for b, r in enumerate(results):
            for k, v in r.items():
                out["res%d.%s" % (b, k)] = v
        out["n_results"] = torch.tensor(len(results)); out["rng_after"] = after
        return {k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in out.items()}
    with torch.no_grad():
        for b, d in enumerate(head.preprocess(det, tg)):
            out["pre%d.boxes" % b] = d["boxes"]; out["pre%d.labels" % b] = d["labels"]; out["pre%d.scores" % b] = d["scores"]
        torch.manual_seed(case["rng_seed"])
        results = head(feats, det, case["shapes"], tg)
    torch.cuda.synchronize()
    last = head.engine().last
    for b, r in enumerate(results):
        for k, v in r.items():
            out["res%d.%s" % (b, k)] = v
    out["n_results"] = torch.tensor(len(results))
    lay = last["layout"]
    out["n_tables"] = torch.tensor(lay.n_active)
    if lay.n_active:
        out["logits_p"] = last["logits"][:, :K]; out["logits_s"] = last["logits"][:, K:K + 1]
        out["pair_features"] = last["pair_features"]
        ent = last["tables"][0]
        for a in range(lay.n_active):
            m = lay.meta[a]
            g0, G = int(m["grid_off"]), int(m["n_h"]) * int(m["n"])
            out["timg%d.ent" % a] = ent[a]
            out["timg%d.spatial46" % a] = last["spatial46"][g0:g0 + G, :46]
            out["timg%d.h_node" % a] = last["h_node"][int(m["hum_off"]):int(m["hum_off"]) + int(m["n_h"])]
            out["timg%d.node" % a] = last["node"][int(m["node_off"]):int(m["node_off"]) + int(m["n"])]
            if "adjacency" in last:
                out["timg%d.adjacency" % a] = last["adjacency"][g0:g0 + G].reshape(-1, 1)
    return {k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in out.items()}


def _run_train(case, head, det, tg, feats, backward=False):
    """Training-mode forward (and optionally backward of the summed losses).  Captures graph_train's lists."""
    import skghoi_amd.train_graph as tgm
    from skghoi_amd import train_fused
    K = case["cfg"]["K"]
    if getattr(head, "fused_training", False) and train_fused.supported(head):
        return _run_train_fused(case, head, det, tg, feats, backward)
    cap = {}
    orig = tgm.graph_train

    def wrapped(*a, **k):
        r = orig(*a, **k); cap["lists"], cap["lay"] = r[0], r[1]; return r

    tgm.graph_train = wrapped
    try:
        out = {}
        for b, d in enumerate(head.preprocess(det, tg)):
            out["pre%d.boxes" % b] = d["boxes"]; out["pre%d.labels" % b] = d["labels"]; out["pre%d.scores" % b] = d["scores"]
        torch.manual_seed(case["rng_seed"])
        results = head(feats, det, case["shapes"], tg)
    finally:
        tgm.graph_train = orig
    losses = results[-1]
    for b, r in enumerate(results[:-1]):
        for k, v in r.items():
            out["res%d.%s" % (b, k)] = v
    out["n_results"] = torch.tensor(len(results) - 1)
    for k, v in losses.items():
        out[k] = v
    feats_l, bh, bo, oc, lab, prior, pos, neg, he, te, re, rn = cap["lists"]
    out["pair_features"] = torch.cat(feats_l)
    for i in range(len(pos)):
        out["timg%d.pos_scores" % i] = pos[i]; out["timg%d.neg_scores" % i] = neg[i]
        out["timg%d.head_ent" % i] = he[i]; out["timg%d.tail_ent" % i] = te[i]
        out["timg%d.rel" % i] = re[i]; out["timg%d.rel_norm" % i] = rn[i]
    out["n_tables"] = torch.tensor(len(pos))
    grads = None
    if backward:
        head.zero_grad()
        sum(losses.values()).backward()
        grads = {k: p.grad.detach().cpu().numpy() for k, p in head.named_parameters() if p.grad is not None}
    flat = {k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in out.items()}
    return (flat, grads) if backward else flat


def _run_train_fused(case, head, det, tg, feats, backward):
    """Training forward on the fused step (skghoi_amd/train_fused.py): result dicts, losses, pair features and the
    TransH pos / neg scores; gradients of all parameters when `backward`."""
    out = {}
    for b, d in enumerate(head.preprocess(det, tg)):
        out["pre%d.boxes" % b] = d["boxes"]; out["pre%d.labels" % b] = d["labels"]; out["pre%d.scores" % b] = d["scores"]
    head.engine().debug = True
    torch.manual_seed(case["rng_seed"])
    results = head(feats, det, case["shapes"], tg)
    losses = results[-1]
    for b, r in enumerate(results[:-1]):
        for k, v in r.items():
            out["res%d.%s" % (b, k)] = v
    out["n_results"] = torch.tensor(len(results) - 1)
    for k, v in losses.items():
        out[k] = v
    last = getattr(head, "_last_train", None)
    if last is not None:
        out["pair_features"] = last["pair_features"].clone()
        for i in range(len(last["pos_scores"])):
            out["timg%d.pos_scores" % i] = last["pos_scores"][i]; out["timg%d.neg_scores" % i] = last["neg_scores"][i]
        out["n_tables"] = torch.tensor(len(last["pos_scores"]))
    grads = None
    if backward:
        head.zero_grad()
        sum(losses.values()).backward()
        grads = {k: p.grad.detach().cpu().numpy() for k, p in head.named_parameters() if p.grad is not None}
    flat = {k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in out.items()}
    return (flat, grads) if backward else flat


def run_train_with_grads(case, fused=True):
    head = build_head(case)
    head.fused_training = bool(fused)
    if fused == "direct":
        head.grad_mode = "direct"
    det = to_cuda(case["detections"]); tg = to_cuda(case["targets"])
    feats = OrderedDict((k, case["feat3"].cuda()) for k in "0123")
    return _run_train(case, head, det, tg, feats, backward=True)
