"""ORACLE -- test infrastructure, NOT product code.

CPU (PyTorch fp32) restatement of SKGHOI's interaction-head hot path, written function-by-function from the reference
sources cited in each docstring (paths relative to /root/reference; HEAD = heads/adamixer_transH_spatial_r50_head.py).
It keeps the reference's arithmetic *as written* -- every redundant GEMM over all G grid rows, two message-passing
iterations that recompute each other, the TransH tables drawn from the global CPU RNG (Q1/Q2), the skipped
image offset bug (Q9) -- so that it can stand in for the reference on the GPU box, where /root/reference does not exist.

Pinning: tests/test_oracle_vs_reference.py (build container only) runs this file against the imported, unmodified
reference head on every golden case and requires bit-exact indices and <=1e-6 float agreement; tests/golden/*.npz
hold the reference's outputs for those cases (generator: tests/golden/make_golden.py), and
tests/test_oracle_golden.py re-checks the oracle against them anywhere.  The torchvision box ops are restated in
oracle/tv_boxes.py ("parity unpinned at the torchvision boundary", see there).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product
(skghoi_amd/) never does and fails loudly when its HIP library is missing.
"""
import math
from typing import List, Optional

import torch
import torch.nn.functional as F

from oracle import tv_boxes

TRANSH_DIM = 50      # HEAD:686
TRANSH_ENT = 80      # HEAD:690
CARD = 16            # HEAD:650,655,675,682


# ----------------------------------------------------------------------------- preprocess (HEAD:92-151)
def preprocess(detections, targets, human_idx, box_score_thresh=0.2, box_nms_thresh=0.5, max_human=15,
               max_object=15, append_gt=False):
    """HEAD:92-151.  Score filter -> class-wise NMS -> sort by score -> top-k humans / objects -> humans first."""
    results = []
    for b, det in enumerate(detections):
        boxes, labels, scores = det["boxes"], det["labels"], det["scores"]
        if append_gt:                                                         # HEAD:107-116
            t = targets[b]
            n = t["boxes_h"].shape[0]
            boxes = torch.cat([t["boxes_h"], t["boxes_o"], boxes])
            scores = torch.cat([torch.ones(2 * n), scores])
            labels = torch.cat([human_idx * torch.ones(n).long(), t["object"], labels])
        active = torch.nonzero(scores >= box_score_thresh).squeeze(1)          # HEAD:119-121
        keep = tv_boxes.batched_nms(boxes[active], scores[active], labels[active], box_nms_thresh)  # HEAD:123-128
        active = active[keep]
        order = torch.sort(scores[active], descending=True, stable=True)[1]    # HEAD:131 (ties: stable, see tv_boxes)
        active = active[order]
        h_idx = torch.nonzero(labels[active] == human_idx).squeeze(1)[:max_human]   # HEAD:134-139
        o_idx = torch.nonzero(labels[active] != human_idx).squeeze(1)[:max_object]
        active = active[torch.cat([h_idx, o_idx])]                             # HEAD:141-142
        results.append(dict(boxes=boxes[active].view(-1, 4), labels=labels[active].view(-1),
                            scores=scores[active].view(-1), index=active))
    return results


# ----------------------------------------------------------------------------- pairs (HEAD:847-860)
def pair_grid(n_h: int, n: int):
    """HEAD:847-860: full n_h x n meshgrid (row-major, self pairs included) and the kept (x != y) pairs."""
    x = torch.arange(n_h).view(-1, 1).expand(n_h, n)
    y = torch.arange(n).view(1, -1).expand(n_h, n)
    x_keep, y_keep = torch.nonzero(x != y).unbind(1)
    return x.reshape(-1), y.reshape(-1), x_keep, y_keep


# ----------------------------------------------------------------------------- spatial (ops.py:85-157)
def spatial_ratio_encoding(b1, b2, hw, eps=1e-10):
    """ops.py:107-157 for one image: 23 ratio features and their log -> [M,46]; IoU is the diagonal of box_iou
    (ops.py:119).  Feature order: SURVEY Appendix B / ops.py:134-152."""
    h, w = hw
    c1x = (b1[:, 0] + b1[:, 2]) / 2; c1y = (b1[:, 1] + b1[:, 3]) / 2
    c2x = (b2[:, 0] + b2[:, 2]) / 2; c2y = (b2[:, 1] + b2[:, 3]) / 2
    b1w = b1[:, 2] - b1[:, 0]; b1h = b1[:, 3] - b1[:, 1]
    b2w = b2[:, 2] - b2[:, 0]; b2h = b2[:, 3] - b2[:, 1]
    dx = torch.abs(c2x - c1x) / (b1w + eps)
    dy = torch.abs(c2y - c1y) / (b1h + eps)
    # diagonal of the pairwise IoU == element-wise IoU (same fp32 operations per element)
    lt = torch.max(b1[:, :2], b2[:, :2]); rb = torch.min(b1[:, 2:], b2[:, 2:])
    wh = (rb - lt).clamp(min=0)
    inter = wh[:, 0] * wh[:, 1]
    iou = inter / (b1w * b1h + b2w * b2h - inter)
    c1xw = c1x / w; c1yh = c1y / h; c2xw = c2x / w; c2yh = c2y / h
    b1ww = b1w / w; b1hh = b1h / h; b2ww = b2w / w; b2hh = b2h / h
    a1 = b1w * b1h / (h * w); a2 = b2w * b2h / (h * w)
    r1 = b1w / (b1h + eps); r2 = b2w / (b2h + eps)
    f = torch.stack([
        c1xw, c1yh, c2xw, c2yh, c1xw / (c2xw + eps), c1yh / (c2yh + eps),
        b1ww, b1hh, b2ww, b2hh, b1ww / (b2ww + eps), b1hh / (b2hh + eps),
        a1, a2, a1 / (a2 + eps), b2w * b2h / (b1w * b1h + eps),
        r1, r2, iou,
        (c2x > c1x).float() * dx, (c2x < c1x).float() * dx,
        (c2y > c1y).float() * dy, (c2y < c1y).float() * dy], 1)
    return torch.cat([f, torch.log(f + eps)], 1)


# ----------------------------------------------------------------------------- TransH (heads/TransH/TransH.py)
def draw_transh_tables(K: int, ent_tot: int = TRANSH_ENT, dim: int = TRANSH_DIM):
    """TransH.py:20-28 consumes the global CPU RNG in this order: three nn.Embedding default inits
    (normal_) then three xavier_uniform_ (U(+-sqrt(6/(rows+dim))))."""
    torch.empty(ent_tot, dim).normal_(); torch.empty(K, dim).normal_(); torch.empty(K, dim).normal_()
    a_e = math.sqrt(6.0 / (ent_tot + dim)); a_r = math.sqrt(6.0 / (K + dim))
    ent = torch.empty(ent_tot, dim).uniform_(-a_e, a_e)
    rel = torch.empty(K, dim).uniform_(-a_r, a_r)
    nrm = torch.empty(K, dim).uniform_(-a_r, a_r)
    return ent, rel, nrm


def transh_forward(ent, rel, nrm, heads, relations, tails):
    """TransH.py:75-106 (mode 'normal', norm_flag True, p_norm 2)."""
    h_ = ent[heads]; t_ = ent[tails]; r = rel[relations]; rn = nrm[relations]
    w = F.normalize(rn, p=2, dim=-1)                                    # TransH.py:76
    h = h_ - torch.sum(h_ * w, -1, True) * w                            # TransH.py:85
    t = t_ - torch.sum(t_ * w, -1, True) * w
    hn = F.normalize(h, 2, -1); rr = F.normalize(r, 2, -1); tn = F.normalize(t, 2, -1)   # TransH.py:58-60
    score = torch.norm((hn + rr) - tn, 2, -1).flatten()                 # TransH.py:68-70
    return h_, r, rn, t_, score


# ----------------------------------------------------------------------------- MBF (HEAD:431-530)
class Params:
    """Thin view on a reference-keyed state dict."""

    def __init__(self, sd, prefix="box_pair_head."):
        self.sd = sd
        self.p = prefix

    def lin(self, name, x):
        return F.linear(x, self.sd[self.p + name + ".weight"], self.sd[self.p + name + ".bias"])

    def w(self, name):
        return self.sd[self.p + name]


def mbf(P: Params, name, appearance, spatial):
    """MultiBranchFusion.forward, HEAD:469-474."""
    return F.relu(torch.stack([
        P.lin("%s.fc_3.%d" % (name, b), F.relu(P.lin("%s.fc_1.%d" % (name, b), appearance) *
                                              P.lin("%s.fc_2.%d" % (name, b), spatial)))
        for b in range(CARD)]).sum(dim=0))


def message_mbf_object(P: Params, name, appearance, spatial):
    """MessageMBF._forward_object_nodes, HEAD:518-527: appearance [n,D] broadcast over n_h; no outer ReLU."""
    n_h, n = spatial.shape[:2]
    assert len(appearance) == n
    return torch.stack([
        P.lin("%s.fc_3.%d" % (name, b), F.relu(P.lin("%s.fc_1.%d" % (name, b), appearance).repeat(n_h, 1, 1) *
                                              P.lin("%s.fc_2.%d" % (name, b), spatial)))
        for b in range(CARD)]).sum(dim=0)


def message_mbf_human(P: Params, name, appearance, spatial):
    """MessageMBF._forward_human_nodes, HEAD:509-517: appearance [n_h,D] broadcast over n, spatial transposed."""
    n_h, n = spatial.shape[:2]
    assert len(appearance) == n_h
    return torch.stack([
        P.lin("%s.fc_3.%d" % (name, b), F.relu(P.lin("%s.fc_1.%d" % (name, b), appearance).repeat(n, 1, 1) *
                                              P.lin("%s.fc_2.%d" % (name, b), spatial).permute([1, 0, 2])))
        for b in range(CARD)]).sum(dim=0)


# ----------------------------------------------------------------------------- GraphHead pieces
def associate_with_ground_truth(boxes_h, boxes_o, target, K, fg_iou_thresh=0.5):
    """HEAD:703-719."""
    labels = torch.zeros(boxes_h.shape[0], K)
    x, y = torch.nonzero(torch.min(tv_boxes.box_iou(boxes_h, target["boxes_h"]),
                                   tv_boxes.box_iou(boxes_o, target["boxes_o"])) >= fg_iou_thresh).unbind(1)
    labels[x, target["labels"][y]] = 1
    return labels


def compute_prior_scores(x, y, scores, object_class, o2v, K, training):
    """HEAD:721-767: prior_h/prior_o = detection scores ** p on the verbs valid for the object's class."""
    prior_h = torch.zeros(len(x), K); prior_o = torch.zeros_like(prior_h)
    p = 1.0 if training else 2.8                                         # HEAD:742
    s_h = scores[x].pow(p); s_o = scores[y].pow(p)
    tgt = [o2v[int(o)] for o in object_class[y]]
    pair_idx = [i for i, tar in enumerate(tgt) for _ in tar]
    flat = [t for tar in tgt for t in tar]
    prior_h[pair_idx, flat] = s_h[pair_idx]
    prior_o[pair_idx, flat] = s_o[pair_idx]
    return torch.stack([prior_h, prior_o])


def graph_head_forward(sd, feat3, image_shapes, pooled, box_coords, box_labels, box_scores, K, human_idx, o2v,
                       targets=None, training=False, num_iter=2, fg_iou_thresh=0.5, tables=None,
                       row_loop=False, capture=None):
    """GraphHead.forward, HEAD:769-993.  `tables`: optional list of (ent, rel, norm) per *processed* image; when
    None they are drawn from the global CPU RNG exactly as the reference does (one fresh TransH per image).
    `row_loop=True` keeps the reference's Python loop over all G*K embedding rows (HEAD:877-883) instead
    of the equivalent strided slice -- used only when timing the CPU baseline.  `capture`: dict that receives
    per-image intermediates."""
    P = Params(sd)
    if training:
        assert targets is not None
    global_features = F.adaptive_avg_pool2d(feat3, 1).flatten(start_dim=1)               # HEAD:811
    enc_all = F.relu(P.lin("box_head.3", F.relu(P.lin("box_head.1", pooled.flatten(1)))))   # HEAD:812
    num_boxes = [len(b) for b in box_coords]
    counter = 0
    out = dict(pair_features=[], boxes_h=[], boxes_o=[], object_class=[], labels=[], prior=[],
               pos_scores=[], neg_scores=[], head_ent=[], tail_ent=[], rel=[], rel_norm=[])
    t_i = 0
    # HEAD:822 zips over the rows of box_features as well -> the loop stops at min(B, sum N)
    n_loop = min(len(box_coords), enc_all.shape[0])
    for b in range(n_loop):
        coords, labels, scores = box_coords[b], box_labels[b], box_scores[b]
        n = num_boxes[b]
        n_h = int(torch.sum(labels == human_idx))
        if n_h == 0 or n <= 1:                                                   # HEAD:829-839 (no counter += n: Q9)
            out["pair_features"].append(torch.zeros(0, 2048)); out["boxes_h"].append(torch.zeros(0, 4))
            out["boxes_o"].append(torch.zeros(0, 4)); out["object_class"].append(torch.zeros(0, dtype=torch.int64))
            out["prior"].append(torch.zeros(2, 0, K)); out["labels"].append(torch.zeros(0, K))
            continue
        if not torch.all(labels[:n_h] == human_idx):
            raise ValueError("Human detections are not permuted to the top")
        node = enc_all[counter:counter + n]
        h_node = node[:n_h]
        x, y, x_keep, y_keep = pair_grid(n_h, n)
        if len(x_keep) == 0:
            raise ValueError("There are no valid human-object pairs")
        sp = spatial_ratio_encoding(coords[x], coords[y], image_shapes[b])        # HEAD:863-865
        sp_raw = sp
        if torch.isnan(sp).sum() > 0:
            sp = torch.nan_to_num(sp)                                            # HEAD:866-868
        # ---- TransH (HEAD:558-582): fresh tables per image
        if tables is None:
            ent, rel, nrm = draw_transh_tables(K)
        else:
            ent, rel, nrm = tables[t_i]
        t_i += 1
        G = n_h * n
        relations = torch.arange(K).repeat(G)
        heads = torch.full((G * K,), human_idx, dtype=torch.int64)
        tails = y.repeat_interleave(K)
        th_h, th_r, th_rn, th_t, th_score = transh_forward(ent, rel, nrm, heads, relations, tails)
        if row_loop:                                                             # HEAD:877-883
            hl, tl = [], []
            for idx, (ex, ey) in enumerate(zip(th_h, th_t)):
                if idx % K == 0:
                    hl.append(ex); tl.append(ey)
            head_rows = torch.stack(hl); tail_rows = torch.stack(tl)
        else:
            head_rows = th_h[::K]; tail_rows = th_t[::K]
        g_h = F.relu(P.lin("fc_head.0", torch.cat((h_node[x], head_rows), 1)))   # HEAD:884
        g_o = F.relu(P.lin("fc_tail.0", torch.cat((node[y], tail_rows), 1)))     # HEAD:885
        s = F.relu(P.lin("spatial_head.4", F.relu(P.lin("spatial_head.2", F.relu(P.lin("spatial_head.0", sp))))))
        s3 = s.reshape(n_h, n, -1)                                               # HEAD:888-889
        adjacency = torch.ones(n_h, n)
        for _ in range(num_iter):                                                # HEAD:892-925 (never feeds back)
            weights = mbf(P, "attention_head", torch.cat([g_h, g_o], 1), s)
            adjacency = P.lin("adjacency", weights).reshape(n_h, n)
            all_ent = g_o[0:int(g_o.size(0) / n_h)]                              # HEAD:900,903
            h_ent = g_h[[i for i in range(g_h.size(0)) if i % n == 0]]           # HEAD:901,904
            msg_h = F.relu(torch.sum(adjacency.softmax(dim=1)[..., None] *
                                     message_mbf_object(P, "obj_to_sub", all_ent, s3), dim=1))
            h_node = F.layer_norm(h_ent + msg_h, (1024,), P.w("norm_h.weight"), P.w("norm_h.bias"))
            msg_o = F.relu(torch.sum(adjacency.t().softmax(dim=1)[..., None] *
                                     message_mbf_human(P, "sub_to_obj", h_ent, s3), dim=1))
            node = F.layer_norm(all_ent + msg_o, (1024,), P.w("norm_o.weight"), P.w("norm_o.bias"))
        sc_keep = th_score.reshape(n_h, n, K)[x_keep, y_keep]                    # HEAD:926
        if targets is not None:                                                  # HEAD:933-963
            tl_ = associate_with_ground_truth(coords[x_keep], coords[y_keep], targets[b], K, fg_iou_thresh)
            px, py = torch.nonzero(tl_).unbind(1)
            neg_xy = (tl_ == 0).nonzero()
            rand = torch.randperm(neg_xy.size(0))[:len(px)]                      # HEAD:939 (global CPU RNG)
            nx, ny = neg_xy[rand].unbind(1)
            he = th_h.reshape(n_h, n, K, -1)[x_keep, y_keep]; te = th_t.reshape(n_h, n, K, -1)[x_keep, y_keep]
            re = th_r.reshape(n_h, n, K, -1)[x_keep, y_keep]; rne = th_rn.reshape(n_h, n, K, -1)[x_keep, y_keep]
            out["labels"].append(tl_)
            out["pos_scores"].append(sc_keep[px, py]); out["neg_scores"].append(sc_keep[nx, ny])
            out["head_ent"].append(torch.cat((he[px, py], he[nx, ny]), 0))
            out["tail_ent"].append(torch.cat((te[px, py], te[nx, ny]), 0))
            out["rel"].append(torch.cat((re[px, py], re[nx, ny]), 0))
            out["rel_norm"].append(torch.cat((rne[px, py], rne[nx, ny]), 0))
            if capture is not None:
                capture.setdefault("pos_xy", []).append(torch.stack([px, py]))
                capture.setdefault("neg_xy", []).append(torch.stack([nx, ny]))
        sk = s3[x_keep, y_keep]
        att1 = mbf(P, "attention_head", torch.cat([h_node[x_keep], node[y_keep]], 1), sk)      # HEAD:970
        att2 = mbf(P, "attention_head_g", global_features[b, None], sk)                          # HEAD:971-972
        out["pair_features"].append(torch.cat([att1, att2], dim=1))
        out["boxes_h"].append(coords[x_keep]); out["boxes_o"].append(coords[y_keep])
        out["object_class"].append(labels[y_keep])
        out["prior"].append(compute_prior_scores(x_keep, y_keep, scores, labels, o2v, K, training))
        if capture is not None:
            capture.setdefault("x", []).append(x); capture.setdefault("y", []).append(y)
            capture.setdefault("x_keep", []).append(x_keep); capture.setdefault("y_keep", []).append(y_keep)
            capture.setdefault("spatial46", []).append(sp_raw); capture.setdefault("adjacency", []).append(adjacency)
            capture.setdefault("h_node", []).append(h_node); capture.setdefault("node", []).append(node)
            capture.setdefault("transh_score", []).append(th_score)
            capture.setdefault("tables", []).append((ent, rel, nrm))
        counter += n
    return out


# ----------------------------------------------------------------------------- losses
def binary_focal_loss(x, y, alpha=0.5, gamma=2.0, reduction="mean", eps=1e-6):
    """ops.py:159-211."""
    loss = (1 - y - alpha).abs() * ((y - x).abs() + eps) ** gamma * F.binary_cross_entropy(x, y, reduction="none")
    return loss.mean() if reduction == "mean" else loss.sum() if reduction == "sum" else loss


def margin_loss(p_score, n_score, margin=1.0):
    """heads/MarginLoss.py:28-36 (adv_flag False) == OpenKE/openke/module/loss/MarginLoss.py:24-28."""
    return torch.max(p_score - n_score, torch.tensor([-margin])).mean() + margin


def transh_loss_intended(pos_scores: List[torch.Tensor], neg_scores: List[torch.Tensor], n_p, margin=1.0):
    """Intended semantics of HEAD:207-235 (the committed call raises TypeError): NegativeSampling's
    one-argument forward (heads/NegativeSampling.py:30-63) splits score = cat[pos, neg] in halves shaped [M,1],
    MarginLoss(margin=1) (HEAD:230), divided by n_p."""
    score = torch.cat([torch.cat(pos_scores), torch.cat(neg_scores)])
    half = len(score) // 2
    p = score[:half].view(-1, half).permute(1, 0)
    n = score[half:].view(-1, half).permute(1, 0)
    return margin_loss(p, n, margin) / n_p


def postprocess(logits_p, logits_s, prior, boxes_h, boxes_o, object_class, labels):
    """HEAD:237-337."""
    num_boxes = [len(b) for b in boxes_h]
    weights = torch.sigmoid(logits_s).squeeze(1).split(num_boxes)
    scores = torch.sigmoid(logits_p).split(num_boxes)
    if len(labels) == 0:
        labels = [None] * len(num_boxes)
    results = []
    for w, s, p, bh, bo, o, l in zip(weights, scores, prior, boxes_h, boxes_o, object_class, labels):
        x, y = torch.nonzero(p[0]).unbind(1)
        r = dict(boxes_h=bh, boxes_o=bo, index=x, prediction=y,
                 scores=s[x, y] * p[:, x, y].prod(dim=0) * w[x].detach(),
                 object=o, prior=p[:, x, y], weights=w)
        if l is not None:
            r["labels"] = l[x, y]
            r["unary_labels"] = l.sum(dim=1).clamp(max=1)
        results.append(r)
    return results


def interaction_head_forward(sd, feat3, detections, image_shapes, pooled_fn, K, human_idx, o2v, targets=None,
                             training=False, max_human=15, max_object=15, box_nms_thresh=0.5, box_score_thresh=0.2,
                             num_iter=2, fg_iou_thresh=0.5, tables=None, row_loop=False, capture=None):
    """InteractionHead.forward, HEAD:341-429.  `pooled_fn(box_coords) -> [sum N, C, p, p]` plays box_roi_pool
    (HEAD:387).  Returns (results, extras) where extras carries logits and, in training, the three loss terms
    (transH term per `transh_loss_intended`)."""
    if training:
        assert targets is not None
    det = preprocess(detections, targets, human_idx, box_score_thresh, box_nms_thresh, max_human, max_object,
                     append_gt=training)
    coords = [d["boxes"] for d in det]; labels = [d["labels"] for d in det]; scores = [d["scores"] for d in det]
    pooled = pooled_fn(coords)
    gh = graph_head_forward(sd, feat3, image_shapes, pooled, coords, labels, scores, K, human_idx, o2v,
                            targets=targets, training=training, num_iter=num_iter, fg_iou_thresh=fg_iou_thresh,
                            tables=tables, row_loop=row_loop, capture=capture)
    pf = torch.cat(gh["pair_features"])
    logits_p = F.linear(pf, sd["box_pair_predictor.weight"], sd["box_pair_predictor.bias"])    # HEAD:410
    logits_s = F.linear(pf, sd["box_pair_suppressor.weight"], sd["box_pair_suppressor.bias"])  # HEAD:411
    results = postprocess(logits_p, logits_s, gh["prior"], gh["boxes_h"], gh["boxes_o"], gh["object_class"],
                          gh["labels"])
    extras = dict(logits_p=logits_p, logits_s=logits_s, pair_features=pf, preprocessed=det, graph=gh)
    if training:
        lab = torch.cat([r["labels"] for r in results]); sc = torch.cat([r["scores"] for r in results])
        n_p = len(torch.nonzero(lab))
        hoi = binary_focal_loss(sc, lab, reduction="sum", gamma=0.2) / n_p                    # HEAD:153-177
        wl = torch.cat([r["weights"] for r in results]); ul = torch.cat([r["unary_labels"] for r in results])
        n_pu = len(torch.nonzero(ul))
        inter = binary_focal_loss(wl, ul, reduction="sum", gamma=2.0) / n_pu                   # HEAD:180-205
        th = transh_loss_intended(gh["pos_scores"], gh["neg_scores"], n_pu)                    # HEAD:207-235
        extras["losses"] = dict(hoi_loss=hoi, interactiveness_loss=inter, transH_loss=th)
    return results, extras
