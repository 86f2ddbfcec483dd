"""Synthetic weights and inputs for the interaction-head hot path.

Everything here is drawn with numpy's legacy `RandomState` (bit-stable across numpy/torch versions), so the build
container and the GPU box regenerate identical tensors from a seed and only small outputs need to be committed as
golden fixtures.  The shapes follow SURVEY.md section 8(d): the "20x20" workload is 20 humans + 20 objects on a
disjoint 8x5 cell grid of an 800x1200 image (pairwise IoU 0, so class-wise NMS keeps all 40 boxes).

The state-dict key set is the reference head's (SURVEY.md Appendix A;
/root/reference/heads/adamixer_transH_spatial_r50_head.py:635-701 and models/...:176-177).
"""
import json
import os
from collections import OrderedDict

import numpy as np
import torch

_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")


def hico_object_to_verb():
    """80 lists: verbs valid for each HICO-DET object class (alphabetical ids, 49 = person).

    Derived data (tests/golden/make_hico_tables.py) from the 600 (verb, object) keys of the reference's
    hicodet/hico_text_label.py; equivalent to HICODet.object_to_verb (hicodet/hicodet.py:168-179)."""
    with open(os.path.join(_DATA, "hico_object_to_verb.json")) as f:
        return json.load(f)["object_to_verb"]


def random_object_to_verb(num_obj, K, per_class=8, seed=7):
    rs = np.random.RandomState(seed)
    return [sorted(rs.choice(K, size=min(per_class, K), replace=False).tolist()) for _ in range(num_obj)]


def head_param_shapes(K=117, out_channels=256, pool=7, node=1024, rep=1024, card=16):
    """Ordered (name, shape) list of the InteractionHead state_dict (408 tensors at the defaults)."""
    sub = rep // card
    shp = []

    def lin(prefix, o, i):
        shp.append((prefix + ".weight", (o, i)))
        shp.append((prefix + ".bias", (o,)))

    def mbf(prefix, app):
        for b in range(card):
            lin("%s.fc_1.%d" % (prefix, b), sub, app)
        for b in range(card):
            lin("%s.fc_2.%d" % (prefix, b), sub, 1024)
        for b in range(card):
            lin("%s.fc_3.%d" % (prefix, b), rep, sub)

    g = "box_pair_head."
    lin(g + "box_head.1", node, out_channels * pool * pool)
    lin(g + "box_head.3", node, node)
    lin(g + "adjacency", 1, rep)
    mbf(g + "sub_to_obj", node)
    mbf(g + "obj_to_sub", node)
    shp.append((g + "norm_h.weight", (node,))); shp.append((g + "norm_h.bias", (node,)))
    shp.append((g + "norm_o.weight", (node,))); shp.append((g + "norm_o.bias", (node,)))
    lin(g + "spatial_head.0", 128, 46)
    lin(g + "spatial_head.2", 256, 128)
    lin(g + "spatial_head.4", 1024, 256)
    mbf(g + "attention_head", 2 * node)
    mbf(g + "attention_head_g", 256)
    lin(g + "fc_head.0", 1024, node + 50)
    lin(g + "fc_tail.0", 1024, node + 50)
    lin("box_pair_suppressor", 1, 2 * rep)
    lin("box_pair_predictor", K, 2 * rep)
    return shp


def make_state_dict(K=117, out_channels=256, pool=7, seed=0, gain=1.0):
    """Deterministic weights: U(-a, a), a = gain/sqrt(fan_in) (the nn.Linear default range), LayerNorm affine
    perturbed around (1, 0) so that it is exercised."""
    rs = np.random.RandomState(seed)
    sd = OrderedDict()
    for name, shape in head_param_shapes(K, out_channels, pool):
        if ".norm_" in name:
            base = 1.0 if name.endswith("weight") else 0.0
            arr = base + 0.1 * (rs.random_sample(shape) - 0.5)
        else:
            fan_in = shape[1] if len(shape) == 2 else None
            if fan_in is None:
                # bias: fan_in of the matching weight is the previous entry's second dim
                fan_in = prev_fan_in
            a = gain / np.sqrt(fan_in)
            arr = (rs.random_sample(shape) * 2.0 - 1.0) * a
        if len(shape) == 2:
            prev_fan_in = shape[1]
        sd[name] = torch.from_numpy(arr.astype(np.float32))
    return sd


def make_image(seed, n_h=20, n_o=20, hw=(800, 1200), out_channels=256, pool=7, human_idx=49, num_obj_classes=80,
               grid=(5, 8), feat_hw=(25, 38), score_lo=0.25, score_hi=1.0, shuffle=True):
    """One synthetic cached image: detections on a disjoint cell grid + pooled features (post-preprocess order is
    decided by the head; `pooled` rows are in the order the head's preprocess emits: humans first, each group by
    descending score -- see `pooled_in_head_order`)."""
    rs = np.random.RandomState(seed)
    H, W = hw
    rows, cols = grid
    n = n_h + n_o
    assert n <= rows * cols
    ch, cw = H / rows, W / cols
    cells = rs.permutation(rows * cols)[:n]
    boxes = np.zeros((n, 4), np.float32)
    for k, c in enumerate(cells):
        r, q = divmod(int(c), cols)
        jx, jy = rs.uniform(0, 0.3, 2)
        sx, sy = rs.uniform(0.30, 0.65, 2)
        x1 = q * cw + jx * cw; y1 = r * ch + jy * ch
        boxes[k] = (x1, y1, x1 + sx * cw, y1 + sy * ch)
    others = [c for c in range(num_obj_classes) if c != human_idx]
    obj_labels = rs.permutation(others)[:n_o] if n_o <= len(others) else rs.choice(others, n_o)
    labels = np.concatenate([np.full(n_h, human_idx), obj_labels]).astype(np.int64)
    scores = rs.uniform(score_lo, score_hi, n).astype(np.float32)
    if shuffle:
        p = rs.permutation(n)
        boxes, labels, scores = boxes[p], labels[p], scores[p]
    pooled = rs.standard_normal((n, out_channels, pool, pool)).astype(np.float32)
    feat3 = rs.standard_normal((1, 256, feat_hw[0], feat_hw[1])).astype(np.float32)
    return dict(boxes=torch.from_numpy(boxes), labels=torch.from_numpy(labels), scores=torch.from_numpy(scores),
                pooled=torch.from_numpy(pooled), feat3=torch.from_numpy(feat3), hw=(int(H), int(W)))


def make_batch(seeds, **kw):
    imgs = [make_image(s, **kw) for s in seeds]
    detections = [dict(boxes=i["boxes"], labels=i["labels"], scores=i["scores"]) for i in imgs]
    pooled = torch.cat([i["pooled"] for i in imgs])
    feat3 = torch.cat([i["feat3"] for i in imgs])
    shapes = [i["hw"] for i in imgs]
    return detections, pooled, feat3, shapes


def make_targets(detection, human_idx, o2v, seed, n_gt=4, jitter=2.0):
    """Ground-truth HOI pairs built from the image's own boxes (jittered) so that some pairs associate."""
    rs = np.random.RandomState(seed)
    labels = detection["labels"].numpy()
    boxes = detection["boxes"].numpy()
    hs = np.nonzero(labels == human_idx)[0]
    os_ = np.nonzero(labels != human_idx)[0]
    if len(hs) == 0 or len(os_) == 0:
        z = torch.zeros(0, 4)
        return dict(boxes_h=z, boxes_o=z.clone(), object=torch.zeros(0, dtype=torch.int64),
                    labels=torch.zeros(0, dtype=torch.int64))
    bh, bo, ob, vb = [], [], [], []
    for _ in range(n_gt):
        i = int(rs.choice(hs)); j = int(rs.choice(os_))
        verbs = o2v[int(labels[j])]
        if not verbs:
            continue
        bh.append(boxes[i] + rs.uniform(-jitter, jitter, 4))
        bo.append(boxes[j] + rs.uniform(-jitter, jitter, 4))
        ob.append(int(labels[j])); vb.append(int(rs.choice(verbs)))
    return dict(boxes_h=torch.tensor(np.array(bh), dtype=torch.float32).reshape(-1, 4),
                boxes_o=torch.tensor(np.array(bo), dtype=torch.float32).reshape(-1, 4),
                object=torch.tensor(ob, dtype=torch.int64), labels=torch.tensor(vb, dtype=torch.int64))
