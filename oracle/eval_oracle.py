"""ORACLE (test infrastructure): loop-level restatement of pocket's BoxPairAssociation and DetectionAPMeter ('11P') as
utils.py:148-198 uses them.  pocket is absent and unpinned: parity is unpinned at this boundary."""
import numpy as np


def iou1(a, b):
    aa = (a[2] - a[0]) * (a[3] - a[1]); ab = (b[2] - b[0]) * (b[3] - b[1])
    w = max(min(a[2], b[2]) - max(a[0], b[0]), 0.0); h = max(min(a[3], b[3]) - max(a[1], b[1]), 0.0)
    inter = w * h
    return inter / (aa + ab - inter)


def associate(gt_h, gt_o, det_h, det_o, scores, min_iou=0.5):
    n_det = len(det_h)
    labels = np.zeros(n_det)
    if n_det == 0 or len(gt_h) == 0:
        return labels
    match = -np.ones(n_det, dtype=int)
    for d in range(n_det):
        best, arg = -1.0, -1
        for g in range(len(gt_h)):
            v = min(iou1(gt_h[g], det_h[d]), iou1(gt_o[g], det_o[d]))
            if v > best:
                best, arg = v, g
        if best >= min_iou:
            match[d] = arg
    for g in sorted(set(match.tolist())):
        if g < 0:
            continue
        ds = [d for d in range(n_det) if match[d] == g]
        top = max(ds, key=lambda d: (scores[d], -d))
        labels[top] = 1
    return labels


def ap_11p(scores, labels, num_gt):
    if num_gt == 0 or len(scores) == 0:
        return 0.0
    order = sorted(range(len(scores)), key=lambda i: (-scores[i], i))
    tp = fp = 0.0
    prec, rec = [], []
    for i in order:
        tp += labels[i]; fp += 1 - labels[i]
        prec.append(tp / (tp + fp)); rec.append(tp / num_gt)
    ap = 0.0
    for t in np.linspace(0, 1, 11):
        ps = [p for p, r in zip(prec, rec) if r >= t]
        if ps:
            ap += max(ps) / 11
    return ap
