"""ORACLE (test infrastructure): CPU restatement of torchvision's MultiScaleRoIAlign / roi_align(aligned=False) /
LevelMapper as the reference uses them (models/adamixer_transH_spatial_r50_models.py:158-162, head.py:387).
torchvision is absent and unpinned: parity is unpinned at this boundary; this file is the definition."""
import math

import torch


def infer_scale(feature_hw, original_hw):
    approx = float(feature_hw[0]) / float(original_hw[0])
    return 2.0 ** float(torch.tensor(approx).log2().round())


def level_of(boxes, k_min, k_max, canonical_scale=224, canonical_level=4, eps=1e-6):
    s = torch.sqrt((boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1]))
    lv = torch.floor(canonical_level + torch.log2(s / canonical_scale) + torch.tensor(eps, dtype=s.dtype))
    return (torch.clamp(lv, min=k_min, max=k_max).to(torch.int64) - k_min)


def _bilinear(f, y, x):
    """f [C,H,W]; y, x scalars (python floats)."""
    C, H, W = f.shape
    if y < -1.0 or y > H or x < -1.0 or x > W:
        return torch.zeros(C)
    y = max(y, 0.0); x = max(x, 0.0)
    y_low, x_low = int(y), int(x)
    if y_low >= H - 1:
        y_high = y_low = H - 1; y = float(y_low)
    else:
        y_high = y_low + 1
    if x_low >= W - 1:
        x_high = x_low = W - 1; x = float(x_low)
    else:
        x_high = x_low + 1
    ly, lx = y - y_low, x - x_low
    hy, hx = 1.0 - ly, 1.0 - lx
    return hy * hx * f[:, y_low, x_low] + hy * lx * f[:, y_low, x_high] + ly * hx * f[:, y_high, x_low] + \
        ly * lx * f[:, y_high, x_high]


def roi_align(feat, rois, image_idx, scale, pooled, sampling):
    """feat [B,C,H,W] fp32, rois [K,4] -> [K,C,pooled,pooled].  Scalar loops in fp32 like the C++ kernel."""
    f32 = torch.float32
    K = rois.shape[0]
    out = torch.zeros(K, feat.shape[1], pooled, pooled)
    sc = torch.tensor(scale, dtype=f32)
    for n in range(K):
        b = rois[n].to(f32) * sc
        x1, y1, x2, y2 = [float(v) for v in b]
        rw = max(float(torch.tensor(x2, dtype=f32) - torch.tensor(x1, dtype=f32)), 1.0)
        rh = max(float(torch.tensor(y2, dtype=f32) - torch.tensor(y1, dtype=f32)), 1.0)
        bw = float(torch.tensor(rw, dtype=f32) / pooled); bh = float(torch.tensor(rh, dtype=f32) / pooled)
        gh = sampling if sampling > 0 else int(math.ceil(rh / pooled))
        gw = sampling if sampling > 0 else int(math.ceil(rw / pooled))
        f = feat[int(image_idx[n])]
        for ph in range(pooled):
            for pw in range(pooled):
                acc = torch.zeros(feat.shape[1])
                for iy in range(gh):
                    y = float(torch.tensor(y1, dtype=f32) + torch.tensor(ph * bh, dtype=f32) +
                              torch.tensor((iy + 0.5) * bh / gh, dtype=f32))
                    for ix in range(gw):
                        x = float(torch.tensor(x1, dtype=f32) + torch.tensor(pw * bw, dtype=f32) +
                                  torch.tensor((ix + 0.5) * bw / gw, dtype=f32))
                        acc += _bilinear(f, y, x)
                out[n, :, ph, pw] = acc / max(gh * gw, 1)
    return out


def multiscale_roi_align(feats, boxes, image_shapes, output_size=7, sampling_ratio=2):
    """feats: list of [B,C,H,W] (levels, finest first); boxes: list of [N,4] per image."""
    max_h = max(s[0] for s in image_shapes); max_w = max(s[1] for s in image_shapes)
    scales = [infer_scale(f.shape[-2:], (max_h, max_w)) for f in feats]
    k_min = int(-math.log2(scales[0])); k_max = int(-math.log2(scales[-1]))
    rois = torch.cat(boxes)
    img = torch.cat([torch.full((len(b),), i, dtype=torch.int64) for i, b in enumerate(boxes)])
    out = torch.zeros(rois.shape[0], feats[0].shape[1], output_size, output_size)
    if len(feats) == 1:
        return roi_align(feats[0], rois, img, scales[0], output_size, sampling_ratio)
    lv = level_of(rois, k_min, k_max)
    for l, (f, s) in enumerate(zip(feats, scales)):
        idx = torch.nonzero(lv == l).squeeze(1)
        if len(idx):
            out[idx] = roi_align(f, rois[idx], img[idx], s, output_size, sampling_ratio)
    return out
