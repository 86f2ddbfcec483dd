"""ORACLE (test infrastructure, not product code).

CPU restatement of the three `torchvision.ops.boxes` functions the reference head calls
(`box_iou`: /root/reference/ops.py:119, /root/reference/heads/adamixer_transH_spatial_r50_head.py:711-714;
`batched_nms`: head.py:123-128).  torchvision is NOT installed in the build image and the reference pins no
version, so this file *is* the definition the goldens were generated with ("parity unpinned at the torchvision
boundary").  The published torchvision algorithm is restated:

  box_area  = (x2-x1)*(y2-y1)
  box_iou   = inter / (area1[:,None] + area2 - inter),  inter = prod(clamp(min(rb)-max(lt), 0))   (no eps: 0/0 = NaN)
  nms       = greedy, candidates in descending score order (stable: ties keep ascending index),
              suppress j when IoU(i,j) > thr (strict), IoU = inter / (area_i + area_j - inter)
  batched_nms = nms on boxes shifted by label * (max_coord + 1)  ("coordinate trick"); returns kept indices
              in descending score order.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import torch
from torch import Tensor


def box_area(boxes: Tensor) -> Tensor:
    return (boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1])


def box_iou(boxes1: Tensor, boxes2: Tensor) -> Tensor:
    area1 = box_area(boxes1)
    area2 = box_area(boxes2)
    lt = torch.max(boxes1[:, None, :2], boxes2[:, :2])
    rb = torch.min(boxes1[:, None, 2:], boxes2[:, 2:])
    wh = (rb - lt).clamp(min=0)
    inter = wh[:, :, 0] * wh[:, :, 1]
    union = area1[:, None] + area2 - inter
    return inter / union


def nms(boxes: Tensor, scores: Tensor, iou_threshold: float) -> Tensor:
    n = boxes.shape[0]
    if n == 0:
        return torch.empty((0,), dtype=torch.int64, device=boxes.device)
    b = boxes.detach().cpu().float()
    x1, y1, x2, y2 = b[:, 0].tolist(), b[:, 1].tolist(), b[:, 2].tolist(), b[:, 3].tolist()
    areas_t = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    order = torch.sort(scores.detach().cpu().float(), descending=True, stable=True)[1].tolist()
    suppressed = [False] * n
    keep = []
    f32 = torch.float32
    thr = torch.tensor(iou_threshold, dtype=f32)
    for _i in range(n):
        i = order[_i]
        if suppressed[i]:
            continue
        keep.append(i)
        rest = [j for j in order[_i + 1:] if not suppressed[j]]
        if not rest:
            continue
        rj = torch.tensor(rest, dtype=torch.int64)
        xx1 = torch.maximum(b[i, 0], b[rj, 0]); yy1 = torch.maximum(b[i, 1], b[rj, 1])
        xx2 = torch.minimum(b[i, 2], b[rj, 2]); yy2 = torch.minimum(b[i, 3], b[rj, 3])
        w = (xx2 - xx1).clamp(min=0); h = (yy2 - yy1).clamp(min=0)
        inter = w * h
        ovr = inter / (areas_t[i] + areas_t[rj] - inter)
        for j, s in zip(rest, (ovr > thr).tolist()):
            if s:
                suppressed[j] = True
    return torch.tensor(keep, dtype=torch.int64, device=boxes.device)


def batched_nms(boxes: Tensor, scores: Tensor, idxs: Tensor, iou_threshold: float) -> Tensor:
    if boxes.numel() == 0:
        return torch.empty((0,), dtype=torch.int64, device=boxes.device)
    max_coordinate = boxes.max()
    offsets = idxs.to(boxes) * (max_coordinate + torch.tensor(1).to(boxes))
    boxes_for_nms = boxes + offsets[:, None]
    return nms(boxes_for_nms, scores, iou_threshold)
