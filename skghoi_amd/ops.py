"""Drop-in counterparts of the reference's ops.py for the functions on the hot path (ops.py:85-211)."""
from typing import List, Tuple

import numpy as np
import torch
from torch import Tensor

from . import _capi, layout
from .engine import _stream


def compute_spatial_ratio_encodings(boxes_1: List[Tensor], boxes_2: List[Tensor], shapes: List[Tuple[int, int]],
                                    eps: float = 1e-10) -> Tensor:
    """ops.py:85-157 for explicit box lists (row i of boxes_1 pairs with row i of boxes_2): [sum M, 46].

    Runs the same HIP kernel as the head (one wavefront per list entry) by treating every row pair as a 1 x 2 graph:
    box 1 is node 0, box 2 is node 1, the encoding of interest is grid row (0, 1)."""
    if eps != 1e-10:
        raise ValueError("the HIP kernel hard-codes eps = 1e-10 (ops.py:87)")
    outs = []
    lib = _capi.lib()
    for b1, b2, shape in zip(boxes_1, boxes_2, shapes):
        M = b1.shape[0]
        dev = b1.device
        if dev.type != "cuda":
            raise _capi.SkgError("compute_spatial_ratio_encodings runs on a HIP device only")
        if M == 0:
            outs.append(torch.zeros(0, 46, device=dev)); continue
        boxes = torch.stack([b1.float(), b2.float()], dim=1).reshape(-1, 4).contiguous()      # [2M, 4]
        meta = np.zeros(M, layout.META_DTYPE)
        ar = np.arange(M)
        meta["image"] = 0; meta["n_h"] = 1; meta["n"] = 2
        meta["box_off"] = 2 * ar; meta["node_off"] = 2 * ar; meta["hum_off"] = ar
        meta["grid_off"] = 2 * ar; meta["pair_off"] = ar
        meta["img_h"] = float(shape[0]); meta["img_w"] = float(shape[1])
        md = torch.from_numpy(meta.view(np.int32).reshape(-1).copy()).to(dev)
        i32 = dict(device=dev, dtype=torch.int32)
        gh = torch.empty(2 * M, **i32); go = torch.empty(2 * M, **i32); gp = torch.empty(2 * M, **i32)
        gi = torch.empty(2 * M, **i32); pg = torch.empty(M, **i32); ph = torch.empty(M, **i32); po = torch.empty(M, **i32)
        xk = torch.empty(M, device=dev, dtype=torch.int64); yk = torch.empty_like(xk)
        sp = torch.empty(2 * M, _capi.SPATIAL_LD, device=dev)
        _capi.check(lib.skg_pairs_spatial_f32(boxes.data_ptr(), md.data_ptr(), M, gh.data_ptr(), go.data_ptr(),
                                              gp.data_ptr(), gi.data_ptr(), pg.data_ptr(), xk.data_ptr(), yk.data_ptr(),
                                              ph.data_ptr(), po.data_ptr(), sp.data_ptr(), 0, _stream()),
                    "skg_pairs_spatial_f32")
        outs.append(sp[1::2, :46])
    return torch.cat(outs)


def binary_focal_loss(x: Tensor, y: Tensor, alpha: float = 0.5, gamma: float = 2.0, reduction: str = "mean",
                      eps: float = 1e-6) -> Tensor:
    """ops.py:159-211: L = |1-y-alpha| * (|y-x| + eps)^gamma * BCE(x, y)."""
    loss = (1 - y - alpha).abs() * ((y - x).abs() + eps) ** gamma * \
        torch.nn.functional.binary_cross_entropy(x, y, reduction="none")
    if reduction == "mean":
        return loss.mean()
    elif reduction == "sum":
        return loss.sum()
    elif reduction == "none":
        return loss
    raise ValueError("Unsupported reduction method {}".format(reduction))
