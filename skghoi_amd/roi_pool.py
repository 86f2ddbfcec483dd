"""MultiScaleRoIAlign on the HIP device -- the stage in front of the interaction head.

Drop-in for `torchvision.ops.MultiScaleRoIAlign` as the reference builds it
(models/adamixer_transH_spatial_r50_models.py:158-162: featmap_names ['0','1','2','3'], output_size 7,
sampling_ratio 2) and calls it (heads/adamixer_transH_spatial_r50_head.py:387):
    box_features = box_roi_pool(features: Dict[str, Tensor[B,C,H,W]], boxes: List[Tensor[N,4]], image_shapes)
torchvision is absent from the image; its published algorithm is restated (scale inference 2**round(log2(feat/img)),
LevelMapper with canonical scale 224 / level 4 / eps 1e-6, roi_align with aligned=False).  Differentiable with respect to
the feature maps (`skg_roi_align_bwd_f32`): the reference trains the detector's backbone and neck through this pooling
(configures/hicodet/adamixer_transH_spatial_r50_main.py:109-127 gives them lr * 0.1).
"""
import ctypes as C
import math
from typing import Dict, List, Tuple

import numpy as np
import torch
from torch import nn, Tensor

from . import _capi
from .engine import _stream


def _level_args(feats, scales):
    L = len(feats)
    return ((C.c_void_p * L)(*[f.data_ptr() for f in feats]),
            (C.c_int32 * L)(*[int(f.shape[2]) for f in feats]), (C.c_int32 * L)(*[int(f.shape[3]) for f in feats]),
            (C.c_float * L)(*scales))


class _RoIAlignFn(torch.autograd.Function):
    """out = MultiScaleRoIAlign(feats...); backward scatters d out into zeroed feature gradients (float atomics)."""

    @staticmethod
    def forward(ctx, cfg, rois, img, *feats):
        scales, k_min, k_max, canon_s, canon_l, pooled, sampling = cfg
        fs = [f.float().contiguous() for f in feats]
        n_rois, Cc = rois.shape[0], fs[0].shape[1]
        out = torch.empty(n_rois, Cc, pooled, pooled, device=rois.device, dtype=torch.float32)
        ptrs, Hs, Ws, sc = _level_args(fs, scales)
        _capi.check(_capi.lib().skg_roi_align_f32(ptrs, Hs, Ws, sc, len(fs), Cc, k_min, k_max, float(canon_s), int(canon_l),
                                                  rois.data_ptr(), img.data_ptr(), n_rois, pooled, sampling,
                                                  out.data_ptr(), _stream()), "skg_roi_align_f32")
        ctx.cfg, ctx.rois, ctx.img = cfg, rois, img
        ctx.shapes = [tuple(f.shape) for f in fs]
        ctx.dtypes = [f.dtype for f in feats]
        return out

    @staticmethod
    def backward(ctx, dout):
        scales, k_min, k_max, canon_s, canon_l, pooled, sampling = ctx.cfg
        dout = dout.float().contiguous()
        dfs = [torch.zeros(sh, device=dout.device, dtype=torch.float32) for sh in ctx.shapes]
        ptrs, Hs, Ws, sc = _level_args(dfs, scales)
        _capi.check(_capi.lib().skg_roi_align_bwd_f32(ptrs, Hs, Ws, sc, len(dfs), ctx.shapes[0][1], k_min, k_max,
                                                      float(canon_s), int(canon_l), ctx.rois.data_ptr(), ctx.img.data_ptr(),
                                                      ctx.rois.shape[0], pooled, sampling, dout.data_ptr(), _stream()),
                    "skg_roi_align_bwd_f32")
        return (None, None, None) + tuple(d.to(t) for d, t in zip(dfs, ctx.dtypes))


class MultiScaleRoIAlign(nn.Module):
    def __init__(self, featmap_names: List[str], output_size, sampling_ratio: int, *, canonical_scale: int = 224,
                 canonical_level: int = 4):
        super().__init__()
        if isinstance(output_size, (tuple, list)):
            if output_size[0] != output_size[1]:
                raise ValueError("only square outputs are supported (the reference uses 7x7)")
            output_size = output_size[0]
        self.featmap_names = list(featmap_names)
        self.output_size = int(output_size)
        self.sampling_ratio = int(sampling_ratio)
        self.canonical_scale = canonical_scale
        self.canonical_level = canonical_level
        self.scales = None
        self.k_min = self.k_max = None

    @staticmethod
    def infer_scale(feature_hw, original_hw) -> float:
        """torchvision.ops.poolers._infer_scale: 2 ** round(log2(feature / image)) from the first (height) axis."""
        approx = float(feature_hw[0]) / float(original_hw[0])
        return 2.0 ** float(torch.tensor(approx).log2().round())

    def setup_scales(self, feats: List[Tensor], image_shapes: List[Tuple[int, int]]):
        max_h = max(s[0] for s in image_shapes); max_w = max(s[1] for s in image_shapes)
        self.scales = [self.infer_scale(f.shape[-2:], (max_h, max_w)) for f in feats]
        self.k_min = int(-math.log2(self.scales[0])); self.k_max = int(-math.log2(self.scales[-1]))

    def forward(self, x: Dict[str, Tensor], boxes: List[Tensor], image_shapes: List[Tuple[int, int]]) -> Tensor:
        feats = [x[k] for k in self.featmap_names if k in x]
        if not feats:
            raise KeyError("none of featmap_names %s in the feature dict" % self.featmap_names)
        dev = feats[0].device
        if dev.type != "cuda":
            raise _capi.SkgError("MultiScaleRoIAlign runs on a HIP device only")
        if self.scales is None or len(self.scales) != len(feats):
            self.setup_scales(feats, image_shapes)
        n_per = [int(b.shape[0]) for b in boxes]
        n_rois = sum(n_per)
        Cc = feats[0].shape[1]
        if n_rois == 0:
            return torch.empty(0, Cc, self.output_size, self.output_size, device=dev, dtype=torch.float32)
        rois = torch.cat([b.reshape(-1, 4) for b in boxes]).detach().float().contiguous()
        img = torch.repeat_interleave(torch.arange(len(boxes), dtype=torch.int32),
                                      torch.tensor(n_per)).to(dev, non_blocking=True)
        L = len(feats)
        k_min, k_max = (self.k_min, self.k_max) if L > 1 else (0, 0)
        cfg = (list(self.scales), k_min, k_max, self.canonical_scale, self.canonical_level, self.output_size,
               self.sampling_ratio)
        if torch.is_grad_enabled() and any(f.requires_grad for f in feats):
            return _RoIAlignFn.apply(cfg, rois, img, *feats)
        return _RoIAlignFn.forward(_NoCtx(), cfg, rois, img, *feats)


class _NoCtx:
    """Stand-in for the autograd context on the no-gradient path (the forward stores three attributes on it)."""
