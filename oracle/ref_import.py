"""ORACLE (test infrastructure; BUILD CONTAINER ONLY).

Imports the reference's own, unmodified interaction head from /root/reference so that the CPU restatement in
`oracle/skg_oracle.py` can be pinned against it and golden vectors can be generated (tests/golden/make_golden.py).
/root/reference does not exist on the GPU box: nothing that runs there imports this file.

Recipe:
  * three absent third-party names are stubbed in sys.modules:
      torchvision.ops.boxes.{box_iou,batched_nms,nms,box_area} -> oracle/tv_boxes.py
      pocket.ops.Flatten                                       -> torch.nn.Flatten equivalent
      mmdet.utils.{get_root_logger,get_device}                 -> logging.getLogger / 'cpu'
  * sys.path gets the reference directories its files hard-code (head.py:24, models.py:22-24, MarginLoss.py:7)
  * bytecode writing is disabled so nothing is ever written under /root/reference.
"""
import logging
import os
import sys
import types

sys.dont_write_bytecode = True

import torch
from torch import nn

REF_ROOT = os.environ.get("SKG_REFERENCE_ROOT", "/root/reference")


def reference_available() -> bool:
    return os.path.isfile(os.path.join(REF_ROOT, "heads", "adamixer_transH_spatial_r50_head.py"))


class _Flatten(nn.Module):
    def __init__(self, start_dim=0, end_dim=-1):
        super().__init__()
        self.start_dim = start_dim
        self.end_dim = end_dim

    def forward(self, x):
        return x.flatten(self.start_dim, self.end_dim)


def _install_stubs():
    from oracle import tv_boxes

    def mod(name):
        m = sys.modules.get(name)
        if m is None:
            m = types.ModuleType(name)
            sys.modules[name] = m
        return m

    tv = mod("torchvision"); tvo = mod("torchvision.ops"); tvb = mod("torchvision.ops.boxes")
    tv.ops = tvo; tvo.boxes = tvb
    for fn in ("box_iou", "batched_nms", "nms", "box_area"):
        setattr(tvb, fn, getattr(tv_boxes, fn))
    pk = mod("pocket"); pko = mod("pocket.ops")
    pk.ops = pko; pko.Flatten = _Flatten
    md = mod("mmdet"); mdu = mod("mmdet.utils")
    md.utils = mdu
    mdu.get_root_logger = lambda *a, **k: logging.getLogger("skg_ref")
    mdu.get_device = lambda *a, **k: "cpu"


_REF = None


def load_reference():
    """Returns the imported reference head module (adamixer_transH_spatial_r50_head)."""
    global _REF
    if _REF is not None:
        return _REF
    if not reference_available():
        raise RuntimeError("reference tree not present at %s" % REF_ROOT)
    _install_stubs()
    for p in (REF_ROOT, os.path.join(REF_ROOT, "heads"), os.path.join(REF_ROOT, "heads", "TransH"),
              os.path.join(REF_ROOT, "OpenKE", "openke", "module", "loss"),
              os.path.join(REF_ROOT, "OpenKE", "openke", "module")):
        if p not in sys.path:
            sys.path.append(p)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        import adamixer_transH_spatial_r50_head as ref  # noqa: the reference's own file
    _REF = ref
    return ref


class PoolStub(nn.Module):
    """Stands in for MultiScaleRoIAlign (models.py:158-162): returns the cached pooled features."""

    def __init__(self, pooled):
        super().__init__()
        self.pooled = pooled

    def forward(self, features, boxes, image_shapes):
        return self.pooled


def build_reference_head(K, human_idx, o2v, out_channels, pool, max_human, max_object, num_iter=2,
                         box_nms_thresh=0.5, box_score_thresh=0.2, fg_iou_thresh=0.5):
    """Reference InteractionHead on CPU (construction mirrors models.py:164-191)."""
    ref = load_reference()
    gh = ref.GraphHead(out_channels, pool, 1024, 1024, K, human_idx, o2v, fg_iou_thresh=fg_iou_thresh,
                       num_iter=num_iter)
    gh.transh_head.device = "cpu"   # head.py:548 hard-codes 'cuda'
    head = ref.InteractionHead(PoolStub(None), gh, nn.Linear(2048, 1), nn.Linear(2048, K),
                               human_idx=human_idx, num_classes=K, box_nms_thresh=box_nms_thresh,
                               box_score_thresh=box_score_thresh, max_human=max_human, max_object=max_object)
    return head


class TransHCapture:
    """Context manager: records the (ent, rel, norm) tables of every TransH the reference constructs."""

    def __init__(self):
        self.tables = []

    def __enter__(self):
        ref = load_reference()
        self._cls = ref.TransH
        self._orig = ref.TransH.__init__
        cap = self

        def wrapped(this, *a, **k):
            cap._orig(this, *a, **k)
            cap.tables.append((this.ent_embeddings.weight.detach().clone(),
                               this.rel_embeddings.weight.detach().clone(),
                               this.norm_vector.weight.detach().clone()))

        self._cls.__init__ = wrapped
        return self

    def __exit__(self, *exc):
        self._cls.__init__ = self._orig
        return False
