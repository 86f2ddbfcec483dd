"""On-disk formats around the hot path.

1. Detections: the reference's per-image JSON, written by hicodet/detections/adamixer_preprocessing.py:99-135 and read
   by utils.py:132-138 -- {"boxes": [[x1,y1,x2,y2],...], "scores": [...], "labels": [...]}, one file per image named
   like the image with .json.
2. Feature cache (the reference has no format for pooled features; this one is defined here): one binary shard file

       magic "SKGFC001" | u32 dtype (0 fp32, 1 fp16, 2 bf16) | u32 C | u32 pool | u32 gdim | u64 n_images | u64 n_boxes
       | i64 box_off[n_images + 1] | f32 image_hw[n_images, 2] | f32 global[n_images, gdim]
       | (pad to 4096) | payload[n_boxes, C, pool, pool] in dtype

   `global` holds AdaptiveAvgPool2d(1) of features['3'] (all the head ever reads of the feature maps, HEAD:811), so
   the ~1 MB/image map is replaced by 1 KB; the payload is the MultiScaleRoIAlign output in the head's post-preprocess
   box order.  The payload is page-aligned and memory-mapped: a batch is one contiguous slice -> one pinned staging
   copy -> one H2D transfer (2.0 MB/image fp32, 1.0 MB fp16/bf16).  fp16/bf16 storage is lossy (documented; the parity
   configuration is fp32).
"""
import json
import os
import struct

import numpy as np
import torch

MAGIC = b"SKGFC001"
_DT = {0: np.float32, 1: np.float16, 2: np.uint16}          # bf16 is stored as its upper 16 bits
_HDR = struct.Struct("<8sIIIIQQ")


def write_detections_json(path, boxes, scores, labels):
    with open(path, "w") as f:
        json.dump({"boxes": np.asarray(boxes, dtype=np.float64).reshape(-1, 4).tolist(),
                   "scores": np.asarray(scores, dtype=np.float64).reshape(-1).tolist(),
                   "labels": np.asarray(labels).reshape(-1).astype(int).tolist()}, f)


def read_detections_json(path, device=None):
    with open(path) as f:
        d = json.load(f)
    return dict(boxes=torch.as_tensor(d["boxes"], dtype=torch.float32, device=device).reshape(-1, 4),
                scores=torch.as_tensor(d["scores"], dtype=torch.float32, device=device),
                labels=torch.as_tensor(d["labels"], dtype=torch.int64, device=device))


def _to_storage(x, dtype_code):
    x = np.ascontiguousarray(x, dtype=np.float32)
    if dtype_code == 0:
        return x
    if dtype_code == 1:
        return x.astype(np.float16)
    u = x.view(np.uint32)
    return ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint16)          # round-to-nearest-even bf16 (finite data)


def write_feature_shard(path, pooled_list, global_feats, image_hw, dtype="fp32"):
    """pooled_list: per-image arrays [N_i, C, p, p]; global_feats [n_images, gdim]; image_hw [n_images, 2]."""
    code = {"fp32": 0, "fp16": 1, "bf16": 2}[dtype]
    n_img = len(pooled_list)
    C, p = (pooled_list[0].shape[1], pooled_list[0].shape[2]) if n_img else (0, 0)
    counts = [int(x.shape[0]) for x in pooled_list]
    off = np.zeros(n_img + 1, np.int64); off[1:] = np.cumsum(counts)
    g = np.ascontiguousarray(global_feats, dtype=np.float32).reshape(n_img, -1)
    hw = np.ascontiguousarray(image_hw, dtype=np.float32).reshape(n_img, 2)
    with open(path, "wb") as f:
        f.write(_HDR.pack(MAGIC, code, C, p, g.shape[1] if n_img else 0, n_img, int(off[-1])))
        f.write(off.tobytes()); f.write(hw.tobytes()); f.write(g.tobytes())
        pad = (-f.tell()) % 4096
        f.write(b"\0" * pad)
        for x in pooled_list:
            f.write(_to_storage(np.asarray(x), code).tobytes())


class FeatureShard:
    """Memory-mapped reader.  `batch(lo, hi, device)` returns (pooled [sum N, C, p, p] fp32 on device, global
    [hi-lo, gdim, 1, 1] fp32 -- usable as features['3'] --, image_hw list, boxes-per-image list)."""

    def __init__(self, path):
        self.path = path
        with open(path, "rb") as f:
            magic, self.code, self.C, self.pool, self.gdim, self.n_images, self.n_boxes = _HDR.unpack(f.read(_HDR.size))
            if magic != MAGIC:
                raise ValueError("%s is not a SKGFC001 feature shard" % path)
            self.box_off = np.frombuffer(f.read(8 * (self.n_images + 1)), dtype=np.int64)
            self.image_hw = np.frombuffer(f.read(8 * self.n_images), dtype=np.float32).reshape(-1, 2)
            self.glob = np.frombuffer(f.read(4 * self.n_images * self.gdim), dtype=np.float32).reshape(self.n_images, -1)
            pos = f.tell()
        self.payload_off = pos + ((-pos) % 4096)
        self.row = self.C * self.pool * self.pool
        self.payload = np.memmap(path, dtype=_DT[self.code], mode="r", offset=self.payload_off,
                                 shape=(self.n_boxes, self.row))

    def batch(self, lo, hi, device):
        b0, b1 = int(self.box_off[lo]), int(self.box_off[hi])
        raw = torch.from_numpy(np.array(self.payload[b0:b1], copy=True))
        if device is not None and torch.device(device).type == "cuda":
            raw = raw.pin_memory().to(device, non_blocking=True)
        if self.code == 0:
            x = raw
        elif self.code == 1:
            x = raw.float()
        else:
            x = (raw.to(torch.int32) << 16).view(torch.float32)
        pooled = x.reshape(b1 - b0, self.C, self.pool, self.pool)
        g = torch.from_numpy(np.ascontiguousarray(self.glob[lo:hi])).to(device).reshape(hi - lo, self.gdim, 1, 1)
        hw = [(int(h), int(w)) for h, w in self.image_hw[lo:hi]]
        return pooled, g, hw, np.diff(self.box_off[lo:hi + 1]).tolist()


def produce_shard(head, features, detections, image_shapes, path, dtype="fp32", targets=None):
    """The feature-cache producer: runs the head's own preprocess (NMS / top-k on the device), its `box_roi_pool` on
    the kept boxes and the global average pool of features['3'], and writes one shard.  Returns the kept detections
    (list of dicts) so that the caller can store them next to the shard (write_detections_json)."""
    from . import _capi
    from .engine import _stream
    with torch.no_grad():
        kept = head.preprocess(detections, targets)
        coords = [d["boxes"] for d in kept]
        pooled = head.box_roi_pool(features, coords, image_shapes)
        f3 = features["3"].float().contiguous()
        g = torch.empty(f3.shape[0], f3.shape[1], device=f3.device)
        _capi.check(_capi.lib().skg_global_avgpool_f32(f3.data_ptr(), f3.shape[0], f3.shape[1],
                                                       f3.shape[2] * f3.shape[3], g.data_ptr(), _stream()),
                    "skg_global_avgpool_f32")
    sizes = [len(c) for c in coords]
    parts = [p.cpu().numpy() for p in pooled.split(sizes)]
    write_feature_shard(path, parts, g.cpu().numpy(), image_shapes, dtype=dtype)
    return kept


class CachedPool(torch.nn.Module):
    """`box_roi_pool` stand-in that serves a batch of a FeatureShard (set `.pooled` before each forward)."""

    def __init__(self):
        super().__init__()
        self.pooled = None

    def forward(self, features, boxes, image_shapes):
        n = sum(len(b) for b in boxes)
        if self.pooled is None or self.pooled.shape[0] != n:
            raise RuntimeError("cached features hold %s rows, the head kept %d boxes" % (
                None if self.pooled is None else self.pooled.shape[0], n))
        return self.pooled
