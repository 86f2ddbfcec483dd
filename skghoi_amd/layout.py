"""Host-side batch layout of the interaction-head hot path (pure numpy, no device work).

From the per-image counts the preprocess kernel reports -- humans n_h, nodes n, scored cells L -- this builds the
`skg_image_meta` records (include/skghoi.h) and the small index arrays the kernels gather through.  The reference
walks images in a Python loop (heads/adamixer_transH_spatial_r50_head.py:822-982); here every image of the batch is laid
out in concatenated row spaces:

    boxes     : all selected detections            sum_all N     (box_off)
    enc rows  : box_head output                    sum_all N     (enc_off -- see Q9 below)
    nodes     : graph nodes of ACTIVE images       sum N         (node_off)
    humans    : human nodes of ACTIVE images       sum n_h       (hum_off)
    grid      : n_h x n rows incl. self pairs      sum n_h*n     (grid_off)
    pairs     : kept pairs x != y                  sum n_h*(n-1) (pair_off)
    cells     : scored (pair, verb) cells          sum L         (out_off)

Reference quirks reproduced here (each can be switched off):
  * Q9  (HEAD:829-839): a skipped image (n_h == 0 or n <= 1) does not advance the node-encoding offset, so later
        images of the batch read the wrong rows of box_head's output.  `faithful_skip_offset=True` keeps that.
  * zip truncation (HEAD:822): the image loop also zips over the rows of box_features, so at most sum_all N images
        are visited.
"""
import numpy as np

from . import _capi

META_DTYPE = np.dtype(_capi.META_FIELDS)
assert META_DTYPE.itemsize == 48


class BatchLayout:
    pass


def build(n_h, n, L, image_shapes, human_idx, faithful_skip_offset=True, zip_truncation=True):
    """n_h, n, L: int sequences of length B (L may be None -> zeros).  Returns a BatchLayout."""
    n_h = np.asarray(n_h, dtype=np.int64); n = np.asarray(n, dtype=np.int64)
    B = len(n)
    L = np.zeros(B, np.int64) if L is None else np.asarray(L, dtype=np.int64)
    lay = BatchLayout()
    lay.B = B
    lay.n_h = n_h; lay.n = n
    lay.sum_all = int(n.sum())
    lay.box_off = np.concatenate([[0], np.cumsum(n)]).astype(np.int64)
    lay.n_visit = min(B, lay.sum_all) if zip_truncation else B
    skipped = (n_h == 0) | (n <= 1)
    lay.skipped = skipped
    active = [b for b in range(lay.n_visit) if not skipped[b]]
    lay.active = np.asarray(active, dtype=np.int64)
    A = len(active)
    lay.n_active = A
    meta = np.zeros(A, META_DTYPE)
    act = lay.active
    ah_ = n_h[act]; an_ = n[act]
    shp = np.asarray([(float(s[0]), float(s[1])) for s in image_shapes], dtype=np.float32).reshape(-1, 2)

    def excl(v):
        return np.cumsum(v) - v

    meta["image"] = act; meta["n_h"] = ah_; meta["n"] = an_
    meta["box_off"] = lay.box_off[act]
    meta["enc_off"] = excl(an_) if faithful_skip_offset else lay.box_off[act]
    meta["node_off"] = excl(an_); meta["hum_off"] = excl(ah_)
    meta["grid_off"] = excl(ah_ * an_); meta["pair_off"] = excl(ah_ * (an_ - 1)); meta["out_off"] = excl(L[act])
    if A:
        meta["img_h"] = shp[act, 0]; meta["img_w"] = shp[act, 1]
    node = int(an_.sum()); hum = int(ah_.sum()); grid = int((ah_ * an_).sum())
    pair = int((ah_ * (an_ - 1)).sum()); out = int(L[act].sum())
    lay.meta = meta
    lay.sum_n, lay.sum_h, lay.sum_g, lay.sum_p, lay.sum_l = node, hum, grid, pair, out
    an = meta["n"].astype(np.int64); ah = meta["n_h"].astype(np.int64)
    ar = np.arange(A, dtype=np.int32)
    lay.node_img = np.repeat(ar, an).astype(np.int32)            # active-image index of every graph node
    lay.hum_img = np.repeat(ar, ah).astype(np.int32)
    node_local = (np.arange(node, dtype=np.int64) - np.repeat(meta["node_off"].astype(np.int64), an))
    hum_local = (np.arange(hum, dtype=np.int64) - np.repeat(meta["hum_off"].astype(np.int64), ah))
    lay.node_enc_row = (np.repeat(meta["enc_off"].astype(np.int64), an) + node_local).astype(np.int32)
    lay.hum_enc_row = (np.repeat(meta["enc_off"].astype(np.int64), ah) + hum_local).astype(np.int32)
    lay.node_ent_row = node_local.astype(np.int32)               # tails are indexed by position y
    lay.hum_ent_row = np.full(hum, human_idx, dtype=np.int32)    # heads are the constant human_idx
    lay.pairs_per_image = (ah * (an - 1)).astype(np.int64)
    lay.cells_per_image = L[act].astype(np.int64)
    if A and (meta["n"].max() > _capi.TRANSH_ENT or human_idx >= _capi.TRANSH_ENT or human_idx < 0):
        # the reference indexes an 80-row embedding with y and human_idx (HEAD:570-572, 690)
        raise IndexError("index out of range in self")
    return lay


def build_train(n_h, n, L, image_shapes, human_idx, gt_count=None, faithful_skip_offset=True, zip_truncation=True, pin=True):
    """build() + pack_int_arrays() + the training step's extra tables (hum_of, node_of, pair_img, gt_off) in ONE native
    call (skg_layout_pack_train, csrc/skg_layout.cpp), written straight into a pinned staging tensor.  Returns (lay, buf,
    offs): the BatchLayout (its arrays are views of `buf`), the int32 host tensor to upload in one copy, and the slices
    {name: (offset, length)} inside it.  Same numbers as the numpy route (tests/test_capi_and_host.py)."""
    import ctypes as C
    import torch
    lib = _capi.lib()
    n_h = np.ascontiguousarray(n_h, dtype=np.int64); n = np.ascontiguousarray(n, dtype=np.int64)
    B = len(n)
    L = np.zeros(B, np.int64) if L is None else np.ascontiguousarray(L, dtype=np.int64)
    shp = np.ascontiguousarray([(float(s_[0]), float(s_[1])) for s_ in image_shapes], dtype=np.float32).reshape(-1, 2)
    if len(shp) < B:
        raise IndexError("list index out of range")
    gtc = None if gt_count is None else np.ascontiguousarray(gt_count, dtype=np.int32)
    info = _capi.LayoutInfo()
    # capacity: every table is bounded by a small multiple of the row spaces; a sizing call costs less than guessing wrong
    args = (n_h.ctypes.data, n.ctypes.data, L.ctypes.data, B, shp.ctypes.data, int(human_idx), int(bool(faithful_skip_offset)),
            int(bool(zip_truncation)), (gtc.ctypes.data if gtc is not None else None))
    _capi.check(lib.skg_layout_pack_train(*args, None, 0, C.byref(info)), "skg_layout_pack_train")
    if info.index_error:
        raise IndexError("index out of range in self")
    buf = torch.empty(int(info.ints), dtype=torch.int32, pin_memory=pin)
    _capi.check(lib.skg_layout_pack_train(*args, buf.data_ptr(), buf.numel(), C.byref(info)), "skg_layout_pack_train")
    host = buf.numpy()
    offs = {name: (int(info.off[i]), int(info.len[i])) for i, name in enumerate(_capi.LAYOUT_SLICES)}
    sl = lambda name: host[offs[name][0]:offs[name][0] + offs[name][1]]
    lay = BatchLayout()
    lay.B = B
    lay.n_h, lay.n = n_h, n
    lay.sum_all, lay.n_visit, lay.n_active = int(info.sum_all), int(info.n_visit), int(info.n_active)
    lay.box_off = np.concatenate([[0], np.cumsum(n)]).astype(np.int64)
    lay.skipped = (n_h == 0) | (n <= 1)
    lay.active = sl("active").astype(np.int64)
    lay.meta = sl("meta").view(META_DTYPE)
    lay.sum_n, lay.sum_h, lay.sum_g, lay.sum_p, lay.sum_l = (int(info.sum_n), int(info.sum_h), int(info.sum_g),
                                                             int(info.sum_p), int(info.sum_l))
    for name in ("node_img", "hum_img", "node_enc_row", "hum_enc_row", "node_ent_row", "hum_ent_row"):
        setattr(lay, name, sl(name))
    an = lay.meta["n"].astype(np.int64); ah = lay.meta["n_h"].astype(np.int64)
    lay.pairs_per_image = ah * (an - 1)
    lay.cells_per_image = L[lay.active]
    return lay, buf, offs


def single(n_h, n, L, image_shape):
    """The layout of ONE active image (n_h >= 1, n >= 2) with what result extraction reads (the captured single-image
    plans build it per call: build() costs ~0.1 ms of numpy for the general case, a fifth of such a forward)."""
    lay = BatchLayout()
    lay.B = lay.n_visit = lay.n_active = 1
    lay.n_h = np.asarray([n_h], np.int64); lay.n = np.asarray([n], np.int64)
    lay.sum_all = lay.sum_n = int(n); lay.sum_h = int(n_h)
    lay.sum_g, lay.sum_p, lay.sum_l = int(n_h * n), int(n_h * (n - 1)), int(L)
    lay.skipped = np.zeros(1, bool)
    lay.active = np.zeros(1, np.int64)
    lay.box_off = np.asarray([0, n], np.int64)
    lay.pairs_per_image = np.asarray([lay.sum_p], np.int64)
    lay.cells_per_image = np.asarray([lay.sum_l], np.int64)
    meta = np.zeros(1, META_DTYPE)
    meta["n_h"] = n_h; meta["n"] = n; meta["img_h"] = float(image_shape[0]); meta["img_w"] = float(image_shape[1])
    lay.meta = meta
    return lay


class ChunkLayout:
    pass


def chunk(lay, a0, a1):
    """Sub-layout of active images [a0, a1): node / human / grid / pair offsets are rebased to the chunk (the graph
    stage works on chunk-local workspaces), box_off / enc_off / image / out_off stay global.  P0 / G0 are the global
    offsets of the chunk's first kept pair / grid row (outputs are written through views starting there)."""
    ch = ChunkLayout()
    meta = lay.meta[a0:a1].copy()
    ch.P0 = int(meta["pair_off"][0]); ch.G0 = int(meta["grid_off"][0])
    ch.N0 = int(meta["node_off"][0]); ch.H0 = int(meta["hum_off"][0])
    meta["pair_off"] -= ch.P0; meta["grid_off"] -= ch.G0; meta["node_off"] -= ch.N0; meta["hum_off"] -= ch.H0
    ch.meta = meta
    ch.a0, ch.a1 = a0, a1
    ch.n_active = a1 - a0
    an = meta["n"].astype(np.int64); ah = meta["n_h"].astype(np.int64)
    ch.sum_n = int(an.sum()); ch.sum_h = int(ah.sum())
    ch.sum_g = int((an * ah).sum()); ch.sum_p = int((ah * (an - 1)).sum())
    ch.node_img = (lay.node_img[ch.N0:ch.N0 + ch.sum_n] - a0).astype(np.int32)
    ch.hum_img = (lay.hum_img[ch.H0:ch.H0 + ch.sum_h] - a0).astype(np.int32)
    ch.node_enc_row = lay.node_enc_row[ch.N0:ch.N0 + ch.sum_n]
    ch.hum_enc_row = lay.hum_enc_row[ch.H0:ch.H0 + ch.sum_h]
    ch.node_ent_row = lay.node_ent_row[ch.N0:ch.N0 + ch.sum_n]
    ch.hum_ent_row = lay.hum_ent_row[ch.H0:ch.H0 + ch.sum_h]
    return ch


def pack_int_arrays(lay):
    """One contiguous int32 host buffer (single H2D copy) + the slices of each array inside it."""
    # `*_hn` = [human rows | node rows]: the row tables of the fc_head | fc_tail input gather (HEAD:884-885), laid out
    # back to back so that one slice serves the kernel (no device-side concatenation)
    parts = [("meta", lay.meta.view(np.int32).reshape(-1)), ("node_img", lay.node_img), ("hum_img", lay.hum_img),
             ("node_enc_row", lay.node_enc_row), ("hum_enc_row", lay.hum_enc_row),
             ("node_ent_row", lay.node_ent_row), ("hum_ent_row", lay.hum_ent_row),
             ("enc_row_hn", np.concatenate([lay.hum_enc_row, lay.node_enc_row])),
             ("img_hn", np.concatenate([lay.hum_img, lay.node_img])),
             ("ent_row_hn", np.concatenate([lay.hum_ent_row, lay.node_ent_row]))]
    offs = {}
    cur = 0
    for k, v in parts:
        cur = (cur + 3) & ~3                   # 16-byte align every slice
        offs[k] = (cur, len(v))
        cur += len(v)
    buf = np.zeros(max(cur, 4), np.int32)
    for k, v in parts:
        o, l = offs[k]
        buf[o:o + l] = v
    return buf, offs
