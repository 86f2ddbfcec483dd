"""Host-side TransH tables, drawn exactly as the reference draws them.

The reference constructs a brand-new, randomly initialised TransH (ent 80x50, rel Kx50, norm Kx50) for every image of
every forward (heads/adamixer_transH_spatial_r50_head.py:574-580; heads/TransH/TransH.py:20-28), so its outputs depend
on the global CPU torch RNG.  To be a drop-in the replacement must consume that RNG identically: per
processed image three `normal_` draws (the nn.Embedding default inits: 80x50, Kx50, Kx50) followed by three
xavier-uniform draws U(+-sqrt(6/(rows+50))).  In training one `torch.randperm(#negatives)` per image follows
(HEAD:939) -- drawn by the caller, after the image's labels are known.
"""
import math

import torch

from ._capi import TRANSH_DIM, TRANSH_ENT


def draw_tables(K, need_relations=False, generator=None):
    """Returns (ent [80,50], rel [K,50] | None, norm [K,50] | None) and advances the RNG like one TransH()."""
    g = generator
    torch.empty(TRANSH_ENT, TRANSH_DIM).normal_(generator=g)
    torch.empty(K, TRANSH_DIM).normal_(generator=g)
    torch.empty(K, TRANSH_DIM).normal_(generator=g)
    a_e = math.sqrt(6.0 / (TRANSH_ENT + TRANSH_DIM))
    a_r = math.sqrt(6.0 / (K + TRANSH_DIM))
    ent = torch.empty(TRANSH_ENT, TRANSH_DIM).uniform_(-a_e, a_e, generator=g)
    rel = torch.empty(K, TRANSH_DIM).uniform_(-a_r, a_r, generator=g)
    nrm = torch.empty(K, TRANSH_DIM).uniform_(-a_r, a_r, generator=g)
    if not need_relations:
        return ent, None, None
    return ent, rel, nrm


def draw_batch(K, n_images, need_relations=False, pin=False, out=None):
    """Tables for `n_images` processed images, stacked: ent [A,80,50] (+ rel, norm [A,K,50]).
    `out` = (ent, rel, nrm) pre-allocated (e.g. persistent pinned) buffers with at least n_images rows."""
    if out is not None:
        ent = out[0][:n_images]
        rel = out[1][:n_images] if need_relations else None
        nrm = out[2][:n_images] if need_relations else None
    else:
        ent = torch.empty(n_images, TRANSH_ENT, TRANSH_DIM, pin_memory=pin)
        rel = torch.empty(n_images, K, TRANSH_DIM, pin_memory=pin) if need_relations else None
        nrm = torch.empty(n_images, K, TRANSH_DIM, pin_memory=pin) if need_relations else None
    for a in range(n_images):
        e, r, n = draw_tables(K, need_relations)
        ent[a] = e
        if need_relations:
            rel[a] = r; nrm[a] = n
    return ent, rel, nrm
