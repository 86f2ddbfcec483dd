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


def _normal_draws(n):
    """32-bit draws `torch.empty(n).normal_()` takes from the CPU generator (ATen normal_fill: one uniform draw per
    element, plus a re-drawn last block of 16 when n is not a multiple of 16; n >= 16)."""
    return n + (16 if n % 16 else 0)


def draws_per_image(K):
    """(dead normal draws, ent draws, rel + norm draws) of one TransH() construction."""
    n_e, n_r = TRANSH_ENT * TRANSH_DIM, K * TRANSH_DIM
    return _normal_draws(n_e) + 2 * _normal_draws(n_r), n_e, 2 * n_r


_NATIVE = None      # None: not probed; False: use the torch calls; 0 / 1: skg_transh_draw_f32 with this fused_affine flag


def _torch_draw_tables(K, need_relations, g=None):
    """One TransH() worth of draws through torch itself (also the yardstick the native path is probed against).
    The normal_ inits are overwritten by the xavier draws, so only their RNG consumption matters: a uniform_ over the
    same number of 32-bit draws advances the generator identically."""
    dead, n_e, n_rn = draws_per_image(K)
    torch.empty(dead).uniform_(generator=g)
    a_e = math.sqrt(6.0 / (TRANSH_ENT + TRANSH_DIM))
    a_r = math.sqrt(6.0 / (K + TRANSH_DIM))
    ent = torch.empty(TRANSH_ENT, TRANSH_DIM).uniform_(-a_e, a_e, generator=g)
    if not need_relations:
        torch.empty(n_rn).uniform_(generator=g)
        return ent, None, None
    rel = torch.empty(K, TRANSH_DIM).uniform_(-a_r, a_r, generator=g)
    nrm = torch.empty(K, TRANSH_DIM).uniform_(-a_r, a_r, generator=g)
    return ent, rel, nrm


def _native_draw(state, n_images, K, need_relations, fused, ent, rel, nrm):
    from . import _capi
    _capi.check(_capi.lib().skg_transh_draw_f32(state.data_ptr(), state.numel(), n_images, K, int(need_relations), fused,
                                                ent.data_ptr(), rel.data_ptr() if need_relations else None,
                                                nrm.data_ptr() if need_relations else None), "skg_transh_draw_f32")


def _probe_native():
    """Decides once per process whether skg_transh_draw_f32 reproduces this PyTorch build bit for bit (tables AND
    generator state after two images), trying both roundings of uniform_'s affine step.  Uses a private generator."""
    global _NATIVE
    K = 7
    try:
        for fused in (1, 0):
            g = torch.Generator().manual_seed(20240607)
            g.set_state(g.get_state())
            state = g.get_state().clone()
            want = [_torch_draw_tables(K, True, g) for _ in range(2)]
            ent = torch.empty(2, TRANSH_ENT, TRANSH_DIM); rel = torch.empty(2, K, TRANSH_DIM); nrm = torch.empty(2, K, TRANSH_DIM)
            _native_draw(state, 2, K, True, fused, ent, rel, nrm)
            same = all(torch.equal(ent[i], want[i][0]) and torch.equal(rel[i], want[i][1]) and
                       torch.equal(nrm[i], want[i][2]) for i in range(2)) and torch.equal(state, g.get_state())
            if same:
                _NATIVE = fused
                return
    except Exception:
        pass
    _NATIVE = False


def native_path():
    """False, or the fused_affine flag (0 / 1) the native draw runs with."""
    if _NATIVE is None:
        _probe_native()
    return _NATIVE


def draw_tables(K, need_relations=False, generator=None):
    """Returns (ent [80,50], rel [K,50] | None, norm [K,50] | None) and advances the RNG like one TransH()."""
    fused = native_path()
    if fused is False:
        return _torch_draw_tables(K, need_relations, generator)
    ent = torch.empty(1, TRANSH_ENT, TRANSH_DIM)
    rel = torch.empty(1, K, TRANSH_DIM) if need_relations else None
    nrm = torch.empty(1, K, TRANSH_DIM) if need_relations else None
    state = generator.get_state() if generator is not None else torch.get_rng_state()
    _native_draw(state, 1, K, need_relations, fused, ent, rel, nrm)
    if generator is not None:
        generator.set_state(state)
    else:
        torch.set_rng_state(state)
    return ent[0], (rel[0] if need_relations else None), (nrm[0] if need_relations else None)


def draw_batch(K, n_images, need_relations=False, pin=False, out=None):
    """Tables for `n_images` processed images, stacked: ent [A,80,50] (+ rel, norm [A,K,50]), from the global CPU
    generator.  `out` = (ent, rel, nrm) pre-allocated (e.g. persistent pinned) buffers with at least n_images rows.

    Native path (skg_transh_draw_f32, probed against torch once per process): the generator state is taken out,
    advanced in C without evaluating the dead draws, and put back.  Otherwise the same draws go through torch calls."""
    if out is not None:
        ent = out[0][:n_images]
        rel = out[1][:n_images] if need_relations else None
        nrm = out[2][:n_images] if need_relations else None
    else:
        ent = torch.empty(n_images, TRANSH_ENT, TRANSH_DIM, pin_memory=pin)
        rel = torch.empty(n_images, K, TRANSH_DIM, pin_memory=pin) if need_relations else None
        nrm = torch.empty(n_images, K, TRANSH_DIM, pin_memory=pin) if need_relations else None
    if n_images == 0:
        return ent, rel, nrm
    fused = native_path()
    if fused is not False:
        state = torch.get_rng_state()
        _native_draw(state, n_images, K, need_relations, fused, ent, rel, nrm)
        torch.set_rng_state(state)
        return ent, rel, nrm
    for a in range(n_images):
        e, r, n = _torch_draw_tables(K, need_relations)
        ent[a] = e
        if need_relations:
            rel[a] = r; nrm[a] = n
    return ent, rel, nrm


def draw_train(K, n_neg, n_take, pin=False):
    """Host RNG of a training forward for len(n_neg) processed images, in the reference's order: per image the TransH
    tables (HEAD:574-580) then torch.randperm(n_neg[a])[:n_take[a]] (HEAD:938-939).  Returns ent [A,80,50], rel, nrm
    [A,K,50] and the concatenated permutation heads (int64 CPU tensor).  Native (skg_transh_draw_train_f32) when the
    probe against torch passed, through torch calls otherwise."""
    A = len(n_neg)
    ent = torch.empty(A, TRANSH_ENT, TRANSH_DIM, pin_memory=pin)
    rel = torch.empty(A, K, TRANSH_DIM, pin_memory=pin)
    nrm = torch.empty(A, K, TRANSH_DIM, pin_memory=pin)
    perm = torch.empty(int(sum(n_take)), dtype=torch.int64)
    if A == 0:
        return ent, rel, nrm, perm
    fused = native_path()
    if fused is not False:
        from . import _capi
        nn_ = torch.tensor([int(v) for v in n_neg], dtype=torch.int64)
        nt_ = torch.tensor([int(v) for v in n_take], dtype=torch.int64)
        state = torch.get_rng_state()
        _capi.check(_capi.lib().skg_transh_draw_train_f32(state.data_ptr(), state.numel(), A, K, fused, nn_.data_ptr(),
                                                          nt_.data_ptr(), ent.data_ptr(), rel.data_ptr(), nrm.data_ptr(),
                                                          perm.data_ptr()), "skg_transh_draw_train_f32")
        torch.set_rng_state(state)
        return ent, rel, nrm, perm
    o = 0
    for a in range(A):
        e, r, n = _torch_draw_tables(K, True)
        ent[a] = e; rel[a] = r; nrm[a] = n
        m = int(n_take[a])
        perm[o:o + m] = torch.randperm(int(n_neg[a]))[:m]
        o += m
    return ent, rel, nrm, perm
