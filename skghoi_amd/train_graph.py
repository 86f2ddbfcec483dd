"""Training-mode GraphHead.forward (heads/adamixer_transH_spatial_r50_head.py:769-993) on the HIP device.

Differentiable: every dense layer goes through skghoi_amd.autograd.linear (forward AND backward GEMMs on the fp32-MFMA
kernel); gathers, products, segment softmax, LayerNorm and the losses are element-wise torch device ops recorded by
autograd.  The same algebra as the inference engine is used: message passing once, fc_head/fc_tail
and fc_1 on unique node rows, aggregation before the linear fc_3 -- all exact re-associations, so gradients equal the
reference's up to fp32 rounding.  The whole batch is processed in concatenated row spaces (no per-image Python loop on
the numerical path); per-image lists are views split off at the end.

Host RNG is consumed exactly like the reference: per processed image the six TransH draws (HEAD:574-580) and then
`torch.randperm(#negatives)` (HEAD:939).
"""
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch
import torch.nn.functional as F

from . import _capi, layout, transh
from . import autograd as _ag
from .engine import _stream


def _stack_mbf(m):
    w1 = torch.cat([l.weight for l in m.fc_1]); b1 = torch.cat([l.bias for l in m.fc_1])
    w2 = torch.cat([l.weight for l in m.fc_2]); b2 = torch.cat([l.bias for l in m.fc_2])
    w3 = torch.cat([l.weight for l in m.fc_3], dim=1); b3 = torch.stack([l.bias for l in m.fc_3]).sum(dim=0)
    return w1, b1, w2, b2, w3, b3


def _perm(state, n, m):
    g = torch.Generator(); g.set_state(state)
    return torch.randperm(n, generator=g)[:m]


def segment_softmax(x, seg, n_seg):
    """softmax of x within the groups given by `seg` (int64, values < n_seg); differentiable."""
    mx = torch.full((n_seg,), float("-inf"), device=x.device, dtype=x.dtype).scatter_reduce(
        0, seg, x.detach(), reduce="amax", include_self=True)
    e = torch.exp(x - mx[seg])
    den = torch.zeros(n_seg, device=x.device, dtype=x.dtype).index_add(0, seg, e)
    return e / den[seg]


def dense_prior(vt, pair_scores_h, pair_scores_o, pair_cls, K, power):
    """compute_prior_scores (HEAD:721-767) for all pairs at once: [2, sumP, K]."""
    dev = pair_scores_h.device
    off = vt.off.long(); flat = vt.flat.long()
    cls = pair_cls.long()
    cnt = off[cls + 1] - off[cls]
    n = cls.shape[0]
    pair = torch.repeat_interleave(torch.arange(n, device=dev), cnt)
    within = torch.arange(pair.shape[0], device=dev) - torch.repeat_interleave(torch.cumsum(cnt, 0) - cnt, cnt)
    verb = flat[torch.repeat_interleave(off[cls], cnt) + within]
    prior = torch.zeros(2, n, K, device=dev)
    prior[0, pair, verb] = pair_scores_h.pow(power)[pair]
    prior[1, pair, verb] = pair_scores_o.pow(power)[pair]
    return prior


def graph_train(eng, gh, feat3, image_shapes, pooled, pre, targets, on_counts=None):
    """Returns (the reference's 12 training lists, layout, packed extras).

    on_counts(counts): called as soon as the labels and priors of the batch are known -- before any of the dense
    layers is enqueued -- with an int64 device tensor {#positive scored cells, #positive pairs, #positive pairs}: the
    loss normalisers of HEAD:162-165, 190-192, 219-221.  The head starts its one fused all-reduce there, so that the
    exchange overlaps the GEMMs."""
    lib = _capi.lib()
    dev = pre.device
    K = gh.num_cls
    st = _stream()
    linear = _ag.linear_bf16 if getattr(eng, "precision", "fp32") == "bf16" else _ag.linear
    lay = layout.build(pre.n_h, pre.n, None, image_shapes, gh.human_idx,
                       faithful_skip_offset=eng.faithful_skip_offset)
    A = lay.n_active
    R2 = 2 * gh.representation_size
    if pooled.shape[0] != lay.sum_all:
        raise _capi.SkgError("box_roi_pool returned %d rows for %d boxes" % (pooled.shape[0], lay.sum_all))
    # HEAD:811-812
    gfeat = F.adaptive_avg_pool2d(feat3.float(), 1).flatten(start_dim=1)
    x0 = pooled.float().flatten(start_dim=1)
    enc = linear(linear(x0, gh.box_head[1].weight, gh.box_head[1].bias, True),
                 gh.box_head[3].weight, gh.box_head[3].bias, True)
    packed = None
    if not A and on_counts is not None:
        on_counts(torch.zeros(3, dtype=torch.int64, device=dev))
    if A:
        buf, offs = layout.pack_int_arrays(lay)
        ibuf = torch.from_numpy(buf).to(dev)

        def isl(name):
            o, l = offs[name]
            return ibuf[o:o + l]

        meta = isl("meta")
        Mg, Mp, Mh, Mn = lay.sum_g, lay.sum_p, lay.sum_h, lay.sum_n
        i32 = dict(device=dev, dtype=torch.int32)
        grid_h = torch.empty(Mg, **i32); grid_o = torch.empty(Mg, **i32); grid_pair = torch.empty(Mg, **i32)
        grid_img = torch.empty(Mg, **i32); pair_grid = torch.empty(max(Mp, 1), **i32)
        x_keep = torch.empty(max(Mp, 1), device=dev, dtype=torch.int64); y_keep = torch.empty_like(x_keep)
        pair_h = torch.empty(max(Mp, 1), **i32); pair_o = torch.empty(max(Mp, 1), **i32)
        sp48 = torch.empty(Mg, _capi.SPATIAL_LD, device=dev)
        _capi.check(lib.skg_pairs_spatial_f32(pre.boxes.data_ptr(), meta.data_ptr(), A, grid_h.data_ptr(),
                                              grid_o.data_ptr(), grid_pair.data_ptr(), grid_img.data_ptr(),
                                              pair_grid.data_ptr(), x_keep.data_ptr(), y_keep.data_ptr(),
                                              pair_h.data_ptr(), pair_o.data_ptr(), sp48.data_ptr(), 1, st),
                    "skg_pairs_spatial_f32")
        x_keep, y_keep = x_keep[:Mp], y_keep[:Mp]
        gh_l, go_l, gi_l = grid_h.long(), grid_o.long(), grid_img.long()
        ph_l, po_l, pg_l = pair_h[:Mp].long(), pair_o[:Mp].long(), pair_grid[:Mp].long()
        # ---- GT association for every image in one launch (HEAD:703-719), one D2H of the positive counts
        act_imgs = [int(b) for b in lay.active]
        gt_cnt = [int(targets[b]["boxes_h"].shape[0]) for b in act_imgs]
        gt_off_h = np.zeros(A + 1, np.int32); gt_off_h[1:] = np.cumsum(gt_cnt)
        if gt_off_h[-1]:
            gt_h = torch.cat([targets[b]["boxes_h"].reshape(-1, 4) for b in act_imgs]).float().contiguous()
            gt_o = torch.cat([targets[b]["boxes_o"].reshape(-1, 4) for b in act_imgs]).float().contiguous()
            gt_l = torch.cat([targets[b]["labels"].reshape(-1) for b in act_imgs]).long().contiguous()
        else:
            gt_h = torch.zeros(1, 4, device=dev); gt_o = torch.zeros(1, 4, device=dev)
            gt_l = torch.zeros(1, dtype=torch.int64, device=dev)
        gt_off = torch.from_numpy(gt_off_h).to(dev)
        labels_all = torch.zeros(max(Mp, 1), K, device=dev)
        npos_d = torch.empty(A, **i32)
        _capi.check(lib.skg_associate_f32(pre.boxes.data_ptr(), meta.data_ptr(), A, x_keep.data_ptr(),
                                          y_keep.data_ptr(), gt_h.data_ptr(), gt_o.data_ptr(), gt_l.data_ptr(),
                                          gt_off.data_ptr(), K, float(gh.fg_iou_thresh), labels_all.data_ptr(),
                                          npos_d.data_ptr(), st), "skg_associate_f32")
        labels_all = labels_all[:Mp]
        n_pos = npos_d.cpu().tolist()
        ppi = [int(v) for v in lay.pairs_per_image]
        # ---- per-pair detection attributes and priors for the whole batch
        pair_img = torch.repeat_interleave(torch.arange(A, device=dev), torch.tensor(ppi, device=dev))
        box_off = torch.from_numpy(lay.meta["box_off"].astype(np.int64)).to(dev)
        bx_h = box_off[pair_img] + x_keep; bx_o = box_off[pair_img] + y_keep
        boxes_h_all = pre.boxes[bx_h]; boxes_o_all = pre.boxes[bx_o]; obj_all = pre.labels[bx_o]
        prior_all = dense_prior(eng.verbs(dev), pre.scores[bx_h], pre.scores[bx_o], obj_all, K,
                                1.0 if gh.training else 2.8)
        if on_counts is not None:
            on_counts(torch.stack([((prior_all[0] != 0) & (labels_all != 0)).sum(),
                                   (labels_all.sum(dim=1) != 0).sum(), (labels_all.sum(dim=1) != 0).sum()]))
        # ---- host RNG in the reference's order (per image: six TransH draws, HEAD:574-580, then randperm(#negatives),
        # HEAD:938-939).  randperm(n) consumes n-1 32-bit draws: the stream is advanced with a cheap random_() of that
        # length and the permutation itself is computed from the saved generator state by worker threads with private
        # generators, overlapped with the GPU work below (tests pin the equivalence against the reference goldens).
        tabs, perms = [], []
        pool = ThreadPoolExecutor(max_workers=min(8, A))
        for a in range(A):
            tabs.append(transh.draw_tables(K, need_relations=True))
            n_neg = ppi[a] * K - n_pos[a]
            state = torch.get_rng_state()
            if n_neg > 1:
                torch.empty(n_neg - 1, dtype=torch.int32).random_()
            perms.append(pool.submit(_perm, state, n_neg, n_pos[a]))
        pool.shutdown(wait=False)
        ent = torch.stack([t[0] for t in tabs]).to(dev); rel = torch.stack([t[1] for t in tabs]).to(dev)
        nrm = torch.stack([t[2] for t in tabs]).to(dev)
        scores_all = torch.empty(max(Mp, 1), K, device=dev)
        _capi.check(lib.skg_transh_scores_f32(ent.data_ptr(), rel.data_ptr(), nrm.data_ptr(), K, gh.human_idx,
                                              meta.data_ptr(), A, scores_all.data_ptr(), st), "skg_transh_scores_f32")
        scores_all = scores_all[:Mp]
        # ---- spatial head (HEAD:888)
        spw = gh.spatial_head[0].weight
        spw48 = torch.cat([spw, spw.new_zeros(spw.shape[0], _capi.SPATIAL_LD - spw.shape[1])], dim=1)
        S = linear(linear(linear(sp48, spw48, gh.spatial_head[0].bias, True), gh.spatial_head[2].weight,
                          gh.spatial_head[2].bias, True), gh.spatial_head[4].weight, gh.spatial_head[4].bias, True)
        a_w1, a_b1, a_w2, a_b2, a_w3, a_b3 = _stack_mbf(gh.attention_head)
        F2 = linear(S, a_w2, a_b2)
        hum_rows = isl("hum_enc_row").long(); node_rows = isl("node_enc_row").long()
        if gh.num_iter > 0:
            ent_h = ent[isl("hum_img").long(), isl("hum_ent_row").long()]
            ent_o = ent[isl("node_img").long(), isl("node_ent_row").long()]
            GH = linear(torch.cat([enc[hum_rows], ent_h], 1), gh.fc_head[0].weight, gh.fc_head[0].bias, True)
            GO = linear(torch.cat([enc[node_rows], ent_o], 1), gh.fc_tail[0].weight, gh.fc_tail[0].bias, True)
            A1h = linear(GH, a_w1[:, :1024]); A1o = linear(GO, a_w1[:, 1024:])
            T = F.relu((A1h[gh_l] + A1o[go_l] + a_b1) * F2)
            Wt = linear(T, a_w3, a_b3, True)
            adj = (Wt @ gh.adjacency.weight.reshape(-1, 1)).squeeze(1) + gh.adjacency.bias          # HEAD:897
            o_w1, o_b1, o_w2, o_b2, o_w3, o_b3 = _stack_mbf(gh.obj_to_sub)
            s_w1, s_b1, s_w2, s_b2, s_w3, s_b3 = _stack_mbf(gh.sub_to_obj)
            Tos = F.relu(linear(GO, o_w1, o_b1)[go_l] * linear(S, o_w2, o_b2))
            Tso = F.relu(linear(GH, s_w1, s_b1)[gh_l] * linear(S, s_w2, s_b2))
            # softmax over the senders of every receiver (HEAD:909-922) as segment ops over the flat grid rows
            alpha = segment_softmax(adj, gh_l, Mh)                # over objects j, per human (image, i)
            beta = segment_softmax(adj, go_l, Mn)                 # over humans i, per node (image, j)
            U = torch.zeros(Mh, Tos.shape[1], device=dev).index_add(0, gh_l, alpha[:, None] * Tos)
            V = torch.zeros(Mn, Tso.shape[1], device=dev).index_add(0, go_l, beta[:, None] * Tso)
            h_node = F.layer_norm(GH + linear(U, o_w3, o_b3, True), (1024,), gh.norm_h.weight, gh.norm_h.bias)
            node = F.layer_norm(GO + linear(V, s_w3, s_b3, True), (1024,), gh.norm_o.weight, gh.norm_o.bias)
        else:
            h_node = enc[hum_rows]; node = enc[node_rows]
        # ---- read-out (HEAD:966-973)
        B1h = linear(h_node, a_w1[:, :1024]); B1o = linear(node, a_w1[:, 1024:])
        att1 = linear(F.relu((B1h[ph_l] + B1o[po_l] + a_b1) * F2[pg_l]), a_w3, a_b3, True)
        g_w1, g_b1, g_w2, g_b2, g_w3, g_b3 = _stack_mbf(gh.attention_head_g)
        G1 = linear(gfeat, g_w1, g_b1)
        att2 = linear(F.relu(G1[gi_l[pg_l]] * linear(S, g_w2, g_b2)[pg_l]), g_w3, g_b3, True)
        PF = torch.cat([att1, att2], dim=1)
        # ---- positives / sampled negatives (HEAD:936-963), all images at once
        pos_p, pos_k = torch.nonzero(labels_all).unbind(1)                       # row-major == per-image order
        zero_p, zero_k = torch.nonzero(labels_all == 0).unbind(1)
        neg_cnt = [ppi[a] * K - n_pos[a] for a in range(A)]
        neg_base = np.concatenate([[0], np.cumsum(neg_cnt)])
        sel = torch.cat([perms[a].result() + int(neg_base[a]) for a in range(A)]).to(dev) if A else None
        neg_p, neg_k = zero_p[sel], zero_k[sel]
        pos_img = pair_img[pos_p]
        hrow = ent[:, gh.human_idx]                                              # [A, 50]
        packed = dict(PF=PF, boxes_h=boxes_h_all, boxes_o=boxes_o_all, object=obj_all, prior=prior_all,
                      labels=labels_all, pair_img=pair_img, ppi=ppi, n_pos=n_pos,
                      pos_scores=scores_all[pos_p, pos_k], neg_scores=scores_all[neg_p, neg_k],
                      head_pos=hrow[pos_img], head_neg=hrow[pair_img[neg_p]],
                      tail_pos=ent[pos_img, y_keep[pos_p]], tail_neg=ent[pair_img[neg_p], y_keep[neg_p]],
                      rel_pos=rel[pos_img, pos_k], rel_neg=rel[pair_img[neg_p], neg_k],
                      nrm_pos=nrm[pos_img, pos_k], nrm_neg=nrm[pair_img[neg_p], neg_k])
    # ---- the reference's per-image lists (views of the packed tensors)
    feats, bh, bo, oc, labels_l, prior = [], [], [], [], [], []
    pos_l, neg_l, he_l, te_l, re_l, rn_l = [], [], [], [], [], []
    if packed is not None:
        P_ = packed
        sp = lambda t: t.split(P_["ppi"])
        f_s, bh_s, bo_s, oc_s, lb_s = sp(P_["PF"]), sp(P_["boxes_h"]), sp(P_["boxes_o"]), sp(P_["object"]), sp(P_["labels"])
        pr_s = P_["prior"].split(P_["ppi"], dim=1)
        npl = P_["n_pos"]
        ps, ns = P_["pos_scores"].split(npl), P_["neg_scores"].split(npl)
        hp, hn = P_["head_pos"].split(npl), P_["head_neg"].split(npl)
        tp, tn = P_["tail_pos"].split(npl), P_["tail_neg"].split(npl)
        rp, rn_ = P_["rel_pos"].split(npl), P_["rel_neg"].split(npl)
        np_, nn_ = P_["nrm_pos"].split(npl), P_["nrm_neg"].split(npl)
    a = 0
    for b in range(lay.n_visit):
        if lay.skipped[b]:                                                      # HEAD:829-839
            feats.append(torch.zeros(0, R2, device=dev)); bh.append(torch.zeros(0, 4, device=dev))
            bo.append(torch.zeros(0, 4, device=dev)); oc.append(torch.zeros(0, device=dev, dtype=torch.int64))
            prior.append(torch.zeros(2, 0, K, device=dev)); labels_l.append(torch.zeros(0, K, device=dev))
            continue
        feats.append(f_s[a]); bh.append(bh_s[a]); bo.append(bo_s[a]); oc.append(oc_s[a])
        labels_l.append(lb_s[a]); prior.append(pr_s[a])
        pos_l.append(ps[a]); neg_l.append(ns[a])
        he_l.append(torch.cat((hp[a], hn[a]), 0)); te_l.append(torch.cat((tp[a], tn[a]), 0))       # HEAD:946-955
        re_l.append(torch.cat((rp[a], rn_[a]), 0)); rn_l.append(torch.cat((np_[a], nn_[a]), 0))
        a += 1
    return (feats, bh, bo, oc, labels_l, prior, pos_l, neg_l, he_l, te_l, re_l, rn_l), lay, packed
