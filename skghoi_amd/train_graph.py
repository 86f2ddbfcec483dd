"""Training-mode GraphHead.forward (heads/adamixer_transH_spatial_r50_head.py:769-993) on the HIP device.

Differentiable: every dense layer goes through skghoi_amd.autograd.linear (forward AND backward GEMMs on the fp32-MFMA
kernel); gathers, products, softmax, LayerNorm and the losses are small element-wise torch device ops recorded by
autograd.  The same algebra as the inference engine is used: message passing once, fc_head/fc_tail
and fc_1 on unique node rows, aggregation before the linear fc_3 -- all exact re-associations, so gradients equal the
reference's up to fp32 rounding.

Host RNG is consumed exactly like the reference: per processed image the six TransH draws (HEAD:574-580) and then
`torch.randperm(#negatives)` (HEAD:939).
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import _capi, layout, transh
from .autograd import linear
from .engine import _stream


def _stack_mbf(m):
    w1 = torch.cat([l.weight for l in m.fc_1]); b1 = torch.cat([l.bias for l in m.fc_1])
    w2 = torch.cat([l.weight for l in m.fc_2]); b2 = torch.cat([l.bias for l in m.fc_2])
    w3 = torch.cat([l.weight for l in m.fc_3], dim=1); b3 = torch.stack([l.bias for l in m.fc_3]).sum(dim=0)
    return w1, b1, w2, b2, w3, b3


def box_iou(b1, b2):
    """torchvision.ops.boxes.box_iou (HEAD:711-714), restated with device tensor ops."""
    a1 = (b1[:, 2] - b1[:, 0]) * (b1[:, 3] - b1[:, 1]); a2 = (b2[:, 2] - b2[:, 0]) * (b2[:, 3] - b2[:, 1])
    lt = torch.max(b1[:, None, :2], b2[:, :2]); rb = torch.min(b1[:, None, 2:], b2[:, 2:])
    wh = (rb - lt).clamp(min=0)
    inter = wh[:, :, 0] * wh[:, :, 1]
    return inter / (a1[:, None] + a2 - inter)


def associate_with_ground_truth(boxes_h, boxes_o, target, K, thresh):
    """HEAD:703-719."""
    labels = torch.zeros(boxes_h.shape[0], K, device=boxes_h.device)
    x, y = torch.nonzero(torch.min(box_iou(boxes_h, target["boxes_h"]), box_iou(boxes_o, target["boxes_o"]))
                         >= thresh).unbind(1)
    labels[x, target["labels"][y]] = 1
    return labels


def graph_train(eng, gh, feat3, image_shapes, pooled, pre, targets):
    """Returns the reference's 12 training lists plus the layout."""
    lib = _capi.lib()
    dev = pre.device
    K = gh.num_cls
    st = _stream()
    lay = layout.build(pre.n_h, pre.n, None, image_shapes, gh.human_idx,
                       faithful_skip_offset=eng.faithful_skip_offset)
    A = lay.n_active
    R2 = 2 * gh.representation_size
    feats, bh, bo, oc, labels_l, prior = [], [], [], [], [], []
    pos_l, neg_l, he_l, te_l, re_l, rn_l = [], [], [], [], [], []
    if pooled.shape[0] != lay.sum_all:
        raise _capi.SkgError("box_roi_pool returned %d rows for %d boxes" % (pooled.shape[0], lay.sum_all))
    # HEAD:811-812
    gfeat = F.adaptive_avg_pool2d(feat3.float(), 1).flatten(start_dim=1)
    x0 = pooled.float().flatten(start_dim=1)
    enc = linear(linear(x0, gh.box_head[1].weight, gh.box_head[1].bias, True),
                 gh.box_head[3].weight, gh.box_head[3].bias, True)
    if A:
        buf, offs = layout.pack_int_arrays(lay)
        ibuf = torch.from_numpy(buf).to(dev)

        def isl(name):
            o, l = offs[name]
            return ibuf[o:o + l]

        meta = isl("meta")
        Mg, Mp = lay.sum_g, lay.sum_p
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
        gh_l, go_l, gi_l = grid_h.long(), grid_o.long(), grid_img.long()
        ph_l, po_l, pg_l = pair_h[:Mp].long(), pair_o[:Mp].long(), pair_grid[:Mp].long()
        # ---- labels first (needed to size the randperm), then the RNG draws in the reference's per-image order
        lab_imgs, tabs, perms = [], [], []
        for a in range(A):
            m = lay.meta[a]
            b = int(m["image"]); p0 = int(m["pair_off"]); P = int(m["n_h"]) * (int(m["n"]) - 1)
            b0 = int(m["box_off"]); n = int(m["n"])
            coords = pre.boxes[b0:b0 + n]
            xk = x_keep[p0:p0 + P]; yk = y_keep[p0:p0 + P]
            lab = associate_with_ground_truth(coords[xk], coords[yk], targets[b], K, gh.fg_iou_thresh)
            lab_imgs.append(lab)
        n_pos = [int(torch.count_nonzero(l)) for l in lab_imgs]
        # The global CPU RNG is advanced exactly like the reference (per image: six TransH draws, HEAD:574-580, then
        # randperm(#negatives), HEAD:938-939).  randperm(n) consumes n-1 32-bit draws: the stream is advanced with a
        # cheap random_() of that length and the permutation itself is computed from the saved state by worker threads
        # with private generators, overlapped with the GPU work below (tests pin the equivalence).
        from concurrent.futures import ThreadPoolExecutor

        def _perm(state, n, m):
            g = torch.Generator(); g.set_state(state)
            return torch.randperm(n, generator=g)[:m]

        pool = ThreadPoolExecutor(max_workers=min(8, max(A, 1)))
        for a in range(A):
            tabs.append(transh.draw_tables(K, need_relations=True))
            n_neg = lab_imgs[a].numel() - n_pos[a]
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
            U, V = [], []
            for a in range(A):
                m = lay.meta[a]
                g0, nh, n = int(m["grid_off"]), int(m["n_h"]), int(m["n"])
                adj_a = adj[g0:g0 + nh * n].reshape(nh, n)
                U.append((adj_a.softmax(dim=1)[..., None] * Tos[g0:g0 + nh * n].reshape(nh, n, -1)).sum(dim=1))
                V.append((adj_a.t().softmax(dim=1)[..., None] *
                          Tso[g0:g0 + nh * n].reshape(nh, n, -1).permute(1, 0, 2)).sum(dim=1))
            h_node = F.layer_norm(GH + linear(torch.cat(U), o_w3, o_b3, True), (1024,), gh.norm_h.weight,
                                  gh.norm_h.bias)
            node = F.layer_norm(GO + linear(torch.cat(V), s_w3, s_b3, True), (1024,), gh.norm_o.weight,
                                gh.norm_o.bias)
        else:
            h_node = enc[hum_rows]; node = enc[node_rows]
        # ---- read-out (HEAD:966-973)
        B1h = linear(h_node, a_w1[:, :1024]); B1o = linear(node, a_w1[:, 1024:])
        att1 = linear(F.relu((B1h[ph_l] + B1o[po_l] + a_b1) * F2[pg_l]), a_w3, a_b3, True)
        g_w1, g_b1, g_w2, g_b2, g_w3, g_b3 = _stack_mbf(gh.attention_head_g)
        G1 = linear(gfeat, g_w1, g_b1)
        att2 = linear(F.relu(G1[gi_l[pg_l]] * linear(S, g_w2, g_b2)[pg_l]), g_w3, g_b3, True)
        PF = torch.cat([att1, att2], dim=1)
    a = 0
    for b in range(lay.n_visit):
        if lay.skipped[b]:                                                      # HEAD:829-839
            feats.append(torch.zeros(0, R2, device=dev)); bh.append(torch.zeros(0, 4, device=dev))
            bo.append(torch.zeros(0, 4, device=dev)); oc.append(torch.zeros(0, device=dev, dtype=torch.int64))
            prior.append(torch.zeros(2, 0, K, device=dev)); labels_l.append(torch.zeros(0, K, device=dev))
            continue
        m = lay.meta[a]
        p0 = int(m["pair_off"]); P = int(m["n_h"]) * (int(m["n"]) - 1); b0 = int(m["box_off"]); n = int(m["n"])
        coords = pre.boxes[b0:b0 + n]; lab = pre.labels[b0:b0 + n]; sc = pre.scores[b0:b0 + n]
        xk = x_keep[p0:p0 + P]; yk = y_keep[p0:p0 + P]
        tl = lab_imgs[a]
        px, py = torch.nonzero(tl).unbind(1)
        neg_xy = (tl == 0).nonzero()
        nx, ny = neg_xy[perms[a].result().to(dev)].unbind(1)
        sk = scores_all[p0:p0 + P]
        e_a, r_a, n_a = ent[a], rel[a], nrm[a]
        hrow = e_a[gh.human_idx]
        he_l.append(torch.cat((hrow.expand(len(px), -1), hrow.expand(len(nx), -1)), 0))          # HEAD:946
        te_l.append(torch.cat((e_a[yk[px]], e_a[yk[nx]]), 0))                                      # HEAD:949
        re_l.append(torch.cat((r_a[py], r_a[ny]), 0)); rn_l.append(torch.cat((n_a[py], n_a[ny]), 0))
        labels_l.append(tl); pos_l.append(sk[px, py]); neg_l.append(sk[nx, ny])
        feats.append(PF[p0:p0 + P]); bh.append(coords[xk]); bo.append(coords[yk]); oc.append(lab[yk])
        prior.append(gh.compute_prior_scores(xk, yk, sc, lab))
        a += 1
    return (feats, bh, bo, oc, labels_l, prior, pos_l, neg_l, he_l, te_l, re_l, rn_l), lay
