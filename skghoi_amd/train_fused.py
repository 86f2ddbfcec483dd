"""The fused TRAINING step of the interaction head: forward and hand-written backward on the HIP kernels.

Reference: InteractionHead.forward in training mode (heads/adamixer_transH_spatial_r50_head.py:380-429) over
GraphHead.forward (HEAD:769-993) and the losses (HEAD:153-235, ops.py:159-211), with autograd doing the backward.  The
first MI355X version kept autograd and only replaced nn.Linear (skghoi_amd/train_graph.py + autograd.py): ~1000 kernel
launches per step (gathers, products, masks, transposes, segment softmax, LayerNorm, the losses), host-bound at 13 ms for
four images.  Here the differentiable part of the step is ONE autograd.Function:

    forward   ~32 launches: every dense layer on skg_gemmx_f32 (precision "bf16": skg_gemmx_bf16 -- operands rounded to
              bf16 on the way to the matrix core, everything in memory fp32) / skg_gemm_f32 (MBF fc_1 * fc_2 fused in the
              fc_2 epilogue; the 16 branch weights of every MBF are gathered once per step into stacked copies, fc_3
              branch-major), message aggregation, add + LayerNorm, read-out, classifier
    backward  ~50 launches: dX and dW (+ bias gradient as row sums) of a layer in ONE grouped launch with the ReLU mask
              of the layer below in its epilogue -- no transposes, no mask kernels; the graph stages by the kernels of
              skg_train.hip (per-destination reductions, deterministic)
    losses    one kernel: both focal terms forward + d/dlogits (skg_hoi_loss_f32)

The same algebra as the inference engine (message passing once, fc_head / fc_tail / fc_1 on unique
node rows, aggregation before the linear fc_3), so gradients equal the reference's up to fp32 summation order.
Parameter gradients are returned to autograd as views of one gradient arena laid out like the stacked copies (16
branches = 16 contiguous slices), so `loss.backward()`, DDP and the optimizer see the reference's 408 parameters.
"""
import ctypes as C
import os
import threading
import weakref
from operator import attrgetter, is_ as _is

import numpy as np
import torch

from . import _capi, gemmx, layout
from .engine import _ptr, _stream, current_stream_of, gemm, gemm_desc

_grad_of = attrgetter("grad")
MBF_NAMES = ("attention_head", "obj_to_sub", "sub_to_obj", "attention_head_g")      # order of the stacked fc_2 block
ATT, OS, SO, GL = range(4)
EPS_LN = 1e-5
# SKG_INLAUNCH_REDUCE=1: split-K of the plan's products reduced inside the product launch (skg_gemmx_desc.split_ctr) instead
# of by a second launch per split product.  Bit-identical results, 14 launches less per batch-4 step -- and SLOWER on MI355X
# (round 5, profiles/r05_inlaunch_splitk_*): the tile's last arriver drains its write-through partials, takes the counter's
# round trip and reads S slices at one workgroup's memory-level parallelism (10-17 us on top of the product's own time),
# while the reduce launch it replaces costs ~5 us and back-to-back launches of one stream leave no gap between them:
# bf16 step 1.26 -> 1.37 ms, fp32 2.47 -> 2.59.  Opt-in.
INLAUNCH_SPLIT_REDUCE = os.environ.get("SKG_INLAUNCH_REDUCE", "0") == "1"
# SKG_TWO_BRANCH=1: the bf16 plan's node chain (fc_head / fc_tail, fc_1 projections; from backward stage 6 on their gradients
# and box_head's) on a second stream beside the spatial chain (skg_train_plan.two_branch).  Same kernels, same results -- and
# no faster on MI355X (round 5, profiles/r05_two_branch_*): the chains do overlap, but the spatial chain's products fill every
# CU (456-800 workgroups at two per CU), so a node-chain launch beside them waits for CUs to drain (stage 7's node products:
# 105 us beside stage 6 instead of 20 alone) and the big products stretch (61 vs 49 us) -- what the overlap hides, contention
# and seven more launches give back: 1.24-1.25 vs 1.22-1.27 ms.  Opt-in.
TWO_BRANCH = os.environ.get("SKG_TWO_BRANCH", "0") == "1"


def _check(rc, what):
    _capi.check(rc, what)


# Plain (un-stacked) parameters of the head, by arena segment name: (owner path, attribute)
_PLAIN = (("nh_w", ("norm_h", "weight")), ("nh_b", ("norm_h", "bias")), ("no_w", ("norm_o", "weight")),
          ("no_b", ("norm_o", "bias")), ("adj_w", ("adjacency", "weight")), ("adj_b", ("adjacency", "bias")),
          ("sp4_w", ("spatial_head", 4, "weight")), ("sp4_b", ("spatial_head", 4, "bias")),
          ("sp2_w", ("spatial_head", 2, "weight")), ("sp2_b", ("spatial_head", 2, "bias")),
          ("fh_w", ("fc_head", 0, "weight")), ("fh_b", ("fc_head", 0, "bias")),
          ("ft_w", ("fc_tail", 0, "weight")), ("ft_b", ("fc_tail", 0, "bias")),
          ("sp0_w", ("spatial_head", 0, "weight")), ("sp0_b", ("spatial_head", 0, "bias")),
          ("bh3_w", ("box_head", 3, "weight")), ("bh3_b", ("box_head", 3, "bias")),
          ("bh1_w", ("box_head", 1, "weight")), ("bh1_b", ("box_head", 1, "bias")))

# Arena order = the order in which the backward FINISHES the gradients (read-out layers first, box_head last): after
# milestone m of TrainJob.backward the arena prefix [0, milestone_end[m]) is final, which is what lets the gradient
# exchange of a data-parallel step run chunk by chunk behind the backward (skghoi_amd/trainer.py, ArenaExchange).
_ORDER = (("clsW", "clsb"), ("W3_3",), ("nh_w", "nh_b", "no_w", "no_b"), ("W3_1", "W3_2"), ("W3_0", "adj_w", "adj_b"),
          ("b1_0",), ("W2", "b2"), ("W1_0", "sp4_w", "sp4_b"), ("W1_2", "W1_1", "b1_2", "b1_1", "sp2_w", "sp2_b"),
          ("fh_w", "fh_b", "ft_w", "ft_b", "sp0_w", "sp0_b", "W1_3", "b1_3"), ("bh3_w", "bh3_b"), ("bh1_w", "bh1_b", "b3"))


def _resolve(gh, path):
    o = gh
    for k in path:
        o = o[k] if isinstance(k, int) else getattr(o, k)
    return o


class Stacked:
    """The PARAMETER ARENA of the head: one flat fp32 buffer holding all 408 parameters in the layout the kernels of the
    training step read, and the live Parameters re-pointed into it (`p.data` = a view of the arena).  Nothing is copied
    per step: the optimizer updates the arena in place, the GEMMs read it.  A gradient arena with the same layout gives
    every `p.grad` as a view, so that the optimizer and the data-parallel gradient exchange work on two flat buffers.

    Segments:  per MBF m (order attention_head, obj_to_sub, sub_to_obj, attention_head_g): W1_m [1024, in_m] (16 x
    [64, in_m] row blocks = fc_1[b].weight), b1_m [1024], W3_m [16][1024][64] (branch-major: branch b is the contiguous
    fc_3[b].weight); b3 [4][16][1024]; W2 [4 x 1024, 1024] and b2 [4 x 1024] of all four MBFs back to back (one GEMM
    for their input gradient); clsW [K + 1, 2048] = predictor rows then the suppressor row, clsb [K + 1]; every other
    parameter (box_head, spatial_head, fc_head / fc_tail, adjacency, LayerNorms) as a segment of its own shape.
    Segment order: _ORDER (gradient readiness).

    The aliasing is re-checked every step (408 data_ptr reads, ~40 us): `module.to()`, `p.data = t` or a replaced
    Parameter re-point storage; the arena then takes the new values and the parameters are re-pointed into it."""

    def __init__(self, head, device):
        gh = head.box_pair_head
        self.device = device
        self.K = K = head.num_classes
        mbfs = [getattr(gh, n) for n in MBF_NAMES]
        self.in_dim = [m.fc_1[0].weight.shape[1] for m in mbfs]
        shapes = {}
        for i in range(4):
            shapes["W1_%d" % i] = (1024, self.in_dim[i]); shapes["b1_%d" % i] = (1024,)
            shapes["W3_%d" % i] = (16, 1024, 64)
        shapes.update(b3=(4, 16, 1024), W2=(4096, 1024), b2=(4096,), clsW=(K + 1, 2048), clsb=(K + 1,))
        plain = {}
        for name, path in _PLAIN:
            plain[name] = _resolve(gh, path)
            shapes[name] = tuple(plain[name].shape)
        seg, off, ends = {}, 0, []
        for group in _ORDER:
            for name in group:
                n = int(np.prod(shapes[name]))
                seg[name] = (off, shapes[name])
                off += (n + 7) // 8 * 8                        # every segment 32-byte aligned (its bf16 twin: 16 bytes)
            ends.append(off)
        assert set(seg) == set(shapes)
        self.seg, self.total, self.milestone_end = seg, off, ends
        self.numel = {k: int(np.prod(v[1])) for k, v in seg.items()}
        self.buf = torch.zeros(off, device=device, dtype=torch.float32)
        # (live parameter, segment, selector inside the segment's view) in a fixed order
        self.entries = []
        for i, m in enumerate(mbfs):
            for b in range(16):
                self.entries.append((m.fc_1[b].weight, "W1_%d" % i, (slice(64 * b, 64 * b + 64),)))
                self.entries.append((m.fc_1[b].bias, "b1_%d" % i, (slice(64 * b, 64 * b + 64),)))
                self.entries.append((m.fc_2[b].weight, "W2", (slice(1024 * i + 64 * b, 1024 * i + 64 * b + 64),)))
                self.entries.append((m.fc_2[b].bias, "b2", (slice(1024 * i + 64 * b, 1024 * i + 64 * b + 64),)))
                self.entries.append((m.fc_3[b].weight, "W3_%d" % i, (b,)))
                self.entries.append((m.fc_3[b].bias, "b3", (i, b)))
        self.entries.append((head.box_pair_predictor.weight, "clsW", (slice(0, K),)))
        self.entries.append((head.box_pair_suppressor.weight, "clsW", (slice(K, K + 1),)))
        self.entries.append((head.box_pair_predictor.bias, "clsb", (slice(0, K),)))
        self.entries.append((head.box_pair_suppressor.bias, "clsb", (slice(K, K + 1),)))
        self.n_stacked = len(self.entries)
        for name, _ in _PLAIN:
            self.entries.append((plain[name], name, ()))
        self.src = [e[0] for e in self.entries]                       # the LIVE parameters
        self.dst = [self.view(self.buf, e[1])[e[2]] for e in self.entries]
        self.dst_ptrs = [t.data_ptr() for t in self.dst]
        self.ids = {id(p): k for k, p in enumerate(self.src)}
        self.adoptions = 0

    def view(self, flat, name):
        off, shape = self.seg[name]
        return flat[off:off + self.numel[name]].view(shape)

    def seg_off_array(self):
        """int64 offsets (floats) of the arena segments in SKG_SEG_* order (include/skghoi.h)."""
        a = getattr(self, "_seg_off", None)
        if a is None:
            a = self._seg_off = (C.c_int64 * len(_capi.TRAIN_SEGS))(*[self.seg[n][0] for n in _capi.TRAIN_SEGS])
        return a

    def aliased(self):
        return [p.data_ptr() for p in self.src] == self.dst_ptrs

    def adopt(self):
        """Copies the live parameter values into the arena and re-points every Parameter into it."""
        with torch.no_grad():
            torch._foreach_copy_(self.dst, [p.detach() for p in self.src])
            for p, d in zip(self.src, self.dst):
                p.data = d
                p._skg_arena = self          # (the optimizer's plan: this parameter's storage is checked by the step's forward)
        self.adoptions += 1

    def grad_arena(self):
        """(arena, per-entry views) for one backward.  The arena of the previous step and its 408 view objects are reused
        when nothing else holds them any more -- zero_grad(set_to_none=True) has dropped every p.grad (C++ side: the
        tensors' use counts) and no caller kept a gradient tensor (Python side: the reference counts of the cached view
        objects) -- which saves ~0.4 ms of view construction per step; otherwise (gradient accumulation, a caller
        holding on to a gradient, autograd still owning one) a fresh arena is made."""
        c = getattr(self, "_ga", None)
        if c is not None and self._holders(c[1]) == c[2]:
            return c[0], c[1]
        ga = torch.empty(self.total, device=self.device, dtype=torch.float32)
        views = self.grad_views(ga)
        self._ga = (ga, views, self._holders(views))
        return ga, views

    def scratch(self, name, n, dtype, dev):
        """A buffer of at least n elements that lives with this arena and is handed out again to the next step (grown when a
        step needs more).  Only for memory whose every use is ordered on the step's stream."""
        pool = self.__dict__.setdefault("_scratch", {})
        t = pool.get(name)
        if t is None or t.numel() < n or t.dtype != dtype or t.device != dev:
            t = pool[name] = torch.empty(max(n, 1), device=dev, dtype=dtype)
        return t

    def counters(self):
        """Tile counters of the plan's split-K products (include/skghoi.h, skg_train_plan.counters): zero when made, and every
        launch that uses them leaves them zero -- so ONE array serves every step of this arena's stream."""
        t = getattr(self, "_ctr", None)
        if t is None:
            t = self._ctr = torch.zeros(1 << 14, device=self.device, dtype=torch.int32)
        return t

    def twin(self):
        """bf16 twin of the parameter arena (same element offsets), rewritten by the forward's first part every bf16 step."""
        t = getattr(self, "_twin", None)
        if t is None:
            t = self._twin = torch.empty(self.total, device=self.device, dtype=torch.int16)
        return t

    def persistent_grads(self):
        """(arena, views) kept for the life of this Stacked: the gradient home of `fused_step`, which OVERWRITES every
        gradient each step (nothing accumulates across steps there), so the views can stay assigned to `p.grad`."""
        c = getattr(self, "_pga", None)
        if c is None:
            ga = torch.zeros(self.total, device=self.device, dtype=torch.float32)
            c = self._pga = (ga, self.grad_views(ga))
        return c

    @staticmethod
    def _holders(views):
        import sys
        return sum(map(sys.getrefcount, views)) + sum(map(torch.Tensor._use_count, views))

    def refresh(self):
        if not self.aliased():
            self.adopt()
        self.b3sum = self.view(self.buf, "b3").sum(dim=1)                 # [4, 1024]: fc_3 biases summed over branches

    def grad_views(self, garena):
        """Views of a gradient arena (same layout) for all parameters, in `entries` order.  Built with one unbind
        per (MBF, kind) -- 16 branch slices per call -- instead of 388 separate slicing calls."""
        K = self.K
        W2 = self.view(garena, "W2").view(4, 16, 64, 1024); b2 = self.view(garena, "b2").view(4, 16, 64)
        b3 = self.view(garena, "b3")
        out = []
        for i in range(4):
            w1 = self.view(garena, "W1_%d" % i).view(16, 64, self.in_dim[i]).unbind(0)
            c1 = self.view(garena, "b1_%d" % i).view(16, 64).unbind(0)
            w2 = W2[i].unbind(0); c2 = b2[i].unbind(0)
            w3 = self.view(garena, "W3_%d" % i).unbind(0); c3 = b3[i].unbind(0)
            for b in range(16):
                out += [w1[b], c1[b], w2[b], c2[b], w3[b], c3[b]]
        cw, cb = self.view(garena, "clsW"), self.view(garena, "clsb")
        out += [cw[:K], cw[K:K + 1], cb[:K], cb[K:K + 1]]
        out += [self.view(garena, name) for name, _ in _PLAIN]
        return out


def _lin(x, W, out, bias=None, relu=False, **kw):
    return gemmx.forward(x, W, out, bias=bias, relu=relu, **kw)


class StepFn(torch.autograd.Function):
    """The whole differentiable part of the training step as ONE autograd node:
    (pooled box features, global features, parameters) -> the three loss scalars (HEAD:419-427).

    forward: dense forward (job.forward) -> scoring (HEAD:721-767, 237-337) -> both focal terms + d(sum)/d(logits) in one
    kernel -> TransH scores, sampling, margin term -> normalisers (one fused all-reduce when data parallel) -> the three
    scalars.  backward: d(total)/d(logits) = d(sum)/d(logits) * (upstream gradient / n_p) per loss column (one kernel),
    then the hand-written backward (job.backward).  The TransH term reaches only the step's throw-away embeddings
    (SURVEY 8a-17): it has no gradient towards the inputs.

    grad_mode "autograd" (default): the parameters are inputs of the Function and their gradients are returned to the
    autograd engine (hooks, DDP and torch.autograd.grad see them).  grad_mode "direct": the Function's only parameter-side
    input is a one-element anchor; the backward writes `p.grad` itself (assign, or add to an existing gradient) -- the
    engine's per-leaf bookkeeping for 408 parameters costs ~1 ms of host time per step; the single-process trainer
    (skghoi_amd/trainer.py) switches it on."""

    @staticmethod
    def forward(ctx, run, prep, x0, gfeat, *params):
        job = run.job
        S = job.forward(x0, gfeat)
        losses = run.tail(prep, S["logits"])
        ctx.job = job
        ctx.in_meta = ((x0.shape, x0.dtype), (gfeat.shape, gfeat.dtype))
        return losses[0], losses[1], losses[2]

    @staticmethod
    def backward(ctx, g0, g1, g2):
        job = ctx.job
        if job.S is None:
            raise RuntimeError("Trying to backward through the fused interaction-head step a second time: its saved "
                               "activations are freed by the first backward (retain_graph is not supported by the "
                               "fused step; use head.fused_training = False for repeated backward passes)")
        src = job.dlogits                                  # d(sum of the focal terms)/d(logits): stays as the kernel left it
        d = torch.empty_like(src)
        z = None
        if g0 is None or g1 is None:
            z = torch.zeros(1, device=src.device, dtype=torch.float32)
        g0 = z if g0 is None else g0.reshape(-1)[:1].float()
        g1 = z if g1 is None else g1.reshape(-1)[:1].float()
        _check(_capi.lib().skg_scale_dlogits_f32(src.data_ptr(), src.stride(0), src.shape[0], job.K,
                                                 job.loss_scale.data_ptr(), g0.data_ptr(), g1.data_ptr(), d.data_ptr(),
                                                 _stream()), "skg_scale_dlogits_f32")
        dx0, dgfeat, pgrads = job.backward(d, ctx.needs_input_grad[2], ctx.needs_input_grad[3])
        # the step works on flattened fp32 copies of its inputs: hand the gradients back in the inputs' own shape / dtype
        # (pooled box features arrive as [N, 256, 7, 7] when the RoI pooling in front is differentiable)
        (s0, t0), (s1, t1) = ctx.in_meta
        if dx0 is not None:
            dx0 = dx0.reshape(s0).to(t0)
        if dgfeat is not None:
            dgfeat = dgfeat.reshape(s1).to(t1)
        if job.direct:
            for p, g in zip(job.params, pgrads):
                if g is None or not p.requires_grad:
                    continue
                if p.grad is None:
                    p.grad = g
                else:
                    p.grad.add_(g)
            return None, None, dx0, dgfeat, torch.zeros_like(job.anchor)
        return (None, None, dx0, dgfeat) + tuple(pgrads)


class TrainJob:
    """One training forward / backward of the dense part of the head on a packed batch."""

    def __init__(self, head, eng, stacked, lay, pre, ibuf, offs, ent, image_meta_dev):
        self.head, self.eng, self.st = head, eng, stacked
        self.gh = head.box_pair_head
        self.lay, self.pre = lay, pre
        self.ibuf, self.offs = ibuf, offs
        self.ent = ent
        self.meta = image_meta_dev
        self.K = head.num_classes
        self.dev = pre.device
        self.S = {}
        # precision="bf16": every dense layer of the step (forward, dX, dW) rounds its operands to bf16 on the way to
        # the matrix core and accumulates in fp32 -- the reference under torch.autocast(bfloat16); tensors stay fp32.
        self.bf16 = head.precision == "bf16"
        self.twins = getattr(head, "bf16_twins", True) and os.environ.get("SKG_BF16_TWINS", "1") != "0"

    def gx(self, ops):
        gemmx.launch(ops, bf16=self.bf16)

    def isl(self, name):
        o, l = self.offs[name]
        return self.ibuf[o:o + l]

    # ------------------------------------------------------------------------------------------------ forward
    def forward_a(self, x0, gfeat):
        """The part of the forward that needs neither the TransH tables nor the label counts: parameter stacking,
        box_head, the global branch's fc_1 and the spatial head.  The step driver enqueues it BEFORE its host
        synchronisation, so that the GPU works through it while the host draws the tables."""
        st, gh, lay, dev, S = self.st, self.gh, self.lay, self.dev, self.S
        f32 = dict(device=dev, dtype=torch.float32)
        NA, Mg = lay.sum_all, lay.sum_g
        st.refresh()
        W1g, b1g = st.view(st.buf, "W1_%d" % GL), st.view(st.buf, "b1_%d" % GL)
        x0 = x0.detach().float().reshape(x0.shape[0], -1).contiguous()
        gfeat = gfeat.detach().float().contiguous()
        S["x0"], S["gfeat"] = x0, gfeat
        Bf = gfeat.shape[0]
        bh1, bh3 = gh.box_head[1], gh.box_head[3]
        # ---- box_head (HEAD:812), fc_1 of the global branch (HEAD:971) and the first two layers of the spatial head
        #      (HEAD:888, on the 46-d encodings the step driver produced with the pair enumeration): two grouped launches
        E1 = torch.empty(NA, 1024, **f32); enc = torch.empty(NA, 1024, **f32); G1 = torch.empty(Bf, 1024, **f32)
        sp48 = S["sp48"]
        sp = gh.spatial_head
        s1 = torch.empty(Mg, 128, **f32); s2 = torch.empty(Mg, 256, **f32)
        self.gx([_lin(x0, bh1.weight, E1, bh1.bias, True), _lin(sp48, sp[0].weight, s1, sp[0].bias, True, K=46)])
        self.gx([_lin(E1, bh3.weight, enc, bh3.bias, True), _lin(gfeat, W1g, G1, b1g),
                 _lin(s1, sp[2].weight, s2, sp[2].bias, True)])
        S.update(E1=E1, enc=enc, G1=G1, s1=s1, s2=s2)
        self.part_a_done = True

    def forward(self, x0, gfeat):
        lib = _capi.lib()
        st, gh, lay, dev = self.st, self.gh, self.lay, self.dev
        f32 = dict(device=dev, dtype=torch.float32)
        i32 = dict(device=dev, dtype=torch.int32)
        stream = _stream()
        S = self.S
        NA, Mg, Mp, Mh, Mn, A = lay.sum_all, lay.sum_g, lay.sum_p, lay.sum_h, lay.sum_n, lay.n_active
        K = self.K
        if not getattr(self, "part_a_done", False):
            self.forward_a(x0, gfeat)
        W2, b2 = st.view(st.buf, "W2"), st.view(st.buf, "b2")
        W1 = [st.view(st.buf, "W1_%d" % i) for i in range(4)]
        b1 = [st.view(st.buf, "b1_%d" % i) for i in range(4)]
        W3 = [st.view(st.buf, "W3_%d" % i) for i in range(4)]
        b3 = st.b3sum
        blk = (6, 1024 * 64)
        x0, gfeat = S["x0"], S["gfeat"]
        Bf = gfeat.shape[0]
        E1, enc, G1 = S["E1"], S["enc"], S["G1"]
        s1, s2 = S["s1"], S["s2"]
        grid_h, grid_o, grid_pair, grid_img, pair_grid, pair_h, pair_o, sp48 = (
            S[k] for k in ("grid_h", "grid_o", "grid_pair", "grid_img", "pair_grid", "pair_h", "pair_o", "sp48"))
        # ---- fc_head / fc_tail on unique node rows (HEAD:884-885)
        Xhn = torch.empty(Mh + Mn, 1088, **f32)
        _check(lib.skg_concat_entity_f32(enc.data_ptr(), 1024, self.isl("enc_row_hn").data_ptr(), self.ent.data_ptr(),
                                         self.isl("img_hn").data_ptr(), self.isl("ent_row_hn").data_ptr(), Mh + Mn,
                                         Xhn.data_ptr(), 1088, stream), "skg_concat_entity_f32")
        GH = torch.empty(Mh, 1024, **f32); GO = torch.empty(Mn, 1024, **f32); Sp = torch.empty(Mg, 1024, **f32)
        fh, ft = gh.fc_head[0], gh.fc_tail[0]
        sp = gh.spatial_head
        self.gx([_lin(Xhn[:Mh], fh.weight, GH, fh.bias, True, K=1074), _lin(Xhn[Mh:], ft.weight, GO, ft.bias, True, K=1074),
                 _lin(s2, sp[4].weight, Sp, sp[4].bias, True)])           # + the last spatial layer (HEAD:888)
        S.update(Xhn=Xhn, GH=GH, GO=GO, Sp=Sp)
        # ---- fc_1 projections on node rows (HEAD:894-896 separable over [human | object]; HEAD:514, 524)
        A1h = torch.empty(Mh, 1024, **f32); A1o = torch.empty(Mn, 1024, **f32)
        C1o = torch.empty(Mn, 1024, **f32); C1h = torch.empty(Mh, 1024, **f32)
        Wa1 = W1[ATT]
        self.gx([_lin(GH, Wa1, A1h, K=1024), _lin(GO, Wa1[:, 1024:], A1o, K=1024),
                      _lin(GO, W1[OS], C1o, b1[OS]), _lin(GH, W1[SO], C1h, b1[SO])])
        S.update(A1h=A1h, A1o=A1o, C1o=C1o, C1h=C1h)
        # ---- fc_2 on the grid rows with the fc_1 * fc_2 -> ReLU product in the epilogue; the raw fc_2 output is kept
        F = torch.empty(Mg, 4096, **f32)                     # [F2 | F_os | F_so | F_g], leading dimension 4096
        T = torch.empty(Mg, 1024, **f32); Tos = torch.empty(Mg, 1024, **f32); Tso = torch.empty(Mg, 1024, **f32)
        Tg = torch.empty(max(Mp, 1), 1024, **f32)

        def fc2(i, out, **kw):
            gemm(Sp, W2, b2[1024 * i:1024 * i + 1024], out, Mg, 1024, 1024, _capi.EPI_MUL_RELU, W_off=1024 * 1024 * i,
                 C_raw=F[:, 1024 * i:], ldc_raw=4096, **kw)
        def prod(P, p_idx, Q, q_idx, mbias, i, f_idx, rows, out):
            _check(lib.skg_rows_mul_relu_f32(P.data_ptr(), p_idx.data_ptr(), 1024, _ptr(Q), _ptr(q_idx), 1024, _ptr(mbias),
                                             F.data_ptr() + 4 * 1024 * i, _ptr(f_idx), 4096, rows, 1024, out.data_ptr(),
                                             1024, stream), "skg_rows_mul_relu_f32")
        if self.bf16:
            # all four fc_2 as ONE product (N = 4096) on the bf16 matrix pipe, then the fc_1 * fc_2 -> ReLU rows
            self.gx([_lin(Sp, W2, F, b2)])
            prod(A1h, grid_h, A1o, grid_o, b1[ATT], ATT, None, Mg, T)
            prod(C1o, grid_o, None, None, None, OS, None, Mg, Tos)
            prod(C1h, grid_h, None, None, None, SO, None, Mg, Tso)
            prod(G1, self.pair_img, None, None, None, GL, pair_grid, Mp, Tg)
        else:
            fc2(ATT, T, P=A1h, p_idx=grid_h, ldp=1024, Q=A1o, q_idx=grid_o, ldq=1024, mbias=b1[ATT])
            fc2(OS, Tos, P=C1o, p_idx=grid_o, ldp=1024)
            fc2(SO, Tso, P=C1h, p_idx=grid_h, ldp=1024)
            fc2(GL, Tg, P=G1, p_idx=grid_img, ldp=1024, out_rows=grid_pair)
        S.update(F=F, T=T, Tos=Tos, Tso=Tso, Tg=Tg)
        # ---- attention fc_3 + ReLU, adjacency logits (HEAD:896-897)
        Wt = torch.empty(Mg, 1024, **f32)
        self.gx([_lin(T, W3[ATT], Wt, b3[ATT], True, w_blocks=blk)])
        adj_raw = torch.empty(Mg, **f32)
        wadj = gh.adjacency.weight.detach().reshape(-1)
        _check(lib.skg_rowdot_f32(Wt.data_ptr(), 1024, wadj.data_ptr(), Mg, 1024, adj_raw.data_ptr(), stream),
               "skg_rowdot_f32")
        # ---- softmax-weighted aggregation before the linear fc_3 (HEAD:907-922)
        U = torch.empty(Mh, 1024, **f32); V = torch.empty(Mn, 1024, **f32)
        adj = torch.empty(Mg, **f32); alpha = torch.empty(Mg, **f32); beta = torch.empty(Mg, **f32)
        # (the adjacency bias shifts every logit of a softmax alike: it cancels in alpha / beta, and its gradient -- the sum of
        #  a softmax gradient -- is zero up to rounding, which is what the GEMM row sum below returns)
        _check(lib.skg_graph_aggregate_train_f32(adj_raw.data_ptr(), 1, Mg, 0.0, self.meta.data_ptr(), A,
                                                 self.isl("hum_img").data_ptr(), self.isl("node_img").data_ptr(), Mh, Mn,
                                                 Tos.data_ptr(), Tso.data_ptr(), 1024, 1024, U.data_ptr(), V.data_ptr(),
                                                 1024, adj.data_ptr(), alpha.data_ptr(), beta.data_ptr(), stream),
               "skg_graph_aggregate_train_f32")
        S.update(Wt=Wt, U=U, V=V, alpha=alpha, beta=beta, adj=adj)
        # ---- message fc_3 + ReLU, residual, LayerNorm (HEAD:909-914, 916-925)
        M1 = torch.empty(Mh, 1024, **f32); M2 = torch.empty(Mn, 1024, **f32)
        self.gx([_lin(U, W3[OS], M1, b3[OS], True, w_blocks=blk), _lin(V, W3[SO], M2, b3[SO], True, w_blocks=blk)])
        Hp = torch.empty(Mh, 1024, **f32); h_node = torch.empty(Mh, 1024, **f32); st_h = torch.empty(Mh, 2, **f32)
        Op = torch.empty(Mn, 1024, **f32); node = torch.empty(Mn, 1024, **f32); st_o = torch.empty(Mn, 2, **f32)
        nh, no = gh.norm_h, gh.norm_o
        _check(lib.skg_add_layernorm_f32(GH.data_ptr(), 1024, M1.data_ptr(), 1024, nh.weight.data_ptr(), nh.bias.data_ptr(),
                                         Mh, EPS_LN, Hp.data_ptr(), h_node.data_ptr(), st_h.data_ptr(), stream),
               "skg_add_layernorm_f32")
        _check(lib.skg_add_layernorm_f32(GO.data_ptr(), 1024, M2.data_ptr(), 1024, no.weight.data_ptr(), no.bias.data_ptr(),
                                         Mn, EPS_LN, Op.data_ptr(), node.data_ptr(), st_o.data_ptr(), stream),
               "skg_add_layernorm_f32")
        S.update(M1=M1, M2=M2, Hp=Hp, Op=Op, h_node=h_node, node=node, st_h=st_h, st_o=st_o)
        # ---- read-out on the kept pairs (HEAD:966-973)
        B1h = torch.empty(Mh, 1024, **f32); B1o = torch.empty(Mn, 1024, **f32)
        self.gx([_lin(h_node, Wa1, B1h, K=1024), _lin(node, Wa1[:, 1024:], B1o, K=1024)])
        Tp = torch.empty(max(Mp, 1), 1024, **f32)
        _check(lib.skg_rows_mul_relu_f32(B1h.data_ptr(), pair_h.data_ptr(), 1024, B1o.data_ptr(), pair_o.data_ptr(), 1024,
                                         b1[ATT].data_ptr(), F.data_ptr(), pair_grid.data_ptr(), 4096, Mp, 1024,
                                         Tp.data_ptr(), 1024, stream), "skg_rows_mul_relu_f32")
        PF = torch.empty(max(Mp, 1), 2048, **f32)
        self.gx([_lin(Tp, W3[ATT], PF, b3[ATT], True, M=Mp, N=1024, w_blocks=blk),
                      _lin(Tg, W3[GL], PF[:, 1024:], b3[GL], True, M=Mp, N=1024, w_blocks=blk)])
        # ---- classifier: predictor | suppressor as one product (HEAD:410-411)
        ld = (K + 1 + 3) // 4 * 4
        logits = torch.zeros(max(Mp, 1), ld, **f32)
        self.gx([_lin(PF, st.view(st.buf, "clsW"), logits, st.view(st.buf, "clsb"), M=Mp, N=K + 1)])
        S.update(B1h=B1h, B1o=B1o, Tp=Tp, PF=PF, logits=logits[:Mp])
        return S

    # ------------------------------------------------------------------------------------------------ losses
    def loss_forward(self, logits):
        """Both focal terms summed (HEAD:162-165, 190-192 before the division by n_p) + d/dlogits, from the packed result
        arrays `self.result` (skg_postprocess_f32 on these logits) and the label matrix `self.labels`."""
        lib = _capi.lib()
        lay, dev, K = self.lay, self.dev, self.K
        r = self.result
        Mp, Lt = lay.sum_p, lay.sum_l
        f32 = dict(device=dev, dtype=torch.float32)
        self.dlogits = torch.zeros_like(logits)
        self.cell_labels = torch.empty(max(Lt, 1), **f32)
        self.unary = torch.empty(max(Mp, 1), **f32)
        partial = torch.empty(lay.n_active * _capi.LOSS_CHUNKS, 4, **f32)
        _check(lib.skg_hoi_loss_f32(logits.data_ptr(), logits.stride(0), K, self.meta.data_ptr(), lay.n_active, Lt,
                                    r["index"].data_ptr(), r["prediction"].data_ptr(), r["scores"].data_ptr(),
                                    self.labels.data_ptr(), self.cell_labels.data_ptr(), self.unary.data_ptr(),
                                    partial.data_ptr(), self.dlogits.data_ptr(), _stream()), "skg_hoi_loss_f32")
        self.partial = partial                            # rows of {cell loss, pair loss, #positive cells, #positive pairs}
        return partial

    # ------------------------------------------------------------------------------------------------ backward
    def backward(self, dlogits, need_dx0, need_dgfeat):
        lib = _capi.lib()
        st, gh, lay, dev, S = self.st, self.gh, self.lay, self.dev, self.S
        f32 = dict(device=dev, dtype=torch.float32)
        stream = _stream()
        NA, Mg, Mp, Mh, Mn, A = lay.sum_all, lay.sum_g, lay.sum_p, lay.sum_h, lay.sum_n, lay.n_active
        K = self.K
        blk = (6, 1024 * 64)
        IG, WG = gemmx.input_grad, gemmx.weight_grad
        ga, sviews = st.grad_arena()                            # gradient arena of the stacked parameters + p.grad views
        gv = lambda name: st.view(ga, name)
        W2 = st.view(st.buf, "W2")
        W1 = [st.view(st.buf, "W1_%d" % i) for i in range(4)]
        b1 = [st.view(st.buf, "b1_%d" % i) for i in range(4)]
        W3 = [st.view(st.buf, "W3_%d" % i) for i in range(4)]
        dW1 = [gv("W1_%d" % i) for i in range(4)]
        db1 = [gv("b1_%d" % i) for i in range(4)]
        dW3 = [gv("W3_%d" % i) for i in range(4)]
        db3 = torch.empty(4, 1024, **f32)                        # one bias gradient per MBF, replicated over its branches
        dlogits = dlogits.contiguous()
        ld = dlogits.stride(0)
        PF, Tp, Tg = S["PF"], S["Tp"], S["Tg"]
        # ---- classifier
        dPF = torch.empty(max(Mp, 1), 2048, **f32)
        self.gx([IG(dlogits, st.view(st.buf, "clsW"), dPF, mask=PF, M=Mp, N_in=2048, K_out=K + 1),
                      WG(dlogits, PF, gv("clsW"), db=gv("clsb"), rows=Mp, n_out=K + 1, k_in=2048)])
        # ---- read-out fc_3 (both branches): dT = dPF W3 cut by the product's ReLU; dW3 = dPF^T T
        dTp = torch.empty(max(Mp, 1), 1024, **f32); dTg = torch.empty(max(Mp, 1), 1024, **f32)
        self.gx([IG(dPF, W3[ATT], dTp, mask=Tp, M=Mp, N_in=1024, K_out=1024, w_blocks=blk),
                      IG(dPF[:, 1024:], W3[GL], dTg, mask=Tg, M=Mp, N_in=1024, K_out=1024, w_blocks=blk),
                      WG(dPF, Tp, dW3[ATT], db=db3[ATT], rows=Mp, n_out=1024, k_in=1024, w_blocks=blk),
                      WG(dPF[:, 1024:], Tg, dW3[GL], db=db3[GL], rows=Mp, n_out=1024, k_in=1024, w_blocks=blk)])
        # ---- read-out fc_1 * fc_2 products: dF at the pairs' grid rows, dm in place
        F = S["F"]
        dF = torch.zeros(Mg, 4096, **f32)                        # [dF2 | dF_os | dF_so | dF_g]; self-pair rows stay zero
        pair_grid, pair_h, pair_o, grid_h, grid_o, grid_img = (S[k] for k in ("pair_grid", "pair_h", "pair_o", "grid_h",
                                                                              "grid_o", "grid_img"))
        _check(lib.skg_mul_bwd_f32(dTp.data_ptr(), 1024, F.data_ptr(), pair_grid.data_ptr(), 4096, S["B1h"].data_ptr(),
                                   pair_h.data_ptr(), 1024, S["B1o"].data_ptr(), pair_o.data_ptr(), 1024,
                                   b1[ATT].data_ptr(), Mp, dF.data_ptr(), 4096, 0, stream), "skg_mul_bwd_f32")
        pair_img = self.pair_img
        _check(lib.skg_mul_bwd_f32(dTg.data_ptr(), 1024, F.data_ptr() + 4 * 3072, pair_grid.data_ptr(), 4096,
                                   S["G1"].data_ptr(), pair_img.data_ptr(), 1024, None, None, 0, None, Mp,
                                   dF.data_ptr() + 4 * 3072, 4096, 0, stream), "skg_mul_bwd_f32")
        dB1h = torch.empty(Mh, 1024, **f32); dB1o = torch.empty(Mn, 1024, **f32)
        hum_img, node_img = self.isl("hum_img"), self.isl("node_img")
        _check(lib.skg_segment_sum_f32(dTp.data_ptr(), 1024, self.meta.data_ptr(), A, hum_img.data_ptr(),
                                       node_img.data_ptr(), Mh, Mn, 1, dB1h.data_ptr(), dB1o.data_ptr(), 0, stream),
               "skg_segment_sum_f32")
        dG1 = torch.zeros(S["G1"].shape[0], 1024, **f32)
        _check(lib.skg_segment_sum_f32(dTg.data_ptr(), 1024, self.meta.data_ptr(), A, None, None, 0, 0, 2,
                                       dG1.data_ptr(), None, 0, stream), "skg_segment_sum_f32")
        # ---- read-out fc_1 on the normalised nodes: dh_node, dnode; dW1[att] from both halves
        dh_node = torch.empty(Mh, 1024, **f32); dnode = torch.empty(Mn, 1024, **f32)
        Wa1, dWa1 = W1[ATT], dW1[ATT]
        self.gx([IG(dB1h, Wa1, dh_node, N_in=1024), IG(dB1o, Wa1[:, 1024:], dnode, N_in=1024),
                      WG(dB1h, S["h_node"], dWa1, k_in=1024), WG(dB1o, S["node"], dWa1[:, 1024:], k_in=1024)])
        # ---- LayerNorm + residual: dHp continues to the node, dHp cut by the message's ReLU goes to fc_3
        nh, no = gh.norm_h, gh.norm_o
        dHp = torch.empty(Mh, 1024, **f32); dHm = torch.empty(Mh, 1024, **f32)
        dOp = torch.empty(Mn, 1024, **f32); dOm = torch.empty(Mn, 1024, **f32)
        g_nh = (gv("nh_w"), gv("nh_b")); g_no = (gv("no_w"), gv("no_b"))
        _check(lib.skg_layernorm_bwd_f32(dh_node.data_ptr(), 1024, S["Hp"].data_ptr(), S["st_h"].data_ptr(),
                                         nh.weight.data_ptr(), Mh, dHp.data_ptr(), S["M1"].data_ptr(), dHm.data_ptr(),
                                         g_nh[0].data_ptr(), g_nh[1].data_ptr(), stream), "skg_layernorm_bwd_f32")
        _check(lib.skg_layernorm_bwd_f32(dnode.data_ptr(), 1024, S["Op"].data_ptr(), S["st_o"].data_ptr(),
                                         no.weight.data_ptr(), Mn, dOp.data_ptr(), S["M2"].data_ptr(), dOm.data_ptr(),
                                         g_no[0].data_ptr(), g_no[1].data_ptr(), stream), "skg_layernorm_bwd_f32")
        # ---- message fc_3
        dU = torch.empty(Mh, 1024, **f32); dV = torch.empty(Mn, 1024, **f32)
        self.gx([IG(dHm, W3[OS], dU, N_in=1024, w_blocks=blk), IG(dOm, W3[SO], dV, N_in=1024, w_blocks=blk),
                      WG(dHm, S["U"], dW3[OS], db=db3[OS], w_blocks=blk), WG(dOm, S["V"], dW3[SO], db=db3[SO], w_blocks=blk)])
        # ---- aggregation + softmax
        dTos = torch.empty(Mg, 1024, **f32); dTso = torch.empty(Mg, 1024, **f32)
        da = torch.empty(4, Mg, **f32)                            # da | db | dadj_h | dadj_n
        _check(lib.skg_aggregate_bwd_f32(dU.data_ptr(), dV.data_ptr(), S["Tos"].data_ptr(), S["Tso"].data_ptr(),
                                         S["alpha"].data_ptr(), S["beta"].data_ptr(), grid_h.data_ptr(), grid_o.data_ptr(),
                                         Mg, self.meta.data_ptr(), hum_img.data_ptr(), node_img.data_ptr(), Mh, Mn,
                                         dTos.data_ptr(), dTso.data_ptr(), da[0].data_ptr(), da[1].data_ptr(),
                                         da[2].data_ptr(), da[3].data_ptr(), stream), "skg_aggregate_bwd_f32")
        # ---- adjacency Linear(1024 -> 1) over relu(fc_3(T))
        dadj = torch.empty(Mg, 1, **f32); dWt = torch.empty(Mg, 1024, **f32)
        wadj = gh.adjacency.weight.detach().reshape(-1)
        _check(lib.skg_adjacency_bwd_f32(da[2].data_ptr(), da[3].data_ptr(), wadj.data_ptr(), S["Wt"].data_ptr(), Mg,
                                         dadj.data_ptr(), dWt.data_ptr(), stream), "skg_adjacency_bwd_f32")
        g_adj_w, g_adj_b = gv("adj_w"), gv("adj_b")
        dT = torch.empty(Mg, 1024, **f32)
        T = S["T"]
        self.gx([IG(dWt, W3[ATT], dT, mask=T, N_in=1024, w_blocks=blk),
                      WG(dWt, T, dW3[ATT], db=db3[ATT], accumulate=True, w_blocks=blk),
                      WG(dadj, S["Wt"], g_adj_w, db=g_adj_b)])
        # ---- in-loop fc_1 * fc_2 products
        _check(lib.skg_mul_bwd_f32(dT.data_ptr(), 1024, F.data_ptr(), None, 4096, S["A1h"].data_ptr(), grid_h.data_ptr(),
                                   1024, S["A1o"].data_ptr(), grid_o.data_ptr(), 1024, b1[ATT].data_ptr(), Mg,
                                   dF.data_ptr(), 4096, 1, stream), "skg_mul_bwd_f32")
        _check(lib.skg_mul_bwd_f32(dTos.data_ptr(), 1024, F.data_ptr() + 4 * 1024, None, 4096, S["C1o"].data_ptr(),
                                   grid_o.data_ptr(), 1024, None, None, 0, None, Mg, dF.data_ptr() + 4 * 1024, 4096, 0,
                                   stream), "skg_mul_bwd_f32")
        _check(lib.skg_mul_bwd_f32(dTso.data_ptr(), 1024, F.data_ptr() + 4 * 2048, None, 4096, S["C1h"].data_ptr(),
                                   grid_h.data_ptr(), 1024, None, None, 0, None, Mg, dF.data_ptr() + 4 * 2048, 4096, 0,
                                   stream), "skg_mul_bwd_f32")
        dA1h = torch.empty(Mh, 1024, **f32); dA1o = torch.empty(Mn, 1024, **f32)
        dC1o = torch.empty(Mn, 1024, **f32); dC1h = torch.empty(Mh, 1024, **f32)
        seg = lambda src, oh, on: _check(lib.skg_segment_sum_f32(
            src.data_ptr(), 1024, self.meta.data_ptr(), A, hum_img.data_ptr(), node_img.data_ptr(), Mh, Mn, 0,
            _ptr(oh), _ptr(on), 0, stream), "skg_segment_sum_f32")
        seg(dT, dA1h, dA1o); seg(dTos, None, dC1o); seg(dTso, dC1h, None)
        # the multiplier bias of attention_head's fc_1 is added once per row: its gradient is the sum over all rows
        g_ab1 = dA1h.sum(dim=0) + dB1h.sum(dim=0)
        db1[ATT].copy_(g_ab1)
        # ---- fc_2 of all four MBFs: ONE product for the input gradient (K = 4096), one for the weights
        Sp = S["Sp"]
        dS = torch.empty(Mg, 1024, **f32)
        self.gx([IG(dF, W2, dS, mask=Sp, N_in=1024), WG(dF, Sp, gv("W2"), db=gv("b2"))])
        # ---- fc_1 projections on node rows: gradients of the nodes accumulate on top of the residual path.  The spatial
        #      head's backward (dS -> ds2 -> ds1 -> first layer) is independent of the node chain: its three steps ride in
        #      the same grouped launches.
        GH, GO = S["GH"], S["GO"]
        sp = gh.spatial_head
        s1, s2, sp48 = S["s1"], S["s2"], S["sp48"]
        ds2 = torch.empty(Mg, 256, **f32); ds1 = torch.empty(Mg, 128, **f32)
        g_sp = [(gv("sp%d_w" % i), gv("sp%d_b" % i)) for i in (0, 2, 4)]
        self.gx([IG(dA1h, Wa1, dHp, accumulate=True, N_in=1024), IG(dA1o, Wa1[:, 1024:], dOp, accumulate=True, N_in=1024),
                 WG(dA1h, GH, dWa1, accumulate=True, k_in=1024), WG(dA1o, GO, dWa1[:, 1024:], accumulate=True, k_in=1024),
                 IG(dS, sp[4].weight, ds2, mask=s2), WG(dS, s2, g_sp[2][0], db=g_sp[2][1])])
        self.gx([IG(dC1h, W1[SO], dHp, mask=GH, accumulate=True), IG(dC1o, W1[OS], dOp, mask=GO, accumulate=True),
                 WG(dC1h, GH, dW1[SO], db=db1[SO]), WG(dC1o, GO, dW1[OS], db=db1[OS]),
                 IG(ds2, sp[2].weight, ds1, mask=s1), WG(ds2, s1, g_sp[1][0], db=g_sp[1][1])])
        # ---- fc_head / fc_tail, with the first spatial layer and the global branch's fc_1 (HEAD:971)
        Xhn = S["Xhn"]
        fh, ft = gh.fc_head[0], gh.fc_tail[0]
        dXhn = torch.empty(Mh + Mn, 1088, **f32)
        g_fh_w, g_fh_b, g_ft_w, g_ft_b = gv("fh_w"), gv("fh_b"), gv("ft_w"), gv("ft_b")
        gfeat = S["gfeat"]
        ops = [IG(dHp, fh.weight, dXhn[:Mh], N_in=1074), IG(dOp, ft.weight, dXhn[Mh:], N_in=1074),
               WG(dHp, Xhn[:Mh], g_fh_w, db=g_fh_b, k_in=1074), WG(dOp, Xhn[Mh:], g_ft_w, db=g_ft_b, k_in=1074),
               WG(ds1, sp48, g_sp[0][0], db=g_sp[0][1], k_in=46), WG(dG1, gfeat, dW1[GL], db=db1[GL])]
        dgfeat = None
        if need_dgfeat:
            dgfeat = torch.empty_like(gfeat)
            ops.append(IG(dG1, W1[GL], dgfeat))
        self.gx(ops)
        d_enc = torch.empty(NA, 1024, **f32)
        _check(lib.skg_entity_rows_bwd_f32(dXhn.data_ptr(), 1088, self.hum_of.data_ptr(), self.node_of.data_ptr(), Mh, NA,
                                           S["enc"].data_ptr(), d_enc.data_ptr(), stream), "skg_entity_rows_bwd_f32")
        # ---- box_head
        bh1, bh3 = gh.box_head[1], gh.box_head[3]
        E1, x0 = S["E1"], S["x0"]
        dE1 = torch.empty(NA, 1024, **f32)
        g_bh3_w, g_bh3_b, g_bh1_w, g_bh1_b = gv("bh3_w"), gv("bh3_b"), gv("bh1_w"), gv("bh1_b")
        self.gx([IG(d_enc, bh3.weight, dE1, mask=E1), WG(d_enc, E1, g_bh3_w, db=g_bh3_b)])
        ops = [WG(dE1, x0, g_bh1_w, db=g_bh1_b)]
        dx0 = None
        if need_dx0:
            dx0 = torch.empty_like(x0)
            ops.append(IG(dE1, bh1.weight, dx0))
        self.gx(ops)
        # ---- the fc_3 bias of branch b is added once per row whatever b: every branch gets the MBF's bias gradient
        gv("b3").copy_(db3.unsqueeze(1).expand(4, 16, 1024))
        # ---- hand the gradients back in the order of the Function's parameter inputs
        out = []
        for p in self.params:
            k = st.ids.get(id(p))
            out.append(sviews[k] if k is not None else None)
        self.S = None                                            # the saved activations die with the step
        return dx0, dgfeat, out


class NativeJob(TrainJob):
    """TrainJob whose dense forward / backward are ONE native call each (include/skghoi.h: skg_train_plan,
    skghoi_amd/csrc/skg_train_plan.hip): the same launch sequence on the same operands, issued from C++ out of a plan
    struct -- no per-launch Python, no per-activation tensor: every activation lives at a fixed offset of one workspace.
    Data parallel (`head.grad_exchange`, skghoi_amd/trainer.py ArenaExchange): the backward runs stage by stage and after
    every stage the gradient-arena prefix that is final (Stacked.milestone_end[s]) is handed to the exchange."""

    def _plan(self, x0, gfeat):
        st, lay, S = self.st, self.lay, self.S
        pl = _capi.TrainPlan()
        pl.NA, pl.Mg, pl.Mp, pl.Mh, pl.Mn, pl.A = lay.sum_all, lay.sum_g, lay.sum_p, lay.sum_h, lay.sum_n, lay.n_active
        pl.K, pl.Bf, pl.Cf, pl.x0_k = self.K, gfeat.shape[0], gfeat.shape[1], x0.shape[1]
        pl.bf16 = 1 if self.bf16 else 0
        pl.ld_logits = (self.K + 1 + 3) // 4 * 4
        pl.params = st.buf.data_ptr()
        pl.seg_off = st.seg_off_array()
        pl.x0, pl.gfeat, pl.sp48 = x0.data_ptr(), gfeat.data_ptr(), S["sp48"].data_ptr()
        pl.meta = self.meta.data_ptr()
        for k in ("enc_row_hn", "img_hn", "ent_row_hn", "hum_img", "node_img"):
            setattr(pl, k, self.isl(k).data_ptr())
        for k in ("grid_h", "grid_o", "grid_pair", "grid_img", "pair_grid", "pair_h", "pair_o"):
            setattr(pl, k, S[k].data_ptr())
        pl.pair_img, pl.hum_of, pl.node_of = self.pair_img.data_ptr(), self.hum_of.data_ptr(), self.node_of.data_ptr()
        pl.timer = self.head.__dict__.get("_train_timer")      # measurement aid (bench.py): events around every gemmx launch
        tgt = os.environ.get("SKG_SPLIT_TARGET_BF16" if self.bf16 else "SKG_SPLIT_TARGET_F32")     # developer knobs: they travel
        if tgt:                                                                                    # in the plan, not in the library
            pl.split_target = int(tgt)
        if os.environ.get("SKG_SPLIT_MAX"):
            pl.split_max = int(os.environ["SKG_SPLIT_MAX"])
        # the step's node chain on a second stream beside the spatial chain (skg_train_plan.two_branch; bf16 step, single
        # process: a staged data-parallel backward issues its stages in runs that end at the chunk events)
        pl.two_branch = 1 if (self.bf16 and TWO_BRANCH) else 0
        if INLAUNCH_SPLIT_REDUCE:
            ctr = st.counters()                                # split-K reduced inside the product launches (skg_gemmx_desc.split_ctr)
            pl.counters, pl.n_counters = ctr.data_ptr(), ctr.numel()
        return pl

    def forward_a(self, x0, gfeat):
        lib = _capi.lib()
        st, lay, dev, S = self.st, self.lay, self.dev, self.S
        if not st.aliased():
            st.adopt()
        x0 = x0.detach().float().reshape(x0.shape[0], -1).contiguous()
        gfeat = gfeat.detach().float().contiguous()
        if x0.shape[1] != st.seg["bh1_w"][1][1]:
            raise RuntimeError("mat1 and mat2 shapes cannot be multiplied (%dx%d and %dx%d)" % (
                x0.shape[0], x0.shape[1], st.seg["bh1_w"][1][1], 1024))
        S["x0"], S["gfeat"] = x0, gfeat
        pl = self.plan = self._plan(x0, gfeat)
        self.head._last_train_plan = pl            # (bench.py reads the step's arithmetic off it: skg_train_flops)
        n = int(lib.skg_train_ws_floats(C.byref(pl)))
        if n < 0:
            _check(n, "skg_train_ws_floats")
        # workspaces are kept by the parameter arena object and reused from step to step (steps follow each other on one
        # stream: step i + 1's forward is ordered behind step i's backward, and the worker thread of step i has been joined)
        keep = getattr(self, "reuse_ws", False)
        alloc = st.scratch if keep else (lambda name, m, dt, d: torch.empty(max(m, 1), device=d, dtype=dt))
        self.ws = alloc("ws", n, torch.float32, dev)
        pl.ws, pl.ws_floats = self.ws.data_ptr(), n
        self.ws16 = None
        if self.bf16 and self.twins:
            # bf16 twins: a second workspace with the same element offsets (every product output and per-row kernel output is
            # also stored rounded there), the twin of the parameter arena (rewritten by part 0 every step) and of the pair
            # features -- the products then read 2-byte operands straight into LDS (skg_gemmx_t16_kernel)
            self.ws16 = alloc("ws16", n, torch.int16, dev)
            pl.ws16 = self.ws16.data_ptr()
            pl.params16 = st.twin().data_ptr()
            pl.params_floats = st.total
        Mp1 = max(lay.sum_p, 1)
        self.PF = torch.empty(Mp1, 2048, device=dev, dtype=torch.float32)
        if self.ws16 is not None:
            self.PF16 = alloc("pf16", Mp1 * 2048, torch.int16, dev)
            pl.pf16 = self.PF16.data_ptr()
        self.logits_full = torch.zeros(Mp1, pl.ld_logits, device=dev, dtype=torch.float32)
        pl.pair_features, pl.logits = self.PF.data_ptr(), self.logits_full.data_ptr()
        _check(lib.skg_ctx_train_forward_f32(context_for(self.head).handle(), C.byref(pl), 0, _stream()), "skg_train_forward_f32[0]")
        self.part_a_done = True

    def forward(self, x0, gfeat):
        if not getattr(self, "part_a_done", False):
            self.forward_a(x0, gfeat)
        pl = self.plan
        pl.ent = self.ent.data_ptr()
        _check(_capi.lib().skg_ctx_train_forward_f32(context_for(self.head).handle(), C.byref(pl), 1, _stream()),
               "skg_train_forward_f32[1]")
        self.S.update(PF=self.PF, logits=self.logits_full[:self.lay.sum_p])
        return self.S

    def flops(self):
        """2 M N K over every dense product of this step (forward + backward), from the plan itself."""
        return float(_capi.lib().skg_train_flops(C.byref(self.plan), 2))

    def saved(self, which):
        """A saved activation out of the workspace (tests): 'enc', 'h_node', 'node', 'adjacency'."""
        idx = dict(enc=0, h_node=1, node=2, adjacency=3)[which]
        off = int(_capi.lib().skg_train_ws_offset(C.byref(self.plan), idx))
        lay = self.lay
        rows, cols = dict(enc=(lay.sum_all, 1024), h_node=(lay.sum_h, 1024), node=(lay.sum_n, 1024),
                          adjacency=(lay.sum_g, 1))[which]
        return self.ws[off:off + rows * cols].view(rows, cols)

    # single process: the stages behind which the optimizer inside the backward updates an arena chunk (two events on the
    # backward's queue, ~12 us each; the last chunk -- box_head.1, 44 % of the parameters -- follows its stage in-stream)
    ADAMW_STAGES = (6, 9)

    def backward(self, dlogits, need_dx0, need_dgfeat, arena=None, defer=False, span=None, adamw=None):
        """defer=True (engine-free step): the launches are issued by the worker thread of the head's context
        (skg_ctx_train_backward_async_f32) while this thread goes on with host work; `join_backward()` must run before
        anything is enqueued behind the gradients.  span: a dict that receives "b1", a timing event the worker records behind
        the last stage (measurement).  Returns (dx0, dgfeat, gradient views in parameter order | None)."""
        lib = _capi.lib()
        st, S, pl = self.st, self.S, self.plan
        context_for(self.head).join()
        ga, sviews = arena if arena is not None else st.grad_arena()
        pl.grads = ga.data_ptr()
        dlogits = dlogits.contiguous()
        if dlogits.shape[1] != pl.ld_logits:
            raise _capi.SkgError("dlogits has %d columns, the plan %d" % (dlogits.shape[1], pl.ld_logits))
        pl.dlogits = dlogits.data_ptr()
        dx0 = torch.empty_like(S["x0"]) if need_dx0 else None
        dgfeat = torch.empty_like(S["gfeat"]) if need_dgfeat else None
        pl.dx0, pl.dgfeat = _ptr(dx0), _ptr(dgfeat)
        stream = _stream()
        ex = getattr(self.head, "grad_exchange", None)
        n = _capi.TRAIN_BWD_STAGES
        if defer:
            # ONE call hands the whole backward to the worker thread of this head's context.  Data parallel: the worker
            # records an event behind every stage; the exchange (skghoi_amd/trainer.py, ArenaExchange.drive) waits for
            # "stage s issued" and orders the collective of the arena prefix that stage completed behind its event.
            ctx = context_for(self.head)
            ctx.join()
            ctx.owner = threading.get_ident()
            events, mask = None, 0
            native = ex is not None and ex.native is not None
            if ex is not None:
                ex.begin(ga)
                mask = ex.stage_mask(st.milestone_end)       # stages that complete an arena chunk: the context's own events
            after = None
            if span is not None:
                after = span["b1"] = torch.cuda.Event(enable_timing=True)
            elif native and ex.timing:
                after = torch.cuda.Event(enable_timing=True)     # (bench.py: exposed wait = this event -> the last collective)
            if after is not None:
                after.record()                                   # (torch creates the HIP event at its first record)
                events = (C.c_void_p * n)()
                events[n - 1] = after.cuda_event
            xd = None
            if native:
                # the library's own RCCL communicator: the WORKER all-reduces every arena chunk behind the stage that
                # completes it and orders this stream behind the last collective -- nothing left to drive from here
                ex._after = after
                xd = ex.native_exchange(ga, st.milestone_end)
            elif ex is None and adamw is not None:
                xd = _capi.Exchange()                            # single process: chunks for the optimizer only
                cs = [(s_, st.milestone_end[s_]) for s_ in self.ADAMW_STAGES] + [(n - 1, ga.numel())]
                xd.comm, xd.arena, xd.n_chunks = None, ga.data_ptr(), len(cs)
                for i, (s_, end) in enumerate(cs):
                    xd.stage[i], xd.end[i] = s_, end
            sl = None
            if xd is not None and adamw is not None:
                sl = adamw(st, ga, [(int(xd.stage[i]), int(xd.end[i])) for i in range(xd.n_chunks)])
                if sl is not None:
                    xd.adamw, xd.adamw_steps, xd.adamw_n_steps = sl["adamw"], sl["steps"], sl["n_steps"]
                    for i, f in enumerate(sl["first"]):
                        xd.adamw_first[i] = f
                    for k in ("lr", "beta1", "beta2", "eps", "weight_decay", "bias1", "bias2"):
                        setattr(xd, k, sl[k])
                elif not native:
                    xd = None
            if xd is not None:
                _check(lib.skg_ctx_train_backward_exchange_f32(ctx.handle(), C.byref(pl), 0, n, stream, events, C.byref(xd)),
                       "skg_ctx_train_backward_exchange_f32")
                if sl is not None:
                    self.head.__dict__["_adamw_in_backward"] = sl
            else:
                _check(lib.skg_ctx_train_backward_async_f32(ctx.handle(), C.byref(pl), 0, n, stream, events, mask),
                       "skg_train_backward_async_f32")
            # everything the plan names stays alive until the worker has enqueued the last launch
            ctx.pending.append((self, self.S, self.ws, self.ws16, dlogits, ga, dx0, dgfeat))
            if native:
                ctx.driven = ex                                  # finish() at the join: bookkeeping only
            elif ex is not None:
                ctx.exchange = (ex, list(st.milestone_end))
            self.S = None
            self.ws = None
            return dx0, dgfeat, None
        if ex is None:
            _check(lib.skg_train_backward_f32(C.byref(pl), 0, n, stream), "skg_train_backward_f32")
        else:
            # data parallel on the autograd route (a trainable detector in front of the head): the stages are issued from
            # this thread, and after every stage the gradient-arena prefix that stage completed goes out to the peers
            ex.begin(ga)
            if ex.native is not None:
                _check(lib.skg_train_backward_f32(C.byref(pl), 0, n, stream), "skg_train_backward_f32")
                ex.native_all_reduce(ga, st.milestone_end)       # the same chunk sequence as the peers' staged route
            else:
                for s_ in range(n):
                    _check(lib.skg_train_backward_f32(C.byref(pl), s_, s_ + 1, stream), "skg_train_backward_f32[%d]" % s_)
                    ex.on_stage(s_, ga, st.milestone_end[s_], last=(s_ == n - 1))
            ex.finish()
        out = []
        for p in self.params:
            k = st.ids.get(id(p))
            out.append(sviews[k] if k is not None else None)
        self.S = None                                            # the saved activations die with the step
        self.ws = None
        return dx0, dgfeat, out


class TrainContext:
    """The skg_context (include/skghoi.h) of one head: the library-side worker thread that issues a deferred backward, its
    one job slot, and -- on this side -- what that job still reads.  One per head, so two trainers in one process (two
    host threads, two devices) never share a job slot."""
    live = None

    def __init__(self):
        self._h = None
        self.pid = os.getpid()
        self.pending = []             # what the deferred backward still reads (tensors the plan names)
        self.exchange = None          # (ArenaExchange, milestone ends) of a data-parallel job whose collectives are not out yet
        self.driven = None            # the exchange once its collectives are out: finish() at the join
        self.owner = None             # host thread that submitted the pending job (its stream is the one to order)
        if TrainContext.live is None:
            TrainContext.live = weakref.WeakSet()
        TrainContext.live.add(self)

    def handle(self):
        if self._h is None or self.pid != os.getpid():          # (a forked child: the parent's worker does not exist here)
            self._h = _capi.lib().skg_context_create()
            self.pid = os.getpid()
            if not self._h:
                raise _capi.SkgError("skg_context_create failed")
        return self._h

    def drive(self):
        """Data parallel: sends the gradient-arena chunks out behind the stages the worker is issuing (no-op otherwise)."""
        if self.exchange is not None:
            ex, ends = self.exchange
            self.exchange = None
            self.driven = ex
            ex.drive(self, ends)

    def join(self):
        """Waits until the worker has enqueued every launch of the pending job; with a gradient exchange, sends whatever
        chunks are still to go and orders the caller's stream behind all of them."""
        if not self.pending:
            return
        try:
            self.drive()
            _check(_capi.lib().skg_ctx_train_backward_join(self.handle()), "skg_train_backward_f32 (deferred)")
        finally:
            self.pending.clear()
            ex, self.driven, self.exchange = self.driven, None, None
        if ex is not None:
            ex.finish()

    def stage_wait(self, s, stream=None):
        """Blocks (without the GIL) until the worker has issued stage s; with `stream`: also orders that stream behind the
        stage on the device (the context's own event, recorded by the worker)."""
        lib = _capi.lib()
        _check(lib.skg_ctx_train_backward_stage_wait(self.handle(), s), "skg_train_backward_f32 (deferred, stage %d)" % s)
        if stream is not None:
            _check(lib.skg_ctx_stream_wait_stage(self.handle(), s, stream.cuda_stream), "skg_ctx_stream_wait_stage(%d)" % s)

    def close(self, _getpid=os.getpid, _lib=_capi.lib):
        h, self._h = self._h, None
        if h and self.pid == _getpid():
            try:
                _lib().skg_context_destroy(h)
            except Exception:                       # noqa: BLE001  (interpreter shutdown)
                pass

    def __del__(self):
        self.close()


def context_for(head):
    c = head.__dict__.get("_train_ctx")
    if c is None:
        c = head.__dict__["_train_ctx"] = TrainContext()
    return c


def _mine(every_thread=False):
    me = threading.get_ident()
    return [c for c in list(TrainContext.live or ()) if c.pending and (every_thread or c.owner == me)]


def drive_exchanges():
    """Data parallel: hands the gradient chunks of every deferred backward this thread has in flight to the process group
    (the trainer calls it as soon as the step's other host work is queued; join_backward() would do it too, later)."""
    for c in _mine():
        c.drive()


def join_backward(every_thread=False):
    """Waits until the library's worker threads have enqueued every launch of the deferred backwards THIS host thread
    submitted (NativeJob.backward defer=True) and, data parallel, until the step's stream is ordered behind the gradient
    exchange; a no-op otherwise.  Called before the optimizer step, before the next backward and by anything that enqueues
    work behind the gradients."""
    for c in _mine(every_thread):
        c.join()


import atexit
atexit.register(join_backward, True)          # (a worker still issuing launches must not race the runtime's teardown)


def job_class(head):
    """NativeJob (default) or the Python-issued TrainJob (`head.train_plan = "python"` / SKG_TRAIN_PLAN=python: the same
    kernels launched one by one from Python -- kept as the readable statement of the sequence and as a cross-check)."""
    import os
    mode = os.environ.get("SKG_TRAIN_PLAN") or getattr(head, "train_plan", "native")
    return TrainJob if mode == "python" else NativeJob


# ---------------------------------------------------------------------------------------------------- step driver
def _perm(state, n, m):
    g = torch.Generator(); g.set_state(state)
    return torch.randperm(n, generator=g)[:m]


def supported(head):
    """The fused step covers the reference's configuration -- message passing on (num_iter >= 1), plain Linear
    predictor / suppressor -- in exact fp32 (the default) or with bf16 operands on every dense layer
    (precision="bf16").  Anything else takes the autograd path (skghoi_amd/train_graph.py)."""
    gh = head.box_pair_head
    return (head.precision in ("fp32", "bf16") and gh.num_iter > 0 and isinstance(head.box_pair_predictor, torch.nn.Linear)
            and isinstance(head.box_pair_suppressor, torch.nn.Linear)
            and head.box_pair_predictor.in_features == 2048 and head.box_pair_suppressor.in_features == 2048
            and head.box_pair_suppressor.out_features == 1 and head.box_pair_predictor.out_features == head.num_classes
            and head.box_pair_predictor.bias is not None and head.box_pair_suppressor.bias is not None)


_ESIZE = {torch.float32: 4, torch.int32: 4, torch.int64: 8, torch.uint8: 1, torch.int16: 2, torch.float64: 8, torch.bool: 1}


class PrepArena:
    """One reusable block of device memory for everything a PREPARED batch owns (index tables, labels, TransH tables and
    scores ...: ~17 tensors per step).  Allocated one by one they cost the step's host thread ~0.1 ms -- torch.empty, a
    record_stream per tensor for the stream that consumes them, and at release one event per tensor from the allocator --
    on a step whose host thread IS the bound (1.38 ms of host work per 1.40 ms batch-4 bf16 step, measured).  A slot is
    bump-allocated by one preparation on the side stream, read by one training step on the step's stream, and reused two
    batches later: the side stream first waits for `done`, the event trainer.train_step records on the step's stream once
    the step that consumed the slot has enqueued its last kernel.  Only the trainer's look-ahead uses slots
    (prefetch_train(..., arena=True)): it alone knows when a prepared batch is dead.  A request that does not fit falls
    back to torch.empty (returned through `spill`, handled like any cross-stream tensor) and enlarges the next block."""
    ALIGN = 256

    def __init__(self, dev):
        self.dev, self.buf, self.off, self.need, self.done, self.spill = dev, None, 0, 1 << 20, None, []
        self.typed = {}                           # dtype -> the block viewed as that type

    def begin(self, stream, consumer):
        """Start of a preparation on `stream` (current); `consumer`: the stream the training step runs on."""
        if self.done is not None:
            stream.wait_event(self.done)
        if self.buf is None or self.buf.numel() < self.need:
            self.buf = torch.empty(int(self.need * 1.25) // self.ALIGN * self.ALIGN + self.ALIGN, dtype=torch.uint8,
                                   device=self.dev)
            self.buf.record_stream(consumer)          # once per block: whenever it is dropped, both streams are honoured
            self.typed = {}
        self.off, self.spill = 0, []

    def take(self, shape, dtype):
        shape = (shape,) if isinstance(shape, int) else tuple(shape)
        n = 1
        for v in shape:
            n *= int(v)
        nbytes = n * (_ESIZE.get(dtype) or torch.empty(0, dtype=dtype).element_size())
        end = self.off + (nbytes + self.ALIGN - 1) // self.ALIGN * self.ALIGN
        if end > self.buf.numel():
            self.need = max(self.need, end * 2)
            t = torch.empty(shape, dtype=dtype, device=self.dev)
            self.spill.append(t)
            self.off = end
            return t
        base = self.typed.get(dtype)
        if base is None:
            base = self.typed[dtype] = self.buf.view(dtype)
        es = nbytes // n if n else 1
        strides, acc = [], 1
        for v in reversed(shape):
            strides.append(acc)
            acc *= int(v)
        t = base.as_strided(shape, strides[::-1], self.off // es)
        self.off = end
        self.need = max(self.need, end)
        return t


def release_prepared(head):
    """trainer.train_step, once the step's last kernel is enqueued: the arena slot of the batch this step consumed may be
    reused behind everything the current stream holds now."""
    slot = head.__dict__.pop("_prep_slot_in_use", None)
    if slot is not None:
        if slot.done is None:
            slot.done = torch.cuda.Event()
        slot.done.record(current_stream_of(slot.dev))


class Prepared:
    """Everything of a training forward that depends on the BATCH but not on the weights: selected detections (NMS,
    top-k, GT boxes prepended), layout, pair / spatial arrays, label matrix, positive counts, the TransH tables and
    negative permutations drawn from the host RNG, all uploaded.  Made inline by the forward -- or ahead of time, for the
    NEXT batch, on a side stream while the GPU is busy with the current step (InteractionHead.prefetch_train)."""
    empty = False
    norm = None           # data parallel: dist.PreparedNormalisers, the n_p all-reduce started by the preparation
    slot = None           # PrepArena the batch's device tensors live in (the trainer's look-ahead), else None
    ready = None          # event behind the preparation's device work when it ran on a side stream
    cross = ()            # tensors allocated on the side stream and consumed on the step's stream


def _stacked_for(head, dev):
    st = getattr(head, "_stacked", None)
    if st is None or st.device != dev or any(a is not b for a, b in zip(st.src[st.n_stacked - 4:st.n_stacked], (
            head.box_pair_predictor.weight, head.box_pair_suppressor.weight, head.box_pair_predictor.bias,
            head.box_pair_suppressor.bias))) or st.epoch != _reg_epoch():
        st = Stacked(head, dev)
        st.epoch = _reg_epoch()
        head._stacked = st
    return st


def prepare_train(head, eng, detections, image_shapes, targets, before_sync=None, training=True):
    """prepare_steps() run to the end in one go (the inline forward)."""
    gen = prepare_steps(head, eng, detections, image_shapes, targets, before_sync, training=training)
    try:
        while True:
            next(gen)
    except StopIteration as done:
        return done.value


def prepare_steps(head, eng, detections, image_shapes, targets, before_sync=None, arena=None, training=True):
    """The weight-independent part of InteractionHead.forward in training mode (HEAD:92-151 preprocess with GT boxes
    appended, HEAD:847-868 pairs + spatial encoding, HEAD:703-719 label association, the host RNG of HEAD:574-580 / 939,
    and the whole TransH term HEAD:207-235 / 936-963: its scores depend only on the step's throw-away embeddings).
    Two host synchronisations: the per-image counts after the preprocess kernel and the positive counts after the
    association kernel.  A GENERATOR that yields right before each of them, so that a prefetching caller can do other host
    work (enqueue the current step's backward, then its optimizer) while the kernels run, and returns the Prepared batch.
    before_sync(prep): called right before the second one (the inline forward enqueues the table-independent part of the
    dense forward there, so that the GPU works while the host waits and draws).
    arena: a PrepArena slot (already begun on the current stream) that holds the batch's device tensors instead of one
    torch allocation each.
    training=False: the same for an eval-mode forward WITH targets (validation, utils.py:283-299): no GT boxes appended
    (HEAD:105-106), the eval score power in the cell count (HEAD:742); labels are associated and the host RNG is consumed
    exactly as in training -- the reference does both whenever targets are given (HEAD:933-963)."""
    from . import transh
    from . import dist as skd
    lib = _capi.lib()
    gh = head.box_pair_head
    K = head.num_classes
    launched = eng.pre_launch(detections, targets, training, training, defer=True)
    yield 1
    pre = eng.pre_pack(eng.pre_launch_end(launched))
    dev = pre.device
    stream = _stream()
    # ---- the batch layout and every index table of the step: one native call into a pinned block, one upload
    gt_all = [int(t["boxes_h"].shape[0]) for t in targets]
    lay, hbuf, offs = layout.build_train(pre.n_h, pre.n, pre.L, image_shapes, gh.human_idx, gt_count=gt_all,
                                         faithful_skip_offset=eng.faithful_skip_offset)
    prep = Prepared()
    prep.pre, prep.lay, prep.inputs = pre, lay, (detections, image_shapes, targets)
    prep.training = training
    A = lay.n_active
    if A == 0 or lay.sum_p == 0:
        prep.empty = True
        return prep
    Mg, Mp, Mh, Mn, NA = lay.sum_g, lay.sum_p, lay.sum_h, lay.sum_n, lay.sum_all
    f32 = dict(device=dev, dtype=torch.float32)
    i32 = dict(device=dev, dtype=torch.int32)
    act_imgs = [int(b) for b in lay.active]
    gt_off_h = hbuf.numpy()[offs["gt_off"][0]:offs["gt_off"][0] + offs["gt_off"][1]]
    # device tensors of the batch: views of the arena slot when the trainer's look-ahead supplied one, else one allocation each
    if arena is not None:
        E = arena.take
    else:
        def E(shape, dtype):
            return torch.empty(shape, dtype=dtype, device=dev)

    def upload(h):                                                   # pinned host tensor -> device, asynchronously
        d = E(tuple(h.shape), h.dtype)
        d.copy_(h, non_blocking=True)
        return d
    ibuf = upload(hbuf)
    isl = lambda name: ibuf[offs[name][0]:offs[name][0] + offs[name][1]]
    meta = isl("meta")
    # ---- pairs + spatial encoding, GT association (HEAD:847-868, 703-719): ahead of the dense part because the
    # number of positives per image sizes the host RNG draws below
    grid = E(4 * Mg + 3 * max(Mp, 1), torch.int32)
    grid_h, grid_o, grid_pair, grid_img = grid[:Mg], grid[Mg:2 * Mg], grid[2 * Mg:3 * Mg], grid[3 * Mg:4 * Mg]
    pair_grid, pair_h, pair_o = grid[4 * Mg:].view(3, max(Mp, 1)).unbind(0)
    keep = E((2, max(Mp, 1)), torch.int64)
    x_keep, y_keep = keep[0], keep[1]
    sp48 = E((Mg, _capi.SPATIAL_LD), torch.float32)
    _check(lib.skg_pairs_spatial_f32(pre.boxes.data_ptr(), meta.data_ptr(), A, grid_h.data_ptr(), grid_o.data_ptr(),
                                     grid_pair.data_ptr(), grid_img.data_ptr(), pair_grid.data_ptr(), x_keep.data_ptr(),
                                     y_keep.data_ptr(), pair_h.data_ptr(), pair_o.data_ptr(), sp48.data_ptr(), 1, stream),
           "skg_pairs_spatial_f32")
    n_gt = int(gt_off_h[-1])
    own_gt = ()
    if n_gt and arena is not None and all(targets[b]["boxes_h"].dtype == torch.float32 and
                                          targets[b]["boxes_o"].dtype == torch.float32 and
                                          targets[b]["labels"].dtype == torch.int64 for b in act_imgs):
        gt_h = torch.cat([targets[b]["boxes_h"].reshape(-1, 4) for b in act_imgs], out=E((n_gt, 4), torch.float32))
        gt_o = torch.cat([targets[b]["boxes_o"].reshape(-1, 4) for b in act_imgs], out=E((n_gt, 4), torch.float32))
        gt_l = torch.cat([targets[b]["labels"].reshape(-1) for b in act_imgs], out=E(n_gt, torch.int64))
    elif n_gt:
        gt_h = torch.cat([targets[b]["boxes_h"].reshape(-1, 4) for b in act_imgs]).float().contiguous()
        gt_o = torch.cat([targets[b]["boxes_o"].reshape(-1, 4) for b in act_imgs]).float().contiguous()
        gt_l = torch.cat([targets[b]["labels"].reshape(-1) for b in act_imgs]).long().contiguous()
        own_gt = (gt_h, gt_o, gt_l)                                  # (allocations of their own even beside an arena slot)
    else:
        gt_h = E((1, 4), torch.float32).zero_(); gt_o = E((1, 4), torch.float32).zero_()
        gt_l = E(1, torch.int64).zero_()
    labels_all = E((max(Mp, 1), K), torch.float32).zero_()
    npos_d = E(A, torch.int32)
    _check(lib.skg_associate_f32(pre.boxes.data_ptr(), meta.data_ptr(), A, x_keep.data_ptr(), y_keep.data_ptr(),
                                 gt_h.data_ptr(), gt_o.data_ptr(), gt_l.data_ptr(), isl("gt_off").data_ptr(), K,
                                 float(gh.fg_iou_thresh), labels_all.data_ptr(), npos_d.data_ptr(), stream),
           "skg_associate_f32")
    prep.ibuf, prep.offs, prep.meta = ibuf, offs, meta
    prep.arrays = dict(grid_h=grid_h, grid_o=grid_o, grid_pair=grid_pair, grid_img=grid_img, pair_grid=pair_grid,
                       pair_h=pair_h, pair_o=pair_o, x_keep=x_keep[:Mp], y_keep=y_keep[:Mp], sp48=sp48)
    prep.labels = labels_all
    # data parallel: the three n_p normalisers (HEAD:167-172, 194-199, 223-228) depend on labels and detections, not on the
    # logits -- counted here and all-reduced by the second half of the preparation (ONE 3-element collective), a whole
    # forward ahead of the loss that divides by them; round 3 formed them inside the loss, with the collective between its
    # two halves on the step's stream
    prep.norm = None
    counts = None
    force = getattr(head, "force_collectives", False)
    if training and head.distributed and skd.dist.is_available() and skd.dist.is_initialized() and \
            (skd.dist.get_world_size() > 1 or force):
        vt = eng.verbs(dev)
        counts = E(3, torch.float32)
        _check(lib.skg_count_positives_f32(labels_all.data_ptr(), K, pre.scores.data_ptr(), pre.labels.data_ptr(),
                                           meta.data_ptr(), A, x_keep.data_ptr(), y_keep.data_ptr(), vt.off.data_ptr(),
                                           vt.flat.data_ptr(), vt.num_obj, 1.0, counts.data_ptr(), stream),
               "skg_count_positives_f32")
    # the positive counts travel to the host behind the association kernel
    npos_h = torch.empty(A, dtype=torch.int32, pin_memory=True)
    npos_h.copy_(npos_d, non_blocking=True)
    npos_ev = torch.cuda.Event(); npos_ev.record(current_stream_of(dev))
    if before_sync is not None:
        before_sync(prep)
    yield 2
    if counts is not None:
        # the collective leaves from HERE, not from the first half: a prefetching trainer runs the first half while the
        # current step's gradient chunks are still going out, this half behind its optimizer -- every rank, whichever route
        # its current batch took, issues [chunks of step i, normalisers of batch i + 1] in that order on the one communicator
        ex = getattr(head, "grad_exchange", None)
        prep.norm = skd.PreparedNormalisers(counts, group=getattr(ex, "group", None), force=force,
                                            native=getattr(ex, "native", None))
    # host RNG in the reference's order: per image six TransH draws (HEAD:574-580), then randperm(#negatives) (HEAD:939).
    npos_ev.synchronize()                                           # the step's second host synchronisation
    n_pos = npos_h.tolist()
    ppi = [int(v) for v in lay.pairs_per_image]
    neg_cnt = [ppi[a] * K - n_pos[a] for a in range(A)]
    ent_h, rel_h, nrm_h, perm_h = transh.draw_train(K, neg_cnt, n_pos, pin=True)      # tables + randperm heads, natively
    prep.ent = upload(ent_h); prep.rel = upload(rel_h)
    prep.nrm = upload(nrm_h)
    pos_off_h = np.zeros(A + 1, np.int32); pos_off_h[1:] = np.cumsum(n_pos)
    M_pos = int(pos_off_h[-1])
    o_perm = (A + 1 + 1) // 2 * 2                                   # one staging block: pos_off | perm (int64, 8-byte aligned)
    samp_d = E(o_perm + 2 * max(M_pos, 1), torch.int32)
    sh = torch.empty(o_perm + 2 * max(M_pos, 1), dtype=torch.int32, pin_memory=True)
    sh[:A + 1] = torch.from_numpy(pos_off_h)
    sh[o_perm:o_perm + 2 * M_pos].view(torch.int64).copy_(perm_h)
    samp_d.copy_(sh, non_blocking=True)
    prep.n_pos, prep.M_pos = n_pos, M_pos
    prep.pos_off_d = samp_d[:A + 1]
    prep.perm_d = samp_d[o_perm:o_perm + 2 * max(M_pos, 1)].view(torch.int64)
    # ---- TransH term (HEAD:207-235, intended semantics): the scores of the positives and of as many sampled negatives per
    # image, and per image sum_i max(p_i - n_i, -margin).  Nothing here depends on the head's weights (the embeddings are
    # drawn fresh per image), so it belongs to the preparation -- off the step's critical path when prefetched.
    scores_all = E((max(Mp, 1), K), torch.float32)
    _check(lib.skg_transh_scores_f32(prep.ent.data_ptr(), prep.rel.data_ptr(), prep.nrm.data_ptr(), K, gh.human_idx,
                                     meta.data_ptr(), A, scores_all.data_ptr(), stream), "skg_transh_scores_f32")
    tr = E(2 * max(M_pos, 1) + A, torch.float32)                     # pos scores | neg scores | margin partials
    pos_s, neg_s, mpart = tr[:max(M_pos, 1)], tr[max(M_pos, 1):2 * max(M_pos, 1)], tr[2 * max(M_pos, 1):]
    max_pos = max(n_pos) if n_pos else 0
    sws = E(int(lib.skg_transh_sample_ws_ints(A, max_pos)) + max(M_pos, 1), torch.int32)
    _check(lib.skg_transh_sample_f32(labels_all.data_ptr(), scores_all.data_ptr(), K, meta.data_ptr(), A,
                                     prep.pos_off_d.data_ptr(), max_pos, prep.perm_d.data_ptr(), 1.0, sws.data_ptr(),
                                     sws.data_ptr() + 4 * (sws.numel() - max(M_pos, 1)), pos_s.data_ptr(),
                                     neg_s.data_ptr(), mpart.data_ptr(), stream), "skg_transh_sample_f32")
    prep.pos_s, prep.neg_s, prep.mpart = pos_s[:M_pos], neg_s[:M_pos], mpart
    if arena is not None:
        # the slot's block is known to both streams; what is left to hand over one by one: the selection's outputs and
        # whatever did not fit the block this time
        prep.cross = (pre.boxes, pre.scores, pre.labels) + tuple(arena.spill) + own_gt
        prep.slot = arena
    else:
        prep.cross = (pre.boxes, pre.scores, pre.labels, ibuf, grid, keep, sp48, labels_all, prep.ent, prep.rel, prep.nrm,
                      samp_d, gt_h, gt_o, gt_l, npos_d, scores_all, tr, sws)
        if prep.norm is not None:
            prep.cross += (prep.norm.vals,)
    prep.keep = (ent_h, rel_h, nrm_h, sh, npos_h, hbuf)             # pinned staging: alive until the copies have run
    return prep


class TrainRun:
    """The weight-dependent part of one training forward on a Prepared batch: start() = RoI pooling (caller's module),
    global pool, the table-independent dense layers; finish() = the rest of the dense forward, scoring, the three losses."""

    def __init__(self, head, eng, features, image_shapes):
        self.head, self.eng, self.features, self.image_shapes = head, eng, features, image_shapes
        self.job = None
        self.reuse_ws = False        # fused_step: the backward follows the forward at once -> the arena's persistent workspaces

    def start(self, prep):
        head, eng, pre, lay = self.head, self.eng, prep.pre, prep.lay
        dev = pre.device
        box_coords = list(pre.boxes.split(pre.sizes))
        self.box_features = box_features = head.box_roi_pool(self.features, box_coords, self.image_shapes)
        if box_features.shape[0] != lay.sum_all:
            raise _capi.SkgError("box_roi_pool returned %d rows for %d boxes" % (box_features.shape[0], lay.sum_all))
        st = _stacked_for(head, dev)
        job = job_class(head)(head, eng, st, lay, pre, prep.ibuf, prep.offs, None, prep.meta)
        if getattr(self, "exact", False):
            job.bf16 = False                                            # (validation: exact fp32 products)
        job.params = _head_params(head)
        job.S.update(prep.arrays)
        isl = job.isl
        job.pair_img, job.hum_of, job.node_of = isl("pair_img"), isl("hum_of"), isl("node_of")
        job.labels = prep.labels
        # (a differentiable RoI pooling in front hands the step to autograd: its backward may come after another forward)
        job.reuse_ws = self.reuse_ws and not box_features.requires_grad
        self.gfeat = torch.nn.functional.adaptive_avg_pool2d(self.features["3"].float(), 1).flatten(start_dim=1)  # HEAD:811
        with torch.no_grad():
            job.forward_a(box_features, self.gfeat)
        self.job = job

    def tail(self, prep, logits):
        """Everything between the logits and the three loss scalars (runs inside StepFn.forward: no autograd): scoring +
        result packing, both focal terms with d(sum)/d(logits), TransH scores / sampling / margin term, normalisers.
        Returns losses [3] = (hoi, interactiveness, transH)."""
        from . import dist as skd
        lib = _capi.lib()
        head, eng, job, pre, lay = self.head, self.eng, self.job, prep.pre, prep.lay
        gh = head.box_pair_head
        dev = pre.device
        K = head.num_classes
        stream = _stream()
        A, Mp = lay.n_active, lay.sum_p
        f32 = dict(device=dev, dtype=torch.float32)
        i32 = dict(device=dev, dtype=torch.int32)
        meta = prep.meta
        n_pos, M_pos = prep.n_pos, prep.M_pos
        # ---- scoring + result packing (HEAD:721-767, 237-337)
        g = dict(layout=lay, meta=meta, x_keep=prep.arrays["x_keep"], y_keep=prep.arrays["y_keep"])
        self.r = job.result = eng.score(logits, pre, g, True)
        partial = job.loss_forward(logits)
        mpart = prep.mpart                                              # TransH margin partials: made by the preparation
        tail = torch.empty(8, **f32)
        losses, job.loss_scale = tail[:3], tail[4:6]
        # ---- normalisers (HEAD:167-172, 194-199, 223-228) and the scalars: MarginLoss(margin = 1) is
        #      mean(max(p - n, -margin)) + margin, divided by n_p (HEAD:228-234)
        rows = partial.shape[0]
        norm = None
        # (force_collectives: a process group of ONE rank still runs its collectives -- bench.py times the data-parallel
        #  route on a single GPU that way)
        force = getattr(head, "force_collectives", False)
        if prep.norm is not None:
            norm = prep.norm.get()                                      # all-reduced since the preparation: no wait left here
        elif head.distributed and skd.dist.is_available() and skd.dist.is_initialized() and \
                (skd.dist.get_world_size() > 1 or force):
            counts = torch.empty(3, **f32)
            _check(lib.skg_loss_finish_f32(partial.data_ptr(), rows, mpart.data_ptr(), A, M_pos, 1.0, None, None, None,
                                           counts.data_ptr(), stream), "skg_loss_finish_f32")
            ex_ = getattr(head, "grad_exchange", None)
            norm = skd.start_normalisers(counts, True, group=getattr(ex_, "group", None), force=force,
                                         native=getattr(ex_, "native", None)).get().contiguous()     # ONE fused 3-element all-reduce
        ex = getattr(head, "grad_exchange", None)
        share = 1.0
        if ex is not None and norm is not None:
            # data parallel: the exchange SUMS the ranks' gradient arenas; with this rank's share of the mean folded into the
            # scale of its logit gradients (two floats, inside the same kernel) the sum IS the mean -- no averaging pass over
            # the 118 MB arena
            share = 1.0 / ex.world
            ex.prescaled = True
        _check(lib.skg_loss_finish_f32(partial.data_ptr(), rows, mpart.data_ptr(), A, M_pos, 1.0, share, _ptr(norm),
                                       losses.data_ptr(), job.loss_scale.data_ptr(), None, stream), "skg_loss_finish_f32")
        self.pos_s, self.neg_s = prep.pos_s, prep.neg_s
        return losses

    def job_PF(self):
        return self.job.PF[:self.job.lay.sum_p]

    def finish(self, prep):
        head, eng, job, pre, lay = self.head, self.eng, self.job, prep.pre, prep.lay
        dev = pre.device
        job.ent = prep.ent
        job.direct = getattr(head, "grad_mode", "autograd") == "direct"
        if job.direct:
            job.anchor = torch.zeros(1, device=dev, requires_grad=True)      # a leaf that makes autograd call the backward
            hoi, inter, transh = StepFn.apply(self, prep, self.box_features, self.gfeat, job.anchor)
        else:
            hoi, inter, transh = StepFn.apply(self, prep, self.box_features, self.gfeat, *job.params)
        if eng.debug:                                                   # parity tests read these
            head._last_train = dict(pair_features=job.S["PF"][:lay.sum_p], pos_scores=self.pos_s.split(prep.n_pos),
                                    neg_scores=self.neg_s.split(prep.n_pos), job=job)
        # ---- per-image result dicts (views of the packed arrays)
        results = head._results(lay, self.r, dev, train_extras=(job.cell_labels, job.unary))
        results.append(dict(hoi_loss=hoi, interactiveness_loss=inter, transH_loss=transh.detach()))
        return results


def fused_step(head, eng, features, detections, image_shapes, targets, prep=None, after_forward=None,
               defer_backward=False, adamw=None):
    """One whole forward + backward of the training step WITHOUT the autograd engine, for a trainer that owns the loop
    (skghoi_amd.trainer.train_step): the same kernels in the same order as `StepFn`, with the upstream gradient of the
    three summed losses (utils.py:221: their plain sum) fixed at one, the gradients written into a persistent arena whose
    views stay assigned to `p.grad`.  Saves the Function / engine bookkeeping, the thread hop into the engine's device
    thread and 2 x 408 attribute writes per step (~0.2 ms of a 2 ms step).  Returns (results with the loss dict appended --
    detached scalars --, prep) or (None, prep) when this batch / configuration needs the autograd route: inputs that require
    grad (a trainable detector in front), the Python launch plan, a batch without pairs.
    after_forward(): called once the forward is enqueued (the trainer starts the next batch's preparation there).
    defer_backward: the backward's launch calls go to the library's worker thread (NativeJob.backward defer=True) and this
    thread carries on -- the next batch's preparation, the result dicts -- while they are issued; without it the call
    returns with everything enqueued."""
    if job_class(head) is not NativeJob or getattr(head, "grad_mode", "autograd") != "direct":
        return None, prep
    if any(getattr(t, "requires_grad", False) for t in features.values()):
        return None, prep
    run = TrainRun(head, eng, features, image_shapes)
    run.reuse_ws = True
    # measurement (bench.py): HIP events on the step's stream at the phase boundaries -- f0 forward begins, b0 backward
    # begins (= forward + losses done), b1 behind the backward's last launch, o1 behind the optimizer (trainer.train_step)
    spans = head.__dict__.get("_train_spans")
    sp = None
    if spans is not None:
        sp = dict(f0=torch.cuda.Event(enable_timing=True), b0=torch.cuda.Event(enable_timing=True))
        spans.append(sp)
    if prep is None:
        if sp is not None:
            sp["f0"].record()
        prep = prepare_train(head, eng, detections, image_shapes, targets, before_sync=run.start)
        if prep.empty:
            return None, prep
    else:
        if prep.empty:
            return None, prep
        if prep.ready is not None:
            main = current_stream_of(None)
            main.wait_event(prep.ready)
            for t in prep.cross:
                t.record_stream(main)
        if sp is not None:
            sp["f0"].record()
        run.start(prep)
    if run.box_features.requires_grad:
        # a differentiable RoI pooling in front: its backward belongs to autograd.  (start() has run: finish on that route)
        return run.finish(prep), prep
    job, lay = run.job, prep.lay
    dev = prep.pre.device
    with torch.no_grad():
        job.ent = prep.ent
        job.direct = True
        S = job.forward(run.box_features, run.gfeat)
        losses = run.tail(prep, S["logits"])
        if after_forward is not None and not defer_backward:
            after_forward()
        st = job.st
        ga, views = st.persistent_grads()
        src = job.dlogits
        d = torch.empty_like(src)
        one = st.__dict__.get("_one")
        if one is None:
            one = st._one = torch.ones(1, device=dev, dtype=torch.float32)
        _check(_capi.lib().skg_scale_dlogits_f32(src.data_ptr(), src.stride(0), src.shape[0], job.K,
                                                 job.loss_scale.data_ptr(), one.data_ptr(), one.data_ptr(), d.data_ptr(),
                                                 _stream()), "skg_scale_dlogits_f32")
        if sp is not None:
            sp["b0"].record()
        job.backward(d, False, False, arena=(ga, views), defer=defer_backward, span=sp if defer_backward else None,
                     adamw=adamw if defer_backward else None)
        if sp is not None and "b1" not in sp:
            sp["b1"] = torch.cuda.Event(enable_timing=True); sp["b1"].record()
        ctx = context_for(head)
        if ctx.pending:
            # the index arrays and tables of the prepared batch are named by the plan too: dropping them now would let the
            # allocator hand their blocks out behind an event recorded in the MIDDLE of the backward
            ctx.pending.append((prep, run, features))
        if after_forward is not None and defer_backward:
            after_forward()                                       # (host work beside the worker's launch calls)
        if not all(map(_is, map(_grad_of, st.src), views)):       # (first step, or someone re-pointed / cleared a .grad)
            for p, v in zip(st.src, views):
                if p.requires_grad:
                    p.grad = v
    if eng.debug:
        head._last_train = dict(pair_features=run.job_PF(), pos_scores=run.pos_s.split(prep.n_pos),
                                neg_scores=run.neg_s.split(prep.n_pos), job=job)
    results = head._results(lay, run.r, dev, train_extras=(job.cell_labels, job.unary))
    results.append(dict(hoi_loss=losses[0], interactiveness_loss=losses[1], transH_loss=losses[2]))
    return results, prep


def train_forward(head, eng, features, detections, image_shapes, targets, prep=None):
    """InteractionHead.forward in training mode (HEAD:380-429) on the fused step.  prep: a Prepared batch made ahead of
    time (prefetch_train) or None (prepared inline).  Returns (results with the loss dict appended, prep); results is
    None when the batch has no image with pairs (the caller takes the generic path from prep.pre)."""
    run = TrainRun(head, eng, features, image_shapes)
    if prep is None:
        prep = prepare_train(head, eng, detections, image_shapes, targets, before_sync=run.start)
        if prep.empty:
            return None, prep
    else:
        if prep.empty:
            return None, prep
        if prep.ready is not None:                     # made on the side stream: order it before this stream's use
            main = current_stream_of(None)
            main.wait_event(prep.ready)
            for t in prep.cross:
                t.record_stream(main)
        run.start(prep)
    return run.finish(prep), prep


def validate_forward(head, eng, features, detections, image_shapes, targets, prep=None):
    """InteractionHead.forward in EVAL mode with targets (validation: utils.py:283-299 runs the net in eval mode on batches
    that carry their targets, at batch 4 -- main:55-63) on the fused step's machinery: the native preparation (selection
    without GT boxes, pairs + spatial encoding, label association, the reference's host RNG stream) and the native launch
    plan's dense FORWARD in exact fp32 -- no losses, no backward, the eval score power (HEAD:742).  The results carry
    `labels` / `unary_labels` like the reference's (HEAD:323-327).  Until round 5 this mode went through the round-1 autograd
    graph (train_graph.py): ~100 torch ops per batch.  prep: a Prepared batch made ahead of time (prefetch_train in eval
    mode) or None.  Returns the result list, or None when the batch has no image with pairs (the caller's generic path)."""
    run = TrainRun(head, eng, features, image_shapes)
    run.reuse_ws = True                      # no backward will read the workspace: the arena's persistent one serves every batch
    run.exact = True                         # validation scores in the reference's own arithmetic, whatever the training precision
    if prep is None:
        prep = prepare_train(head, eng, detections, image_shapes, targets, before_sync=run.start, training=False)
        if prep.empty:
            return None
    else:
        if prep.empty or getattr(prep, "training", True):
            return None
        if prep.ready is not None:
            main = current_stream_of(None)
            main.wait_event(prep.ready)
            for t in prep.cross:
                t.record_stream(main)
        run.start(prep)
    job, pre, lay = run.job, prep.pre, prep.lay
    job.ent = prep.ent
    with torch.no_grad():
        S = job.forward(run.box_features, run.gfeat)
        logits = S["logits"]
        g = dict(layout=lay, meta=prep.meta, x_keep=prep.arrays["x_keep"], y_keep=prep.arrays["y_keep"])
        job.result = r = eng.score(logits, pre, g, False)            # (eval: prior scores to the power 2.8)
        job.loss_forward(logits)                                     # the labels at the scored cells / per pair (HEAD:323-327)
    if eng.debug:
        head._last_train = dict(pair_features=S["PF"][:lay.sum_p], pos_scores=prep.pos_s.split(prep.n_pos),
                                neg_scores=prep.neg_s.split(prep.n_pos), job=job)
    results = head._results(lay, r, pre.device, train_extras=(job.cell_labels, job.unary))
    job.S = None
    return results


def _head_params(head):
    """box_pair_head | suppressor | predictor parameters in registration order; the module walk (~0.7 ms for 408
    tensors) is redone only after a module / parameter registration anywhere (engine.module_registration_hook)."""
    ep = _reg_epoch()
    c = getattr(head, "_train_params", None)
    if c is None or c[0] != ep:
        c = (ep, list(head.box_pair_head.parameters()) + list(head.box_pair_suppressor.parameters()) +
             list(head.box_pair_predictor.parameters()))
        head._train_params = c
    return c[1]


def _reg_epoch():
    from .engine import _REG_EPOCH
    return _REG_EPOCH[0]
