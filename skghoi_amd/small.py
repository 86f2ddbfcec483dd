"""Small-batch inference: the eval forward replayed from a captured hipGraph.

The reference evaluates ONE image per forward (utils.py:166-167 asserts it).  At that size the head is launch-bound:
~40 kernels of a few microseconds each, every one paying Python + ctypes + allocator time on the host.  Here the part of
the forward whose launch sequence depends only on the batch's SHAPE -- (humans, nodes) per image -- is captured once
per shape into a hipGraph (torch.cuda.CUDAGraph) and replayed:

    per call (eager, ~6 launches):  preprocess/NMS  -> [the one D2H of the counts] -> plan lookup by shape
                                    -> one H2D of the per-call records (image meta, cell count, TransH entity tables,
                                       drawn straight into the pinned staging buffer with the reference's RNG stream)
                                    -> pack detections into the plan's static buffers -> box_roi_pool (caller's module)
                                    -> global average pool + box_head layer 1 (they read caller-owned tensors)
    graph replay (~25 kernels):     box_head layer 2, pairs + spatial encoding, spatial head, the MBF GEMMs, softmax
                                    aggregation, LayerNorms, read-out, classifier, prior + scoring + compaction
    per call:                       one copy of the result arena out of the plan's static output buffer

Everything inside the graph is the same kernel sequence the eager path enqueues for a single chunk (HeadEngine.
_chunk_phase_a / _chunk_phase_b / _classify / score), so results are bit-identical to it.  What varies between calls of
the same shape never enters a launch argument: image sizes and result offsets travel in the device-side meta records,
the number of scored cells in a device word (skg_postprocess_f32's L_total_dev).

SHAPE BUCKETS (single images, the reference's evaluation mode): a dataset walks hundreds of distinct (humans, nodes)
shapes -- up to 15 x 30 at the default caps -- and a plan per exact shape means a capture (eager pass + capture + teardown
of an evicted plan, ~10-20 ms) on most images of the first epoch.  A single image therefore takes the plan of its BUCKET
(ch, cn) = the next capacities >= (n_h, n) out of a geometric ladder (1, 2, 3, 4, 6, 8, 11, 15, 21, 30, ...): every row
space of the plan is sized for (ch, cn), the launches walk all capacity rows, and the image's true (n_h, n) travel in the
device-side meta record like everything else that varies per call.  The kernels that work per graph (pair enumeration,
softmax aggregation, scoring) read (n_h, n) from the record; the row-wise ones (GEMMs, LayerNorm, products) compute finite
garbage in the unused rows, which nothing reads: the pair kernel fills the index arrays of the unused tail with safe
entries and zero features (skg_pairs_spatial_padded_f32).  A few dozen plans cover every shape; padded work costs
at most (1.4)^2 of the rows.  Results equal the eager path's up to the summation order of split-K reductions (their
factors follow the capacity), i.e. ~1e-7 relative; integer outputs are identical.  Batches of 2..8 images keep one plan per
exact shape tuple, captured when the tuple is seen for the second time (engine.small_capture_after): the first call with a
new tuple takes the eager path, so a stream whose tuples never repeat never pays a capture.

Reference path replaced: heads/adamixer_transH_spatial_r50_head.py:341-429 (InteractionHead.forward, eval mode).
"""
import copy
import threading
import ctypes as C
from collections import OrderedDict

import numpy as np
import torch

from . import _capi, layout, transh
from .engine import Preprocessed, current_stream_of, enqueue_row_exponents, gemm_desc, gemm_group, pick_split_k, _stream

META_WORDS = layout.META_DTYPE.itemsize // 4


# ---- lifetime of the captured graphs
# A hipGraphExec is destroyed HERE and nowhere else: `reap()`, on an idle device, before anything of a new capture exists.
# A plan only borrows its graph.  Plans die whenever Python says so -- by reference count when a test drops its head, by the
# cyclic collector at any allocation (engine <-> runner is a cycle), i.e. possibly between another plan's capture and its
# first replay -- and the HIP runtime has crashed in hipGraphLaunch next to such a teardown (once more
# in round 5: gpurun_out/r5y, a full test run, the replay right behind a fresh capture).  So the graph and the events its
# capture recorded are held by this table; a dying plan merely marks its entry.
_KEPT = {}                 # token -> (CUDAGraph, events of the capture)
_DEAD = []                 # tokens whose plan is gone or retired: destroyed by the next reap()
_NEXT = [0]
_LOCK = threading.RLock()  # (a plan's __del__ may run on any thread, also inside reap(): re-entrant)


def keep_graph(graph, events):
    with _LOCK:
        _NEXT[0] += 1
        _KEPT[_NEXT[0]] = (graph, events)
        return _NEXT[0]


def release_graph(token):
    """The graph will not be launched again; its destruction waits for reap()."""
    with _LOCK:
        if token is not None and token in _KEPT and token not in _DEAD:
            _DEAD.append(token)


def reap():
    """Destroys every released graph -- with the device idle, on the calling thread, now.  Returns how many."""
    with _LOCK:
        if not _DEAD:
            return 0
        dead = list(_DEAD)
        del _DEAD[:]
    torch.cuda.synchronize()
    with _LOCK:
        gone = [_KEPT.pop(t, None) for t in dead]
    n = len(gone)
    del gone               # hipGraphExecDestroy / hipGraphDestroy / hipEventDestroy run here
    return n


def _at_exit():
    """Interpreter exit: the graphs go while the HIP runtime is still up (atexit runs before module globals are torn down)."""
    try:
        with _LOCK:
            _DEAD[:] = list(_KEPT)
        if _DEAD and torch.cuda.is_initialized():
            reap()
        else:
            _KEPT.clear(); del _DEAD[:]
    except Exception:      # noqa: BLE001
        pass


import atexit  # noqa: E402
atexit.register(_at_exit)


class _Plan:
    caps = None            # (grid rows, pair rows) the image owns when the plan serves a bucket of shapes
    token = None           # entry of the graph table above

    @property
    def graph(self):       # borrowed: None before the capture and once the graph has been reaped
        ent = _KEPT.get(self.token)
        return ent[0] if ent is not None else None

    def __del__(self):     # (whenever: reference count, cyclic GC, interpreter exit -- never destroys a graph itself)
        try:
            release_graph(self.token)
        except Exception:  # noqa: BLE001  (module globals already torn down at exit)
            pass


def capacity(v, limit):
    """Smallest rung >= v of the ladder 1, 2, 3, 4, 6, 8, 11, 15, 21, 30, 42, ... (ratio ~1.4), capped at `limit`."""
    c, step = 1, 0
    ladder = (1, 2, 3, 4, 6, 8, 11, 15, 21, 30, 42, 60, 80, 112, 160)
    for c in ladder:
        if c >= v:
            break
    return max(min(c, limit), v)


class SmallBatchRunner:
    """Owns the captured plans of one HeadEngine (LRU by shape key)."""

    def __init__(self, engine, max_plans=128):
        self.eng = engine
        self.max_plans = max_plans
        self.plans = OrderedDict()
        self.pool = None                   # graph memory pool shared by the plans (one forward at a time per engine)
        self.hits = self.misses = 0
        self.deferred = 0                  # calls that took the eager path because their exact shape had not been seen often enough
        self.sightings = OrderedDict()     # exact-shape keys seen but not captured yet -> count
        self.side = None                   # second stream of the plan bodies (second branch of the captured graphs)
        self.epoch = engine.plan_epoch
        self.retired = []                  # dropped plans: destroyed only on an idle device, never next to a capture
        self.captures = self.evictions = 0
        self._ent_stage = None             # one image's entity table, drawn ahead of the count synchronisation
        self._predrawn = None              # generator state from before that draw, while the draw is still unclaimed
        self._ahead = None                 # the NEXT image's selection, launched beside the current forward (look_ahead)
        self._ahead_stage = None
        self.ahead_hits = 0

    def _retire(self, plans):
        plans = list(plans)
        for p in plans:
            release_graph(getattr(p, "token", None))
        self.retired.extend(plans)

    def stats(self):
        n = self.hits + self.misses
        return dict(calls=n, hits=self.hits, misses=self.misses, hit_rate=(self.hits / n if n else None),
                    captures=self.captures, evictions=self.evictions, plans=len(self.plans), deferred=self.deferred,
                    look_ahead_hits=self.ahead_hits)

    def close(self):
        """Drops every plan through the idle-device path (engine replaced, head torn down, tests)."""
        self.drop_look_ahead()
        self._retire(self.plans.values())
        self.plans.clear()
        self._bury_retired()

    def _bury_retired(self):
        """Destroys dropped plans (their hipGraphExec, pool blocks, events, pinned staging) -- with the device idle and
        before anything of a new capture exists.  Tearing graphs down while the next one is being captured or replayed
        on the same streams and pool is where the runtime has crashed (rarely) in hipGraphLaunch."""
        if self.retired or _DEAD:
            import gc
            torch.cuda.synchronize()       # the idle point: the dropped plans' buffers go first ...
            self.retired.clear()
            gc.collect()                   # ... dead cycles (engine <-> runner <-> plans of heads long dropped) mark their graphs
            reap()                         # ... and every released graph of the process is destroyed here
            if self.pool is not None and not any(q.graph is not None for q in self.plans.values()):
                # the last graph of this runner's memory pool is gone (weights changed, engine closed): torch retires a graph
                # pool with its last graph -- the next capture starts a new one
                self.pool = None

    # ------------------------------------------------------------------------------------------------ plan
    def _build_plan(self, key, pre, lay, pw, feat3, pooled):
        eng = self.eng
        dev = pre.device
        f32 = dict(device=dev, dtype=torch.float32)
        p = _Plan()
        p.key = key
        p.pw = pw
        p.lay = lay
        A = lay.n_active
        NA = lay.sum_all
        K = eng.K
        p.ch = layout.chunk(lay, 0, A)
        buf, p.offs = layout.pack_int_arrays(p.ch)
        p.ibuf = torch.from_numpy(buf).to(dev)
        # per-call records: [meta A x 12 words | L_total, pad x3 | ent A x 80 x 50 floats], one pinned block, one H2D
        n_meta = A * META_WORDS
        words = n_meta + 4 + A * _capi.TRANSH_ENT * _capi.TRANSH_DIM
        p.dyn_host = torch.empty(words, dtype=torch.int32, pin_memory=True)
        p.dyn_dev = torch.empty(words, dtype=torch.int32, device=dev)
        p.meta_host = p.dyn_host[:n_meta].numpy().view(layout.META_DTYPE)
        p.meta_host[:] = lay.meta                        # everything but out_off / img_h / img_w is fixed by the shape
        p.meta_i32 = p.dyn_host[:n_meta].numpy().reshape(A, META_WORDS)
        p.meta_f32 = p.meta_i32.view(np.float32)
        p.lt_host = p.dyn_host[n_meta:n_meta + 4].numpy()
        p.ent_host = p.dyn_host[n_meta + 4:].view(torch.float32).view(A, _capi.TRANSH_ENT, _capi.TRANSH_DIM)
        p.meta_dev = p.dyn_dev[:n_meta]
        p.lt_dev = p.dyn_dev[n_meta:n_meta + 4]
        p.ent_dev = p.dyn_dev[n_meta + 4:].view(torch.float32).view(A, _capi.TRANSH_ENT, _capi.TRANSH_DIM)
        p.h2d_done = None
        # static inputs of the graph
        p.boxes = torch.zeros(max(NA, 1), 4, **f32)[:NA]
        p.scores = torch.zeros(max(NA, 1), **f32)[:NA]
        p.labels = torch.zeros(max(NA, 1), dtype=torch.int64, device=dev)[:NA]
        p.sel_off = eng._det_offsets(pre.sizes, dev)
        p.gfeat = torch.zeros(feat3.shape[0], feat3.shape[1], **f32)
        p.enc1 = torch.zeros(max(NA, 1), 1024, **f32)
        kp = pw.bh1_w.shape[1]
        p.x0_pad = torch.zeros(NA, kp, **f32) if pooled[0].numel() != kp else None
        p.sk = pick_split_k(NA, 1024, kp)
        p.ws = torch.empty(p.sk, NA, 1024, **f32) if p.sk > 1 else None
        p.bh1_desc = None                                # filled on first use (needs the weight-twin context)
        p.pre = Preprocessed()
        p.pre.device = dev; p.pre.B = pre.B
        p.pre.boxes, p.pre.scores, p.pre.labels = p.boxes, p.scores, p.labels
        # static outputs: one byte arena [index | prediction | object | scores | prior (2 rows) | weights | boxes_h | boxes_o]
        Mp = lay.sum_p
        Lmax = (max(Mp * K, 1) + 3) // 4 * 4        # region sizes in multiples of 4 words: every array stays 16-byte aligned
        Mq = (max(Mp, 1) + 3) // 4 * 4
        p.Mp, p.Lmax, p.Mq = Mp, Lmax, Mq
        n_i64 = 2 * Lmax + Mq
        n_f32 = 3 * Lmax + Mq + 8 * Mq
        p.arena = torch.zeros(8 * n_i64 + 4 * n_f32, dtype=torch.uint8, device=dev)
        p.n_i64 = n_i64
        i64 = p.arena[:8 * n_i64].view(torch.int64)
        f = p.arena[8 * n_i64:].view(torch.float32)
        p.r = dict(index=i64[:Lmax], prediction=i64[Lmax:2 * Lmax], object=i64[2 * Lmax:],
                   scores=f[:Lmax], prior=f[Lmax:3 * Lmax], weights=f[3 * Lmax:3 * Lmax + Mq],
                   boxes_h=f[3 * Lmax + Mq:3 * Lmax + 5 * Mq].view(Mq, 4), boxes_o=f[3 * Lmax + 5 * Mq:].view(Mq, 4))
        p.out = None
        return p

    def _body(self, p):
        """The shape-dependent part of the forward (HEAD:812-982, 408-411, 237-337) for a single chunk.

        Two branches run side by side (two streams while capturing = two branches of the hipGraph): at one image every
        kernel fills a fraction of the chip, and the spatial chain (pairs -> spatial head -> global read-out branch) shares
        nothing with the box_head -> fc_head/fc_tail -> fc_1 chain until the fc_2 GEMMs."""
        eng, pw, lay = self.eng, p.pw, p.lay
        dev = p.pre.device
        f32 = dict(device=dev, dtype=torch.float32)
        NA, Mp = lay.sum_all, lay.sum_p
        main = torch.cuda.current_stream()
        if self.side is None:
            from .engine import shared_side_stream
            self.side = shared_side_stream(dev, 0, slot=0)      # (process-wide: every live stream costs a hardware queue)
        side = self.side
        if not eng.small_two_branches:
            side = main                        # one chain: the captured graph needs no stream of its own beside the caller's
        with eng._split_ctx(pw):
            enc = torch.empty(max(NA, 1), 1024, **f32)
            Bf, Cf = p.gfeat.shape
            G1 = torch.empty(Bf, 1024, **f32)
            x_keep = torch.empty(max(Mp, 1), device=dev, dtype=torch.int64); y_keep = torch.empty_like(x_keep)
            PF = torch.empty(max(Mp, 1), 2048, **f32)
            fork = torch.cuda.Event(); fork.record(main)
            bh3 = ((p.enc1, pw.bh3_w, pw.bh3_b, enc, NA, 1024, 1024, _capi.EPI_BIAS_RELU), {})       # box_head layer 2 (HEAD:812)
            g1 = ((p.gfeat, pw.att_g["w1"], pw.att_g["b1"], G1, Bf, 1024, Cf, _capi.EPI_BIAS), {})   # attention_head_g fc_1 (HEAD:971)
            if eng.g1_on_side(Bf):
                # The graph forks ONCE, at its root: the global branch's fc_1 opens the side chain instead of riding in the main
                # chain's first launch.  Grouped with box_head layer 2 it gave that launch two successors -- the main chain's
                # entity rows and the side chain's read-out -- and the replayed graph continued BOTH on other queues: a
                # cross-queue hand-over of 15-20 us in the middle of the critical chain (round 5, r05_b1_eval_timeline*.txt).
                # (capture ORDER matters too: hipGraphLaunch enqueues a graph chain by chain, the chain of the root captured
                # first going out first.  With box_head layer 2 captured first the main chain starts 55 us earlier and the side
                # chain 38 us later; the three-product group then runs beside the global branch's two products, all of them
                # slower for it -- 0.480-0.488 against 0.471-0.473 ms, profiles/r05_b1_eval_capture_order.txt.)
                with torch.cuda.stream(side):
                    side.wait_event(fork)
                    gemm_group([g1])
                    cx = eng._chunk_phase_a(p.ch, pw, p.pre, None, x_keep, y_keep, PF, ibuf=p.ibuf, offs=p.offs,
                                            meta=p.meta_dev, caps=p.caps)
                    s_ready = torch.cuda.Event(); s_ready.record(side)
                    eng._chunk_phase_a2(cx, pw, p.pre, G1, PF)
                    g_done = torch.cuda.Event(); g_done.record(side)
                gemm_group([bh3])
                g1_ready = fork
            else:
                with torch.cuda.stream(side):
                    side.wait_event(fork)
                    cx = eng._chunk_phase_a(p.ch, pw, p.pre, None, x_keep, y_keep, PF, ibuf=p.ibuf, offs=p.offs,
                                            meta=p.meta_dev, caps=p.caps)
                    s_ready = torch.cuda.Event(); s_ready.record(side)
                # box_head layer 2 and attention_head_g's fc_1 on the global features: independent, one launch
                gemm_group([bh3, g1])
                g1_ready = torch.cuda.Event(); g1_ready.record(main)
                with torch.cuda.stream(side):
                    side.wait_event(g1_ready)
                    eng._chunk_phase_a2(cx, pw, p.pre, G1, PF)
                    g_done = torch.cuda.Event(); g_done.record(side)
            eng._chunk_phase_b(cx, (p.ent_dev, None, None), pw, p.pre, enc, PF, None, None,
                               need_S=lambda: main.wait_event(s_ready), need_Tg=lambda: main.wait_event(g_done))
            main.wait_event(g_done)
            logits = eng._classify(PF[:Mp], pw)
            g = dict(layout=lay, meta=p.meta_dev, x_keep=x_keep[:Mp], y_keep=y_keep[:Mp])
            eng.score(logits, p.pre, g, False, out=p.r, L_dev=p.lt_dev)
        # everything both branches touched stays referenced until the plan dies: a block handed back to the allocator
        # inside the capture could be given to the other branch while this one still uses it
        # (the fork / join events too: a captured graph must not outlive anything its capture touched)
        return dict(logits=logits, pair_features=PF[:Mp], x_keep=x_keep[:Mp], y_keep=y_keep[:Mp], enc=enc, _cx=cx, _G1=G1,
                    _events=(fork, s_ready, g1_ready, g_done))

    def _capture(self, p):
        self._bury_retired()
        if self.pool is None:
            self.pool = torch.cuda.graph_pool_handle()
        p.out = self._body(p)                      # eager once: first-use work (weight twins, lazy module init) happens here
        torch.cuda.current_stream().synchronize()
        g = torch.cuda.CUDAGraph()
        # No cyclic garbage collection while the stream captures: a collection that happens to run inside the capture
        # frees whatever dead cycles hold -- pinned staging buffers, HIP events, other plans' graphs -- and those frees are
        # not allowed on a capturing thread (the process aborts).  torch.cuda.graph collects once before it starts.
        import gc
        gc_on = gc.isenabled()
        gc.disable()
        from . import engine as _engine
        _engine.AMAX_CAPTURE_KEEP = p.amax_keep = []
        try:
            with torch.cuda.graph(g, pool=self.pool, capture_error_mode="thread_local"):
                p.out = self._body(p)
        finally:
            _engine.AMAX_CAPTURE_KEEP = None
            if gc_on:
                gc.enable()
        p.token = keep_graph(g, p.out.get("_events") if isinstance(p.out, dict) else None)
        self.captures += 1

    # ------------------------------------------------------------------------------------------------ forward
    def eligible(self, head, detections, targets):
        eng = self.eng
        return (not head.training and targets is None and 0 < len(detections) <= eng.small_batch_max
                and not eng.debug)

    def look_ahead(self, head, detections, after=None):
        """The selection of the NEXT single image (score filter, class-wise NMS, top-k: HEAD:92-151), its count read-back, the
        parameter checksum and its TransH entity-table draw, started NOW on the head's side stream -- beside the forward
        that was enqueued a moment ago -- instead of at the top of the next forward, where the GPU would sit idle for the
        selection kernel plus one device -> host round trip (~50 us of a 0.45 ms forward).  The next `forward` called
        with this same `detections` object picks the result up; any other call puts the global CPU generator back where
        it stood and starts over, so results and RNG position are those of a loop without look-ahead.

        after: an event recorded behind whatever produced `detections` (a loader's uploads).  The side stream waits for it
        and for nothing else; None = everything the current stream holds right now, which is always safe but puts the
        selection BEHIND the forward in flight.  The parameters must not change between this call and that forward (the
        checksum that notices a changed weight is taken here)."""
        eng = self.eng
        self.drop_look_ahead()
        if len(detections) != 1 or not self.eligible(head, detections, None):
            return False
        dev = detections[0]["boxes"].device
        side = head._prefetch_stream(dev)
        if after is None:
            after = torch.cuda.Event()
            after.record(current_stream_of(dev))
        side.wait_event(after)
        for t in detections[0].values():
            if torch.is_tensor(t) and t.is_cuda:
                t.record_stream(side)
        from .engine import on_stream
        try:
            with on_stream(side):
                st = eng.pre_launch(detections, None, False, False, check_weights=True, defer=True)
        except Exception:                          # noqa: BLE001
            # whatever is wrong with these detections (too many of them, a tensor on another device) is the NEXT forward's to
            # report, from the place a loop without look-ahead sees it -- not this call's, which runs while the results of
            # the forward before are still on their way to the caller
            return False
        state = torch.get_rng_state()
        if self._ahead_stage is None:
            self._ahead_stage = torch.empty(1, _capi.TRANSH_ENT, _capi.TRANSH_DIM)
        transh.draw_batch(eng.K, 1, need_relations=False, out=(self._ahead_stage, None, None))
        self._ahead = (detections, detections[0], st, state, side)
        return True

    def drop_look_ahead(self):
        """An unclaimed look-ahead is abandoned: the generator goes back to where it stood before its table draw."""
        if self._ahead is not None:
            torch.set_rng_state(self._ahead[3])
            self._ahead = None

    def _claim_look_ahead(self, detections):
        a = self._ahead
        if a is None:
            return None
        if a[0] is not detections or len(detections) != 1 or detections[0] is not a[1]:
            self.drop_look_ahead()
            return None
        self._ahead = None
        st, state = a[2], a[3]
        main = current_stream_of(st["dev"])
        kind, _host, ev = st["pending"]
        if ev is not None:
            main.wait_event(ev)                    # the selection's outputs were written on the side stream ...
        else:
            main.wait_stream(a[4])                 # (blocking-copy mode: no event of its own)
        for t in (st["index"], st["countx"], st["raw"][3]):
            t.record_stream(main)                  # ... into blocks of ITS pool, which this stream's launches now read
        self.ahead_hits += 1
        return st, state

    def forward(self, head, features, detections, image_shapes):
        """Returns the list of result dicts, or None when this batch has to take the eager path."""
        eng = self.eng
        predrawn = None
        ahead = self._claim_look_ahead(detections)
        if ahead is not None:
            st, predrawn = ahead
            self._ent_stage, self._ahead_stage = self._ahead_stage, self._ent_stage      # the table drawn with the look-ahead
            pre = eng.pre_launch_end(st)
        elif len(detections) == 1:
            # one image (the reference's evaluation mode): its TransH entity table (HEAD:574-580: ~4 000 normal draws from the
            # global CPU generator, ~20 us) is drawn while the selection kernel runs, not after the counts have arrived with
            # the GPU idle.  Same generator, same order -- unless the image turns out to have no pairs or the call leaves for
            # the eager path: then the generator is put back and that path draws for itself.
            st = eng.pre_launch(detections, None, False, False, check_weights=True, defer=True)
            state = torch.get_rng_state()
            if self._ent_stage is None:
                self._ent_stage = torch.empty(1, _capi.TRANSH_ENT, _capi.TRANSH_DIM)
            transh.draw_batch(eng.K, 1, need_relations=False, out=(self._ent_stage, None, None))
            predrawn = state
            pre = eng.pre_launch_end(st)
        else:
            pre = eng.pre_launch(detections, None, False, False, check_weights=True)
        self._predrawn = predrawn
        dev = pre.device
        pw = eng.weights(dev, wsum=pre.wsum, walk=False)
        if not pw.fused_cls:
            return self._fallback(head, pre, features, image_shapes)
        if self.epoch != eng.plan_epoch:           # the packed weights were rebuilt: every plan points into the old copies
            self._retire(self.plans.values())
            self.plans.clear()
            self.epoch = eng.plan_epoch
        feat3 = features["3"]
        # single images: one plan per BUCKET of shapes (module docstring); the true (n_h, n) go into the per-call record
        bucket = None
        if pre.B == 1 and eng.small_batch_buckets:
            nh1, n1 = int(pre.n_h[0]), int(pre.n[0])
            if nh1 >= 1 and 2 <= n1 <= _capi.TRANSH_ENT:
                bucket = (capacity(nh1, max(eng.max_human, nh1)),
                          capacity(n1, min(max(eng.max_human + eng.max_object, n1), _capi.TRANSH_ENT)))
        shape_key = ("bucket",) + bucket if bucket else (tuple(pre.n_h.tolist()), tuple(pre.n.tolist()))
        key = shape_key + (tuple(feat3.shape[:2]), eng.precision, eng.gh.num_iter, eng.faithful_skip_offset,
                           eng.plan_epoch, dev.index)
        p = self.plans.get(key)
        if p is None and not bucket and eng.small_capture_after > 1:
            # exact-shape plans (batches of 2..8 images, single images without buckets): a capture costs 10-20 ms -- an eager
            # forward 1-2 -- and pays only for a shape that comes back.  A stream of batches whose shape tuples never repeat
            # (any real dataset at batch 4) would capture on EVERY call; so a shape is captured when it is seen for the
            # `small_capture_after`-th time and served by the eager path until then.
            seen = self.sightings.get(key, 0) + 1
            if seen < eng.small_capture_after:
                self.sightings[key] = seen
                self.sightings.move_to_end(key)
                while len(self.sightings) > 4096:
                    self.sightings.popitem(last=False)
                self.deferred += 1
                return self._fallback(head, pre, features, image_shapes)
            self.sightings.pop(key, None)
        call_lay = None
        if bucket:
            call_lay = layout.single(nh1, n1, int(pre.L[0]), image_shapes[0])
            if call_lay.sum_p == 0:
                return self._fallback(head, pre, features, image_shapes)
        elif p is None:
            call_lay = layout.build(pre.n_h, pre.n, pre.L, image_shapes, eng.human_idx,
                                    faithful_skip_offset=eng.faithful_skip_offset)
            if call_lay.n_active == 0 or call_lay.n_visit == 0 or call_lay.sum_p == 0:
                return self._fallback(head, pre, features, image_shapes)
        if p is not None:
            lay = p.lay
        elif bucket:                                # the plan's layout: one image of exactly the bucket's capacity
            lay = layout.build([bucket[0]], [bucket[1]], None, image_shapes, eng.human_idx,
                               faithful_skip_offset=eng.faithful_skip_offset)
        else:
            lay = call_lay
        n_act = int(sum(pre.sizes))                 # selected boxes of this call (<= lay.sum_all for a bucket plan)
        # ---- static inputs of this call
        new = p is None
        if new:
            self.misses += 1
            eng.pre_pack(pre)
            box_coords = list(pre.boxes.split(pre.sizes))
            pooled = head.box_roi_pool(features, box_coords, image_shapes)
            if pooled.shape[0] != n_act:
                raise _capi.SkgError("box_roi_pool returned %d rows for %d boxes" % (pooled.shape[0], n_act))
            if pooled[0].numel() != pw.bh1_k:
                raise RuntimeError("mat1 and mat2 shapes cannot be multiplied (%dx%d and %dx%d)" % (
                    pooled.shape[0], pooled[0].numel(), pw.bh1_k, 1024))
            p = self._build_plan(key, pre, lay, pw, feat3, pooled)
            if bucket:
                p.caps = (bucket[0] * bucket[1], bucket[0] * (bucket[1] - 1))
            p.boxes[:n_act].copy_(pre.boxes); p.scores[:n_act].copy_(pre.scores); p.labels[:n_act].copy_(pre.labels)
            while len(self.plans) >= self.max_plans:
                self._retire([self.plans.popitem(last=False)[1]])
                self.evictions += 1
            self.plans[key] = p
        else:
            self.hits += 1
            self.plans.move_to_end(key)
            sel_off = p.sel_off if not bucket else eng._det_offsets(pre.sizes, dev)
            eng.pre_pack(pre, out=(p.boxes[:n_act], p.scores[:n_act], p.labels[:n_act]), sel_off=sel_off)
            pooled = head.box_roi_pool(features, list(p.boxes[:n_act].split(pre.sizes)), image_shapes)
            if pooled.shape[0] != n_act:
                raise _capi.SkgError("box_roi_pool returned %d rows for %d boxes" % (pooled.shape[0], n_act))
        # the two launches that read caller-owned tensors go first: global average pool (HEAD:811), box_head layer 1
        # (HEAD:812, 51 MB of weights) -- the GPU works on them while the host fills in the per-call records below
        lib = _capi.lib()
        f3 = feat3 if (feat3.dtype == torch.float32 and feat3.is_contiguous()) else feat3.float().contiguous()
        _capi.check(lib.skg_global_avgpool_f32(f3.data_ptr(), f3.shape[0], f3.shape[1], f3.shape[2] * f3.shape[3],
                                               p.gfeat.data_ptr(), _stream()), "skg_global_avgpool_f32")
        x0 = pooled.reshape(pooled.shape[0], -1)
        if x0.dtype != torch.float32:
            x0 = x0.float()
        if p.x0_pad is not None:
            p.x0_pad[:n_act, :x0.shape[1]] = x0
            x0 = p.x0_pad
        elif not x0.is_contiguous():
            x0 = x0.contiguous()
        if p.bh1_desc is None:
            with eng._split_ctx(pw):
                p.bh1_desc = gemm_desc(x0, pw.bh1_w, pw.bh1_b, p.enc1, lay.sum_all, 1024, x0.shape[1],
                                       _capi.EPI_BIAS_RELU, split_k=p.sk, split_ws=p.ws)
        p.bh1_desc.A = x0.data_ptr()               # the fields that change from call to call: the caller's tensor and (bucket
        p.bh1_desc.M = n_act                       # plans) its row count
        if p.bh1_desc.w_split:
            p.bh1_exp = enqueue_row_exponents(p.bh1_desc, x0.device)
        _capi.check(lib.skg_gemm_f32(C.byref(p.bh1_desc), _stream()), "skg_gemm_f32[box_head 1]")
        # per-call records: meta (image sizes, result offsets), cell count, TransH entity tables
        if p.h2d_done is not None:
            p.h2d_done.synchronize()               # the staging block's previous copy (normally long finished)
        act = lay.active
        L = pre.L[act]
        if lay.n_active == 1:
            b0 = int(act[0])
            p.meta_i32[0, 9] = 0
            p.meta_f32[0, 10] = float(image_shapes[b0][0]); p.meta_f32[0, 11] = float(image_shapes[b0][1])
            Lt = int(L[0])
            if bucket:                             # the image's true graph inside the bucket's capacity
                p.meta_i32[0, 1] = nh1; p.meta_i32[0, 2] = n1
        else:
            p.meta_i32[:, 9] = np.cumsum(L) - L
            p.meta_f32[:, 10] = [float(image_shapes[int(b)][0]) for b in act]
            p.meta_f32[:, 11] = [float(image_shapes[int(b)][1]) for b in act]
            Lt = int(L.sum())
        p.lt_host[0] = Lt
        if self._predrawn is not None and lay.n_active == 1:
            p.ent_host[0].copy_(self._ent_stage[0])          # drawn beside the selection kernel (top of this function)
            self._predrawn = None
        else:
            self._undo_predraw()
            transh.draw_batch(eng.K, lay.n_active, need_relations=False, out=(p.ent_host, None, None))
        p.dyn_dev.copy_(p.dyn_host, non_blocking=True)
        if p.h2d_done is None:
            p.h2d_done = torch.cuda.Event()
        p.h2d_done.record(current_stream_of(dev))
        if p.graph is None:
            self._capture(p)
        p.graph.replay()
        out = p.arena.clone()                      # the plan's output buffer is overwritten by the next replay
        # ---- per-call views of the copied arena
        Lmax, Mq = p.Lmax, p.Mq
        Mp = call_lay.sum_p if bucket else p.Mp    # kept pairs of THIS image (a prefix of the plan's pair rows)
        i64 = out[:8 * p.n_i64].view(torch.int64)
        f = out[8 * p.n_i64:].view(torch.float32)
        r = dict(index=i64[:Lt], prediction=i64[Lmax:Lmax + Lt], object=i64[2 * Lmax:2 * Lmax + Mp],
                 scores=f[:Lt], prior=f[Lmax:Lmax + 2 * max(Lt, 1)].view(2, max(Lt, 1))[:, :Lt],
                 weights=f[3 * Lmax:3 * Lmax + Mp], boxes_h=f[3 * Lmax + Mq:3 * Lmax + Mq + 4 * Mp].view(Mp, 4),
                 boxes_o=f[3 * Lmax + 5 * Mq:3 * Lmax + 5 * Mq + 4 * Mp].view(Mp, 4))
        if bucket:
            eng.last = dict(p.out, layout=call_lay, plan=p, logits=p.out["logits"][:Mp],
                            pair_features=p.out["pair_features"][:Mp], x_keep=p.out["x_keep"][:Mp],
                            y_keep=p.out["y_keep"][:Mp], enc=p.out["enc"][:n_act])
            # one active image, nothing skipped: the views above ARE its result dict (HEAD:317-322) -- what
            # head._results(call_lay, r, dev) returns after re-slicing and splitting each of the eight arrays once more
            return [dict(boxes_h=r["boxes_h"], boxes_o=r["boxes_o"], index=r["index"], prediction=r["prediction"],
                         scores=r["scores"], object=r["object"], prior=r["prior"], weights=r["weights"])]
        call_lay = copy.copy(lay)
        cells = np.zeros(lay.n_active, np.int64); cells[:] = L
        call_lay.cells_per_image = cells
        call_lay.sum_l = Lt
        eng.last = dict(p.out, layout=call_lay, plan=p)
        return head._results(call_lay, r, dev)

    def _undo_predraw(self):
        """Puts the global CPU generator back where it stood before this call's early table draw."""
        if self._predrawn is not None:
            torch.set_rng_state(self._predrawn)
            self._predrawn = None

    def _fallback(self, head, pre, features, image_shapes):
        """Batches without a single kept pair, or with injected non-Linear classifiers: the eager path, continuing from
        the preprocess that has already run."""
        self._undo_predraw()
        self.eng.pre_pack(pre)
        return head._forward_eager(pre, features, image_shapes)
