"""Host orchestration of the interaction-head hot path on MI355X.

Everything numerical runs in libskghoi_hip.so (include/skghoi.h) on the caller's current HIP stream; PyTorch only owns
device memory.  The per-image Python loop of the reference (heads/adamixer_transH_spatial_r50_head.py:822-982) becomes a
fixed sequence of batched launches over concatenated row spaces (skghoi_amd/layout.py):

  preprocess (NMS/top-k) -> [one D2H of per-image counts] -> pack -> roi-pool (injected module) -> box_head GEMMs
  -> pairs + 46-d spatial -> spatial_head GEMMs -> fc_head/fc_tail on UNIQUE node rows
  -> message passing ONCE (iterations never feed back):
        attention_head: fc_1 split into human/object halves on unique rows; fc_2 GEMM with the fc_1*fc_2*ReLU product
        fused in its epilogue; fc_3 GEMM with the adjacency dot fused in its epilogue
        messages: fc_2 GEMMs with fused product, softmax-weighted aggregation BEFORE fc_3 (linear), fc_3 on node rows
        with fused ReLU + residual, LayerNorm
  -> read-out MBFs on kept pairs -> predictor|suppressor GEMM -> prior + scoring + compaction.

There is no CPU fallback: the library must be built and tensors must live on a HIP device.
"""
import ctypes as C
import os
import threading

import numpy as np
import torch

from . import _capi, layout, transh

EPS_LN = 1e-5

# bench.py sets this to a list to time every GEMM launch with HIP events on the launch stream:
# entries are (start_event, end_event, M, N, K, epilogue).  None = no instrumentation (the default).
GEMM_TIMER = None          # bench.py: list collecting (start event, end event, M, N, K, epilogue) per GEMM launch
GEMM_TIMER_EPI = None      # ... restricted to these epilogue ids (None: every launch)

import os as _os
_PIN_TABLES = _os.environ.get("SKG_PIN_TABLES", "1") == "1"     # developer switch
_NO_SPIN = _os.environ.get("SKG_NO_SPIN", "0") == "1"           # developer switch: blocking count read at every batch size


_SHARED_STREAMS = {}


def shared_side_stream(dev, priority=0, slot=0):
    """ONE side stream per (device, priority, slot) for the whole process.

    Every stream of a new priority class / every few streams make the HIP runtime open another hardware queue, and a
    process that keeps more than a handful of them alive gets its kernels scheduled noticeably worse.  Measured on the
    training step: every head used to make its own high-priority look-ahead stream; with the streams of three earlier heads
    still alive (bench.py's training legs after its eval legs; any loop that rebuilds the head) the 4th and 5th head ran the
    SAME kernels at 1.98 instead of 1.40 ms per step (fp32: 3.13 instead of 2.65) -- and so did every head when six such
    streams were opened up front (tools/_trainleg_probe.py).  One stream, made once, keeps the first head's speed."""
    dev = torch.device(dev)
    key = (dev.index if dev.index is not None else torch.cuda.current_device(), priority, slot)
    st = _SHARED_STREAMS.get(key)
    if st is None:
        st = _SHARED_STREAMS[key] = torch.cuda.Stream(device=dev, priority=priority)
    return st


def _ptr(t):
    return 0 if t is None else t.data_ptr()


_RAW_STREAM = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream():
    """hipStream_t of torch's current stream on the current device (one C call; ~45 launches per forward need it)."""
    if _RAW_STREAM is not None:
        return _RAW_STREAM(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def current_stream_of(dev):
    """torch.cuda.current_stream(dev) through the integer fast path: with a torch.device -- or nothing -- torch resolves the
    index through _get_available_device_type() -> torch.cuda.is_available(), ~10 us per call (measured: seven calls, 74 us of
    a 1.4 ms training step's host thread)."""
    idx = dev if isinstance(dev, int) else (dev.index if dev is not None and dev.index is not None else None)
    return torch.cuda.current_stream(torch.cuda.current_device() if idx is None else idx)


_SET_STREAM = getattr(torch._C, "_cuda_setStream", None)


class on_stream:
    """`with torch.cuda.stream(s)` for a stream of the CURRENT device, without the context manager's device-less
    current_stream() lookup (above) and its device switch: two C calls in, one out."""
    __slots__ = ("s", "prev")

    def __init__(self, s):
        self.s = s

    def __enter__(self):
        s = self.s
        if _SET_STREAM is None or s.device_index != torch.cuda.current_device():
            self.prev = torch.cuda.stream(s)          # (another device's stream: the stock manager handles the switch)
            self.prev.__enter__()
            return s
        self.prev = torch.cuda.current_stream(s.device_index)
        _SET_STREAM(stream_id=s.stream_id, device_index=s.device_index, device_type=s.device_type)
        return s

    def __exit__(self, *exc):
        p = self.prev
        if isinstance(p, torch.cuda.StreamContext):
            return p.__exit__(*exc)
        _SET_STREAM(stream_id=p.stream_id, device_index=p.device_index, device_type=p.device_type)
        return False


class on_device:
    """`with torch.cuda.device(dev)` that does nothing when dev already is the current device (the usual case: one
    process per GPU)."""
    __slots__ = ("ctx",)

    def __init__(self, dev):
        idx = dev if isinstance(dev, int) else dev.index
        self.ctx = None if (idx is None or idx == torch.cuda.current_device()) else torch.cuda.device(idx)

    def __enter__(self):
        if self.ctx is not None:
            self.ctx.__enter__()

    def __exit__(self, *exc):
        if self.ctx is not None:
            return self.ctx.__exit__(*exc)
        return False


# Bumped whenever ANY nn.Module registers a parameter or a sub-module (module.weight = nn.Parameter(...),
# load_state_dict(assign=True), head.box_pair_predictor = nn.Linear(...)): the engine then re-enumerates its parameters
# (by identity) before trusting its packed copies.  O(1) per forward while nothing is registered.
_REG_EPOCH = [0]


def _registration_hook(*_args):
    _REG_EPOCH[0] += 1


torch.nn.modules.module.register_module_parameter_registration_hook(_registration_hook)
torch.nn.modules.module.register_module_module_registration_hook(_registration_hook)

_CHUNK_DTYPE = np.dtype([("ptr", "u8"), ("count", "u4"), ("first", "u4")])      # skg_param_chunk


class ParamWatch:
    """Notices any change of the live parameters behind a PackedWeights: replaced Parameter objects (identity, checked
    when the registration epoch moved), re-pointed storage (data_ptr walk) and changed VALUES -- by an on-device
    checksum of the parameter bytes (skg_param_checksum), so that writes that bypass autograd's version counters
    (`p.data.mul_(2)`) are seen too.  Parameters that are not contiguous fp32 on the engine's device fall back to
    the version counters."""

    CHUNK_WORDS = 1 << 13

    def __init__(self, plist, device):
        self.plist = plist
        self.epoch = _REG_EPOCH[0]
        self.ids = tuple(map(id, plist))
        self.ptrs = tuple(map(torch.Tensor.data_ptr, plist))
        self.versions = sum(p._version for p in plist)
        self.device = device
        self.table = None
        self.sum = None
        if all(p.device == device and p.dtype == torch.float32 and p.is_contiguous() and p.data_ptr() % 16 == 0
               for p in plist):
            rows = []
            first = 0
            for p in plist:
                n, base = p.numel(), p.data_ptr()
                for o in range(0, n, self.CHUNK_WORDS):
                    rows.append((base + 4 * o, min(self.CHUNK_WORDS, n - o), (first + o) & 0xffffffff))
                first += n
            arr = np.array(rows, dtype=_CHUNK_DTYPE)
            self.n_chunks = len(rows)
            self.table = torch.from_numpy(arr.view(np.uint8).copy()).to(device)
            self._out = torch.zeros(_capi.CHECKSUM_PARTIALS, dtype=torch.int64, device=device)

    def enqueue(self, out_ptr):
        """Checksum of the live parameters onto the current stream: CHECKSUM_PARTIALS int64 partial sums at device
        address out_ptr (their sum modulo 2^64 is the checksum, see fold())."""
        _capi.check(_capi.lib().skg_param_checksum(self.table.data_ptr(), self.n_chunks, out_ptr, _stream()),
                    "skg_param_checksum")

    @staticmethod
    def fold(partials):
        """numpy int64 / uint64 partial sums -> the checksum (Python int, modulo 2^64)."""
        return int(partials.view(np.uint64).sum(dtype=np.uint64))

    def checksum_sync(self):
        self.enqueue(self._out.data_ptr())
        return self.fold(self._out.cpu().numpy())


class PackedWeights:
    """Device copies of the head's parameters in the layouts the kernels want (stacked MBF branches, padded K)."""

    def __init__(self, graph_head, predictor, suppressor, device):
        gh = graph_head
        f32 = dict(device=device, dtype=torch.float32)

        def w(p):
            return p.detach().to(**f32).contiguous()

        def pad_k(wt, mult=4, to=None):
            k = wt.shape[1]
            kp = to if to is not None else (k + mult - 1) // mult * mult
            if kp == k:
                return wt.contiguous()
            out = torch.zeros(wt.shape[0], kp, **f32)
            out[:, :k] = wt
            return out

        def mbf(m):
            w1 = torch.cat([w(l.weight) for l in m.fc_1]); b1 = torch.cat([w(l.bias) for l in m.fc_1])
            w2 = torch.cat([w(l.weight) for l in m.fc_2]); b2 = torch.cat([w(l.bias) for l in m.fc_2])
            w3 = torch.cat([w(l.weight) for l in m.fc_3], dim=1).contiguous()
            b3 = torch.stack([w(l.bias) for l in m.fc_3]).sum(dim=0)
            return dict(w1=w1.contiguous(), b1=b1, w2=w2.contiguous(), b2=b2, w3=w3, b3=b3)

        self.device = device
        self.splits = SplitWeights()
        self.K = gh.num_cls
        self.bh1_w = pad_k(w(gh.box_head[1].weight)); self.bh1_b = w(gh.box_head[1].bias)
        self.bh1_k = gh.box_head[1].weight.shape[1]
        self.bh3_w = w(gh.box_head[3].weight); self.bh3_b = w(gh.box_head[3].bias)
        self.sp0_w = pad_k(w(gh.spatial_head[0].weight), to=_capi.SPATIAL_LD); self.sp0_b = w(gh.spatial_head[0].bias)
        self.sp2_w = w(gh.spatial_head[2].weight); self.sp2_b = w(gh.spatial_head[2].bias)
        self.sp4_w = w(gh.spatial_head[4].weight); self.sp4_b = w(gh.spatial_head[4].bias)
        self.fh_w = pad_k(w(gh.fc_head[0].weight), to=1088); self.fh_b = w(gh.fc_head[0].bias)
        self.ft_w = pad_k(w(gh.fc_tail[0].weight), to=1088); self.ft_b = w(gh.fc_tail[0].bias)
        self.att = mbf(gh.attention_head)
        self.att_g = mbf(gh.attention_head_g)
        self.os = mbf(gh.obj_to_sub)
        self.so = mbf(gh.sub_to_obj)
        self.adj_w = w(gh.adjacency.weight).reshape(-1).contiguous()
        self.adj_b = float(gh.adjacency.bias.detach().float().cpu().item())
        self.nh_g = w(gh.norm_h.weight); self.nh_b = w(gh.norm_h.bias)
        self.no_g = w(gh.norm_o.weight); self.no_b = w(gh.norm_o.bias)
        self.fused_cls = isinstance(predictor, torch.nn.Linear) and isinstance(suppressor, torch.nn.Linear) \
            and predictor.in_features == 2048 and suppressor.in_features == 2048 and suppressor.out_features == 1 \
            and predictor.out_features == self.K
        if self.fused_cls:
            self.cls_w = torch.cat([w(predictor.weight), w(suppressor.weight)]).contiguous()
            pb = predictor.bias if predictor.bias is not None else torch.zeros(self.K)
            sb = suppressor.bias if suppressor.bias is not None else torch.zeros(1)
            self.cls_b = torch.cat([w(pb), w(sb)]).contiguous()


class VerbTable:
    """CSR form of object_class_to_target_class (HEAD:629, 747-760); verbs ascending and unique per class."""

    def __init__(self, o2v, K, device):
        rows = [sorted(set(int(v) for v in r)) for r in o2v]
        for r in rows:
            if r and (r[0] < 0 or r[-1] >= K):
                raise IndexError("index %d is out of bounds for dimension 1 with size %d" % (r[-1], K))
        off = np.zeros(len(rows) + 1, np.int32)
        off[1:] = np.cumsum([len(r) for r in rows])
        flat = np.asarray([v for r in rows for v in r] or [0], np.int32)
        self.num_obj = len(rows)
        self.nverbs = torch.from_numpy(np.diff(off).astype(np.int32)).to(device)
        self.off = torch.from_numpy(off).to(device)
        self.flat = torch.from_numpy(flat).to(device)


import threading as _threading

_TLS = _threading.local()     # .splits: the SplitWeights in force on this thread (inside a fp16x2 engine call), else None


def _active_splits():
    return getattr(_TLS, "splits", None)


class SplitWeights:
    """fp16x2 twins of nn.Linear weights for skg_gemm_desc.w_split, made on first use and kept with the packed
    weights they mirror (keyed by the weight view: address, N, K, leading dimension)."""

    def __init__(self):
        self.twins = {}

    def get(self, W, W_off, N, K, ldw):
        """-> (twin bytes tensor, w_scale) for the view W.flatten()[W_off:] as [N, K] with leading dimension ldw."""
        key = (W.data_ptr() + 4 * W_off, N, K, ldw)
        t = self.twins.get(key)
        if t is None:
            lib = _capi.lib()
            view = W.reshape(-1)[W_off:].as_strided((N, K), (ldw, 1))
            amax = float(view.abs().max())                       # one-time host sync per weight
            e = 13 - int(np.floor(np.log2(amax))) if np.isfinite(amax) and amax > 0 else 0
            e = max(min(e, 100), -100)
            twin = torch.empty(lib.skg_split_weights_bytes(N, K), dtype=torch.uint8, device=W.device)
            _capi.check(lib.skg_split_weights_f16x2(key[0], N, K, ldw, float(2.0 ** e), twin.data_ptr(), _stream()),
                        "skg_split_weights_f16x2")
            # The twin is made on whichever stream first needs it, other streams (n_streams > 1) may read it later with
            # no event in between: finish it now.  Once per weight; the amax read above has synchronised already.
            torch.cuda.current_stream().synchronize()
            t = (twin, float(2.0 ** -e), W)                       # holding W keeps its address from being reused
            self.twins[key] = t
        return t[0], t[1]

    def __enter__(self):
        self._prev = _active_splits()
        _TLS.splits = self
        return self

    def __exit__(self, *a):
        _TLS.splits = self._prev
        return False


def gemm_desc(A, W, bias, C_out, M, N, K, epilogue, lda=None, ldw=None, ldc=None, a_rows=None, out_rows=None, P=None,
              p_idx=None, ldp=0, Q=None, q_idx=None, ldq=0, mbias=None, C_raw=None, ldc_raw=0, dot_w=None,
              dot_partial=None, res=None, ldres=0, A_off=0, W_off=0, C_off=0, d=None, split_k=0, split_ws=None,
              w_split=None, w_scale=0.0):
    """Fills a skg_gemm_desc.  *_off are element offsets into A / W / C (column sub-views)."""
    d = _capi.GemmDesc() if d is None else d
    d.A = A.data_ptr() + 4 * A_off; d.lda = lda if lda is not None else A.stride(0)
    d.W = W.data_ptr() + 4 * W_off; d.ldw = ldw if ldw is not None else W.stride(0)
    d.bias = _ptr(bias)
    d.C = (C_out.data_ptr() + 4 * C_off) if C_out is not None else 0
    d.ldc = ldc if ldc is not None else (C_out.stride(0) if C_out is not None else 0)
    d.M, d.N, d.K, d.epilogue = M, N, K, epilogue
    d.a_rows = _ptr(a_rows); d.out_rows = _ptr(out_rows)
    d.P = _ptr(P); d.p_idx = _ptr(p_idx); d.ldp = ldp
    d.Q = _ptr(Q); d.q_idx = _ptr(q_idx); d.ldq = ldq
    d.mbias = _ptr(mbias); d.C_raw = _ptr(C_raw); d.ldc_raw = ldc_raw
    d.dot_w = _ptr(dot_w); d.dot_partial = _ptr(dot_partial)
    d.res = _ptr(res); d.ldres = ldres
    d.split_k = split_k; d.split_ws = _ptr(split_ws)
    splits = _active_splits()
    if w_split is None and splits is not None and K % 16 == 0 and M > 0:
        w_split, w_scale = splits.get(W, W_off, N, K, d.ldw)
    d.w_split = _ptr(w_split); d.w_scale = w_scale
    return d


# ---- row exponents of the split-operand (fp16x2) GEMMs' A operands (power-of-two row scale, include/skghoi.h): one small
# pass per operand in front of its GEMM, on the same stream.  The buffer comes from the caching allocator (stream-ordered:
# it may be handed out again only to later work of this stream); inside a graph capture the buffers are kept by the plan,
# because the capture's two branches share one pool.
AMAX_CAPTURE_KEEP = None        # set to a list by small.py while a plan is captured
# False (default): the GEMM workgroups estimate each row's scale from the first 64 k they walk (no extra pass; a low
# estimate costs at worst an exact re-run of a tile).  True: one skg_row_exponents_f32 pass per operand gives the row's
# true maximum (+10 % step time at batch 256) -- for operands whose rows start with long runs of zeros.
EXACT_ROW_SCALE = False


def enqueue_row_exponents(d, device):
    """For a descriptor that takes the split-operand loop (w_split set): enqueues the exponent pass of its A operand on
    the current stream and points d.a_exp at the result.  Returns the buffer (keep it until the GEMM is enqueued)."""
    if not d.w_split or d.M <= 0 or not EXACT_ROW_SCALE:
        d.a_exp = 0
        return None
    t = torch.empty(d.M, device=device, dtype=torch.int32)
    if AMAX_CAPTURE_KEEP is not None:
        AMAX_CAPTURE_KEEP.append(t)
    _capi.check(_capi.lib().skg_row_exponents_f32(d.A, d.lda, d.a_rows, d.M, d.K, t.data_ptr(), _stream()),
                "skg_row_exponents_f32")
    d.a_exp = t.data_ptr()
    return t


_M64_TARGET = int(os.environ.get("SKG_SPLIT_M64_TARGET") or 512)                  # (developer sweeps)
_LONGK_SMALL_TILES = os.environ.get("SKG_LONGK_SMALL_TILES", "1") != "0"        # (developer A/B switch)


def pick_split_k(M, N, K, target_blocks=1024):
    """Split-K factor for a plain layer whose M x N tile grid would leave most of the 256 CUs idle while each
    workgroup walks a long K (box_head: K = 12544).  Slices keep >= 16 k-tiles (256 k) each."""
    blocks = ((M + 127) // 128) * ((N + 127) // 128)
    if M <= 64:
        # one tile high: the launcher takes 64 x 64 tiles (skg_gemm_tile_scale); slices of >= 4 k-tiles
        blocks = (N + 63) // 64
        return int(max(1, min(-(-_M64_TARGET // blocks), K // 64, 64))) if K >= 512 else 1
    if blocks >= target_blocks or K < 2048:
        return 1
    if M <= 512 and K >= 4096 and _LONGK_SMALL_TILES:
        # a few images' box rows against K = 12544 (box_head layer 1 at 2 ... 8 images): 128-row tiles spend up to half their
        # MFMAs on padding rows and ~50 slices of 16 k-steps each cost a launch of 784 workgroups 86 us at M = 160.  Fewer,
        # longer slices keep the launch under the launcher's bounds for 64 x 64 tiles on the latency loop (fewer than 200 tiles
        # of 128 x 128, slices included: csrc/skg_gemm.hip g_route_tiles / g_small_tiles) with ~3 workgroups per CU.
        blocks64 = ((M + 63) // 64) * ((N + 63) // 64)
        return int(max(1, min(199 // blocks, -(-768 // blocks64), K // 256, 64)))
    return int(max(1, min(-(-target_blocks // blocks), K // 256, 64)))


def dot_partials(M, N, K, lda, ldw):
    """Slab count of the dot_partial output of an EPI_RELU_DOT launch (depends on the tile shape the launcher picks)."""
    d = _capi.GemmDesc()
    d.M, d.N, d.K, d.lda, d.ldw, d.epilogue = M, N, K, lda, ldw, _capi.EPI_RELU_DOT
    d.w_split = 1 if (_active_splits() is not None and K % 16 == 0) else 0       # only null / non-null matters here
    d.w_scale = 1.0
    n = _capi.lib().skg_gemm_dot_partials(C.byref(d))
    if n <= 0:
        raise _capi.SkgError("skg_gemm_dot_partials -> %d" % n)
    return n


def gemm(A, W, bias, C_out, M, N, K, epilogue, **kw):
    """One skg_gemm_f32 launch (see gemm_desc for the keywords)."""
    d = gemm_desc(A, W, bias, C_out, M, N, K, epilogue, **kw)
    keep_exp = enqueue_row_exponents(d, A.device) if d.w_split else None
    timed = GEMM_TIMER is not None and (GEMM_TIMER_EPI is None or epilogue in GEMM_TIMER_EPI)
    if timed:
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
    _capi.check(_capi.lib().skg_gemm_f32(C.byref(d), _stream()), "skg_gemm_f32[%dx%dx%d epi %d]" % (M, N, K, epilogue))
    if timed:
        e1.record()
        GEMM_TIMER.append((e0, e1, M, N, K, epilogue))


# Workgroups a small (64 x 64 tile) grouped launch aims for; split-K supplies them.  Two costs pull against each other at a
# few images (measured on MI355X): every workgroup costs ~10 ns of dispatch, and every k-tile a workgroup walks costs ~1 us
# (one HBM / L2 round trip behind a one-tile-deep prefetch).  ~512 workgroups of >= 4 k-tiles sat near the minimum in round 2;
# re-swept at the end of round 4 (64-k steps, two stages in flight): 320 / 384 / 448 / 512 -> single 20 x 20 image 0.483-0.488 /
# 0.474-0.482 / 0.502-0.504 / 0.498-0.499 ms, batches of 2 and 4 level, a stream of mixed shapes level (0.46-0.47 either way).
SMALL_GROUP_BLOCKS = int(os.environ.get("SKG_SMALL_GROUP_BLOCKS") or 384)     # (the env override: developer sweeps)


def gemm_group(specs):
    """Independent small GEMMs in one launch: specs = [(args, kwargs), ...] as for gemm().

    When the launcher takes 64 x 64 tiles for the group (a few images, skg_gemm_group_tile) the members whose grid is
    still tiny -- node-row GEMMs: M = a few dozen rows, N = K = 1024 is 16 workgroups walking 64 k-tiles each -- get
    split-K slices (>= 4 k-tiles each) and a second launch reduces them in slice order."""
    n = len(specs)
    arr = (_capi.GemmDesc * n)()
    flops = 0.0
    keep_exp = []
    for i, (a, kw) in enumerate(specs):
        gemm_desc(*a, d=arr[i], **kw)
        if arr[i].w_split:
            keep_exp.append(enqueue_row_exponents(arr[i], a[0].device))
        flops += 2.0 * a[4] * a[5] * a[6]
    lib = _capi.lib()
    if lib.skg_gemm_group_tile(arr, n) == 1:
        tiles = [((d.M + 63) // 64) * ((d.N + 63) // 64) if d.M else 0 for d in arr]
        total = sum(tiles)
        keep = []
        for d, t in zip(arr, tiles):
            if t == 0 or d.split_k > 1 or d.epilogue not in (_capi.EPI_BIAS, _capi.EPI_BIAS_RELU, _capi.EPI_BIAS_RES_RELU):
                continue
            sk = min(-(-SMALL_GROUP_BLOCKS // max(total, 1)), d.K // 64, 64)
            if sk > 1:
                ws = torch.empty(sk * d.M * d.N, device=specs[0][0][0].device, dtype=torch.float32)
                keep.append(ws)
                d.split_k = sk; d.split_ws = ws.data_ptr()
    timed = GEMM_TIMER is not None and (GEMM_TIMER_EPI is None or 5 in GEMM_TIMER_EPI)
    if timed:
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
    _capi.check(lib.skg_gemm_group_f32(arr, n, _stream()), "skg_gemm_group_f32[%d]" % n)
    if timed:
        e1.record()
        GEMM_TIMER.append((e0, e1, int(flops // 2), 1, 1, 5))       # epilogue id 5 = grouped launch


class _NullCtx:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


class _DrawAborted(Exception):
    pass


class _TableDrawer:
    """Draws the per-image TransH tables of every chunk, in order (the global CPU RNG is consumed exactly as the
    reference does, skghoi_amd/transh.py), into a ring of pinned staging buffers.

    Chunk i uses slot i % len(slots).  A slot may be redrawn only after (a) the consumer has ENQUEUED the H2D copies
    of the chunk that used it last -- `release(i)`, which also records the HIP event behind those copies -- and
    (b) that event has completed.  Large batches draw on a helper thread that runs ahead of the consumer by at most
    len(slots) chunks; a few images (<= INLINE_IMAGES) are drawn lazily inside get(), on the caller's thread."""

    INLINE_IMAGES = 8

    def __init__(self, K, sizes, need_relations, slots):
        import threading
        self.K, self.sizes, self.need = K, sizes, need_relations
        self.slots = slots
        self.out = [None] * len(sizes)
        self.err = None
        self.abort = False
        self.ready = [threading.Event() for _ in sizes]
        self.released = [threading.Event() for _ in sizes]
        self.events = [None] * len(sizes)          # HIP event behind the H2D copies of chunk i (set by release)
        self.drawn = 0                             # inline mode: chunks drawn so far
        if sum(sizes) <= self.INLINE_IMAGES:       # a few images: the draw (20 us each) is cheaper than starting a thread
            self.thread = None
        else:
            self.thread = threading.Thread(target=self._run, daemon=True)
            self.thread.start()

    def _draw(self, i):
        ns = len(self.slots)
        slot = self.slots[i % ns]
        if i >= ns:                                # the chunk that used this staging buffer last, in this forward
            self.released[i - ns].wait()
            if self.abort:
                raise _DrawAborted()
            ev = self.events[i - ns]
        else:                                      # ... or in an earlier forward
            ev = slot["event"]
        if ev is not None:
            ev.synchronize()
        slot["event"] = None
        self.out[i] = transh.draw_batch(self.K, self.sizes[i], need_relations=self.need, out=slot["bufs"])

    def _run(self):
        try:
            for i in range(len(self.sizes)):
                self._draw(i)
                self.ready[i].set()
        except BaseException as e:                 # surface the failure in the caller's thread
            self.err = e
            for ev in self.ready:
                ev.set()

    def get(self, i):
        if self.thread is None:
            while self.drawn <= i:                 # lazily and in order: the RNG stream is the reference's
                self._draw(self.drawn)
                self.drawn += 1
            return self.out[i]
        self.ready[i].wait()
        if self.err is not None:
            raise self.err
        return self.out[i]

    def release(self, i, event):
        """The consumer has enqueued every H2D copy out of chunk i's staging buffers; `event` completes after them."""
        self.events[i] = event
        self.slots[i % len(self.slots)]["event"] = event
        self.released[i].set()

    def join(self):
        self.abort = True
        for ev in self.released:                   # a consumer that failed midway must not leave the thread waiting
            ev.set()
        if self.thread is not None:
            self.thread.join()
        # (inline mode draws on demand: chunks a failed consumer never asked for are not drawn, exactly as a reference
        # forward that raised midway would have left the RNG)
        if self.err is not None and not isinstance(self.err, _DrawAborted):
            raise self.err


class Preprocessed:
    """Packed output of InteractionHead.preprocess for a batch."""
    pass


class HeadEngine:
    GROUP_FC2_BELOW = 32768          # grid rows (40 images of 20 x 20) below which the three fc_2 GEMMs share a launch

    def __init__(self, graph_head, predictor, suppressor, human_idx, num_classes, box_nms_thresh, box_score_thresh,
                 max_human, max_object, faithful_skip_offset=True):
        self.gh = graph_head
        self.predictor = predictor
        self.suppressor = suppressor
        self.human_idx = int(human_idx)
        self.K = int(num_classes)
        self.box_nms_thresh = float(box_nms_thresh)
        self.box_score_thresh = float(box_score_thresh)
        self.max_human = int(max_human)
        self.max_object = int(max_object)
        self.faithful_skip_offset = faithful_skip_offset
        self.chunk_images = 128     # active images per graph chunk (RNG/GPU overlap + cache-sized intermediates)
        self.debug = False          # keep per-chunk intermediates (spatial46, h_node, node, adjacency) in graph()
        self._slots = None
        self._streams = None
        self.n_streams = 2          # chunks alternate over this many side streams: one chunk's kernel tails and small
                                    # launches are filled by the other's GEMMs (+2.6 % at 256 images, measured); 1: off
        self.precision = "fp32"     # "fp32": exact fp32 MFMA; "fp16x2" (opt-in): fp16 matrix pipe from 2-way operand splits
        self._pw = None
        self._vt = None
        self._det_off_cache = {}
        self._gt_const = None
        self._cnt_host = None
        self._cnt_event = None
        self._cnt_host_dev = None
        self.plan_epoch = 0
        self._ck_events = None
        self.small_two_branches = os.environ.get("SKG_SMALL_ONE_BRANCH") != "1"    # captured plans: spatial chain beside the box_head chain
        # OPT-IN (SKG_G1_ON_SIDE=1), single-image batches only: the global branch's fc_1 as a launch of its own that opens the
        # captured plan's side chain (the eager path then issues the same two launches: the paths stay bit-identical).  It
        # takes a cross-queue hand-over out of the replayed graph (B = 1: 0.465 against 0.485 ms) -- but both full GPU test
        # runs made with it ended in a segmentation fault inside hipGraphLaunch, at the first replay of a plan captured
        # after another plan's graph had been destroyed (gpurun_out/r5y, r5z2), where the grouped form
        # has run clean in every full run of rounds 3-5.  The grouped launch stays the default.
        self.g1_on_side_branch = os.environ.get("SKG_G1_ON_SIDE", "0") == "1"
        self.small_batch_max = 8    # eval batches of up to this many images replay a captured hipGraph (skghoi_amd/small.py); 0: off
        self.small_batch_buckets = True   # single images share one plan per BUCKET of (humans, nodes) instead of one per shape
        self.small_capture_after = 2      # an exact-shape plan is captured at the shape's 2nd sighting (eager until then; 1: at once)
        self._small = None
        self.last = None          # intermediates of the last graph pass (parity tests read them)

    # ------------------------------------------------------------------------------------------ caches
    def _enumerate(self):
        return list(self.gh.parameters()) + list(self.predictor.parameters()) + list(self.suppressor.parameters())

    def weights(self, device, wsum=None, walk=True):
        """Packed weights, re-packed when any live parameter was modified in place (by any route), replaced or moved.

        wsum: the parameter checksum this forward already brought back with the preprocess counts (None: computed
        here, which costs a synchronisation).  walk=False skips the data_ptr walk (small-batch graph path)."""
        pw = self._pw
        if pw is not None and pw.device == device:
            w = pw.watch
            stale = False
            if w.epoch != _REG_EPOCH[0]:                    # something was registered somewhere: compare identities
                if tuple(map(id, self._enumerate())) != w.ids:
                    stale = True
                else:
                    w.epoch = _REG_EPOCH[0]
            if not stale and walk and tuple(map(torch.Tensor.data_ptr, w.plist)) != w.ptrs:
                stale = True
            if not stale:
                if w.table is not None:
                    stale = (w.checksum_sync() if wsum is None else wsum) != w.sum
                else:
                    stale = sum(p._version for p in w.plist) != w.versions
            if not stale:
                return pw
        plist = self._enumerate()
        pw = PackedWeights(self.gh, self.predictor, self.suppressor, device)
        pw.watch = ParamWatch(plist, device)
        if pw.watch.table is not None:
            pw.watch.sum = pw.watch.checksum_sync()
        self._pw = pw
        self.plan_epoch += 1                          # cached launch plans hold pointers into the old copies
        return pw

    def _table_slots(self, cap, need_relations):
        """Persistent pinned staging buffers for the TransH tables (a ring of 4): allocating pinned memory per call
        costs a hipHostMalloc (tens of ms) whenever the caching host allocator has no free block."""
        ok = self._slots is not None and self._slots[0]["cap"] >= cap and \
            (not need_relations or self._slots[0]["bufs"][1] is not None)
        if not ok:
            self._slots = []
            for _ in range(4):
                ent = torch.empty(cap, _capi.TRANSH_ENT, _capi.TRANSH_DIM, pin_memory=_PIN_TABLES)
                rel = torch.empty(cap, self.K, _capi.TRANSH_DIM, pin_memory=_PIN_TABLES) if need_relations else None
                nrm = torch.empty(cap, self.K, _capi.TRANSH_DIM, pin_memory=_PIN_TABLES) if need_relations else None
                self._slots.append(dict(cap=cap, bufs=(ent, rel, nrm), event=None))
        return self._slots

    def verbs(self, device):
        if self._vt is None or self._vt.off.device != device:
            self._vt = VerbTable(self.gh.object_class_to_target_class, self.gh.num_cls, device)
        return self._vt

    # ------------------------------------------------------------------------------------------ preprocess
    def _det_offsets(self, sizes, dev):
        """int32 prefix of the per-image detection counts on the device, cached by the count tuple (one H2D copy per
        distinct tuple instead of one per forward)."""
        key = (tuple(sizes), dev)
        t = self._det_off_cache.get(key)
        if t is None:
            if len(self._det_off_cache) >= 4096:
                self._det_off_cache.clear()
            h = np.zeros(len(sizes) + 1, np.int32); h[1:] = np.cumsum(sizes)
            t = torch.from_numpy(h).to(dev)
            self._det_off_cache[key] = t
        return t

    def _read_counts(self, countx, defer=False):
        """The forward's one device -> host read.  Small batches poll a HIP event behind an asynchronous copy into pinned
        memory: a blocking copy parks the thread and the wake-up costs tens of microseconds, which at one image per
        forward is a tenth of the whole latency.  Returns a numpy int32 array (a private copy)."""
        n = countx.numel()
        if n > 4 * 16 + 2 * _capi.CHECKSUM_PARTIALS or (_NO_SPIN and not defer):   # larger batches: the plain blocking copy
            return ("blocking", countx, None) if defer else countx.cpu().numpy()
        if self._cnt_host is None or self._cnt_host.numel() < n or self._cnt_host_dev != countx.device:
            self._cnt_host = torch.empty(max(n, 4 * 16 + 2 * _capi.CHECKSUM_PARTIALS), dtype=torch.int32, pin_memory=True)
            self._cnt_event = torch.cuda.Event()
            self._cnt_host_dev = countx.device
        if defer:                          # read back later: a staging buffer and event of its own (other forwards of
            host = torch.empty(n, dtype=torch.int32, pin_memory=True)       # this engine may run in between)
            host.copy_(countx, non_blocking=True)
            ev = torch.cuda.Event(); ev.record(current_stream_of(None))
            return ("pinned", host, ev)
        host = self._cnt_host[:n]
        host.copy_(countx, non_blocking=True)
        ev = self._cnt_event
        ev.record(current_stream_of(None))
        return self._read_counts_end(("pinned", host, ev))

    def _read_counts_end(self, pending):
        kind, host, ev = pending
        if kind == "blocking":
            return host.cpu().numpy()
        if threading.current_thread() is threading.main_thread():
            while not ev.query():
                pass
        else:
            ev.synchronize()               # a helper thread must not spin: the poll holds the GIL, the wait releases it
        return host.numpy().copy()

    def pre_launch(self, detections, targets, append_gt, training, check_weights=False, defer=False):
        """First half of preprocess: score filter + class-wise NMS + top-k on the device and the ONE host
        synchronisation of a forward (per-image counts, plus the parameter checksum riding on the same copy).
        Returns a Preprocessed holding the raw inputs, the selection and the counts; pre_pack() gathers."""
        lib = _capi.lib()
        dev = detections[0]["boxes"].device if detections else torch.device("cuda")
        if dev.type != "cuda":
            raise _capi.SkgError("the interaction head runs on a HIP device only (got %s)" % dev)
        vt = self.verbs(dev)
        B = len(detections)
        bl, sl, ll, sizes = [], [], [], []
        if append_gt:
            # HEAD:107-116: ground-truth human and object boxes in front of the detections, score 1, labels human_idx /
            # object.  The pieces of all images go into ONE concatenation per array (constants are slices of two cached
            # tensors), instead of three concatenations and two fills per image.
            ngs = [int(targets[b]["boxes_h"].shape[0]) for b in range(B)]
            need = 2 * max(ngs, default=0)
            if self._gt_const is None or self._gt_const[0].numel() < need or self._gt_const[0].device != dev:
                n = max(need, 64)
                self._gt_const = (torch.ones(n, device=dev), torch.full((n,), self.human_idx, dtype=torch.int64, device=dev))
            ones, hum = self._gt_const
            for b, det in enumerate(detections):
                t, ng = targets[b], ngs[b]
                bl += [t["boxes_h"].reshape(-1, 4), t["boxes_o"].reshape(-1, 4), det["boxes"].reshape(-1, 4)]
                sl += [ones[:2 * ng], det["scores"].reshape(-1)]
                ll += [hum[:ng], t["object"].reshape(-1), det["labels"].reshape(-1)]
                sizes.append(2 * ng + int(det["boxes"].shape[0]))
        else:
            for b, det in enumerate(detections):
                boxes, labels, scores = det["boxes"], det["labels"], det["scores"]
                bl.append(boxes.reshape(-1, 4)); sl.append(scores.reshape(-1)); ll.append(labels.reshape(-1))
                sizes.append(int(boxes.shape[0]))
        if max(sizes, default=0) > _capi.MAX_DET_PER_IMAGE:
            raise _capi.SkgError("more than %d detections in one image" % _capi.MAX_DET_PER_IMAGE)
        if B == 1 and not append_gt:                 # the reference's evaluation mode: nothing to concatenate
            boxes, scores, labels = bl[0], sl[0], ll[0]
            if boxes.dtype != torch.float32 or not boxes.is_contiguous():
                boxes = boxes.float().contiguous()
            if scores.dtype != torch.float32 or not scores.is_contiguous():
                scores = scores.float().contiguous()
            if labels.dtype != torch.int64 or not labels.is_contiguous():
                labels = labels.long().contiguous()
        else:
            boxes = torch.cat(bl).float().contiguous() if B else torch.zeros(0, 4, device=dev)
            scores = torch.cat(sl).float().contiguous() if B else torch.zeros(0, device=dev)
            labels = torch.cat(ll).long().contiguous() if B else torch.zeros(0, dtype=torch.int64, device=dev)
        det_off = self._det_offsets(sizes, dev)
        ld = self.max_human + self.max_object
        index = torch.empty(B, max(ld, 1), dtype=torch.int32, device=dev)
        # counts [B,4] | parameter checksum partials (u64 x CHECKSUM_PARTIALS): one buffer, one D2H copy
        countx = torch.empty(4 * B + 2 * _capi.CHECKSUM_PARTIALS, dtype=torch.int32, device=dev)
        watch = None
        side_done = None
        if check_weights and self._pw is not None and self._pw.device == dev and self._pw.watch.table is not None:
            watch = self._pw.watch
            if B <= self.small_batch_max:
                # a few images: the forward is a chain of short kernels, and the 18 us checksum (118 MB read) in front of
                # the selection kernel would be on its critical path.  It runs BESIDE the selection kernel on the process's
                # side stream, ordered behind everything this stream holds now (an optimizer step enqueued a moment ago is
                # seen); the one D2H copy below waits for both.
                ev = self._ck_events
                if ev is None or ev[2] != dev:
                    ev = self._ck_events = (torch.cuda.Event(), torch.cuda.Event(), dev)
                side = shared_side_stream(dev, 0, slot=0)
                ev[0].record(current_stream_of(dev))
                side.wait_event(ev[0])
                with on_stream(side):
                    watch.enqueue(countx.data_ptr() + 16 * B)
                    ev[1].record(side)
                side_done = ev[1]
            else:
                watch.enqueue(countx.data_ptr() + 16 * B)                  # rides on the one D2H copy below
        prior_pow = 1.0 if training else 2.8                                    # HEAD:742
        if boxes.numel() == 0:
            boxes = torch.zeros(1, 4, device=dev); scores = torch.zeros(1, device=dev)
            labels = torch.zeros(1, dtype=torch.int64, device=dev)
        _capi.check(lib.skg_preprocess_f32(boxes.data_ptr(), scores.data_ptr(), labels.data_ptr(), det_off.data_ptr(),
                                           B, self.human_idx, self.box_score_thresh, self.box_nms_thresh,
                                           self.max_human, self.max_object, vt.nverbs.data_ptr(), vt.num_obj,
                                           prior_pow, index.data_ptr(), countx.data_ptr(), _stream()),
                    "skg_preprocess_f32")
        if side_done is not None:
            current_stream_of(dev).wait_event(side_done)
        if defer:
            # the caller goes on with other host work while the kernel runs and comes back with pre_launch_end()
            return dict(pending=self._read_counts(countx, defer=True), countx=countx, watch=watch, B=B, dev=dev,
                        index=index, raw=(boxes, scores, labels, det_off))
        return self.pre_launch_end(dict(pending=None, cntx=self._read_counts(countx), countx=countx, watch=watch, B=B,
                                        dev=dev, index=index, raw=(boxes, scores, labels, det_off)))

    def pre_launch_end(self, st):
        """Second half of pre_launch(defer=True): waits for the counts and builds the Preprocessed record."""
        B, dev, watch, index = st["B"], st["dev"], st["watch"], st["index"]
        boxes, scores, labels, det_off = st["raw"]
        cntx = st["cntx"] if st.get("pending") is None else self._read_counts_end(st["pending"])   # the synchronisation point
        cnt = cntx[:4 * B].reshape(B, 4)
        if B and cnt[:, 0].min() < 0:
            bad = int(np.argmax(cnt[:, 0] < 0))
            raise _capi.SkgError("image %d has %d detections; the preprocess kernel takes at most %d per image"
                                 % (bad, int(cnt[bad, 3]), _capi.MAX_DET_PER_IMAGE))
        pre = Preprocessed()
        pre.wsum = ParamWatch.fold(cntx[4 * B:].view(np.int64)) if watch is not None else None
        pre.device = dev
        pre.B = B
        pre.n_h = cnt[:, 0].astype(np.int64); pre.n = cnt[:, 1].astype(np.int64); pre.L = cnt[:, 2].astype(np.int64)
        pre.sizes = [int(v) for v in pre.n]
        pre.index = index
        pre.raw = (boxes, scores, labels, det_off)
        return pre

    def pre_pack(self, pre, out=None, sel_off=None):
        """Second half: gathers the selected detections, humans first (HEAD:144-149), into packed arrays -- fresh
        ones, or `out` = (boxes [sumN,4], scores [sumN], labels [sumN] int64) with `sel_off` the device prefix of n."""
        dev = pre.device
        total = int(sum(pre.sizes))
        if out is None:
            pre.boxes = torch.empty(max(total, 1), 4, device=dev)[:total]
            pre.scores = torch.empty(max(total, 1), device=dev)[:total]
            pre.labels = torch.empty(max(total, 1), dtype=torch.int64, device=dev)[:total]
        else:
            pre.boxes, pre.scores, pre.labels = out
        if sel_off is None:
            sel_off = self._det_offsets(pre.sizes, dev)
        boxes, scores, labels, det_off = pre.raw
        if total:
            _capi.check(_capi.lib().skg_pack_detections_f32(
                boxes.data_ptr(), scores.data_ptr(), labels.data_ptr(), det_off.data_ptr(), pre.index.data_ptr(),
                pre.index.stride(0), sel_off.data_ptr(), pre.B, pre.boxes.data_ptr(), pre.scores.data_ptr(),
                pre.labels.data_ptr(), _stream()), "skg_pack_detections_f32")
        return pre

    def preprocess(self, detections, targets, append_gt, training, check_weights=False):
        """HEAD:92-151 for the whole batch.  One D2H copy (per-image counts)."""
        return self.pre_pack(self.pre_launch(detections, targets, append_gt, training, check_weights))

    # ------------------------------------------------------------------------------------------ graph head
    def _split_ctx(self, pw):
        return pw.splits if self.precision in ("fp16x2", "bf16") else _NullCtx()

    def graph(self, feat3, image_shapes, pooled, pre, training=False, tables=None, want_scores=False):
        pw = self.weights(pre.device, wsum=getattr(pre, "wsum", None))
        with self._split_ctx(pw):
            return self._graph(feat3, image_shapes, pooled, pre, training, tables, want_scores, pw)

    def classify(self, pair_features, checked=False):
        """checked=True: the caller's graph() pass of this same forward has validated the packed weights already."""
        pw = self._pw if (checked and self._pw is not None) else self.weights(pair_features.device)
        with self._split_ctx(pw):
            return self._classify(pair_features, pw)

    def g1_on_side(self, n_images):
        """Whether fc_1 of the global branch is launched apart from box_head layer 2 (captured AND eager path alike)."""
        return self.g1_on_side_branch and int(n_images) == 1

    def _graph(self, feat3, image_shapes, pooled, pre, training, tables, want_scores, pw):
        """GraphHead.forward (HEAD:769-993) for the batch.  Returns a dict of packed device tensors + layout.

        The active images are processed in chunks of `self.chunk_images`: the TransH tables of chunk c are drawn on
        the host (the reference's RNG stream, ~27k mt19937 draws per image) while the GPU is still busy with chunk
        c-1, and the chunk-sized intermediates ([G,1024] panels) stay closer to the caches."""
        lib = _capi.lib()
        dev = pre.device
        lay = layout.build(pre.n_h, pre.n, pre.L, image_shapes, self.human_idx,
                           faithful_skip_offset=self.faithful_skip_offset)
        A = lay.n_active
        st = _stream()
        f32 = dict(device=dev, dtype=torch.float32)
        out = dict(layout=lay)
        if pooled.shape[0] != lay.sum_all:
            raise _capi.SkgError("box_roi_pool returned %d rows for %d boxes" % (pooled.shape[0], lay.sum_all))
        # ---- global features (HEAD:811) and box_head on every selected box (HEAD:812)
        Bf, Cf = feat3.shape[0], feat3.shape[1]
        feat3 = feat3.float().contiguous()
        gfeat = torch.empty(Bf, Cf, **f32)
        _capi.check(lib.skg_global_avgpool_f32(feat3.data_ptr(), Bf, Cf, feat3.shape[2] * feat3.shape[3],
                                               gfeat.data_ptr(), st), "skg_global_avgpool_f32")
        x0 = pooled.float().reshape(pooled.shape[0], -1)
        if x0.shape[1] != pw.bh1_k:
            raise RuntimeError("mat1 and mat2 shapes cannot be multiplied (%dx%d and %dx%d)" % (
                x0.shape[0], x0.shape[1], pw.bh1_k, 1024))
        if x0.shape[1] != pw.bh1_w.shape[1] or not x0.is_contiguous():
            xp = torch.zeros(x0.shape[0], pw.bh1_w.shape[1], **f32)
            xp[:, :x0.shape[1]] = x0
            x0 = xp
        NA = lay.sum_all
        enc1 = torch.empty(max(NA, 1), 1024, **f32)
        enc = torch.empty(max(NA, 1), 1024, **f32)
        G1 = torch.empty(Bf, 1024, **f32)                       # attention_head_g fc_1(global) per image (HEAD:971)
        if NA:
            sk = pick_split_k(NA, 1024, x0.shape[1])
            ws = torch.empty(sk, NA, 1024, **f32) if sk > 1 else None
            gemm(x0, pw.bh1_w, pw.bh1_b, enc1, NA, 1024, x0.shape[1], _capi.EPI_BIAS_RELU, split_k=sk, split_ws=ws)
        # box_head layer 2 and fc_1(global features) are independent: one launch (the same call the captured small-batch
        # plan makes, skghoi_amd/small.py -- the two paths stay bit-identical)
        bh3 = ((enc1, pw.bh3_w, pw.bh3_b, enc, NA, 1024, 1024, _capi.EPI_BIAS_RELU), {})
        g1 = ((gfeat, pw.att_g["w1"], pw.att_g["b1"], G1, Bf, 1024, Cf, _capi.EPI_BIAS), {})
        if self.g1_on_side(Bf):             # (two launches, as the captured plan issues them: see small.py)
            gemm_group([g1]); gemm_group([bh3])
        else:
            gemm_group([bh3, g1])
        out["enc"] = enc[:NA]
        out["gfeat"] = gfeat
        if A == 0:
            return out
        Mp = lay.sum_p
        meta_g = torch.from_numpy(lay.meta.view(np.int32).reshape(-1).copy()).to(dev, non_blocking=True)
        x_keep = torch.empty(max(Mp, 1), device=dev, dtype=torch.int64); y_keep = torch.empty_like(x_keep)
        PF = torch.empty(max(Mp, 1), 2048, **f32)
        sc = torch.empty(max(Mp, 1), self.K, **f32) if want_scores else None
        keep = {} if self.debug else None
        tabs = []
        step = max(int(self.chunk_images), 1)
        bounds = [(a0, min(A, a0 + step)) for a0 in range(0, A, step)]
        # The reference's RNG stream (~27k mt19937 draws per image) is produced by a helper thread, chunk by chunk and
        # in image order, while this thread enqueues GPU work; torch releases the GIL inside the draws.
        drawer = None
        if tables is None:
            drawer = _TableDrawer(self.K, [b - a for a, b in bounds], want_scores,
                                  self._table_slots(max(b - a for a, b in bounds), want_scores))
        try:
            # software pipeline over chunks: phase A (no dependence on the TransH tables) runs `lookahead` chunks ahead
            # of phase B, so the GPU always has queued work while the host draws the next chunk's tables
            lookahead = 2
            ctxs = {}

            # optional: chunks alternate between `n_streams` HIP streams so that the tail of one chunk's kernels is
            # filled by the other chunk's (same-chunk work stays ordered on its stream)
            main = torch.cuda.current_stream()
            ns = max(1, int(self.n_streams)) if len(bounds) > 1 else 1
            if ns > 1:
                if self._streams is None or len(self._streams) < ns:
                    self._streams = [shared_side_stream(dev, 0, slot=i) for i in range(ns)]    # (process-wide: see there)
                ev0 = torch.cuda.Event(); ev0.record(main)
                for st_ in self._streams[:ns]:
                    st_.wait_event(ev0)

            def on_stream(ci):
                return torch.cuda.stream(self._streams[ci % ns]) if ns > 1 else _NullCtx()

            def phase_a(ci):
                a0, a1 = bounds[ci]
                with on_stream(ci):
                    ctxs[ci] = self._chunk_phase_a(layout.chunk(lay, a0, a1), pw, pre, G1, x_keep, y_keep, PF)

            for ci in range(min(lookahead, len(bounds))):
                phase_a(ci)
            for ci, (a0, a1) in enumerate(bounds):
                if tables is None:
                    t = drawer.get(ci)
                else:
                    t = tuple(None if x is None else x[a0:a1] for x in tables)
                if self.debug:
                    tabs.append(tuple(None if x is None else x.clone() for x in t))
                with on_stream(ci):
                    self._chunk_phase_b(ctxs.pop(ci), t, pw, pre, enc, PF, sc, keep)
                    if tables is None:                  # staging slot is free again once its H2D copies are done
                        ev = torch.cuda.Event(); ev.record(current_stream_of(None))
                        drawer.release(ci, ev)
                if ci + lookahead < len(bounds):
                    phase_a(ci + lookahead)
            if ns > 1:
                for st_ in self._streams[:ns]:
                    ev = torch.cuda.Event(); ev.record(st_)
                    main.wait_event(ev)
        finally:
            if drawer is not None:
                drawer.join()
        out.update(x_keep=x_keep[:Mp], y_keep=y_keep[:Mp], meta=meta_g, pair_features=PF[:Mp])
        if tabs:
            out["tables"] = tuple(None if tabs[0][i] is None else torch.cat([t[i] for t in tabs]) for i in range(3))
        if want_scores:
            out["transh_scores"] = sc[:Mp]
        if keep is not None:
            for k, v in keep.items():
                out[k] = torch.cat(v)
        return out

    def _chunk_phase_a(self, ch, pw, pre, G1, x_keep, y_keep, PF, ibuf=None, offs=None, meta=None, caps=None):
        """Pairs, spatial encoding, spatial head and the global read-out branch of one chunk: everything that does not
        depend on the TransH tables.  ibuf / offs / meta: index arrays already on the device (captured-graph path:
        they are part of the launch plan, `meta` is its per-call record array)."""
        lib = _capi.lib()
        dev = pre.device
        st = _stream()
        f32 = dict(device=dev, dtype=torch.float32)
        i32 = dict(device=dev, dtype=torch.int32)
        A = ch.n_active
        Mh, Mn, Mg, Mp = ch.sum_h, ch.sum_n, ch.sum_g, ch.sum_p
        if ibuf is None:
            buf, offs = layout.pack_int_arrays(ch)
            ibuf = torch.from_numpy(buf).to(dev, non_blocking=True)

        def isl(name):
            o, l = offs[name]
            return ibuf[o:o + l]

        if meta is None:
            meta = isl("meta")
        xk = x_keep[ch.P0:]; yk = y_keep[ch.P0:]               # global arrays, written at the chunk's pair offset
        # ---- pairs + spatial encoding (no dependence on the TransH tables: enqueued before they are drawn)
        grid_h = torch.empty(Mg, **i32); grid_o = torch.empty(Mg, **i32); grid_pair = torch.empty(Mg, **i32)
        grid_img = torch.empty(Mg, **i32); pair_grid = torch.empty(max(Mp, 1), **i32)
        pair_h = torch.empty(max(Mp, 1), **i32); pair_o = torch.empty(max(Mp, 1), **i32)
        sp48 = torch.empty(Mg, _capi.SPATIAL_LD, **f32)
        # caps = (grid rows, pair rows) every image OWNS when the launch plan is sized for a bucket of shapes (small.py)
        gcap, pcap = caps if caps is not None else (0, 0)
        _capi.check(lib.skg_pairs_spatial_padded_f32(pre.boxes.data_ptr(), meta.data_ptr(), A, grid_h.data_ptr(),
                                                     grid_o.data_ptr(), grid_pair.data_ptr(), grid_img.data_ptr(),
                                                     pair_grid.data_ptr(), xk.data_ptr(), yk.data_ptr(),
                                                     pair_h.data_ptr(), pair_o.data_ptr(), sp48.data_ptr(), 1, gcap, pcap,
                                                     st), "skg_pairs_spatial_padded_f32")
        # ---- spatial_head (HEAD:662-669, 888)
        s1 = torch.empty(Mg, 128, **f32); s2 = torch.empty(Mg, 256, **f32); S = torch.empty(Mg, 1024, **f32)
        gemm(sp48, pw.sp0_w, pw.sp0_b, s1, Mg, 128, _capi.SPATIAL_LD, _capi.EPI_BIAS_RELU)
        gemm(s1, pw.sp2_w, pw.sp2_b, s2, Mg, 256, 128, _capi.EPI_BIAS_RELU)
        gemm(s2, pw.sp4_w, pw.sp4_b, S, Mg, 1024, 256, _capi.EPI_BIAS_RELU)
        del s1, s2
        cx = dict(ch=ch, isl=isl, meta=meta, grid_h=grid_h, grid_o=grid_o, grid_img=grid_img, grid_pair=grid_pair,
                  pair_grid=pair_grid, pair_h=pair_h, pair_o=pair_o, sp48=sp48, S=S, ibuf=ibuf)
        if G1 is not None:
            self._chunk_phase_a2(cx, pw, pre, G1, PF)
        return cx

    def _chunk_phase_a2(self, cx, pw, pre, G1, PF):
        """Global read-out branch of one chunk: attention_head_g needs only S and the image's global feature
        (HEAD:971-972).  (Split off phase A so that a captured plan can run it beside the message-passing chain.)"""
        ch = cx["ch"]
        Mg, Mp = ch.sum_g, ch.sum_p
        Tg = torch.empty(max(Mp, 1), 1024, device=pre.device, dtype=torch.float32)
        if Mp:
            gemm(cx["S"], pw.att_g["w2"], pw.att_g["b2"], Tg, Mg, 1024, 1024, _capi.EPI_MUL_RELU, P=G1,
                 p_idx=cx["grid_img"], ldp=1024, out_rows=cx["grid_pair"])
            gemm(Tg, pw.att_g["w3"], pw.att_g["b3"], PF, Mp, 1024, 1024, _capi.EPI_BIAS_RELU, ldc=2048,
                 C_off=ch.P0 * 2048 + 1024)
        cx["Tg"] = Tg

    def _chunk_phase_b(self, cx, tabs, pw, pre, enc, PF, sc, keep, need_S=None, need_Tg=None):
        """Message passing and pair read-out of one chunk (needs the chunk's TransH entity tables).  need_S / need_Tg:
        callables invoked right before the first use of the spatial features S resp. of the global branch's buffer
        (a captured plan produces them on a side stream and joins there)."""
        lib = _capi.lib()
        gh = self.gh
        dev = pre.device
        st = _stream()
        f32 = dict(device=dev, dtype=torch.float32)
        ch, isl, meta = cx["ch"], cx["isl"], cx["meta"]
        grid_h, grid_o, pair_grid, pair_h, pair_o = cx["grid_h"], cx["grid_o"], cx["pair_grid"], cx["pair_h"], cx["pair_o"]
        sp48, S = cx["sp48"], cx["S"]
        A = ch.n_active
        Mh, Mn, Mg, Mp = ch.sum_h, ch.sum_n, ch.sum_g, ch.sum_p
        ent, rel, nrm = tabs
        ent_d = ent.to(dev, non_blocking=True)
        F2 = torch.empty(Mg, 1024, **f32)
        if gh.num_iter > 0:
            # ---- fc_head / fc_tail on unique rows (HEAD:884-885)
            X = torch.empty(Mh + Mn, 1088, **f32)
            rows, eimg, erow = isl("enc_row_hn"), isl("img_hn"), isl("ent_row_hn")
            _capi.check(lib.skg_concat_entity_f32(enc.data_ptr(), 1024, rows.data_ptr(), ent_d.data_ptr(),
                                                  eimg.data_ptr(), erow.data_ptr(), Mh + Mn, X.data_ptr(), 1088, st),
                        "skg_concat_entity_f32")
            GH = torch.empty(Mh, 1024, **f32); GO = torch.empty(Mn, 1024, **f32)
            gemm_group([((X, pw.fh_w, pw.fh_b, GH, Mh, 1024, 1088, _capi.EPI_BIAS_RELU), {}),
                        ((X, pw.ft_w, pw.ft_b, GO, Mn, 1024, 1088, _capi.EPI_BIAS_RELU), dict(A_off=Mh * 1088))])
            # ---- attention_head fc_1, separable over [human | object] halves (HEAD:894-896)
            A1h = torch.empty(Mh, 1024, **f32); A1o = torch.empty(Mn, 1024, **f32)
            # message fc_1 on node rows (HEAD:514, 524)
            C1o = torch.empty(Mn, 1024, **f32); C1h = torch.empty(Mh, 1024, **f32)
            gemm_group([((GH, pw.att["w1"], None, A1h, Mh, 1024, 1024, _capi.EPI_BIAS), dict(ldw=2048)),
                        ((GO, pw.att["w1"], None, A1o, Mn, 1024, 1024, _capi.EPI_BIAS), dict(ldw=2048, W_off=1024)),
                        ((GO, pw.os["w1"], pw.os["b1"], C1o, Mn, 1024, 1024, _capi.EPI_BIAS), {}),
                        ((GH, pw.so["w1"], pw.so["b1"], C1h, Mh, 1024, 1024, _capi.EPI_BIAS), {})])
            # ---- fc_2 GEMMs over the grid rows with the fc_1*fc_2 -> ReLU product fused
            T = torch.empty(Mg, 1024, **f32); Tos = torch.empty(Mg, 1024, **f32); Tso = torch.empty(Mg, 1024, **f32)
            fc2 = [((S, pw.att["w2"], pw.att["b2"], T, Mg, 1024, 1024, _capi.EPI_MUL_RELU),
                    dict(P=A1h, p_idx=grid_h, ldp=1024, Q=A1o, q_idx=grid_o, ldq=1024, mbias=pw.att["b1"], C_raw=F2,
                         ldc_raw=1024)),
                   ((S, pw.os["w2"], pw.os["b2"], Tos, Mg, 1024, 1024, _capi.EPI_MUL_RELU),
                    dict(P=C1o, p_idx=grid_o, ldp=1024)),
                   ((S, pw.so["w2"], pw.so["b2"], Tso, Mg, 1024, 1024, _capi.EPI_MUL_RELU),
                    dict(P=C1h, p_idx=grid_h, ldp=1024))]
            if need_S is not None:
                need_S()
            if Mg < self.GROUP_FC2_BELOW:       # small grids: one launch fills the CUs better than three (+17 % at 4 images)
                gemm_group(fc2)
            else:
                for a, kw in fc2:
                    gemm(*a, **kw)
            # ---- attention fc_3 + ReLU + adjacency dot (HEAD:896-897)
            n_part = dot_partials(Mg, 1024, 1024, T.stride(0), pw.att["w3"].stride(0))
            part = torch.empty(n_part, Mg, **f32)
            gemm(T, pw.att["w3"], pw.att["b3"], None, Mg, 1024, 1024, _capi.EPI_RELU_DOT, dot_w=pw.adj_w,
                 dot_partial=part)
            # ---- softmax-weighted aggregation before fc_3 (HEAD:907-922)
            U = torch.empty(Mh, 1024, **f32); V = torch.empty(Mn, 1024, **f32); adj = torch.empty(Mg, **f32)
            _capi.check(lib.skg_graph_aggregate_f32(part.data_ptr(), n_part, Mg, pw.adj_b, meta.data_ptr(), A,
                                                    isl("hum_img").data_ptr(), isl("node_img").data_ptr(), Mh, Mn,
                                                    Tos.data_ptr(), Tso.data_ptr(), 1024, 1024, U.data_ptr(),
                                                    V.data_ptr(), 1024, adj.data_ptr(), st),
                        "skg_graph_aggregate_f32")
            del T, Tos, Tso
            Hp = torch.empty(Mh, 1024, **f32); Op = torch.empty(Mn, 1024, **f32)
            gemm_group([((U, pw.os["w3"], pw.os["b3"], Hp, Mh, 1024, 1024, _capi.EPI_BIAS_RES_RELU),
                         dict(res=GH, ldres=1024)),
                        ((V, pw.so["w3"], pw.so["b3"], Op, Mn, 1024, 1024, _capi.EPI_BIAS_RES_RELU),
                         dict(res=GO, ldres=1024))])
            h_node = torch.empty(Mh, 1024, **f32); node = torch.empty(Mn, 1024, **f32)
            # norm_h and norm_o (HEAD:912-914, 923-925) as one launch
            _capi.check(lib.skg_layernorm2_f32(Hp.data_ptr(), 1024, pw.nh_g.data_ptr(), pw.nh_b.data_ptr(), Mh,
                                               h_node.data_ptr(), 1024, Op.data_ptr(), 1024, pw.no_g.data_ptr(),
                                               pw.no_b.data_ptr(), Mn, node.data_ptr(), 1024, 1024, EPS_LN, st),
                        "skg_layernorm2_f32")
        else:
            # num_iter == 0: the raw box_head encodings reach the read-out (HEAD:843-845)
            if need_S is not None:
                need_S()
            gemm(S, pw.att["w2"], pw.att["b2"], F2, Mg, 1024, 1024, _capi.EPI_BIAS)
            h_node = enc.index_select(0, isl("hum_enc_row").long())
            node = enc.index_select(0, isl("node_enc_row").long())
            adj = None
        # ---- read-out attention_head on the kept pairs (HEAD:966-970)
        if Mp:
            B1h = torch.empty(Mh, 1024, **f32); B1o = torch.empty(Mn, 1024, **f32)
            gemm_group([((h_node, pw.att["w1"], None, B1h, Mh, 1024, 1024, _capi.EPI_BIAS), dict(ldw=2048)),
                        ((node, pw.att["w1"], None, B1o, Mn, 1024, 1024, _capi.EPI_BIAS), dict(ldw=2048, W_off=1024))])
            if need_Tg is not None:
                need_Tg()
            Tp = cx["Tg"]                                        # the global branch has consumed this buffer
            _capi.check(lib.skg_rows_mul_relu_f32(B1h.data_ptr(), pair_h.data_ptr(), 1024, B1o.data_ptr(),
                                                  pair_o.data_ptr(), 1024, pw.att["b1"].data_ptr(), F2.data_ptr(),
                                                  pair_grid.data_ptr(), 1024, Mp, 1024, Tp.data_ptr(), 1024, st),
                        "skg_rows_mul_relu_f32")
            gemm(Tp, pw.att["w3"], pw.att["b3"], PF, Mp, 1024, 1024, _capi.EPI_BIAS_RELU, ldc=2048,
                 C_off=ch.P0 * 2048)
        if sc is not None:
            rel_d = rel.to(dev, non_blocking=True); nrm_d = nrm.to(dev, non_blocking=True)
            _capi.check(lib.skg_transh_scores_f32(ent_d.data_ptr(), rel_d.data_ptr(), nrm_d.data_ptr(), self.K,
                                                  self.human_idx, meta.data_ptr(), A,
                                                  sc.data_ptr() + 4 * ch.P0 * self.K, st), "skg_transh_scores_f32")
        if keep is not None:
            keep.setdefault("spatial46", []).append(sp48)
            keep.setdefault("h_node", []).append(h_node); keep.setdefault("node", []).append(node)
            if adj is not None:
                keep.setdefault("adjacency", []).append(adj)

    # ------------------------------------------------------------------------------------------ classifier + scoring
    def _classify(self, pair_features, pw):
        """box_pair_predictor | box_pair_suppressor (HEAD:410-411) as one GEMM -> logits [P, K+1 (ld 120)]."""
        dev = pair_features.device
        Mp = pair_features.shape[0]
        ld = (self.K + 1 + 3) // 4 * 4
        logits = torch.empty(max(Mp, 1), ld, device=dev, dtype=torch.float32)
        if pw.fused_cls:
            if Mp:
                sk = pick_split_k(Mp, self.K + 1, 2048)          # a few images: 2 column tiles x K = 2048 would idle the chip
                ws = torch.empty(sk, Mp, self.K + 1, device=dev, dtype=torch.float32) if sk > 1 else None
                gemm(pair_features, pw.cls_w, pw.cls_b, logits, Mp, self.K + 1, 2048, _capi.EPI_BIAS, split_k=sk,
                     split_ws=ws)
        else:                       # injected modules that are not plain Linear layers are honoured as given
            logits[:Mp, :self.K] = self.predictor(pair_features)
            logits[:Mp, self.K:self.K + 1] = self.suppressor(pair_features)
        return logits[:Mp]

    def score(self, logits, pre, g, training, out=None, L_dev=None):
        """compute_prior_scores + postprocess (HEAD:721-767, 237-337) -> packed result tensors.  out: pre-allocated
        result arrays; L_dev: device int32 with the total number of scored cells (captured-graph path)."""
        lib = _capi.lib()
        lay = g["layout"]
        dev = pre.device
        vt = self.verbs(dev)
        Mp, Lt = lay.sum_p, lay.sum_l
        i64 = dict(device=dev, dtype=torch.int64)
        f32 = dict(device=dev, dtype=torch.float32)
        r = out if out is not None else dict(
            index=torch.empty(max(Lt, 1), **i64), prediction=torch.empty(max(Lt, 1), **i64),
            scores=torch.empty(max(Lt, 1), **f32), prior=torch.empty(2, max(Lt, 1), **f32),
            weights=torch.empty(max(Mp, 1), **f32), object=torch.empty(max(Mp, 1), **i64),
            boxes_h=torch.empty(max(Mp, 1), 4, **f32), boxes_o=torch.empty(max(Mp, 1), 4, **f32))
        if lay.n_active and Mp:
            _capi.check(lib.skg_postprocess_f32(
                logits.data_ptr(), logits.stride(0), self.K, pre.boxes.data_ptr(), pre.scores.data_ptr(),
                pre.labels.data_ptr(), g["meta"].data_ptr(), lay.n_active, g["x_keep"].data_ptr(),
                g["y_keep"].data_ptr(), vt.off.data_ptr(), vt.flat.data_ptr(), vt.num_obj,
                1.0 if training else 2.8, max(Lt, 1), _ptr(L_dev), int(lay.pairs_per_image.max()),
                r["index"].data_ptr(), r["prediction"].data_ptr(),
                r["scores"].data_ptr(), r["prior"].data_ptr(), r["weights"].data_ptr(), r["object"].data_ptr(),
                r["boxes_h"].data_ptr(), r["boxes_o"].data_ptr(), _stream()), "skg_postprocess_f32")
        return r
