"""Data-parallel trainer shell for the interaction head: optimizer groups, schedule, DDP wrap, the train
step, and -- below -- the data side (detection filter, horizontal flip, collate, distributed loader) and the epoch loop
with the reference's checkpoint dictionary.

Mirrors the reference's settings: AdamW with two parameter groups -- interaction head at `lr`, everything else (the
detector backbone/neck when it is fine-tuned) at `lr * 0.1`, weight decay 1e-4
(configures/hicodet/adamixer_transH_spatial_r50_main.py:109-127), LambdaLR that multiplies the rate by `lr_decay`
from epoch `milestone` on (main:128-132), total loss = plain sum of the loss dict (utils.py:221), ValueError on a NaN
HOI loss (utils.py:218-219), DDP with find_unused_parameters=True (utils.py:202-205).  One process per GPU;
`backend="nccl"` is RCCL on ROCm.  The three `n_p` all-reduces of the reference are issued by the head itself
(InteractionHead.distributed=True).
"""
import math

import torch
import torch.distributed as dist
from torch import nn

_data_ptr = torch.Tensor.data_ptr


def limit_host_threads(ranks_on_host: int = 1, cap: int = 4) -> int:
    """Sizes torch's intra-op CPU pool for the training loop: this process's share of the usable cores (cgroup quota
    aware, skghoi_amd.dist.host_cpu_share), at most `cap`.  The step is host-bound and its CPU ops are tiny
    (randperm, small cats): on a 256-core host with a 16-core quota the default 128 threads made it 3-4x slower."""
    from .dist import host_cpu_share
    n = max(1, min(cap, host_cpu_share() // max(1, ranks_on_host)))
    torch.set_num_threads(n)
    return n


def build_optimizer(net: nn.Module, lr: float = 1e-4, weight_decay: float = 1e-4, head_key: str = "interaction_head"):
    """main:109-127: parameters whose name contains `head_key` train at lr, the rest at lr * 0.1.  A bare
    InteractionHead (no wrapper, so no 'interaction_head' in its names) is treated as all-head."""
    named = [(n, p) for n, p in net.named_parameters() if p.requires_grad]
    head = [p for n, p in named if head_key in n]
    rest = [p for n, p in named if head_key not in n]
    if not head:
        head, rest = rest, []
    groups = [{"params": head}]
    if rest:
        groups.append({"params": rest, "lr": lr * 0.1})
    # same update rule; on the GPU the multi-tensor "fused" implementation is one launch per step instead of ~10 per
    # parameter group walk (6 ms of host time for the head's 408 tensors)
    on_gpu = bool(named) and all(p.is_cuda for _, p in named)
    if on_gpu:
        return SkgAdamW(groups, lr=lr, weight_decay=weight_decay, fused=True)
    return torch.optim.AdamW(groups, lr=lr, weight_decay=weight_decay)


class CachedFusedAdamW(torch.optim.AdamW):
    """torch.optim.AdamW(fused=True) -- same state, same state_dict, same kernel (torch._fused_adamw_) -- whose step()
    does not rebuild its five per-parameter tensor lists from the state dict on every call: for the head's 408
    parameters that walk is ~0.9 ms of host time per step, on a step whose GPU work is ~3 ms.  The lists are cached per
    group and revalidated cheaply (list lengths, identity of the first / last state tensors: load_state_dict and
    add_param_group replace them); anything the fast path does not cover (first step, a parameter without gradient,
    amsgrad, grad scaling, capturable, tensor lr) goes through the stock implementation."""

    def __init__(self, *args, **kw):
        super().__init__(*args, **kw)
        self._lists = {}

    def _cached(self, gi, group):
        params = group["params"]
        c = self._lists.get(gi)
        if c is not None and c[0] is params and len(c[1]) == len(params):
            s0, s1 = self.state.get(params[0]), self.state.get(params[-1])
            if s0 is not None and s1 is not None and s0.get("exp_avg") is c[2][0] and s1.get("exp_avg") is c[2][-1] \
                    and s0.get("step") is c[4][0]:
                return c
        st = [self.state.get(p) for p in params]
        if any(x is None or "exp_avg" not in x or not torch.is_tensor(x.get("step")) or not x["step"].is_cuda for x in st):
            return None
        c = (params, list(params), [x["exp_avg"] for x in st], [x["exp_avg_sq"] for x in st], [x["step"] for x in st])
        self._lists[gi] = c
        return c

    def zero_grad(self, set_to_none: bool = True):
        if not set_to_none:
            return super().zero_grad(set_to_none)
        for group in self.param_groups:                  # the stock loop spends ~0.2 ms on per-tensor checks for 408 tensors
            for p in group["params"]:
                p.grad = None

    @torch.no_grad()
    def step(self, closure=None):
        if closure is not None or getattr(self, "grad_scale", None) is not None or getattr(self, "found_inf", None) is not None:
            return super().step(closure)
        plan = []
        for gi, group in enumerate(self.param_groups):
            if group.get("amsgrad") or group.get("maximize") or group.get("capturable") or group.get("differentiable") \
                    or not group.get("fused") or torch.is_tensor(group["lr"]) or not group["params"]:
                return super().step()
            c = self._cached(gi, group)
            if c is None:
                return super().step()                               # first step: the stock path creates the state
            grads = [p.grad for p in c[1]]
            if any(g is None for g in grads):
                return super().step()
            plan.append((group, c, grads))
        for group, c, grads in plan:
            beta1, beta2 = group["betas"]
            torch._foreach_add_(c[4], 1)
            torch._fused_adamw_(c[1], grads, c[2], c[3], [], c[4], amsgrad=False, lr=group["lr"], beta1=beta1, beta2=beta2,
                                weight_decay=group["weight_decay"], eps=group["eps"], maximize=False, grad_scale=None,
                                found_inf=None)
        return None


def build_scheduler(optimizer, milestone: int = 6, lr_decay: float = 0.1):
    """main:128-132."""
    return torch.optim.lr_scheduler.LambdaLR(optimizer, lambda epoch: 1.0 if epoch < milestone else lr_decay)


def _interaction_heads(module: nn.Module):
    from .adamixer_transH_spatial_r50_head import InteractionHead
    return [m for m in module.modules() if isinstance(m, InteractionHead)]


class NativeComm:
    """An skg_comm (include/skghoi.h): the HIP library's own RCCL communicator + stream for the gradient exchange, over the
    ranks of a torch.distributed process group whose backend is RCCL ("nccl").  `create` is COLLECTIVE over that group and
    agrees on the outcome: every rank gets a communicator or every rank gets None (and keeps torch.distributed's
    collectives) -- RCCL absent on some rank, ncclCommInitRank refused (two ranks on one GPU), SKG_NATIVE_RCCL=0."""

    def __init__(self, handle, world, rank):
        self.handle, self.world, self.rank = handle, world, rank

    @staticmethod
    def _agree(device, group, *flags):
        """AND of each flag over the ranks of the group."""
        t = torch.tensor([1 if f else 0 for f in flags], device=device, dtype=torch.int32)
        dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
        out = [bool(v) for v in t.tolist()]
        return out[0] if len(out) == 1 else out

    @classmethod
    def create(cls, device, group=None):
        import ctypes as C
        import os
        import warnings
        from . import _capi
        # SKG_RCCL_LIB: the library to bind instead of PyTorch's librccl (another RCCL build -- or the blocking shared-memory
        # stand-in of tests/fake_rccl, which lets two ranks on ONE GPU run this route end to end over a gloo group)
        named = os.environ.get("SKG_RCCL_LIB")
        if os.environ.get("SKG_NATIVE_RCCL", "1") == "0" or device.type != "cuda" or \
                (dist.get_backend(group) != "nccl" and not named):
            return None
        lib = _capi.lib()
        path = named or os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
        loaded = lib.skg_comm_load(path.encode() if os.path.exists(path) else None) == 0
        if not cls._agree(device, group, loaded):
            return None
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        ident = torch.zeros(_capi.COMM_ID_BYTES + 1, dtype=torch.uint8, device=device)      # id | "id is valid"
        if rank == 0:
            buf = (C.c_char * _capi.COMM_ID_BYTES)()
            if lib.skg_comm_unique_id(buf) == 0:
                ident = torch.cat([torch.frombuffer(bytearray(bytes(buf)), dtype=torch.uint8),
                                   torch.ones(1, dtype=torch.uint8)]).to(device)
        dist.broadcast(ident, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        raw = bytes(ident.cpu().tolist())
        h = C.c_void_p()
        res = {}
        if raw[-1] == 1:
            # ncclCommInitRank is collective: a rank that never arrives would park its peers inside it for good.  It runs on
            # a helper thread with a deadline; a rank that misses it reports so, everybody falls back, and the stuck
            # communicator is left alone (never destroyed: its peers may still be inside the call).
            import threading

            def init():
                with torch.cuda.device(device):
                    res["rc"] = lib.skg_comm_create(raw[:_capi.COMM_ID_BYTES], rank, world, C.byref(h))
                    res["err"] = (lib.skg_comm_last_error() or b"").decode(errors="replace")
            t = threading.Thread(target=init, daemon=True)
            t.start()
            t.join(float(os.environ.get("SKG_NATIVE_RCCL_TIMEOUT", "120")))
        rc = res.get("rc", -1)
        in_time = raw[-1] != 1 or "rc" in res
        ok, all_in_time = cls._agree(device, group, rc == 0, in_time)
        if not ok:
            if rc == 0 and all_in_time:
                lib.skg_comm_destroy(h)
            if rank == 0:
                warnings.warn("skghoi_amd: the library's own RCCL communicator could not be created on every rank (%s); the "
                              "gradient exchange stays on torch.distributed"
                              % (res.get("err") or ("no answer within the deadline" if not in_time else "a peer failed")))
            return None
        comm = cls(h, world, rank)
        import atexit
        import weakref
        ref = weakref.ref(comm)
        atexit.register(lambda: ref() is not None and ref().close())     # RCCL torn down before the HIP runtime is
        return comm

    def close(self):
        """Destroys the communicator (drains its stream first).  The trainer's backward must have been joined."""
        h, self.handle = self.handle, None
        if h:
            from . import _capi, train_fused
            train_fused.join_backward(True)
            _capi.lib().skg_comm_destroy(h)

    def __del__(self):
        try:
            self.close()
        except Exception:                                   # noqa: BLE001  (interpreter shutdown)
            pass


class ArenaExchange:
    """Gradient exchange of the fused training step over its flat gradient ARENA (skghoi_amd/train_fused.py, Stacked).

    The reference wraps the net in DistributedDataParallel(find_unused_parameters=True) (utils.py:202-205): per-parameter
    hooks fill buckets while autograd walks the graph.  The fused step is ONE autograd node -- all 408 gradients would
    become ready at the same instant and every bucket's all-reduce would queue up behind the whole backward.  Here the
    backward itself drives the exchange: the arena is laid out in the order in which the backward FINISHES gradients
    (read-out layers first, box_head last), and after every stage the newly final prefix goes out as an all-reduce on the
    process group's own stream (RCCL over xGMI), concurrent with the stages still to run; the optimizer waits for the
    last one.  Collectives per step: the fused 3-element normaliser all-reduce + one per chunk (prefixes are coalesced to
    at least `min_chunk` floats; the 12.8 M-float box_head.1 gradient closes the backward and is the tail that cannot
    overlap).  Gradients are averaged (sum, then 1 / world folded into ONE scaling pass: gloo has no AVG).

    A rank whose step did not run through the fused node (the Python-issued launch plan, an unsupported configuration on
    that rank) still has to meet its peers in the same collectives: `after_backward` sends its parameter gradients (zeros
    where a parameter has none -- DDP's find_unused_parameters semantics) through the same chunk sequence.  (A batch without
    a single pair cannot complete a step on either side: the reference's torch.cat over its empty score lists raises.)"""

    def __init__(self, head, group=None, min_chunk=1 << 23, native=None):
        self.head, self.group, self.min_chunk = head, group, int(_os.environ.get("SKG_DP_MIN_CHUNK") or min_chunk)
        self.n_stages = 12                    # (_capi.TRAIN_BWD_STAGES: the arena's milestones)
        self.native = native                  # NativeComm: the library's worker thread issues the collectives itself
        self._after = None                    # with timing on the native route: event behind the backward's last launch
        self.world = dist.get_world_size(group)
        self.works, self.done, self.ga = [], 0, None
        self.ran = False
        self.collectives = 0                  # arena collectives of the last step
        self.timing = False
        self.last_ms = None                   # with timing: (device time the step's stream waited for the exchange, ms)
        self._xstream = None                  # stream the collectives are issued on: ordered behind ONE stage event each
        self.prescaled = False                # this step's gradients already carry the 1 / world of the mean

    # -- driven by NativeJob.backward
    def begin(self, ga):
        self.works, self.done, self.ga, self.collectives = [], 0, ga, 0
        self.ran = True

    def _goes_out(self, s, pending, last):
        """Whether the arena prefix pending behind stage s leaves as a chunk: coalesced to min_chunk floats (every chunk costs
        the backward one event -- a ~12 us bubble on its queue, measured at world size 1: 6 / 4 / 2 chunks -> +0.08 / +0.05 /
        +0.03 ms); the stage before the last sends what it has from a quarter of that on, so that the tail nobody can overlap
        -- the 12.8 M-float box_head.1 gradient of the last stage -- is not made longer; the last stage sends the rest."""
        return pending >= self.min_chunk or (s == self.n_stages - 2 and pending >= self.min_chunk // 4) or \
            (last and pending > 0)

    def on_stage(self, s, ga, end, last=False):
        if self._goes_out(s, end - self.done, last):
            if _PROBE != "nocoll":           # (developer probe: the staged backward without its collectives)
                self.works.append(dist.all_reduce(ga[self.done:end], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
            self.done = end
            self.collectives += 1

    def chunk_stages(self, ends):
        """[(stage, end)]: the stages of the backward behind which an arena chunk goes out (prefixes coalesced to min_chunk).
        The same milestone list comes back every step: the answer is kept (three walks per step otherwise, on the host
        thread, between the loss and the backward's submission -- where the GPU is waiting for this thread)."""
        key = (tuple(ends), self.min_chunk, self.n_stages)
        c = self.__dict__.get("_chunks_cache")
        if c is not None and c[0] == key:
            return c[1]
        out = self._chunk_stages(ends)
        self._chunks_cache = (key, out)
        self.__dict__.pop("_mask_cache", None)
        self.__dict__.pop("_ex_cache", None)
        return out

    def _chunk_stages(self, ends):
        out, done, n = [], 0, len(ends)
        for s_, end in enumerate(ends):
            last = s_ == n - 1
            if self._goes_out(s_, end - done, last):
                out.append((s_, end))
                done = end
        return out

    def stage_mask(self, ends):
        """Bit mask of chunk_stages() for skg_ctx_train_backward_async_f32: the worker records the context's own (device-
        scope) event behind exactly these stages."""
        chunks = self.chunk_stages(ends)
        c = self.__dict__.get("_mask_cache")
        if c is not None and c[0] is chunks and c[1] == self.world:
            return c[2]
        m = 0
        for s_, _ in chunks:
            m |= 1 << s_
        if m and self.world > 1:
            # the chunk goes to OTHER GPUs: the stage's writes must leave this GPU's L2 before the collective reads them
            # (bit 31: the context records events with the default system-scope release instead of its device-scope set)
            m |= 1 << 31
        self._mask_cache = (chunks, self.world, m)
        return m

    def native_exchange(self, ga, ends):
        """skg_exchange of this step for skg_ctx_train_backward_exchange_f32: the chunk table of chunk_stages(), the arena
        padding behind the last milestone included in the last chunk (as finish() sends it on the torch route)."""
        from . import _capi
        chunks = self.chunk_stages(ends)
        c = self.__dict__.get("_ex_cache")
        if c is not None and c[0] is chunks and c[1] == (ga.data_ptr(), ga.numel(), self.native.handle):
            ex = c[2]                        # the gradient arena is handed out again from step to step: the same table
            ex.adamw = None                  # (the caller fills the optimizer's part per step, or leaves it off)
        else:
            ex = _capi.Exchange()
            ex.comm, ex.arena, ex.n_chunks = self.native.handle, ga.data_ptr(), len(chunks)
            for i, (s_, end) in enumerate(chunks):
                ex.stage[i], ex.end[i] = s_, end
            ex.end[len(chunks) - 1] = max(int(ex.end[len(chunks) - 1]), ga.numel())
            self._ex_cache = (chunks, (ga.data_ptr(), ga.numel(), self.native.handle), ex)
        self.done = ga.numel()
        self.collectives = len(chunks)
        return ex

    def native_all_reduce(self, ga, ends):
        """The step's chunk sequence on the library's communicator from THIS thread, behind the current stream's tail (a rank
        off the staged route meets its peers' collectives one for one); the current stream is ordered behind the last."""
        import ctypes as C
        from . import _capi
        chunks = self.chunk_stages(ends)
        arr = (C.c_int64 * len(chunks))(*[int(e) for _, e in chunks])
        arr[len(chunks) - 1] = max(int(arr[len(chunks) - 1]), ga.numel())
        _capi.check(_capi.lib().skg_comm_all_reduce_chunks_f32(self.native.handle, ga.data_ptr(), arr, len(chunks),
                                                               torch.cuda.current_stream(ga.device).cuda_stream),
                    "skg_comm_all_reduce_chunks_f32")
        self.done = ga.numel()
        self.collectives = len(chunks)
        self._after = None

    def drive(self, ctx, ends):
        """The backward was handed to the library's worker thread in ONE call (train_fused.NativeJob.backward, defer=True).
        For every arena chunk: wait (on the host, without the GIL) until the worker has issued the stage that completes the
        chunk, order the exchange stream behind THAT stage's event -- not behind whatever else the worker has put on the
        step's stream since -- and start the collective there."""
        ga = self.ga
        if self._xstream is None:
            from .engine import shared_side_stream
            self._xstream = shared_side_stream(ga.device, 0, slot=1)
        xs = self._xstream
        if _PROBE == "mainstream":           # (developer probe: collectives on the step's own stream, behind whatever it holds)
            xs = torch.cuda.current_stream(ga.device)
        n = len(ends)
        for s_, end in self.chunk_stages(ends):
            ctx.stage_wait(s_, xs)
            with torch.cuda.stream(xs):
                self.on_stage(s_, ga, end, last=(s_ == n - 1))

    def finish(self):
        """Orders the step's stream behind every chunk and forms the average."""
        ga = self.ga
        if self.native is not None and not self.works and self.done >= ga.numel():
            # the worker thread issued the collectives and ordered the step's stream behind the last one
            if not self.prescaled:
                ga.mul_(1.0 / self.world)
            self.prescaled = False
            return
        if self.done < ga.numel():                              # (arena padding at the very end)
            self.on_stage(-1, ga, ga.numel(), last=True)
        ev0 = ev1 = None
        if self.timing and ga.is_cuda:
            ev0 = torch.cuda.Event(enable_timing=True); ev1 = torch.cuda.Event(enable_timing=True)
            ev0.record()
        for w in self.works:
            w.wait()
        if not self.prescaled:               # (the fused step folds 1 / world into its logit gradients: train_fused.TrainRun.tail)
            ga.mul_(1.0 / self.world)
        self.prescaled = False
        if ev0 is not None:
            ev1.record()
            self._events = (ev0, ev1)
        self.works = []

    def read_timing(self):
        if self.native is not None and self._after is not None:
            import ctypes as C
            from . import _capi
            ms = C.c_float()
            _capi.check(_capi.lib().skg_comm_exposed_ms(self.native.handle, self._after.cuda_event, C.byref(ms)),
                        "skg_comm_exposed_ms")
            self.last_ms = float(ms.value)
            return self.last_ms
        ev = getattr(self, "_events", None)
        if ev is None:
            return None
        ev[1].synchronize()
        self.last_ms = ev[0].elapsed_time(ev[1])
        return self.last_ms

    # -- driven by train_step
    def after_backward(self):
        """Call between backward and optimizer.step().  No-op when the fused node ran; otherwise this rank joins its peers'
        collectives with whatever gradients its parameters hold."""
        if self.ran:
            self.ran = False
            return
        from . import train_fused
        head = self.head
        dev = next(head.parameters()).device
        st = train_fused._stacked_for(head, dev)
        if not st.aliased():
            st.adopt()
        ga = torch.zeros(st.total, device=dev, dtype=torch.float32)
        views = st.grad_views(ga)
        with torch.no_grad():
            for p, v in zip(st.src, views):
                if p.grad is not None:
                    v.copy_(p.grad)
        # this rank's share of the mean goes in before the sum, like its peers' (whose fused step folded 1 / world into
        # their logit gradients): every rank then ends with the same sum.  A step that left the fused node AFTER the shared
        # loss tail (the Python-issued launch plan: train_fused.TrainRun.tail) already carries the share in its gradients.
        if not self.prescaled:
            ga.mul_(1.0 / self.world)
        self.begin(ga)
        self.prescaled = True
        if self.native is not None:
            self.native_all_reduce(ga, st.milestone_end)
        else:
            for s_, end in enumerate(st.milestone_end):
                self.on_stage(s_, ga, end, last=(s_ == len(st.milestone_end) - 1))
        self.finish()
        for p, v in zip(st.src, views):
            if p.requires_grad:
                p.grad = v
        self.ran = False


import os as _os
_PROBE = _os.environ.get("SKG_DP_PROBE", "")


def exchanges(module: nn.Module):
    """The gradient exchanges installed under `module` (wrap_ddp).  The module walk (~240 sub-modules: 0.2 ms) is redone only
    after a module / parameter registration anywhere or a new wrap_ddp call."""
    from .engine import _REG_EPOCH
    key = (_REG_EPOCH[0], _WRAP_EPOCH[0])
    c = module.__dict__.get("_skg_exchanges")
    if c is None or c[0] != key:
        c = (key, [h.grad_exchange for h in _interaction_heads(module) if getattr(h, "grad_exchange", None) is not None])
        module.__dict__["_skg_exchanges"] = c
    return c[1]


_WRAP_EPOCH = [0]


def wrap_ddp(module: nn.Module, device=None, force_exchange=False):
    """utils.py:202-205 (pocket's engine wraps the net in DDP with find_unused_parameters=True).  A single process needs
    no gradient hooks: the head's fused step then writes p.grad directly (grad_mode "direct", ~1 ms of autograd
    bookkeeping per step saved); under DDP the gradients go through the autograd engine, whose hooks DDP listens to.
    force_exchange: install the arena exchange even in a process group of ONE rank (bench.py times the data-parallel route
    -- RCCL's streams, kernels and the staged backward -- on a single GPU that way)."""
    _WRAP_EPOCH[0] += 1
    up = dist.is_available() and dist.is_initialized()
    if not up or (dist.get_world_size() == 1 and not force_exchange):
        for h in _interaction_heads(module):
            h.grad_mode = "direct"
        return module
    # data parallel: every interaction head exchanges its own gradients over its arena, behind its backward
    # (ArenaExchange); whatever else the module trains (a detector backbone) stays with DistributedDataParallel
    from . import train_fused
    ignore = []
    heads = _interaction_heads(module)
    names = {id(m): n for n, m in module.named_modules()}
    fused = []
    for h in heads:
        dev = next(h.parameters()).device
        if train_fused.supported(h) and h.fused_training and dev.type == "cuda":
            fused.append(h)
            prefix = names[id(h)]
            ignore += [(prefix + "." if prefix else "") + n for n, _ in h.named_parameters()]
        else:
            h.grad_mode = "autograd"
    rest = [n for n, p in module.named_parameters() if p.requires_grad and n not in set(ignore)]
    # With a DDP remainder the arena collectives get a process group of their OWN: a rank whose batch bypassed the fused node
    # issues them in after_backward(), i.e. AFTER DDP's bucket all-reduces, its peers from inside the backward, BEFORE --
    # on one communicator that order mismatch hangs or mixes buffers; two communicators are independent.
    group = dist.new_group() if (rest and fused and dist.get_world_size() > 1) else None
    # The head alone is trained (BASELINE config 4: cached detections): its exchange runs on a communicator of the HIP
    # library's own, issued by the backward's worker thread (NativeComm; collective, all ranks agree or all decline).  With a
    # DDP remainder two host threads would feed two communicators at once -- kept on torch.distributed, one issue order.
    native = NativeComm.create(next(fused[0].parameters()).device, group) if (fused and not rest) else None
    for h in fused:
        dev = next(h.parameters()).device
        h.grad_mode = "direct"
        st = train_fused._stacked_for(h, dev)
        if not st.aliased():
            st.adopt()
        dist.broadcast(st.buf, 0, group=group)              # DDP's constructor broadcast: rank 0's parameters everywhere
        h.grad_exchange = ArenaExchange(h, group=group, native=native)
        if force_exchange:
            h.force_collectives = True                      # (a world of one rank still runs the normaliser collective)
    if not rest or dist.get_world_size() == 1:
        return module
    if ignore:
        nn.parallel.DistributedDataParallel._set_params_and_buffers_to_ignore_for_model(module, ignore)
    ids = [device.index] if device is not None and device.type == "cuda" else None
    return nn.parallel.DistributedDataParallel(module, device_ids=ids, find_unused_parameters=True)


def _join_backward():
    from .train_fused import join_backward
    join_backward()


def _plain_step(optimizer):
    """A SkgAdamW without step hooks whose update may be issued by the backward's worker thread (train_step) -- OPT-IN
    (SKG_ADAMW_IN_BACKWARD=1).  Measured on one MI355X, batch-4 step: the update of the first 16.6 M parameters running
    beside backward stages 7-11 costs those stages more (HBM / L2 traffic of a 0.5 GB stream under GEMMs that live on L2
    hits, two more events on their queue) than the 0.07 ms it takes off the tail: bf16 1.27 -> 1.33 ms, fp32 2.48 -> 2.51,
    data-parallel route at world size 1 1.35 -> 1.38.  Results are bit-identical either way (tests/test_trainer.py)."""
    from torch.optim import optimizer as _topt
    return isinstance(optimizer, SkgAdamW) and not optimizer._optimizer_step_pre_hooks and \
        not optimizer._optimizer_step_post_hooks and not _topt._global_optimizer_pre_hooks and \
        not _topt._global_optimizer_post_hooks and _os.environ.get("SKG_ADAMW_IN_BACKWARD", "0") == "1"


def _optimizer_step(optimizer):
    """optimizer.step() -- for SkgAdamW without torch's per-call profiler wrapper (Optimizer.profile_hook_step: a
    record_function scope + hook bookkeeping, ~35 us of a step whose host thread is the bound) when no step hooks are
    registered; what the wrapper and the LR schedulers' call counter record is kept."""
    from torch.optim import optimizer as _topt
    if isinstance(optimizer, SkgAdamW) and not optimizer._optimizer_step_pre_hooks and \
            not optimizer._optimizer_step_post_hooks and not _topt._global_optimizer_pre_hooks and \
            not _topt._global_optimizer_post_hooks:
        # (Optimizer.__init__ patched the class: step = profile_hook_step(step), marked `hooked`; one level down is the
        #  @torch.no_grad-decorated method itself)
        raw = getattr(SkgAdamW.step, "__wrapped__", None) if getattr(SkgAdamW.step, "hooked", False) else None
        if raw is not None:
            optimizer._opt_called = True                    # (lr_scheduler's "step() before optimizer.step()" check)
            return raw(optimizer)
    return optimizer.step()


def _release_prepared(net):
    mod = net.module if isinstance(net, nn.parallel.DistributedDataParallel) else net
    if "_prep_slot_in_use" in mod.__dict__:
        from .train_fused import release_prepared
        release_prepared(mod)


def _drive_exchanges():
    from .train_fused import drive_exchanges
    drive_exchanges()


_ACCEPTS = {}
_CUDA_HERE = []


def _cuda_here():
    """torch.cuda.is_available(), asked once (it re-reads the environment and counts devices on every call: ~4 us)."""
    if not _CUDA_HERE:
        _CUDA_HERE.append(bool(torch.cuda.is_available()))
    return _CUDA_HERE[0]



def _accepts(fn, name):
    """Whether the (bound) method takes a keyword `name` -- asked once per function (a module other than the interaction
    head may offer prefetch_train with the older signature)."""
    f = getattr(fn, "__func__", fn)
    key = (f, name)
    if key not in _ACCEPTS:
        import inspect
        try:
            ps = inspect.signature(f).parameters
            _ACCEPTS[key] = name in ps or any(p.kind is p.VAR_KEYWORD for p in ps.values())
        except (TypeError, ValueError):
            _ACCEPTS[key] = False
    return _ACCEPTS[key]


def prefetch_batch(net, *batch, after=None, arena=False, deep=False):
    """Hands the NEXT batch to the head for preparation on its side stream (InteractionHead.prefetch_train) while the GPU
    works on the step just enqueued; returns the handle of the preparation in progress (advance() / finish()) or None.
    Only when `net` IS the interaction head (training from cached detections / features) and the batch has the head's own
    call shape (features, detections, image_shapes, targets): inside a full detector the head's inputs exist only after the
    detector has run, and a loader that yields (images, detections, targets) (utils.py:34-42) has nothing the head could
    prepare.  A look-ahead is an optimisation: anything it cannot use it leaves alone -- it never raises for the batch's
    shape."""
    mod = net.module if isinstance(net, nn.parallel.DistributedDataParallel) else net
    fn = getattr(mod, "prefetch_train", None)
    if fn is None or len(batch) != 4:
        return None
    features, detections, image_shapes, targets = batch
    if not isinstance(detections, (list, tuple)) or not detections or not isinstance(detections[0], dict) \
            or not torch.is_tensor(detections[0].get("boxes")) or not detections[0]["boxes"].is_cuda or targets is None:
        return None
    kw = {}
    if after is not None:
        kw["after"] = after
    if arena and _accepts(fn, "arena"):
        kw["arena"] = True
    if deep:
        if not _accepts(fn, "deep"):
            return None
        kw["deep"] = True
    return fn(detections, image_shapes, targets, **kw) or None


def _promote_prefetched(net, batch):
    """The preparation `train_step(prefetch2=batch)` started one step ago, now the next batch's (None if there is none)."""
    mod = net.module if isinstance(net, nn.parallel.DistributedDataParallel) else net
    fn = getattr(mod, "promote_prefetched", None)
    if fn is None or batch is None or len(batch) != 4:
        return None
    return fn(batch[1], batch[3])


def train_step(net, optimizer, *inputs, targets, lazy=False, prefetch=None, prefetch2=None):
    """utils.py:213-229: zero_grad -> forward -> NaN guard -> sum of the loss dict -> backward -> step.  Returns the loss
    dict (detached floats) and the per-image results.

    When `net` is the bare interaction head in its fused training configuration the forward and the backward run as one
    call without the autograd engine (`InteractionHead.fused_step`: same kernels, same order, the upstream gradient of the
    summed losses fixed at one, gradients OVERWRITE `p.grad` -- so nothing is zeroed first); every other case goes
    `zero_grad -> net(...) -> backward()` like the reference.

    prefetch = the NEXT batch (features, detections, image_shapes, targets), or None: once this step is enqueued the
    head prepares that batch on a side stream (selection, pairs, labels, host RNG draws), so that the next call starts
    its dense forward at once instead of paying two host synchronisations with an idle GPU (`prefetch_batch`).

    prefetch2 = the batch AFTER that one, or None (round 5).  With one batch of look-ahead the preparation's last part -- the
    read-back of the positive counts, ~0.16 ms of reference RNG draws on this thread, the uploads, the TransH term -- sits
    between this step's optimizer launch and the next step's forward, and it takes longer than AdamW runs: the GPU idles
    ~85 us at every step boundary (profiles/r05_train_bf16_b4_step_timeline.txt).  With two, batch i + 2's preparation is
    STARTED during step i (selection, pairs, association: device work on the side stream) and FINISHED during step i + 1,
    right behind the submission of that step's backward, where this thread would otherwise wait for the worker; the next
    step's forward then follows the optimizer launch at once.  Same batches, same order of host RNG draws, same results.

    lazy=True keeps the host off the GPU's heels: the losses come back as detached DEVICE tensors and the NaN guard is
    left to whoever reads them (Trainer does, at its print interval / end of epoch) -- the reference's per-iteration
    `isnan` test and `.item()` are two host synchronisations per step, ~0.5 ms of a ~4 ms batch-4 step during which
    nothing is queued behind the optimizer kernels."""
    # the next batch's preparation is interleaved with this step's host work: each of its two device round trips (selection
    # counts, positive counts) runs on the side stream while this thread enqueues the backward resp. the optimizer
    ahead = [None]
    entry = None
    updated = None
    if prefetch is not None and _cuda_here():
        # behind whatever produced the next batch (a loader's non-blocking uploads were enqueued before this call), in front
        # of this step's own kernels: what the side stream's preparation has to wait for, and no more
        from .engine import current_stream_of
        entry = torch.cuda.Event()
        entry.record(current_stream_of(None))

    ahead2 = [None]
    promoted = [False]
    if prefetch2 is not None and exchanges(net):
        # data parallel: one batch of look-ahead.  Finishing a preparation in the middle of the backward puts its fused
        # normaliser all-reduce (torch's communicator) beside the arena chunks' collectives (the library's): measured at world
        # size 1 the step got SLOWER with two (1.424 against 1.333 ms), where the single-process step gains 3 %
        prefetch2 = None

    def look_ahead():
        if prefetch is not None and ahead[0] is None:
            h = _promote_prefetched(net, prefetch) if _cuda_here() else None
            if h is not None:
                # started a step ago (prefetch2 of the previous call): its device work is long done -- finish it NOW (counts,
                # host RNG draws, uploads, TransH term), behind this step's submitted backward instead of behind its optimizer
                ahead[0] = h
                promoted[0] = True
                h.finish()
            else:
                ahead[0] = prefetch_batch(net, *prefetch, after=entry, arena=True)      # selection kernel launched
        if prefetch2 is not None and ahead2[0] is None and ahead[0] is not None and entry is not None:
            ahead2[0] = prefetch_batch(net, *prefetch2, after=entry, arena=True, deep=True)     # its selection kernel
    out = None
    fused = getattr(net, "fused_step", None)            # the bare interaction head: forward + backward without the autograd
    try:
        if fused is not None and len(inputs) == 3:      # engine (gradients overwrite p.grad: nothing to zero beforehand)
            kw = {}
            if lazy and _accepts(fused, "adamw") and _plain_step(optimizer):
                # the optimizer inside the backward: the worker thread updates each arena chunk's parameters as soon as the
                # chunk's gradients are final (and all-reduced), beside the stages still to run.  Only with lazy losses: the
                # eager NaN guard below must be able to stop the step BEFORE any parameter moves.
                kw["adamw"] = optimizer.backward_slices
            out = fused(*inputs, targets, after_forward=look_ahead, defer_backward=True, **kw)
        fused_ran = out is not None
        updated = fused_ran and net.__dict__.pop("_adamw_in_backward", None)
        if out is None:
            optimizer.zero_grad(set_to_none=True)
            out = net(*inputs, targets)
        loss_dict = out.pop()
        if not lazy and torch.isnan(loss_dict["hoi_loss"]):
            raise ValueError(f"The HOI loss is NaN")
        look_ahead()
        advanced = False
        if fused_ran and ahead[0] is not None and exchanges(net):
            # data parallel: handing out the chunks blocks this thread until the worker has enqueued the backward's last stage
            # (~0.4 ms) -- the look-ahead's first half (counts read, pairs + association launched) goes in front of that wait,
            # not behind it, or the next batch's preparation and this step's optimizer both start late
            h = ahead2[0] if promoted[0] else ahead[0]       # (the batch whose preparation began in THIS call)
            if h is not None:
                h.advance()
            advanced = True
        if fused_ran:
            _drive_exchanges()           # data parallel: the arena chunks go out behind the stages the worker is issuing
        if any(v.requires_grad for v in loss_dict.values()):
            if fused_ran:
                # fused_step handed the losses back on the autograd route (a differentiable RoI pooling in front of the
                # head): whatever the parameters still hold from the step before must not be added to
                optimizer.zero_grad(set_to_none=True)
            total = sum(loss for loss in loss_dict.values())
            total.backward()
        if ahead[0] is not None and not advanced:
            # counts read (ready by now), pairs + association launched -- for the batch whose preparation began in this call
            h = ahead2[0] if promoted[0] else ahead[0]
            if h is not None:
                h.advance()
        for ex in exchanges(net):        # data parallel: a rank whose batch bypassed the fused node joins its peers here
            ex.after_backward()
    finally:
        # whatever happened above (the NaN guard, a failing look-ahead): a backward handed to the worker thread is joined
        # before this frame's tensors go away -- the plan names them
        _join_backward()                 # the backward's launches are all on the stream before the optimizer's
        _release_prepared(net)           # ... and the arena slot of the batch this step consumed may be reused behind them
        if updated is None and fused is not None:
            updated = getattr(net, "__dict__", {}).pop("_adamw_in_backward", None)      # (fused() raised after the submit)
        if updated:
            optimizer.backward_done(updated)                 # (whatever else happened: the update IS on the stream)
    if not updated:
        _optimizer_step(optimizer)
    spans = getattr(net, "_train_spans", None)
    if spans and "o1" not in spans[-1]:
        spans[-1]["o1"] = torch.cuda.Event(enable_timing=True)
        spans[-1]["o1"].record()
    if ahead[0] is not None and not promoted[0]:
        ahead[0].finish()                # positive counts read, host RNG draws, uploads, TransH term
    if lazy:
        return {k: v.detach() for k, v in loss_dict.items()}, out
    return {k: float(v.detach()) for k, v in loss_dict.items()}, out


def read_losses(loss_dict: dict) -> dict:
    """Floats of a lazy loss dict in ONE device-to-host copy, with the reference's NaN guard (utils.py:219)."""
    keys = list(loss_dict)
    vals = torch.stack([torch.as_tensor(loss_dict[k]).reshape(()).float() for k in keys]).tolist()
    out = dict(zip(keys, vals))
    if "hoi_loss" in out and out["hoi_loss"] != out["hoi_loss"]:
        raise ValueError(f"The HOI loss is NaN")
    return out


class SkgAdamW(CachedFusedAdamW):
    """AdamW with the update of ALL parameters in one launch of `skg_adamw_f32` (include/skghoi.h) instead
    of torch's multi-tensor kernel (12 launches, 0.58 ms for the head's 29.6 M parameters; this one moves the same 28
    bytes per parameter in 0.13 ms = 6 TB/s).  Same state (`exp_avg`, `exp_avg_sq`, `step` tensors) and `state_dict` as
    `torch.optim.AdamW(fused=True)`, the same decoupled update rule evaluated in fp32 (results agree to rounding: 1e-6
    relative after ten steps, `tests/test_trainer.py`).  A chunk table (parameter / gradient / moment pointers per 16 Ki
    elements) is rebuilt and uploaded through pinned memory only when the gradients' addresses changed (the fused step hands
    out the same gradient arena from step to step); the state's `step` tensors are views of one flat buffer the launch bumps
    itself.  Anything outside the fast path (first step, a missing or non-contiguous gradient, parameters with
    different step counts, amsgrad, ...) takes the stock implementation."""

    CHUNK = 16384
    _DT = None

    def __init__(self, *args, **kw):
        super().__init__(*args, **kw)
        self._plans = {}

    def _plan(self, gi, c):
        """Static part of a group's chunk table; None if the group does not qualify."""
        import numpy as np
        pl = self._plans.get(gi)
        params, exp_avgs, exp_avg_sqs, steps = c[1], c[2], c[3], c[4]
        if pl is not None and pl["lists"] is c:
            if not pl["ok"]:
                return pl
            # .data re-pointed -> rebuild.  Parameters that live in a head's parameter ARENA (train_fused.Stacked marks them)
            # were checked against it, all 408, by this step's forward -- which re-adopts them if anything moved -- so here
            # only the arena's own identity is left to check (its first and last parameter); everything else is compared
            # pointer by pointer.
            free, probe, base = pl["free"], pl["probe"], pl["pbase"]
            if all(params[i].data_ptr() == base[i] for i in probe) and \
                    (not free or [params[i].data_ptr() for i in free] == [base[i] for i in free]) and \
                    all(getattr(params[i], "_skg_arena", None) is not None for i in probe):
                return pl
        ok = all(p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() for p in params) and \
            all(t.dtype == torch.float32 and t.is_contiguous() for t in exp_avgs + exp_avg_sqs) and \
            all(s_.dtype == torch.float32 and s_.is_cuda for s_ in steps) and \
            len({p.device for p in params}) == 1
        if not ok:
            self._plans[gi] = dict(lists=c, ok=False)
            return self._plans[gi]
        st = torch.stack([s.reshape(()) for s in steps]).tolist()           # one synchronisation when the plan is made
        if len(set(st)) != 1:
            self._plans[gi] = dict(lists=c, ok=False)
            return self._plans[gi]
        # the per-parameter `step` tensors become views of ONE flat tensor: counting a step is then one tiny kernel instead
        # of a multi-tensor launch over 408 pointers (same state_dict: 0-dim tensors with the same values)
        flat_step = torch.stack([s.reshape(()) for s in steps]).contiguous()
        views = list(flat_step.unbind(0))
        for p_, v_ in zip(params, views):
            self.state[p_]["step"] = v_
        steps[:] = views
        if SkgAdamW._DT is None:
            SkgAdamW._DT = np.dtype([("p", "u8"), ("g", "u8"), ("m", "u8"), ("v", "u8"), ("count", "u4"), ("res", "u4")])
        numel = np.array([p.numel() for p in params], np.int64)
        nch = (numel + self.CHUNK - 1) // self.CHUNK
        tix = np.repeat(np.arange(len(params)), nch)
        first = np.concatenate([[0], np.cumsum(nch)[:-1]])
        off = (np.arange(int(nch.sum())) - np.repeat(first, nch)) * self.CHUNK
        cnt = np.minimum(self.CHUNK, numel[tix] - off).astype(np.uint32)
        keep = cnt > 0
        tix, off, cnt = tix[keep], off[keep], cnt[keep]
        tab = np.zeros(len(tix), SkgAdamW._DT)
        byte = (4 * off).astype(np.uint64)
        tab["p"] = np.array([p.data_ptr() for p in params], np.uint64)[tix] + byte
        tab["m"] = np.array([t.data_ptr() for t in exp_avgs], np.uint64)[tix] + byte
        tab["v"] = np.array([t.data_ptr() for t in exp_avg_sqs], np.uint64)[tix] + byte
        tab["count"] = cnt
        dev = params[0].device
        nb = tab.nbytes
        managed = [i for i, p_ in enumerate(params) if getattr(p_, "_skg_arena", None) is not None]
        free = [i for i, p_ in enumerate(params) if getattr(p_, "_skg_arena", None) is None]
        arenas = {}
        for i in managed:                                   # first and last parameter of every arena
            a = arenas.setdefault(id(params[i]._skg_arena), [i, i])
            a[1] = i
        probe = sorted({i for a in arenas.values() for i in a})
        pl = dict(lists=c, ok=True, tab=tab, tix=tix, byte=byte, host_step=int(st[0]), dev=dev, numel=numel,
                  pbase=[p.data_ptr() for p in params], free=free, probe=probe,
                  flat_step=flat_step,
                  pinned=[torch.empty(nb, dtype=torch.uint8, pin_memory=True) for _ in range(2)],
                  events=[None, None], dtab=torch.empty(nb, dtype=torch.uint8, device=dev), flip=0)
        self._plans[gi] = pl
        return pl

    # ---- the update inside the backward (include/skghoi.h, skg_exchange.adamw)
    def backward_slices(self, st, ga, chunks):
        """For a fused step whose gradients land in the arena `ga` of the parameter arena `st` (train_fused.Stacked) and
        whose backward hands out the arena in `chunks` = [(stage, end in floats)]: the AdamW chunk table sorted by arena
        offset, the table range of every chunk and this step's factors -- what the backward's worker thread needs to
        update chunk i's parameters behind stage i (behind its all-reduce, data parallel) while the later stages still run.
        None when this step has to go through step(): no state yet, several groups, parameters outside the arena, options
        the one-launch kernel does not cover, step hooks.  The caller MUST call backward_done() once the backward has been
        submitted with the slices."""
        import numpy as np
        if len(self.param_groups) != 1 or getattr(self, "grad_scale", None) is not None or \
                getattr(self, "found_inf", None) is not None:
            return None
        group = self.param_groups[0]
        if group.get("amsgrad") or group.get("maximize") or group.get("capturable") or group.get("differentiable") \
                or not group.get("fused") or torch.is_tensor(group["lr"]) or not group["params"]:
            return None
        c = self._cached(0, group)
        pl = self._plan(0, c) if c is not None else None
        if pl is None or not pl["ok"] or pl["free"]:
            return None
        base, gbase = st.buf.data_ptr(), ga.data_ptr()
        key = (id(st), base, gbase, tuple(chunks))
        ov = pl.get("ov")
        if ov is None or ov["key"] != key:
            params, exp_avgs, exp_avg_sqs = c[1], c[2], c[3]
            if any(getattr(p_, "_skg_arena", None) is not st for p_ in params):
                return None
            off = np.array([(p_.data_ptr() - base) // 4 for p_ in params], np.int64)
            if off.min() < 0 or int((off + np.array([p_.numel() for p_ in params])).max()) > st.total:
                return None
            order = np.argsort(off, kind="stable")
            rows, first, k = [], [0], 0
            ends = [int(e) for _, e in chunks]
            for i in order:
                while k < len(ends) - 1 and off[i] >= ends[k]:
                    first.append(len(rows)); k += 1
                n, pp, mp, vp = params[i].numel(), params[i].data_ptr(), exp_avgs[i].data_ptr(), exp_avg_sqs[i].data_ptr()
                gp = gbase + 4 * int(off[i])
                for o in range(0, n, self.CHUNK):
                    rows.append((pp + 4 * o, gp + 4 * o, mp + 4 * o, vp + 4 * o, min(self.CHUNK, n - o), 0))
            while len(first) < len(ends):
                first.append(len(rows))
            first.append(len(rows))
            tab = np.array(rows, dtype=SkgAdamW._DT)
            dtab = torch.from_numpy(tab.view(np.uint8).copy()).to(pl["dev"])
            ov = pl["ov"] = dict(key=key, dtab=dtab, first=first, rows=len(rows))
        t = pl["host_step"] + 1
        beta1, beta2 = group["betas"]
        fs = pl["flat_step"]
        return dict(plan=pl, adamw=ov["dtab"].data_ptr(), first=ov["first"], steps=fs.data_ptr(), n_steps=fs.numel(),
                    lr=float(group["lr"]), beta1=float(beta1), beta2=float(beta2), eps=float(group["eps"]),
                    weight_decay=float(group["weight_decay"]), bias1=1.0 - beta1 ** t, bias2=1.0 - beta2 ** t)

    def backward_done(self, sl):
        """The backward that carried `sl` (backward_slices) has been submitted: this step's update is on its way."""
        sl["plan"]["host_step"] += 1
        self._opt_called = True                              # (lr_scheduler's "step() before optimizer.step()" check)

    @torch.no_grad()
    def step(self, closure=None):
        import numpy as np
        from . import _capi
        from .engine import _stream
        if closure is not None or getattr(self, "grad_scale", None) is not None or getattr(self, "found_inf", None) is not None:
            self._plans.clear()
            return super().step(closure)
        work = []
        for gi, group in enumerate(self.param_groups):
            c = None
            if not (group.get("amsgrad") or group.get("maximize") or group.get("capturable") or group.get("differentiable")
                    or not group.get("fused") or torch.is_tensor(group["lr"]) or not group["params"]):
                c = self._cached(gi, group)
            pl = self._plan(gi, c) if c is not None else None
            if pl is None or not pl["ok"]:
                self._plans.clear()                          # step counts move outside this class: re-read them next time
                return super().step()
            grads = [p.grad for p in c[1]]
            # the fused step hands out the SAME gradient views from step to step (one gradient arena, reused when nothing
            # holds it: train_fused.Stacked.grad_arena): same addresses as in the table already on the device -> nothing
            # to rebuild or upload.  Only the addresses are kept here: a reference to the view objects would count as a
            # holder and make the step allocate a fresh arena every time.
            try:
                ptrs = list(map(_data_ptr, grads))
            except TypeError:                                # a parameter without gradient
                self._plans.clear()
                return super().step()
            same = pl.get("grad_ptrs") == ptrs
            if not same and any(g is None or g.dtype != torch.float32 or not g.is_contiguous() or g.device != pl["dev"]
                                for g in grads):
                self._plans.clear()
                return super().step()
            work.append((group, c, pl, grads, same))
        lib = _capi.lib()
        for group, c, pl, grads, same in work:
            tab = pl["tab"]
            if not same:
                tab["g"] = np.fromiter((g.data_ptr() for g in grads), np.uint64, len(grads))[pl["tix"]] + pl["byte"]
                k = pl["flip"]; pl["flip"] = 1 - k
                if pl["events"][k] is not None:
                    pl["events"][k].synchronize()            # this staging buffer's previous upload (two steps ago)
                pin = pl["pinned"][k]
                pin.numpy()[:] = tab.view(np.uint8)
                pl["dtab"].copy_(pin, non_blocking=True)
                ev = pl["events"][k] or torch.cuda.Event()
                ev.record(); pl["events"][k] = ev
                pl["grad_ptrs"] = ptrs
            t = pl["host_step"] + 1
            beta1, beta2 = group["betas"]
            fs = pl["flat_step"]                             # the state's step tensors (one flat fp32 buffer): + 1 in the same launch
            _capi.check(lib.skg_adamw_f32(pl["dtab"].data_ptr(), len(tab), float(group["lr"]), float(beta1), float(beta2),
                                          float(group["eps"]), float(group["weight_decay"]), 1.0 - beta1 ** t,
                                          1.0 - beta2 ** t, fs.data_ptr(), fs.numel(), _stream()), "skg_adamw_f32")
            pl["host_step"] = t
        return None


# ---------------------------------------------------------------------------------------------------- data side
def seed_everything(seed: int = 42) -> None:
    """main:67 ("Fix random seed for model synchronisation": every rank builds identical initial weights)."""
    torch.manual_seed(seed)


def filter_detections(detection: dict, human_idx: int, box_score_thresh_h: float = 0.2,
                      box_score_thresh_o: float = 0.2) -> dict:
    """DataFactory.filter_detections (utils.py:97-119): humans above their threshold first, then the other classes above
    theirs, original order inside each group."""
    boxes = torch.as_tensor(detection["boxes"]); labels = torch.as_tensor(detection["labels"])
    scores = torch.as_tensor(detection["scores"])
    idx = torch.nonzero(labels == human_idx).squeeze(1)
    keep = idx[torch.nonzero(scores[idx] >= box_score_thresh_h).squeeze(1)]
    idx = torch.nonzero(labels != human_idx).squeeze(1)
    keep = torch.cat([keep, idx[torch.nonzero(scores[idx] >= box_score_thresh_o).squeeze(1)]])
    return dict(boxes=boxes[keep].view(-1, 4), labels=labels[keep].view(-1), scores=scores[keep].view(-1))


def horizontal_flip_boxes(w: float, boxes: torch.Tensor) -> torch.Tensor:
    """pocket.ops.horizontal_flip_boxes(w, boxes) for 'coords' boxes (x1, y1, x2, y2): x1' = w - x2, x2' = w - x1."""
    out = boxes.clone()
    out[:, 0] = w - boxes[:, 2]
    out[:, 2] = w - boxes[:, 0]
    return out


def hflip_sample(image_or_maps, detection: dict, target: dict, width: float):
    """The horizontal-flip augmentation of DataFactory.__getitem__ (utils.py:85-86, 140-143): the image -- or, when
    training from cached feature maps, every map [.., H, W] -- is mirrored along its last axis, and the detection and
    ground-truth boxes with it.  Returns new objects (inputs are left untouched)."""
    if torch.is_tensor(image_or_maps):
        flipped = image_or_maps.flip(-1)
    else:
        flipped = type(image_or_maps)((k, v.flip(-1)) for k, v in image_or_maps.items())
    det = dict(detection, boxes=horizontal_flip_boxes(width, detection["boxes"]))
    tgt = dict(target, boxes_h=horizontal_flip_boxes(width, target["boxes_h"]),
               boxes_o=horizontal_flip_boxes(width, target["boxes_o"]))
    return flipped, det, tgt


def draw_flips(n: int, flip: bool = True) -> torch.Tensor:
    """DataFactory._flip (utils.py:85-86): one coin per sample, drawn once from the global CPU generator."""
    return torch.randint(0, 2, (n,)) if flip else torch.zeros(n)


def custom_collate(batch):
    """utils.py:34-42."""
    images, detections, targets = [], [], []
    for im, det, tar in batch:
        images.append(im); detections.append(det); targets.append(tar)
    return images, detections, targets


def make_loader(dataset, batch_size: int = 4, num_workers: int = 0, world_size: int = 1, rank: int = 0, shuffle=True):
    """main:46-54: DataLoader over a DistributedSampler (shards the sample indices over the ranks, reshuffled per epoch
    by sampler.set_epoch)."""
    from torch.utils.data import DataLoader
    from torch.utils.data.distributed import DistributedSampler
    sampler = DistributedSampler(dataset, num_replicas=world_size, rank=rank, shuffle=shuffle)
    return DataLoader(dataset=dataset, collate_fn=custom_collate, batch_size=batch_size, num_workers=num_workers,
                      pin_memory=True, sampler=sampler)


def relocate_to_device(x, device):
    """pocket.ops.relocate_to_cuda for any device: tensors inside lists / tuples / dicts are moved (non-blocking), everything
    else -- image sizes, strings -- is returned as it is."""
    if torch.is_tensor(x):
        return x.to(device, non_blocking=True)
    if isinstance(x, dict):
        return type(x)((k, relocate_to_device(v, device)) for k, v in x.items())
    if isinstance(x, (list, tuple)):
        y = [relocate_to_device(v, device) for v in x]
        return y if isinstance(x, list) else (tuple(y) if type(x) is tuple else type(x)(*y))
    return x


def _with_lookahead(loader, enabled=True, depth=1):
    """(batch, next batch or None) pairs -- with depth=2 (batch, next, the one after) triples; the loader is advanced `depth`
    batches ahead of the step (nothing else changes: same batches, same order)."""
    it = iter(loader)
    if not enabled:
        for cur in it:
            yield (cur, None) if depth == 1 else (cur, None, None)
        return
    window = []
    for b in it:
        window.append(b)
        if len(window) == depth + 1:
            yield tuple(window)
            window.pop(0)
    while window:
        yield tuple(window) + (None,) * (depth + 1 - len(window))
        window.pop(0)


@torch.no_grad()
def test(net, test_loader, evaluator, device=None, lookahead=True):
    """utils.py:148-198 (`test`): eval mode, ONE image per forward (`assert len(output) == 1`), every result associated
    with the image's ground-truth pairs and logged into the 600-class 11-point meter; returns `evaluator.summary()`
    (`ap` per class, `full`, and `rare` / `non_rare` when the evaluator knows the training counts).

    test_loader yields batches whose LAST element is the list of targets ({boxes_h, boxes_o, hoi}) and whose other
    elements are the network's arguments -- for the bare interaction head (inference from cached features / detections):
    (features, detections, image_shapes, targets).  evaluator: evaluate.DeviceHOIEvaluator (results stay on the device) or
    evaluate.HOIEvaluator (the host restatement).

    lookahead: the loop runs one image ahead of the network -- image i + 1 is uploaded BEFORE forward i is enqueued and its
    detection selection, count read-back and TransH table draw run beside forward i on the head's side stream
    (InteractionHead.prefetch_eval), so forward i + 1 begins at its launch plan.  Same results, same RNG stream."""
    from .evaluate import DeviceHOIEvaluator
    mod = net.module if isinstance(net, nn.parallel.DistributedDataParallel) else net
    if device is None:
        p = next(mod.parameters(), None)
        device = p.device if p is not None else torch.device("cpu")
    device = torch.device(device)
    net.eval()
    ahead = getattr(mod, "prefetch_eval", None) if (lookahead and device.type == "cuda") else None
    batched = isinstance(evaluator, DeviceHOIEvaluator)
    it = iter(test_loader)
    cur = next(it, None)
    inputs = None if cur is None else relocate_to_device(cur[:-1], device)
    while cur is not None:
        nxt = next(it, None)
        nxt_inputs, uploaded = None, None
        if nxt is not None:
            nxt_inputs = relocate_to_device(nxt[:-1], device)
            if ahead is not None:
                from .engine import current_stream_of
                uploaded = torch.cuda.Event()
                uploaded.record(current_stream_of(device))           # behind image i + 1's uploads, in front of forward i
        output = net(*inputs)
        if ahead is not None and nxt_inputs is not None and len(nxt_inputs) >= 2 and isinstance(nxt_inputs[1], list) \
                and len(nxt_inputs[1]) == 1 and isinstance(nxt_inputs[1][0], dict) \
                and torch.is_tensor(nxt_inputs[1][0].get("boxes")) and nxt_inputs[1][0]["boxes"].is_cuda:
            ahead(nxt_inputs[1], after=uploaded)
        if output is not None:
            assert len(output) == 1, "Batch size is not 1"           # utils.py:166-167
            target = cur[-1][0]
            if batched:
                evaluator.add(output, [target])
            else:
                # (blocking copies: pocket.ops.relocate_to_cpu, utils.py:169 -- the host evaluator reads them at once)
                evaluator.add({k: (v.cpu() if torch.is_tensor(v) else v) for k, v in output[0].items()}, target)
        cur, inputs = nxt, nxt_inputs
    return evaluator.summary()


# ---------------------------------------------------------------------------------------------------- engine
class Trainer:
    """The training loop of the reference's CustomisedDLE / pocket DistributedLearningEngine, reduced to what the head's
    training needs (utils.py:200-229, main:85-145): epochs over a (distributed) loader, the step of `train_step`,
    LambdaLR stepped once per epoch, the NaN guard, and checkpoints carrying the reference's keys

        {"iteration", "epoch", "model_state_dict", "optim_state_dict", "scheduler_state_dict"}

    so that a checkpoint written by either side resumes on the other (main:85-93; the model keys are those of the
    wrapped net without DDP's "module." prefix, as pocket saves `net.module.state_dict()`).

    `step_fn(net, optimizer, batch) -> (loss_dict, results)` adapts the loader's batch to the net's call; the default
    expects (features, detections, image_shapes, targets) batches and calls `train_step`."""

    def __init__(self, net, optimizer, scheduler=None, train_loader=None, rank=0, cache_dir=None, step_fn=None,
                 print_interval=0, lazy_losses=False, val_loader=None, num_classes=117, train_meter=None, device=None):
        self.net, self.optimizer, self.scheduler = net, optimizer, scheduler
        self.train_loader = train_loader
        self.val_loader = val_loader
        self.num_classes = num_classes
        self.rank = rank
        self.cache_dir = cache_dir
        self.print_interval = print_interval
        self.epoch = 0
        self.iteration = 0
        # lazy_losses: losses stay on the device between print intervals (train_step(lazy=True)); `history` then holds
        # device tensors until the end of the epoch, where they are read back at once -- and the NaN guard fires there
        self.lazy_losses = lazy_losses
        # the default step takes a one-batch look-ahead and lets the head prepare it while this step runs on the GPU
        # (prefetch_batch: only for batches of the head's own call shape; any other batch shape is simply not prepared)
        self.lookahead = step_fn is None
        self.step_fn = step_fn or (lambda n, o, b, nxt=None, nxt2=None: train_step(
            n, o, *b[:-1], targets=b[-1], lazy=lazy_losses, prefetch=nxt, prefetch2=nxt2))
        self.history = []
        # the training-mAP meter of the reference's engine (utils.py:208, 229): on when there is a validation loader (the
        # end-of-epoch report needs it) unless switched explicitly
        self.device = device
        self.train_meter = (val_loader is not None) if train_meter is None else bool(train_meter)
        self.meter = None
        self.last_report = None

    # -- meters and validation (utils.py:208, 229-299)
    def _meter_device(self):
        if self.device is not None:
            return torch.device(self.device)
        p = next(self._module().parameters(), None)
        return p.device if p is not None else torch.device("cpu")

    def _new_meter(self):
        from .evaluate import DeviceAPMeter
        return DeviceAPMeter(self.num_classes, device=self._meter_device())

    def log_results(self, results, meter):
        """utils.py:263-282 (_synchronise_and_log_results) without its per-iteration host copy and all_gather: the batch's
        (scores, prediction, labels) stay on the device in `meter`; the ranks' logs meet once, in meter.eval()."""
        if results:
            meter.append_results([r for r in results if isinstance(r, dict)])

    @torch.no_grad()
    def validate(self):
        """utils.py:283-299: eval mode over the (sharded) validation loader -- batches carry their targets, so the head takes
        its eval-with-targets pass and the results carry `labels` -- into a fresh 117-class 11-point meter; returns the
        per-class APs (every rank holds them; the reference returns them on rank 0 only)."""
        meter = self._new_meter()
        dev = self._meter_device()
        self.net.eval()
        # one-batch look-ahead, as in training: while the GPU runs batch i's forward the head prepares batch i + 1 on its side
        # stream (selection, pairs, label association, the reference's host RNG draws: InteractionHead.prefetch_train in eval
        # mode); only for batches of the head's own call shape -- any other batch is simply not prepared
        it = iter(self.val_loader)
        cur = next(it, None)
        cur = None if cur is None else relocate_to_device(cur, dev)
        while cur is not None:
            results = self.net(*cur)
            nxt = next(it, None)
            ahead = None
            if nxt is not None:
                nxt = relocate_to_device(nxt, dev)
                if self.lookahead and isinstance(nxt, (list, tuple)) and len(nxt) == 4:
                    ahead = prefetch_batch(self.net, *nxt)
            self.log_results(results, meter)
            if ahead is not None:
                # the whole preparation NOW, while the GPU runs this batch's forward (there is no backward or optimizer to put
                # between its steps here): count read-backs, association, the reference's RNG draws, uploads -- the next
                # forward then starts at its launch plan instead of waiting ~0.3 ms for them with an idle GPU
                ahead.finish()
            cur = nxt
        return meter.eval()

    def on_end_epoch(self):
        """utils.py:232-249: training mAP off the epoch's meter, validation mAP, one report line on rank 0, meter reset."""
        import time
        t0 = time.perf_counter()
        ap_train = self.meter.eval() if self.meter is not None else None
        t1 = time.perf_counter()
        ap_val = self.validate() if self.val_loader is not None else None
        t2 = time.perf_counter()
        self.last_report = dict(epoch=self.epoch, training_map=(None if ap_train is None else float(ap_train.mean())),
                                validation_map=(None if ap_val is None else float(ap_val.mean())),
                                evaluation_time_s=t1 - t0, total_time_s=t2 - t0)
        if self.rank == 0 and (ap_train is not None or ap_val is not None):
            fmt = lambda v: "n/a" if v is None else "%.4f" % v
            print("Epoch: {} | training mAP: {}, evaluation time: {:.2f}s |validation mAP: {}, total time: {:.2f}s\n".format(
                self.epoch, fmt(self.last_report["training_map"]), t1 - t0, fmt(self.last_report["validation_map"]),
                t2 - t0))
        if self.meter is not None:
            self.meter.reset()
        return self.last_report

    # -- checkpoints (main:85-93, pocket engines' save_checkpoint)
    def _module(self):
        return self.net.module if isinstance(self.net, nn.parallel.DistributedDataParallel) else self.net

    def state(self) -> dict:
        return dict(iteration=self.iteration, epoch=self.epoch, model_state_dict=self._module().state_dict(),
                    optim_state_dict=self.optimizer.state_dict(),
                    scheduler_state_dict=(self.scheduler.state_dict() if self.scheduler is not None else None))

    def save_checkpoint(self, path=None) -> str:
        if path is None:
            import os
            os.makedirs(self.cache_dir, exist_ok=True)
            path = os.path.join(self.cache_dir, "ckpt_{:05d}_{:02d}.pt".format(self.iteration, self.epoch))
        torch.save(self.state(), path)
        return path

    def load_checkpoint(self, path_or_dict, map_location="cpu") -> None:
        ckpt = torch.load(path_or_dict, map_location=map_location) if isinstance(path_or_dict, str) else path_or_dict
        self._module().load_state_dict(ckpt["model_state_dict"])
        if ckpt.get("optim_state_dict") is not None:
            self.optimizer.load_state_dict(ckpt["optim_state_dict"])
        if self.scheduler is not None and ckpt.get("scheduler_state_dict") is not None:
            self.scheduler.load_state_dict(ckpt["scheduler_state_dict"])
        self.epoch = int(ckpt.get("epoch", 0))
        self.iteration = int(ckpt.get("iteration", 0))

    # -- loop
    def train_epoch(self) -> None:
        sampler = getattr(self.train_loader, "sampler", None)
        if hasattr(sampler, "set_epoch"):
            sampler.set_epoch(self.epoch)                   # pocket: reshuffle the shards every epoch
        self.net.train()
        if self.train_meter and self.meter is None:
            self.meter = self._new_meter()
        # the default step looks TWO batches ahead (train_step: prefetch / prefetch2): batch i + 2's preparation starts
        # during step i and ends during step i + 1, off the step boundary
        for batch, nxt, nxt2 in _with_lookahead(self.train_loader, self.lookahead, depth=2):
            losses, results = self.step_fn(self.net, self.optimizer, batch, nxt, nxt2) if self.lookahead else \
                self.step_fn(self.net, self.optimizer, batch)
            if self.meter is not None:
                self.log_results(results, self.meter)
            self.iteration += 1
            self.history.append(losses)
            if self.print_interval and self.iteration % self.print_interval == 0:
                if self.lazy_losses:
                    losses = self.history[-1] = read_losses(losses)          # every rank: the NaN guard is collective
                if self.rank == 0:
                    print("Epoch %d iteration %d: %s" % (self.epoch, self.iteration,
                                                         ", ".join("%s %.4f" % kv for kv in losses.items())))
        if self.lazy_losses:
            self.history = [h if all(isinstance(v, float) for v in h.values()) else read_losses(h) for h in self.history]
        if self.meter is not None or self.val_loader is not None:
            self.on_end_epoch()
        self.epoch += 1
        if self.scheduler is not None:
            self.scheduler.step()

    def __call__(self, num_epochs: int) -> None:
        while self.epoch < num_epochs:
            self.train_epoch()
            if self.cache_dir is not None and self.rank == 0:
                self.save_checkpoint()
