"""CPU tests of what the captured-plan path (skghoi_amd/small.py) relies on to stay alive on the GPU box -- the invariants
behind the two round-2 crash records: a capture runs with the cyclic GC off and restores it, dropped
plans are destroyed on an idle device BEFORE anything of the new capture exists, a plan keeps the events its capture
recorded -- plus the bucket ladder and the single-image layout.  torch.cuda is replaced by recording fakes: no GPU."""
import gc
import types

import numpy as np
import pytest
import torch

from skghoi_amd import layout, small


def test_capacity_ladder():
    for limit in (15, 30, 40, 80):
        caps = [small.capacity(v, limit) for v in range(1, limit + 1)]
        assert all(c >= v for v, c in zip(range(1, limit + 1), caps)) and max(caps) == limit
        assert caps == sorted(caps)
        assert all(c <= 1.5 * v or c - v <= 1 for v, c in zip(range(1, limit + 1), caps))     # padded rows: <= 1.5x per axis
    # the default caps (15 humans, 30 nodes): a few dozen plans cover every single-image shape
    buckets = {(small.capacity(h, 15), small.capacity(n, 30)) for h in range(1, 16) for n in range(max(h, 2), h + 16)}
    assert len(buckets) <= 48
    assert small.capacity(35, 30) == 35                     # a count above the limit is never rounded down


def test_single_layout_matches_general_builder():
    a = layout.single(3, 7, 55, (480, 640))
    b = layout.build([3], [7], [55], [(480, 640)], 49)
    for k in ("n_active", "n_visit", "sum_all", "sum_n", "sum_h", "sum_g", "sum_p", "sum_l"):
        assert getattr(a, k) == getattr(b, k), k
    for k in ("skipped", "pairs_per_image", "cells_per_image", "active"):
        assert np.array_equal(getattr(a, k), getattr(b, k)), k
    for f in ("n_h", "n", "img_h", "img_w", "grid_off", "pair_off", "hum_off", "node_off", "out_off"):
        assert a.meta[f][0] == b.meta[f][0], f


class _FakeGraphCtx:
    def __init__(self, log, fail=False):
        self.log, self.fail = log, fail

    def __enter__(self):
        self.log.append(("graph_enter", gc.isenabled()))
        return self

    def __exit__(self, *exc):
        self.log.append(("graph_exit", gc.isenabled()))
        return False


@pytest.fixture
def fake_cuda(monkeypatch):
    log = []
    monkeypatch.setattr(torch.cuda, "synchronize", lambda *a, **k: log.append(("device_synchronize", None)))
    monkeypatch.setattr(torch.cuda, "graph_pool_handle", lambda: ("pool",))
    monkeypatch.setattr(torch.cuda, "CUDAGraph", lambda: types.SimpleNamespace(replay=lambda: None))
    monkeypatch.setattr(torch.cuda, "current_stream",
                        lambda *a, **k: types.SimpleNamespace(synchronize=lambda: log.append(("stream_synchronize", None))))
    monkeypatch.setattr(torch.cuda, "graph", lambda g, **kw: _FakeGraphCtx(log))
    return log


def _runner(log, body_fail=False):
    eng = types.SimpleNamespace(plan_epoch=0)
    r = small.SmallBatchRunner(eng, max_plans=2)
    calls = {"n": 0}

    def body(p):
        calls["n"] += 1
        log.append(("body", gc.isenabled()))
        if body_fail and calls["n"] == 2:
            raise RuntimeError("kernel launch failed inside the capture")
        return dict(logits=None, _events=("fork", "s_ready", "g1_ready", "g_done"))
    r._body = body
    return r


def test_capture_holds_gc_off_buries_retired_plans_first_and_keeps_its_events(fake_cuda):
    log = fake_cuda
    r = _runner(log)
    r.retired = [object(), object()]                         # plans dropped earlier (eviction / weight change)
    assert gc.isenabled()
    p = small._Plan()
    r._capture(p)
    kinds = [k for k, _ in log]
    # (1) the dropped plans go first, on an idle device, before any body (= before any object of the new capture exists)
    assert kinds[0] == "device_synchronize" and r.retired == []
    assert kinds.index("device_synchronize") < kinds.index("body")
    # (2) eager pass with GC as the caller had it, then the capture with the cyclic GC OFF from before the capture begins
    #     until after it ends (the abort of r2h_*.log: a collection inside the capture freed pinned buffers / events)
    bodies = [v for k, v in log if k == "body"]
    assert bodies == [True, False]
    assert dict(log)["graph_enter"] is False and dict(log)["graph_exit"] is False
    assert gc.isenabled()                                    # restored
    # (3) the plan keeps the four events its capture recorded (the segfault of r2e / r2f: hipGraphLaunch walked an event
    #     that had been destroyed when the body's locals died)
    assert len(p.out["_events"]) == 4 and p.graph is not None and r.captures == 1


def test_capture_restores_gc_when_the_body_raises(fake_cuda):
    r = _runner(fake_cuda, body_fail=True)
    with pytest.raises(RuntimeError):
        r._capture(small._Plan())
    assert gc.isenabled()
    gc.disable()
    try:                                                     # a caller that runs with GC off keeps it off
        r2 = _runner(fake_cuda)
        r2._capture(small._Plan())
        assert not gc.isenabled()
    finally:
        gc.enable()


def test_close_and_eviction_route_plans_through_the_idle_teardown(fake_cuda):
    log = fake_cuda
    r = _runner(log)
    r.plans["a"] = small._Plan(); r.plans["b"] = small._Plan()
    r.close()
    assert not r.plans and not r.retired and ("device_synchronize", None) in log
    st = r.stats()
    assert st["plans"] == 0 and st["hit_rate"] is None


def test_a_graph_is_destroyed_only_by_reap_never_by_its_plans_death(fake_cuda, monkeypatch):
    """Round 5 (gpurun_out/r5y: segfault in hipGraphLaunch right behind a fresh capture, in a full test run): plans die whenever
    Python says so -- reference count, cyclic GC (engine <-> runner is a cycle) -- and must not take their hipGraphExec with
    them at that moment.  The graph and the capture's events live in small._KEPT; a dying or retired plan marks its entry;
    small.reap() destroys the marked ones behind a device synchronisation, and the next capture starts with it."""
    import weakref
    log = fake_cuda

    class Graph:
        def replay(self):
            pass

    monkeypatch.setattr(torch.cuda, "CUDAGraph", Graph)
    small.reap()
    r = _runner(log)
    p = small._Plan()
    r._capture(p)
    g = weakref.ref(p.graph)
    tok = p.token
    assert tok in small._KEPT and small._KEPT[tok][1] == ("fork", "s_ready", "g1_ready", "g_done")
    # (1) the plan dies by reference count: the graph object survives, its entry is marked
    del p
    gc.collect()
    assert g() is not None and tok in small._DEAD
    # (2) reap: synchronise, then destroy
    del log[:]
    assert small.reap() == 1
    assert log[0][0] == "device_synchronize" and g() is None and tok not in small._KEPT and not small._DEAD
    # (3) a retired plan that somebody still holds (engine.last does) keeps no graph once the runner has buried it, and the
    #     next capture reaps whatever died in the meantime before its own body runs
    p2 = small._Plan(); r._capture(p2)
    g2 = weakref.ref(p2.graph)
    r.plans["k"] = p2
    r.close()
    assert p2.graph is None and g2() is None
    p3 = small._Plan(); r._capture(p3)
    g3 = weakref.ref(p3.graph)
    tok3 = p3.token
    del p3
    gc.collect()
    assert g3() is not None
    del log[:]
    r._capture(small._Plan())
    kinds = [k for k, _ in log]
    assert g3() is None and tok3 not in small._KEPT
    assert kinds.index("device_synchronize") < kinds.index("body")
    small.reap()
