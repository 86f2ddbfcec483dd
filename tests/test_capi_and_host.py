"""CPU tests of the C-ABI boundary and the host logic (no GPU compute):
  * libskghoi_hip.so builds/loads and exports every symbol include/skghoi.h declares, with the ctypes mirror in sync
  * argument validation returns SKG_E_* instead of launching
  * batch layout (offsets, Q9 enc offset, zip truncation), TransH RNG replication, module surface / state_dict keys."""
import ctypes
import os
import re
import sys

import numpy as np
import pytest
import torch

import cases
import helpers
from skghoi_amd import _capi, layout, synth, transh

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    if not os.path.isfile(_capi.LIB_PATH):
        sys.path.insert(0, ROOT)
        import __graft_entry__
        __graft_entry__.build()
    return _capi.lib()


def test_every_declared_symbol_is_exported(lib):
    hdr = open(os.path.join(ROOT, "include", "skghoi.h")).read()
    declared = set(re.findall(r"\b(skg_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(_capi.PROTOTYPES), declared ^ set(_capi.PROTOTYPES)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.skg_abi_version() == _capi.ABI_VERSION
    assert b"gfx950" in lib.skg_build_info()


def test_struct_mirrors_match_header():
    assert ctypes.sizeof(_capi.GemmDesc) == 224
    assert layout.META_DTYPE.itemsize == 48
    hdr = open(os.path.join(ROOT, "include", "skghoi.h")).read()
    body = hdr[hdr.index("typedef struct {\n    int32_t image"):hdr.index("} skg_image_meta;")]
    names = re.findall(r"(?:int32_t|float)\s+([a-z_, ]+);", body)
    flat = [n.strip() for group in names for n in group.split(",")]
    assert flat == [f[0] for f in _capi.META_FIELDS]


def _header_struct_fields(hdr, name):
    """[(field, kind)] of `typedef struct { ... } name;` in include/skghoi.h, kind in {ptr, i64, i32, f32}."""
    end = hdr.index("} %s;" % name)
    body = hdr[hdr.rindex("typedef struct {", 0, end):end]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    out = []
    for stmt in body.split(";"):
        stmt = stmt.replace("typedef struct {", "").strip()
        m = re.match(r"(?:const\s+)?(void|float|int32_t|int64_t|uint16_t|uint32_t)\s*(\*?)\s*(.+)$", stmt, flags=re.S)
        if not m:
            continue
        base, star, rest = m.groups()
        for nm in rest.split(","):
            nm = nm.strip()
            ptr = bool(star) or nm.startswith("*")
            kind = "ptr" if ptr else {"float": "f32", "int32_t": "i32", "int64_t": "i64"}[base]
            out.append((nm.lstrip("* "), kind))
    return out


_CT_KIND = {ctypes.c_void_p: "ptr", ctypes.c_int64: "i64", ctypes.c_int32: "i32", ctypes.c_float: "f32"}


def test_gemm_desc_mirrors_match_header_field_by_field():
    """skg_gemm_desc in the header == _capi.GemmDesc == the stub INTEGRATION.md publishes (names, order, kinds): a
    mirror that stops short makes the library read past the caller's struct (round-2 finding: the document lacked a_exp)."""
    hdr = open(os.path.join(ROOT, "include", "skghoi.h")).read()
    want = _header_struct_fields(hdr, "skg_gemm_desc")
    assert want[0] == ("A", "ptr") and want[-1] == ("a_exp", "ptr") and len(want) == 31
    assert [(n, _CT_KIND[t]) for n, t in _capi.GemmDesc._fields_] == want
    wantx = _header_struct_fields(hdr, "skg_gemmx_desc")
    assert [(n, _CT_KIND[t]) for n, t in _capi.GemmXDesc._fields_] == wantx
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    stub = doc[doc.index("class GemmDesc(C.Structure)"):]
    stub = stub[:stub.index("]\n") + 1]
    fields = re.findall(r'\("(\w+)",\s*C\.(c_\w+)\)', stub)
    kinds = {"c_void_p": "ptr", "c_int64": "i64", "c_int32": "i32", "c_float": "f32"}
    assert [(n, kinds[t]) for n, t in fields] == want
    assert "C.sizeof(GemmDesc) == %d" % ctypes.sizeof(_capi.GemmDesc) in doc


def test_argument_validation_without_gpu(lib):
    d = _capi.GemmDesc()
    assert lib.skg_gemm_f32(None, None) == -1
    d.M, d.N, d.K = 4, 4, 6
    d.A = 16; d.W = 16; d.C = 16; d.lda = 8; d.ldw = 8; d.ldc = 4
    assert lib.skg_gemm_f32(ctypes.byref(d), None) == -2           # K % 4 != 0
    d.K = 8; d.A = 20
    assert lib.skg_gemm_f32(ctypes.byref(d), None) == -2           # misaligned A
    d.A = 16; d.epilogue = 9
    assert lib.skg_gemm_f32(ctypes.byref(d), None) == -1
    d.epilogue = _capi.EPI_BIAS; d.M = 0
    assert lib.skg_gemm_f32(ctypes.byref(d), None) == 0            # empty problem: nothing launched
    assert lib.skg_preprocess_f32(16, 16, 16, 16, 2, 49, 0.2, 0.5, 100, 100, 16, 80, 2.8, 16, 16, None) == -3
    assert lib.skg_global_avgpool_f32(None, 0, 256, 10, None, None) == 0
    assert lib.skg_layernorm_f32(16, 1024, 16, 16, 3, 2048, 1e-5, 16, 1024, None) == -1
    assert lib.skg_layernorm2_f32(16, 1024, 16, 16, 3, 16, 1024, 16, 1024, 16, 16, 2, 16, 1024, 2048, 1e-5, None) == -1
    assert lib.skg_layernorm2_f32(16, 1024, 16, 16, -1, 16, 1024, 16, 1024, 16, 16, 2, 16, 1024, 1024, 1e-5, None) == -1
    assert lib.skg_layernorm2_f32(None, 1024, 16, 16, 0, 16, 1024, None, 1024, 16, 16, 0, 16, 1024, 1024, 1e-5, None) == 0
    # fp16x2 weight twins: size query is pure host code; the scale must be a positive power of two
    assert lib.skg_split_weights_bytes(1024, 1024) == 32 * 64 * 2048
    assert lib.skg_split_weights_bytes(118, 2048) == 4 * 128 * 2048
    assert lib.skg_split_weights_f16x2(16, 32, 32, 32, 3.0, 16, None) == -1
    assert lib.skg_split_weights_f16x2(16, 32, 32, 16, 4.0, 16, None) == -1     # ldw < K
    assert lib.skg_split_weights_f16x2(16, 0, 32, 32, 4.0, 16, None) == 0       # nothing to do
    # TransH draw (host function): foreign generator-state sizes and inconsistent counters are rejected
    blob = (ctypes.c_uint8 * 5056)()
    ent = (ctypes.c_float * 4000)()
    assert lib.skg_transh_draw_f32(blob, 5000, 1, 117, 0, 1, ent, None, None) == -1
    assert lib.skg_transh_draw_f32(blob, 5056, 1, 117, 0, 1, ent, None, None) == -1      # left_ = 0 is no valid state
    assert lib.skg_transh_draw_f32(blob, 5056, 1, 117, 1, 1, ent, None, None) == -1      # relations without buffers


def test_layout_offsets_and_quirks():
    shapes = [(800, 1200)] * 5
    lay = layout.build([0, 2, 1, 3, 0], [3, 4, 1, 4, 0], [0, 10, 0, 7, 0], shapes, 49)
    assert lay.active.tolist() == [1, 3] and lay.n_visit == 5
    m = lay.meta
    assert m["box_off"].tolist() == [3, 8]
    assert m["enc_off"].tolist() == [0, 4]            # Q9: skipped images do not advance the encoding offset
    assert m["node_off"].tolist() == [0, 4] and m["hum_off"].tolist() == [0, 2]
    assert m["grid_off"].tolist() == [0, 8] and m["pair_off"].tolist() == [0, 6] and m["out_off"].tolist() == [0, 10]
    assert (lay.sum_n, lay.sum_h, lay.sum_g, lay.sum_p, lay.sum_l) == (8, 5, 20, 15, 17)
    assert lay.hum_enc_row.tolist() == [0, 1, 4, 5, 6] and lay.node_ent_row.tolist() == [0, 1, 2, 3, 0, 1, 2, 3]
    fixed = layout.build([0, 2, 1, 3, 0], [3, 4, 1, 4, 0], None, shapes, 49, faithful_skip_offset=False)
    assert fixed.meta["enc_off"].tolist() == [3, 8]
    trunc = layout.build([1, 1, 1], [2, 0, 0], None, shapes[:3], 49)       # sum N = 2 < B = 3 (HEAD:822 zip)
    assert trunc.n_visit == 2
    buf, offs = layout.pack_int_arrays(lay)
    o, l = offs["meta"]
    assert l == 2 * 12 and all(v[0] % 4 == 0 for v in offs.values())
    with pytest.raises(IndexError):
        layout.build([1], [81], None, shapes[:1], 49)                      # TransH ent_tot = 80


@pytest.mark.parametrize("K", [117, 24])
def test_transh_rng_matches_reference_draw_order(K):
    """The tables the reference drew (captured in the fixtures) are reproduced from the seed."""
    name = "tiny" if K == 117 else "vcoco"
    g = helpers.load_golden(name)
    case = cases.build_case(name)
    torch.manual_seed(case["rng_seed"])
    ent, rel, nrm = transh.draw_batch(K, int(g["n_tables"]), need_relations=True)
    for i in range(int(g["n_tables"])):
        assert np.array_equal(ent[i].numpy(), g["timg%d.ent" % i])
        assert np.array_equal(rel[i].numpy(), g["timg%d.rel_table" % i])
        assert np.array_equal(nrm[i].numpy(), g["timg%d.norm_table" % i])
    # skipping the relation tables must still advance the RNG identically
    torch.manual_seed(case["rng_seed"])
    ent2, _, _ = transh.draw_batch(K, int(g["n_tables"]), need_relations=False)
    assert torch.equal(ent, ent2)


@pytest.mark.parametrize("K,need", [(117, True), (117, False), (24, True), (5, False)])
def test_transh_native_draw_equals_torch_calls(K, need):
    """skg_transh_draw_f32 (host C++) against the same draws made through torch: tables and generator state, from
    generator positions inside and across Mersenne-Twister blocks, incl. a freshly seeded generator."""
    assert transh.native_path() in (0, 1), "the native draw must be the path in use with this PyTorch build"
    fused = transh.native_path()
    for pre in (0, 1, 623, 624, 1000, 31431):
        torch.manual_seed(77 + pre)
        if pre:
            torch.empty(pre).uniform_()
        start = torch.get_rng_state()
        got = transh.draw_batch(K, 3, need_relations=need)
        after_native = torch.get_rng_state()
        torch.set_rng_state(start)
        transh._NATIVE = False
        try:
            want = transh.draw_batch(K, 3, need_relations=need)
        finally:
            transh._NATIVE = fused
        assert torch.equal(after_native, torch.get_rng_state())
        assert torch.equal(got[0], want[0])
        if need:
            assert torch.equal(got[1], want[1]) and torch.equal(got[2], want[2])
    # the reference's own construction (normal_ inits, then xavier) consumes the generator identically
    torch.manual_seed(3)
    e1, r1, n1 = transh.draw_tables(K, True)
    s1 = torch.get_rng_state()
    torch.manual_seed(3)
    emb = [torch.nn.Embedding(transh.TRANSH_ENT, transh.TRANSH_DIM), torch.nn.Embedding(K, transh.TRANSH_DIM),
           torch.nn.Embedding(K, transh.TRANSH_DIM)]
    for m in emb:
        torch.nn.init.xavier_uniform_(m.weight.data)
    assert torch.equal(s1, torch.get_rng_state())
    assert torch.equal(e1, emb[0].weight.data) and torch.equal(r1, emb[1].weight.data) and torch.equal(n1, emb[2].weight.data)


def test_module_surface_and_state_dict_keys():
    from skghoi_amd import GraphHead, InteractionHead
    o2v = synth.hico_object_to_verb()
    gh = GraphHead(out_channels=8, roi_pool_size=2, node_encoding_size=1024, representation_size=1024, num_cls=117,
                   human_idx=49, object_class_to_target_class=o2v, fg_iou_thresh=0.5, num_iter=2)
    head = InteractionHead(box_roi_pool=torch.nn.Identity(), box_pair_head=gh,
                           box_pair_suppressor=torch.nn.Linear(2048, 1), box_pair_predictor=torch.nn.Linear(2048, 117),
                           num_classes=117, human_idx=49, box_nms_thresh=0.5, box_score_thresh=0.2, max_human=15,
                           max_object=15, distributed=False)
    sd = head.state_dict()
    want = synth.head_param_shapes(117, 8, 2)
    assert len(sd) == 408 == len(want)
    assert [(k, tuple(v.shape)) for k, v in sd.items()] == [(k, tuple(s)) for k, s in want]
    assert sum(v.numel() for v in sd.values()) == sum(int(np.prod(s)) for _, s in want)
    head.load_state_dict(synth.make_state_dict(117, 8, 2, seed=3))
    # CPU tensors must fail loudly, never fall back
    det = [dict(boxes=torch.zeros(2, 4), labels=torch.tensor([49, 1]), scores=torch.tensor([0.9, 0.8]))]
    head.eval()
    with pytest.raises(_capi.SkgError):
        head({"3": torch.zeros(1, 256, 2, 2)}, det, [(10, 10)])
    head.train()
    with pytest.raises(AssertionError):
        head({"3": torch.zeros(1, 256, 2, 2)}, det, [(10, 10)])
    with pytest.raises(ValueError):
        GraphHead(8, 2, 512, 1024, 117, 49, o2v)


@pytest.mark.reference
def test_state_dict_matches_reference_module():
    from oracle import ref_import
    from skghoi_amd import GraphHead, InteractionHead
    o2v = synth.hico_object_to_verb()
    ref = ref_import.build_reference_head(117, 49, o2v, 256, 7, 15, 15)
    gh = GraphHead(256, 7, 1024, 1024, 117, 49, o2v)
    head = InteractionHead(torch.nn.Identity(), gh, torch.nn.Linear(2048, 1), torch.nn.Linear(2048, 117), 49, 117)
    a = [(k, tuple(v.shape)) for k, v in ref.state_dict().items()]
    b = [(k, tuple(v.shape)) for k, v in head.state_dict().items()]
    assert a == b and len(a) == 408
    head.load_state_dict(ref.state_dict())


class _FakeEvent:
    def __init__(self):
        self.synced = False

    def synchronize(self):
        self.synced = True


@pytest.mark.parametrize("sizes", [[1] * 6, [1, 2, 1, 1, 2, 1], [2] * 6, [3, 1, 4, 1, 5, 2, 6, 1]])
def test_table_drawer_ring_wraps_without_overwriting(sizes):
    """More chunks than staging slots (ring of 4): a slot is redrawn only after the consumer released the chunk that
    used it last.  Inline mode (<= 8 images) and the helper thread both hand out, for every chunk, the tables a plain
    sequential draw gives (chunks 0 and 1 used to come back holding the tables of chunks 4 and 5)."""
    from skghoi_amd import engine
    K = 5
    torch.manual_seed(77)
    want = [tuple(t.clone() for t in transh.draw_batch(K, n, need_relations=True)) for n in sizes]
    state_after = torch.get_rng_state()
    cap = max(sizes)
    slots = [dict(cap=cap, bufs=(torch.empty(cap, _capi.TRANSH_ENT, _capi.TRANSH_DIM),
                                 torch.empty(cap, K, _capi.TRANSH_DIM), torch.empty(cap, K, _capi.TRANSH_DIM)),
                  event=None) for _ in range(4)]
    torch.manual_seed(77)
    drawer = engine._TableDrawer(K, sizes, True, slots)
    assert (drawer.thread is None) == (sum(sizes) <= engine._TableDrawer.INLINE_IMAGES)
    events = []
    try:
        for i in range(len(sizes)):
            got = drawer.get(i)
            for g, w in zip(got, want[i]):
                assert torch.equal(g, w), "chunk %d" % i          # still intact when the consumer reads it
            ev = _FakeEvent(); events.append(ev)
            drawer.release(i, ev)
    finally:
        drawer.join()
    assert torch.equal(torch.get_rng_state(), state_after)
    assert all(e.synced for e in events[:len(sizes) - 4])        # every reused slot waited for its previous reader


def test_table_drawer_consumer_failure_does_not_hang():
    from skghoi_amd import engine
    K = 5
    sizes = [3] * 8
    slots = [dict(cap=3, bufs=(torch.empty(3, _capi.TRANSH_ENT, _capi.TRANSH_DIM), None, None), event=None)
             for _ in range(4)]
    drawer = engine._TableDrawer(K, sizes, False, slots)
    drawer.get(0)                      # the consumer "fails" here: it never releases anything
    drawer.join()                      # must return (the helper thread is parked on chunk 4's slot)
    assert not drawer.thread.is_alive()


def test_registration_epoch_tracks_parameter_and_module_replacement():
    from skghoi_amd import engine
    lin = torch.nn.Linear(4, 4)
    e0 = engine._REG_EPOCH[0]
    lin.weight = torch.nn.Parameter(torch.zeros(4, 4))
    e1 = engine._REG_EPOCH[0]
    assert e1 > e0
    seq = torch.nn.Sequential(lin)
    seq[0] = torch.nn.Linear(4, 4)
    assert engine._REG_EPOCH[0] > e1
    e2 = engine._REG_EPOCH[0]
    with torch.no_grad():
        lin.weight.mul_(2.0)           # in-place writes do not register anything (the device checksum sees those)
    lin.weight.data.mul_(2.0)
    assert engine._REG_EPOCH[0] == e2


def test_copies_and_pickles_of_the_head_drop_runtime_caches():
    """deepcopy / pickle of the module (EMA copies, checkpoints of whole modules) must not drag the run-time caches along --
    the engine holds HIP streams and captured plans, the prefetch state holds events -- and must not share them either."""
    import copy
    import pickle
    import threading
    from skghoi_amd import GraphHead, InteractionHead, synth
    gh = GraphHead(8, 2, 1024, 1024, 117, 49, synth.hico_object_to_verb())
    head = InteractionHead(torch.nn.Identity(), gh, torch.nn.Linear(2048, 1), torch.nn.Linear(2048, 117), 49, 117)
    unpicklable = threading.Lock()                              # stands in for streams / graphs / events
    head._engine = unpicklable; head._stacked = unpicklable; head._pf_stream = unpicklable; head._prefetched = unpicklable
    gh._engine = unpicklable
    twin = copy.deepcopy(head)
    assert twin._engine is None and twin._stacked is None and twin._prefetched is None and twin.box_pair_head._engine is None
    assert head._engine is unpicklable                          # the original keeps its caches
    assert all(torch.equal(a, b) for a, b in zip(head.state_dict().values(), twin.state_dict().values()))
    blob = pickle.dumps(head)
    back = pickle.loads(blob)
    assert back._engine is None and set(back.state_dict()) == set(head.state_dict())


@pytest.mark.parametrize("seed", range(8))
@pytest.mark.parametrize("faithful", [True, False])
def test_native_training_layout_equals_the_numpy_route(seed, faithful):
    """skg_layout_pack_train (one native call into the staging block) against layout.build + pack_int_arrays + the step's
    extra tables built with numpy: skipped images (no human, a single node), the reference's offset bug on them (Q9), the zip
    truncation of HEAD:822, ragged ground-truth counts."""
    import numpy as np
    from skghoi_amd import layout
    rs = np.random.RandomState(seed)
    B = int(rs.randint(1, 7))
    n_h = rs.randint(0, 5, B); n_o = rs.randint(0, 6, B)
    n = n_h + n_o
    if seed == 3:
        n[:] = 0; n_h[:] = 0; n[0] = 2; n_h[0] = 1                  # sum(n) < B: the zip truncation visits two images only
    L = rs.randint(0, 50, B)
    shapes = [(float(rs.randint(300, 900)), float(rs.randint(300, 900))) for _ in range(B)]
    gt = rs.randint(0, 4, B)
    want = layout.build(n_h, n, L, shapes, 49, faithful_skip_offset=faithful)
    wbuf, woffs = layout.pack_int_arrays(want)
    lay, buf, offs = layout.build_train(n_h, n, L, shapes, 49, gt_count=gt, faithful_skip_offset=faithful, pin=False)
    host = buf.numpy()
    for k in ("B", "sum_all", "n_visit", "n_active", "sum_n", "sum_h", "sum_g", "sum_p", "sum_l"):
        assert getattr(lay, k) == getattr(want, k), k
    for k in ("active", "skipped", "box_off", "pairs_per_image", "cells_per_image", "node_img", "hum_img", "node_enc_row",
              "hum_enc_row", "node_ent_row", "hum_ent_row"):
        assert np.array_equal(getattr(lay, k), getattr(want, k)), k
    assert lay.meta.tobytes() == want.meta.tobytes()
    for k, (o, l) in woffs.items():                                  # every slice of the numpy pack, bit for bit
        o2, l2 = offs[k]
        assert l2 == l and o2 % 4 == 0 and np.array_equal(host[o2:o2 + l2], wbuf[o:o + l]), k
    NA, Mh, Mn = max(want.sum_all, 1), want.sum_h, want.sum_n
    hum_of = np.full(NA, -1, np.int32); node_of = np.full(NA, -1, np.int32)
    hum_of[want.hum_enc_row] = np.arange(Mh, dtype=np.int32); node_of[want.node_enc_row] = np.arange(Mn, dtype=np.int32)
    sl = lambda k: host[offs[k][0]:offs[k][0] + offs[k][1]]
    assert np.array_equal(sl("hum_of"), hum_of) and np.array_equal(sl("node_of"), node_of)
    assert np.array_equal(sl("pair_img"), np.repeat(want.meta["image"].astype(np.int32), want.pairs_per_image))
    gt_off = np.zeros(want.n_active + 1, np.int32); gt_off[1:] = np.cumsum(gt[want.active])
    assert np.array_equal(sl("gt_off"), gt_off)


def test_native_training_layout_raises_like_the_reference_on_too_many_nodes():
    from skghoi_amd import layout
    with pytest.raises(IndexError):
        layout.build_train([1], [81], None, [(10., 10.)], 49, pin=False)
    with pytest.raises(IndexError):
        layout.build_train([1], [3], None, [(10., 10.)], 80, pin=False)
