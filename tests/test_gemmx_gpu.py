"""skg_gemmx_f32 / skg_gemmx_bf16 (free operand layouts, fused backward epilogues) against float64 matmuls, through the
C ABI.  Tolerance: fp32 accumulation over K in a different order than the fp64 reference: <= 2e-6 * sqrt(K) * max|A| *
max|B| per element would be generous; measured errors are ~1e-6 relative to the result's scale (asserted at 1e-5).
The bf16 entry point rounds both operands to bf16 (round to nearest even) and accumulates in fp32: its reference is
the float64 product of the operands rounded the same way by torch, at the SAME tolerance -- bias, masks, accumulated
values and the bias gradient are not rounded."""
import numpy as np
import pytest
import torch

from skghoi_amd import gemmx

pytestmark = pytest.mark.gpu


@pytest.fixture(params=[False, True], ids=["f32", "bf16"])
def bf16(request):
    return request.param


@pytest.fixture(autouse=True, params=[True, False], ids=["reduce-in-launch", "reduce-launch"])
def split_reduction(request, monkeypatch):
    """Every test of this file runs with both reductions of split products: by the tiles' last-arriving workgroups inside the
    product launch (skg_gemmx_desc.split_ctr, the default) and by the second launch."""
    monkeypatch.setattr(gemmx, "INLAUNCH", request.param)
    yield
    if request.param:
        assert int(gemmx.counters(torch.device("cuda", torch.cuda.current_device())).abs().sum()) == 0, \
            "a launch left its tile counters non-zero"


def _q(t, bf16):
    """The operand as the matrix core sees it, in float64."""
    return (t.to(torch.bfloat16) if bf16 else t).double()


def _rnd(*shape, seed=0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g).cuda()


def _close(got, want, what, tol=1e-5):
    want = want.to(torch.float64)
    err = (got.to(torch.float64) - want).abs().max().item()
    scale = max(want.abs().max().item(), 1e-30)
    assert err <= tol * scale, "%s: err %.3e at scale %.3e" % (what, err, scale)


def _blocked(W, blk=64):
    """[N, K] -> branch-major storage [K / blk][N][blk] (how the 16 fc_3 weights of an MBF are kept)."""
    N, K = W.shape
    return W.view(N, K // blk, blk).permute(1, 0, 2).contiguous()


@pytest.mark.parametrize("M,N,K", [(300, 256, 128), (37, 117, 50), (128, 128, 16), (1, 1024, 256), (513, 70, 1030)])
@pytest.mark.parametrize("split", [0, 1, 5])
def test_forward_layer(M, N, K, split, bf16):
    x, W, b = _rnd(M, K, seed=1), _rnd(N, K, seed=2), _rnd(N, seed=3)
    out = torch.full((M, N), float("nan")).cuda()
    op = gemmx.forward(x, W, out, bias=b, relu=True)
    op.split_k = split
    gemmx.launch([op], bf16=bf16)
    _close(out, torch.relu(_q(x, bf16) @ _q(W, bf16).t() + b.double()), "forward")


@pytest.mark.parametrize("split", [0, 2])
@pytest.mark.parametrize("M,N,K,col0,ldc", [(200, 132, 96, 0, 132),      # staged epilogue, N edge inside a tile's quads
                                            (200, 132, 96, 0, 136),      # staged, padded rows
                                            (131, 256, 64, 1, 260),      # C not 16-byte aligned: element-wise epilogue
                                            (131, 256, 64, 0, 258)])     # ldc not a multiple of 4: element-wise epilogue
def test_forward_epilogue_variants(M, N, K, col0, ldc, split, bf16):
    """The staged (16-byte stores from LDS) and the element-wise epilogue must agree with the reference on every
    alignment: which one runs is decided per product from N, the alignment of C / bias / mask and ldc."""
    x, W, b = _rnd(M, K, seed=31), _rnd(N, K, seed=32), _rnd(N, seed=33)
    big = torch.full((M, ldc), float("nan")).cuda()
    out = big[:, col0:col0 + N]
    op = gemmx.forward(x, W, out, bias=b, relu=True)
    op.split_k = split
    gemmx.launch([op], bf16=bf16)
    _close(out, torch.relu(_q(x, bf16) @ _q(W, bf16).t() + b.double()), "forward")
    rest = torch.cat([big[:, :col0], big[:, col0 + N:]], 1)
    assert torch.isnan(rest).all(), "stores outside the product's columns"


def _twin(t):
    return t.to(torch.bfloat16)


@pytest.mark.parametrize("which", ["A", "B", "AB"])
@pytest.mark.parametrize("kind,M,N,K", [("fwd", 300, 256, 128), ("fwd", 1, 1024, 256), ("fwd", 513, 72, 1032),
                                        ("dx", 300, 256, 128), ("dx", 130, 1024, 56), ("dw", 3200, 256, 128),
                                        ("dw", 500, 1024, 48), ("dw", 77, 120, 2048)])
@pytest.mark.parametrize("split", [0, 3])
def test_bf16_twins_change_nothing(kind, M, N, K, which, split):
    """skg_gemmx_bf16 with the operands' bf16 twins (A16 / B16: the same values rounded by torch, same layout) returns
    bit-for-bit what it returns from the fp32 arrays -- every operand layout, ragged edges (the twin is read by the
    fast loop only), split-K, the bias gradient's workgroups (which keep the fp32 operand) -- and C16 receives the bf16
    rounding of C."""
    def build(twins):
        if kind == "fwd":
            x, W, b = _rnd(M, K, seed=41), _rnd(N, K, seed=42), _rnd(N, seed=43)
            out = torch.full((M, N), float("nan")).cuda()
            op = gemmx.forward(x, W, out, bias=b, relu=True)
            a, bb = x, W
        elif kind == "dx":
            dz, W, x = _rnd(M, N, seed=44), _rnd(N, K, seed=45), _rnd(M, K, seed=46)
            out = torch.full((M, K), float("nan")).cuda()
            op = gemmx.input_grad(dz, W, out, mask=x)
            a, bb = dz, W
        else:
            dz, x = _rnd(M, N, seed=47), _rnd(M, K, seed=48)
            out = torch.full((N, K), float("nan")).cuda()
            db = torch.full((N,), float("nan")).cuda()
            op = gemmx.weight_grad(dz, x, out, db=db)
            a, bb = dz, x
        op.split_k = split
        c16 = torch.full(out.shape, float("nan"), dtype=torch.bfloat16).cuda()
        op.C16 = c16
        if twins:
            if "A" in which:
                op.A16 = _twin(a)
            if "B" in which:
                op.B16 = _twin(bb)
        gemmx.launch([op], bf16=True)
        return out, c16, (db if kind == "dw" else None)

    ref, ref16, refdb = build(False)
    got, got16, gotdb = build(True)
    assert torch.equal(got16.view(torch.int16), got.to(torch.bfloat16).view(torch.int16)), "C16 is not the rounding of C"
    if which == "AB":
        # both twins: the direct-to-LDS kernel (64-deep k-steps: other split-K slice boundaries; the bias gradient sums the
        # ROUNDED operand there).  Without split-K the MFMA sequence over k is the same: bit-identical products.
        if split == 0 and K % 64 == 0:
            assert torch.equal(got, ref), "C differs with both twins"
        else:
            _close(got, ref.double(), "C with both twins")
        if refdb is not None:
            _close(gotdb, refdb.double(), "db with both twins", tol=4e-3)     # (sum of bf16-rounded values: 2^-9 per element)
        return
    assert torch.equal(got, ref), "C differs with twins (%s)" % which
    assert torch.equal(got16.view(torch.int16), ref16.view(torch.int16))
    if refdb is not None:
        assert torch.equal(gotdb, refdb)


def test_bf16_twins_through_branch_major_weight():
    M, N, K = 200, 1024, 1024
    x, W, b = _rnd(M, K, seed=51), _rnd(N, K, seed=52) * 0.05, _rnd(N, seed=53)
    Wb = _blocked(W)
    outs = []
    for twins in (False, True):
        out = torch.empty(M, N).cuda()
        op = gemmx.forward(x, Wb, out, bias=b, relu=True, K=K, N=N, w_blocks=(6, N * 64))
        dz = _rnd(M, N, seed=54)
        dx = torch.empty(M, K).cuda()
        op2 = gemmx.input_grad(dz, Wb, dx, N_in=K, K_out=N, w_blocks=(6, N * 64))
        if twins:
            op.A16, op.B16 = _twin(x), _twin(Wb)
            op2.A16, op2.B16 = _twin(dz), _twin(Wb)
        gemmx.launch([op, op2], bf16=True)
        outs.append((out, dx))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


@pytest.mark.parametrize("split", [0, 3])
def test_forward_with_branch_major_weight(split, bf16):
    M, N, K = 200, 1024, 1024
    x, W, b = _rnd(M, K, seed=4), _rnd(N, K, seed=5) * 0.05, _rnd(N, seed=6)
    Wb = _blocked(W)
    out = torch.empty(M, N).cuda()
    op = gemmx.forward(x, Wb, out, bias=b, w_blocks=(6, N * 64))
    op.split_k = split
    gemmx.launch([op], bf16=bf16)
    _close(out, _q(x, bf16) @ _q(W, bf16).t() + b.double(), "forward blocked")


@pytest.mark.parametrize("M,N_out,K_in", [(300, 256, 128), (45, 117, 2048), (130, 1024, 52)])
@pytest.mark.parametrize("split", [0, 4])
def test_input_grad_with_relu_mask_and_accumulate(M, N_out, K_in, split, bf16):
    dz, W = _rnd(M, N_out, seed=7), _rnd(N_out, K_in, seed=8)
    y_prev = _rnd(M, K_in, seed=9)                      # the producing layer's output: ReLU mask source
    dx = _rnd(M, K_in, seed=10)
    before = dx.clone()
    op = gemmx.input_grad(dz, W, dx, mask=y_prev, accumulate=True)
    op.split_k = split
    gemmx.launch([op], bf16=bf16)
    want = (before.double() + _q(dz, bf16) @ _q(W, bf16)) * (y_prev > 0)      # accumulate first, mask last (see header)
    _close(dx, want, "input grad")


def test_input_grad_through_branch_major_weight(bf16):
    M, N, K = 150, 1024, 1024
    dz, W = _rnd(M, N, seed=11), _rnd(N, K, seed=12) * 0.05
    dx = torch.empty(M, K).cuda()
    gemmx.launch([gemmx.input_grad(dz, _blocked(W), dx, w_blocks=(6, N * 64))], bf16=bf16)
    _close(dx, _q(dz, bf16) @ _q(W, bf16), "input grad blocked")


@pytest.mark.parametrize("rows,N_out,K_in", [(3200, 256, 128), (77, 117, 2048), (500, 1024, 48), (16, 64, 64)])
@pytest.mark.parametrize("split", [0, 1, 7])
def test_weight_and_bias_grad(rows, N_out, K_in, split, bf16):
    dz, x = _rnd(rows, N_out, seed=13), _rnd(rows, K_in, seed=14)
    dW = torch.full((N_out, K_in), float("nan")).cuda(); db = torch.full((N_out,), float("nan")).cuda()
    op = gemmx.weight_grad(dz, x, dW, db=db)
    op.split_k = split
    gemmx.launch([op], bf16=bf16)
    _close(dW, _q(dz, bf16).t() @ _q(x, bf16), "dW")
    _close(db, dz.double().sum(0), "db")
    # accumulation into existing gradients (a weight used at two call sites)
    op = gemmx.weight_grad(dz, x, dW, db=db, accumulate=True)
    op.split_k = split
    gemmx.launch([op], bf16=bf16)
    _close(dW, 2 * (_q(dz, bf16).t() @ _q(x, bf16)), "dW accumulate")
    _close(db, 2 * dz.double().sum(0), "db accumulate")


def test_weight_grad_into_branch_major_storage(bf16):
    rows, N, K = 700, 1024, 1024
    dz, x = _rnd(rows, N, seed=15), _rnd(rows, K, seed=16)
    dWb = torch.empty(16, N, 64).cuda()
    gemmx.launch([gemmx.weight_grad(dz, x, dWb, w_blocks=(6, N * 64))], bf16=bf16)
    want = _q(dz, bf16).t() @ _q(x, bf16)
    _close(dWb, _blocked(want.float()).double(), "dW blocked")       # same permutation of the fp64 result


def test_grouped_launch_of_a_layers_backward(bf16):
    """dX and dW of one layer (different shapes and layouts) in ONE launch, strided views as operands."""
    rows, N_out, K_in = 900, 1024, 1088
    big = _rnd(rows, 2 * N_out, seed=17)
    dz = big[:, N_out:]                                  # a column view: leading dimension 2048
    x, W, y_prev = _rnd(rows, K_in, seed=18), _rnd(N_out, K_in, seed=19) * 0.05, _rnd(rows, K_in, seed=20)
    dx = torch.empty(rows, K_in).cuda(); dW = torch.empty(N_out, K_in).cuda(); db = torch.empty(N_out).cuda()
    gemmx.launch([gemmx.input_grad(dz, W, dx, mask=y_prev), gemmx.weight_grad(dz, x, dW, db=db)], bf16=bf16)
    _close(dx, (_q(dz, bf16) @ _q(W, bf16)) * (y_prev > 0), "grouped dx")
    _close(dW, _q(dz, bf16).t() @ _q(x, bf16), "grouped dW")
    _close(db, dz.double().sum(0), "grouped db")


def test_many_small_products_one_call(bf16):
    ops, wants, outs = [], [], []
    for i in range(11):                                  # more than one group's worth: launch() cuts it into groups
        M, N, K = 20 + 7 * i, 64 + 32 * (i % 3), 1024
        x, W = _rnd(M, K, seed=30 + i), _rnd(N, K, seed=50 + i)
        out = torch.empty(M, N).cuda()
        ops.append(gemmx.forward(x, W, out)); outs.append(out); wants.append(_q(x, bf16) @ _q(W, bf16).t())
    gemmx.launch(ops, bf16=bf16)
    for o, w in zip(outs, wants):
        _close(o, w, "small product")


@pytest.mark.parametrize("kernel", ["f32", "bf16", "t16"])
@pytest.mark.parametrize("kind,M,N,K,split", [
    ("fwd", 300, 256, 1024, 4), ("fwd", 3200, 1024, 1024, 2), ("fwd", 513, 72, 1030, 5), ("fwd", 131, 258, 640, 3),
    ("dx", 300, 256, 1024, 8), ("dx", 130, 1024, 560, 3), ("dw", 3200, 256, 128, 7), ("dw", 3205, 1024, 48, 9),
    ("dw", 1077, 120, 2048, 2), ("dwb", 700, 1024, 1024, 3)])
def test_split_reduced_inside_the_launch_equals_the_reduce_launch_bit_for_bit(kernel, kind, M, N, K, split):
    """The tiles' last arrivers add the slices in slice order -- the additions of skg_gemmx_reduce_kernel in the same order,
    whichever slice finishes last: C, its bf16 twin and the bias gradient are bit-identical, on every kernel (exact fp32,
    register-staged bf16, direct-to-LDS with both twins), staged and element-wise epilogues (N = 258: no 16-byte rows),
    ragged tiles, accumulation + ReLU mask, branch-major C.  Run three times in a row: the counters wrap back to zero."""
    bf = kernel != "f32"

    def build(inlaunch, rep):
        if kind == "fwd":
            x, W, b = _rnd(M, K, seed=61), _rnd(N, K, seed=62), _rnd(N, seed=63)
            out = torch.full((M, N), float("nan")).cuda()
            op = gemmx.forward(x, W, out, bias=b, relu=True)
            a, bb, db = x, W, None
        elif kind == "dx":
            dz, W, x = _rnd(M, N, seed=64), _rnd(N, K, seed=65), _rnd(M, K, seed=66)
            out = _rnd(M, K, seed=67)
            op = gemmx.input_grad(dz, W, out, mask=x, accumulate=True)
            a, bb, db = dz, W, None
        else:
            dz, x = _rnd(M, N, seed=68), _rnd(M, K, seed=69)
            db = _rnd(N, seed=70)
            if kind == "dwb":
                out = _rnd(K // 64, N, 64, seed=71)
                op = gemmx.weight_grad(dz, x, out, db=db, accumulate=True, w_blocks=(6, N * 64))
            else:
                out = _rnd(N, K, seed=71)
                op = gemmx.weight_grad(dz, x, out, db=db, accumulate=True)
            a, bb = dz, x
        op.split_k = split
        c16 = torch.full(out.shape, float("nan"), dtype=torch.bfloat16).cuda()
        op.C16 = c16
        if kernel == "t16":
            op.A16, op.B16 = _twin(a), _twin(bb)
        gemmx.launch([op], bf16=bf, inlaunch=inlaunch)
        return out, c16, db

    ref = build(False, 0)
    for rep in range(3):
        got = build(True, rep)
        assert torch.equal(got[0], ref[0]), "C differs (run %d)" % rep
        assert torch.equal(got[1].view(torch.int16), ref[1].view(torch.int16)), "C16 differs"
        if ref[2] is not None:
            assert torch.equal(got[2], ref[2]), "bias gradient differs"


def test_grouped_split_products_reduce_inside_one_launch(bf16):
    """dX (ReLU mask) and dW (+ bias gradient) of a layer, both split, in ONE launch: each product has its own counters."""
    rows, N_out, K_in = 900, 1024, 1088
    dz = _rnd(rows, N_out, seed=81)
    x, W, y_prev = _rnd(rows, K_in, seed=82), _rnd(N_out, K_in, seed=83) * 0.05, _rnd(rows, K_in, seed=84)
    res = []
    for inl in (False, True):
        dx = torch.empty(rows, K_in).cuda(); dW = torch.empty(N_out, K_in).cuda(); db = torch.empty(N_out).cuda()
        a, b = gemmx.input_grad(dz, W, dx, mask=y_prev), gemmx.weight_grad(dz, x, dW, db=db)
        a.split_k, b.split_k = 4, 3
        gemmx.launch([a, b], bf16=bf16, inlaunch=inl)
        res.append((dx, dW, db))
    for u, v in zip(*res):
        assert torch.equal(u, v)
    _close(res[1][0], (_q(dz, bf16) @ _q(W, bf16)) * (y_prev > 0), "grouped dx")
    _close(res[1][1], _q(dz, bf16).t() @ _q(x, bf16), "grouped dW")


def test_argument_validation():
    import ctypes
    from skghoi_amd import _capi
    lib = _capi.lib()
    d = _capi.GemmXDesc()
    assert lib.skg_gemmx_f32(None, 1, None) == -1
    d.M, d.N, d.K = 4, 4, 4
    d.A = d.B = d.C = 16; d.ldc = 4
    d.a_sm, d.a_sk, d.b_sn, d.b_sk = 4, 2, 4, 1          # neither A stride is 1
    assert lib.skg_gemmx_f32(ctypes.byref(d), 1, None) == -1
    d.a_sk = 1; d.M = 0
    assert lib.skg_gemmx_f32(ctypes.byref(d), 1, None) == 0            # empty product: nothing launched
    assert lib.skg_gemmx_f32(ctypes.byref(d), 9, None) == -1           # more than SKG_GEMMX_GROUP_MAX
    assert lib.skg_gemmx_bf16(None, 1, None) == -1
    assert lib.skg_gemmx_bf16(ctypes.byref(d), 1, None) == 0
