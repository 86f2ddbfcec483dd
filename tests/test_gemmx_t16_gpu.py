"""skg_gemmx_bf16 with BOTH operands' bf16 twins: the direct-to-LDS kernel (skg_gemmx_t16_kernel: global_load_lds ring,
ds_read_b64_tr_b16 transposed fragment reads for row-contiguous operands, 64-deep k-steps) against float64 products of the
bf16-rounded operands, through the C ABI -- every operand layout of the training step (forward: both k-contiguous; input
gradient: B row-contiguous; weight gradient: both row-contiguous), ragged M / N / K, split-K, branch-major weights, the
fused epilogues, grouped launches and the bias gradient (fp32 sum of the bf16-rounded operand: what the reference's
autocast backward sums).  Exact-integer operands check the fragment maps: with small integers every product and sum is
exact in bf16 / fp32, so a single wrong element shows as an integer difference."""
import os

import pytest
import torch

from skghoi_amd import gemmx

pytestmark = pytest.mark.gpu


def _rnd(*shape, seed=0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g).cuda()


def _ints(*shape, seed=0, lo=-3, hi=4):
    g = torch.Generator().manual_seed(seed)
    return torch.randint(lo, hi, shape, generator=g).float().cuda()


def _q(t):
    return t.to(torch.bfloat16).double()


def _tw(t):
    return t.to(torch.bfloat16)


def _close(got, want, what, tol=1e-5):
    want = want.to(torch.float64)
    err = (got.to(torch.float64) - want).abs().max().item()
    scale = max(want.abs().max().item(), 1e-30)
    assert err <= tol * scale, "%s: err %.3e at scale %.3e" % (what, err, scale)


def _launch(ops):
    for o in ops:
        assert o.A16 is not None and o.B16 is not None
    gemmx.launch(ops, bf16=True)


SHAPES = [(300, 256, 128), (128, 128, 64), (1, 1024, 256), (513, 72, 1032), (3200, 1024, 1024), (37, 120, 50), (200, 136, 200),
          (131, 8, 192), (64, 64, 70)]


@pytest.mark.parametrize("M,N,K", SHAPES)
@pytest.mark.parametrize("split", [0, 1, 3])
def test_forward_both_operands_k_contiguous(M, N, K, split):
    if K % 8:
        pytest.skip("k-contiguous twins need rows of whole 16-byte pieces")
    x, W, b = _rnd(M, K, seed=1), _rnd(N, K, seed=2), _rnd(N, seed=3)
    out = torch.full((M, N), float("nan")).cuda()
    c16 = torch.full((M, N), float("nan"), dtype=torch.bfloat16).cuda()
    op = gemmx.forward(x, W, out, bias=b, relu=True)
    op.split_k = split
    op.A16, op.B16, op.C16 = _tw(x), _tw(W), c16
    _launch([op])
    _close(out, torch.relu(_q(x) @ _q(W).t() + b.double()), "forward")
    assert torch.equal(c16.view(torch.int16), out.to(torch.bfloat16).view(torch.int16)), "C16 is not the rounding of C"


@pytest.mark.parametrize("M,N_out,K_in", [(300, 256, 128), (45, 120, 2048), (130, 1024, 56), (3200, 1024, 1024), (77, 64, 72)])
@pytest.mark.parametrize("split", [0, 4])
def test_input_grad_b_row_contiguous(M, N_out, K_in, split):
    dz, W = _rnd(M, N_out, seed=7), _rnd(N_out, K_in, seed=8)
    y_prev = _rnd(M, K_in, seed=9)
    dx = _rnd(M, K_in, seed=10)
    before = dx.clone()
    op = gemmx.input_grad(dz, W, dx, mask=y_prev, accumulate=True)
    op.split_k = split
    op.A16, op.B16 = _tw(dz), _tw(W)
    _launch([op])
    _close(dx, (before.double() + _q(dz) @ _q(W)) * (y_prev > 0), "input grad")


@pytest.mark.parametrize("rows,N_out,K_in", [(3200, 256, 128), (77, 120, 2048), (500, 1024, 48), (16, 64, 64), (3381, 1024, 1024),
                                             (100, 8, 8)])
@pytest.mark.parametrize("split", [0, 1, 7])
def test_weight_and_bias_grad_both_row_contiguous(rows, N_out, K_in, split):
    dz, x = _rnd(rows, N_out, seed=13), _rnd(rows, K_in, seed=14)
    dW = torch.full((N_out, K_in), float("nan")).cuda(); db = torch.full((N_out,), float("nan")).cuda()
    op = gemmx.weight_grad(dz, x, dW, db=db)
    op.split_k = split
    op.A16, op.B16 = _tw(dz), _tw(x)
    _launch([op])
    _close(dW, _q(dz).t() @ _q(x), "dW")
    _close(db, _q(dz).sum(0), "db")                     # the fp32 sum of the ROUNDED operand (autocast's grad_output is bf16)
    op = gemmx.weight_grad(dz, x, dW, db=db, accumulate=True)
    op.split_k = split
    op.A16, op.B16 = _tw(dz), _tw(x)
    _launch([op])
    _close(dW, 2 * (_q(dz).t() @ _q(x)), "dW accumulate")
    _close(db, 2 * _q(dz).sum(0), "db accumulate")


@pytest.mark.parametrize("kind", ["fwd", "dx", "dw"])
def test_fragment_maps_with_exact_integers(kind):
    """Small-integer operands: every product and every partial sum is exact, the result must equal the integer matmul
    element for element (asymmetric operands: a transposed or permuted fragment cannot hide)."""
    M, N, K = 257, 192, 328
    if kind == "fwd":
        x, W = _ints(M, K, seed=1), _ints(N, K, seed=2)
        out = torch.empty(M, N).cuda()
        op = gemmx.forward(x, W, out)
        op.A16, op.B16 = _tw(x), _tw(W)
        want = x.double() @ W.double().t()
    elif kind == "dx":
        dz, W = _ints(M, K, seed=3), _ints(K, N, seed=4)
        out = torch.empty(M, N).cuda()
        op = gemmx.input_grad(dz, W, out)
        op.A16, op.B16 = _tw(dz), _tw(W)
        want = dz.double() @ W.double()
    else:
        dz, x = _ints(K, M - 1, seed=5), _ints(K, N, seed=6)          # rows = K, outputs [M - 1 = 256, N]
        out = torch.empty(M - 1, N).cuda()
        db = torch.empty(M - 1).cuda()
        op = gemmx.weight_grad(dz, x, out, db=db)
        op.A16, op.B16 = _tw(dz), _tw(x)
        want = dz.double().t() @ x.double()
    op.split_k = 1
    _launch([op])
    assert torch.equal(out.double(), want), "max abs diff %g" % (out.double() - want).abs().max().item()
    if kind == "dw":
        assert torch.equal(db.double(), dz.double().sum(0))


def _blocked(W, blk=64):
    N, K = W.shape
    return W.view(N, K // blk, blk).permute(1, 0, 2).contiguous()


@pytest.mark.parametrize("split", [0, 3])
def test_branch_major_weights_in_all_three_roles(split):
    M, N, K = 200, 1024, 1024
    x, W, b = _rnd(M, K, seed=51), _rnd(N, K, seed=52) * 0.05, _rnd(N, seed=53)
    Wb = _blocked(W)
    out = torch.empty(M, N).cuda()
    op = gemmx.forward(x, Wb, out, bias=b, relu=True, K=K, N=N, w_blocks=(6, N * 64))
    dz = _rnd(M, N, seed=54)
    dx = torch.empty(M, K).cuda()
    op2 = gemmx.input_grad(dz, Wb, dx, N_in=K, K_out=N, w_blocks=(6, N * 64))
    dWb = torch.empty(16, N, 64).cuda()
    op3 = gemmx.weight_grad(dz, x, dWb, w_blocks=(6, N * 64))
    op.A16, op.B16 = _tw(x), _tw(Wb)
    op2.A16, op2.B16 = _tw(dz), _tw(Wb)
    op3.A16, op3.B16 = _tw(dz), _tw(x)
    for o in (op, op2, op3):
        o.split_k = split
    _launch([op, op2, op3])
    _close(out, torch.relu(_q(x) @ _q(W).t() + b.double()), "forward blocked")
    _close(dx, _q(dz) @ _q(W), "input grad blocked")
    _close(dWb, _blocked((_q(dz).t() @ _q(x)).float()).double(), "dW blocked")


def test_grouped_backward_of_a_layer_with_strided_views():
    rows, N_out, K_in = 904, 1024, 1088
    big = _rnd(rows, 2 * N_out, seed=17)
    dz = big[:, N_out:]
    big16 = _tw(big)
    x, W, y_prev = _rnd(rows, K_in, seed=18), _rnd(N_out, K_in, seed=19) * 0.05, _rnd(rows, K_in, seed=20)
    dx = torch.empty(rows, K_in).cuda(); dW = torch.empty(N_out, K_in).cuda(); db = torch.empty(N_out).cuda()
    o1 = gemmx.input_grad(dz, W, dx, mask=y_prev)
    o2 = gemmx.weight_grad(dz, x, dW, db=db)
    o1.A16, o1.B16 = big16[:, N_out:], _tw(W)
    o2.A16, o2.B16 = big16[:, N_out:], _tw(x)
    _launch([o1, o2])
    _close(dx, (_q(dz) @ _q(W)) * (y_prev > 0), "grouped dx")
    _close(dW, _q(dz).t() @ _q(x), "grouped dW")
    _close(db, _q(dz).sum(0), "grouped db")


def test_same_products_on_the_register_staged_kernel_agree():
    """A/B within one process: the products of the step on the direct-to-LDS kernel and (twins withheld from one operand) on
    the register-staged kernel -- same MFMA sequence over k, so without split-K the results are bit-identical."""
    M, N, K = 640, 512, 1024
    x, W = _rnd(M, K, seed=61), _rnd(N, K, seed=62)
    outs = []
    for both in (True, False):
        out = torch.empty(M, N).cuda()
        op = gemmx.forward(x, W, out)
        op.split_k = 1
        op.A16 = _tw(x)
        if both:
            op.B16 = _tw(W)
        gemmx.launch([op], bf16=True)
        outs.append(out)
    assert torch.equal(outs[0], outs[1])


def _padded(t, pitch):
    """bf16 twin of a [rows, cols] tensor with rows `pitch` elements apart (cols <= pitch; the pad holds NaN: it must never
    reach a stored output)."""
    out = torch.full((t.shape[0], pitch), float("nan"), dtype=torch.bfloat16).cuda()
    out[:, :t.shape[1]] = t.to(torch.bfloat16)
    return out


@pytest.mark.parametrize("split", [0, 3])
def test_padded_twins_of_operands_without_16_byte_rows(split):
    """fc_head / fc_tail are [1024, 1074] (HEAD:694-701): fp32 rows of 4296 bytes, no 16-byte pieces -- their products stayed
    on the register-staged loop.  With a PADDED twin (pitch 1088: skg_gemmx_desc.a16_ld / b16_ld) all three products of the
    layer run on the direct-to-LDS kernel: forward (W k-contiguous, K = 1074 ragged), input gradient (W row-contiguous,
    N = 1074: the last 8-row piece reaches into the pad), weight gradient into the unpadded fp32 gradient (element-wise
    epilogue: ldc 1074)."""
    M, N, K = 240, 1024, 1074
    x, W, b = _rnd(M, 1088, seed=21)[:, :K], _rnd(N, K, seed=22) * 0.05, _rnd(N, seed=23)
    x16 = torch.full((M, 1088), float("nan"), dtype=torch.bfloat16).cuda(); x16[:, :K] = x.to(torch.bfloat16)
    W16 = _padded(W, 1088)
    gemmx.path_counts(reset=True)
    out = torch.full((M, N), float("nan")).cuda()
    op = gemmx.forward(x, W, out, bias=b, relu=True, K=K)
    op.A16, op.B16, op.b16_ld, op.split_k = x16, W16, 1088, split
    _launch([op])
    _close(out, torch.relu(_q(x) @ _q(W).t() + b.double()), "forward through a padded weight twin")
    dz = _rnd(M, N, seed=24)
    dx = torch.full((M, 1088), float("nan")).cuda()
    op = gemmx.input_grad(dz, W, dx[:, :K], N_in=K)
    op.A16, op.B16, op.b16_ld, op.split_k = _tw(dz), W16, 1088, split
    dW = torch.full((N, K), float("nan")).cuda(); db = torch.full((N,), float("nan")).cuda()
    op2 = gemmx.weight_grad(dz, x, dW, db=db, k_in=K)
    op2.A16, op2.B16, op2.split_k = _tw(dz), x16, split
    _launch([op, op2])
    _close(dx[:, :K], _q(dz) @ _q(W), "input gradient through a padded weight twin")
    assert torch.isnan(dx[:, K:]).all()
    _close(dW, _q(dz).t() @ _q(x), "weight gradient, ragged N")
    _close(db, _q(dz).sum(0), "bias gradient", tol=1e-5)
    assert gemmx.path_counts() == (0, 0, 2), "a product left the direct-to-LDS kernel: %r" % (gemmx.path_counts(),)


def test_row_contiguous_operand_with_ragged_rows():
    """dW of the classifier: A(m, k) = dlogits[k][m] with M = 118 rows of a [rows, 120] array (HEAD:410-411: 117 verbs + the
    suppressor) -- not a multiple of the 8-row pieces a row-contiguous twin is staged in, but the pitch leaves room for the
    last piece.  Exact integers: any misplaced row shows."""
    rows, M, N = 3120, 118, 2048
    dl = _ints(rows, 120, seed=31); dl[:, M:] = 0
    pf = _ints(rows, N, seed=32)
    dW = torch.full((M, N), float("nan")).cuda(); db = torch.full((M,), float("nan")).cuda()
    gemmx.path_counts(reset=True)
    op = gemmx.weight_grad(dl, pf, dW, db=db, n_out=M)
    op.A16, op.B16 = _tw(dl), _tw(pf)
    _launch([op])
    assert gemmx.path_counts() == (0, 0, 1)
    assert torch.equal(dW.double(), dl[:, :M].double().t() @ pf.double())
    assert torch.equal(db.double(), dl[:, :M].double().sum(0))
    # V-COCO: 25 columns stored 28 apart; the twin is a padded copy with a pitch of 32
    M2 = 25
    dl2 = _ints(rows, 28, seed=33); dl2[:, M2:] = 0
    dW2 = torch.full((M2, N), float("nan")).cuda()
    op = gemmx.weight_grad(dl2, pf, dW2, n_out=M2)
    op.A16, op.a16_ld, op.B16 = _padded(dl2, 32), 32, _tw(pf)
    _launch([op])
    assert gemmx.path_counts() == (0, 0, 2)
    assert torch.equal(dW2.double(), dl2[:, :M2].double().t() @ pf.double())
    dx = torch.full((rows, N), float("nan")).cuda()
    W = _ints(M2, N, seed=34)
    op = gemmx.input_grad(dl2, W, dx, K_out=M2)
    op.A16, op.a16_ld, op.B16 = _padded(dl2, 32), 32, _tw(W)
    _launch([op])
    assert gemmx.path_counts() == (0, 0, 3)
    assert torch.equal(dx.double(), dl2[:, :M2].double() @ W.double())
