"""GPU parity tests, kernel by kernel, through the C ABI (cruse_amd.ops -> libcruse_hip.so)
against torch-CPU f32 references of the same op at the reference's call sites."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from tests.util import max_abs, rel_l2, t

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "gpu tests need a HIP device"
    from cruse_amd import ops as o
    return o


# ------------------------------------------------------------------ STFT / iSTFT
def test_stft_golden_fixture(ops, golden):
    g = golden("g1_stft.npz")
    for L, T in ((3200, 21), (3199, 20), (3201, 21)):
        re, im, mag = ops.stft(t(g[f"x_{L}"]), 320, 160, mag_bins=160, mag_eps=1e-8)
        X = torch.from_numpy(g[f"X_{L}"])               # [B,F,T,2]
        assert re.shape == (2, T, 161)                  # frame / bin indexing exact
        assert max_abs(re.transpose(1, 2), X[..., 0]) < 2e-5
        assert max_abs(im.transpose(1, 2), X[..., 1]) < 2e-5
        ref_mag = torch.sqrt(X[..., 0] ** 2 + X[..., 1] ** 2 + 1e-8)[:, :160].transpose(1, 2)
        assert max_abs(mag, ref_mag) < 2e-5


def test_stft_impulse_indexing(ops):
    """one-hot impulses pin frame/bin placement and the reflect padding exactly."""
    L = 1600
    for pos in (0, 1, 159, 160, 161, 799, 1598, 1599):
        x = torch.zeros(1, L); x[0, pos] = 1.0
        ref = torch.stft(x, 320, 160, 320, window=torch.hann_window(320), return_complex=True, center=True)
        re, im, _ = ops.stft(x.cuda(), 320, 160)
        assert max_abs(re.transpose(1, 2), ref.real) < 1e-5, pos
        assert max_abs(im.transpose(1, 2), ref.imag) < 1e-5, pos


def test_stft_full_size_and_hop320(ops):
    g = torch.Generator().manual_seed(3)
    x = 0.1 * torch.randn(3, 64000, generator=g)
    for hop, T in ((160, 401), (320, 201)):
        ref = torch.stft(x, 320, hop, 320, window=torch.hann_window(320), return_complex=True, center=True)
        re, im, _ = ops.stft(x.cuda(), 320, hop)
        assert re.shape == (3, T, 161)
        assert rel_l2(torch.complex(re, im).transpose(1, 2), ref) < 1e-5              # as one complex tensor
        assert rel_l2(re.transpose(1, 2), ref.real) < 1e-5 and rel_l2(im.transpose(1, 2), ref.imag) < 1e-5


def test_stft_generic_dft_path(ops):
    x = 0.1 * torch.randn(2, 4000, generator=torch.Generator().manual_seed(4))
    ref = torch.stft(x, 512, 128, 512, window=torch.hann_window(512), return_complex=True, center=True)
    re, im, _ = ops.stft(x.cuda(), 512, 128)
    assert re.shape == (2, ref.shape[2], 257)
    assert rel_l2(re.transpose(1, 2), ref.real) < 1e-4 and rel_l2(im.transpose(1, 2), ref.imag) < 1e-4


def test_istft_golden_and_roundtrip(ops, golden):
    g = golden("g7_istft.npz")
    X = torch.from_numpy(g["X"])                        # [B,F,T,2]
    re, im = t(X[..., 0].transpose(1, 2)), t(X[..., 1].transpose(1, 2))
    y = ops.istft(re, im, 320, 160, 3200)
    assert max_abs(y, torch.from_numpy(g["y_rt"])) < 1e-5
    m = torch.from_numpy(g["m"])
    y2 = ops.istft(t((X[..., 0] * m).transpose(1, 2)), t((X[..., 1] * m).transpose(1, 2)), 320, 160, 3200)
    assert max_abs(y2, torch.from_numpy(g["y_masked"])) < 1e-5
    # full size round trip (size-independent property)
    x = 0.1 * torch.randn(4, 64000, generator=torch.Generator().manual_seed(5)).cuda()
    re, im, _ = ops.stft(x, 320, 160)
    assert max_abs(ops.istft(re, im, 320, 160, 64000), x) < 1e-5


def test_istft_backward_is_adjoint(ops):
    gen = torch.Generator().manual_seed(6)
    re = torch.randn(2, 21, 161, generator=gen); im = torch.randn(2, 21, 161, generator=gen)
    dw = torch.randn(2, 3200, generator=gen)
    rr = re.clone().requires_grad_(True); ii = im.clone().requires_grad_(True)
    y = torch.istft(torch.complex(rr, ii).transpose(1, 2), 320, 160, 320, window=torch.hann_window(320), length=3200)
    y.backward(dw)
    dre, dim = ops.istft_bwd(dw.cuda(), 21, 320, 160)
    assert rel_l2(dre, rr.grad) < 1e-5
    # bins 0 and 160 of the imaginary part carry no gradient (c2r ignores them)
    assert rel_l2(dim[..., 1:160], ii.grad[..., 1:160]) < 1e-5


# ------------------------------------------------------------------ convolutions
LEVELS = [(1, 160, 8, 80), (8, 80, 16, 40), (16, 40, 32, 20), (32, 20, 64, 10)]
PRECS = [(None, 1e-5), ("f32", 1e-5), ("bf16x3", 5e-5)]
WPRECS = [(None, 1e-5), ("f32", 1e-5), ("bf16x3", 5e-5), ("bf16", 1e-2)]


def _nchw(x):       # frame-major [B,T,C,F] -> [B,C,T,F]
    return x.permute(0, 2, 1, 3).contiguous()


@pytest.mark.parametrize("lvl", range(4))
def test_encoder_conv_fwd_bwd(ops, lvl):
    Cin, Fin, Cout, Fout = LEVELS[lvl]
    B, T = 3, 21
    gen = torch.Generator().manual_seed(10 + lvl)
    x = torch.randn(B, T, Cin, Fin, generator=gen)
    w = torch.randn(Cout, Cin, 2, 3, generator=gen) * 0.2
    b = torch.randn(Cout, generator=gen)
    xr = _nchw(x).requires_grad_(True); wr = w.clone().requires_grad_(True); br = b.clone().requires_grad_(True)
    y_ref = F.conv2d(xr, wr, br, stride=(1, 2), padding=(1, 1))[..., :-1, :]       # cruse_net.py:138,149 (R4)
    dy = torch.randn(B, T, Cout, Fout, generator=gen)
    y_ref.backward(_nchw(dy))
    for prec, tol in PRECS:          # VALU kernel and the MFMA implicit-GEMM kernel (where eligible)
        y = ops.conv_gather(x.cuda(), w.cuda(), b.cuda(), B, T, Cin, Fin, Cout, Fout, KT=2, S=2, pad=1, prec=prec)
        assert rel_l2(_nchw(y), y_ref) < tol, prec
        dx = ops.conv_scatter2(dy.cuda(), w.cuda(), None, B, T, Cout, Fout, Cin, KT=2, pad=1, prec=prec)
        assert rel_l2(_nchw(dx), xr.grad) < tol, prec
    for prec, tol in WPRECS:
        dw = torch.zeros_like(w).cuda()
        ops.conv_wgrad(dy.cuda(), x.cuda(), dw, B, T, Cout, Fout, Cin, Fin, KT=2, S=2, pad=1, prec=prec)
        assert rel_l2(dw, wr.grad) < tol, prec
    db = torch.zeros(Cout).cuda()
    ops.channel_sum(dy.cuda(), B * T, Cout, Fout, db)
    assert rel_l2(db, br.grad) < 1e-5


@pytest.mark.parametrize("lvl", range(4))
def test_skip_conv_fwd_bwd(ops, lvl):
    _, _, C, Fq = LEVELS[lvl]
    B, T = 2, 13
    gen = torch.Generator().manual_seed(20 + lvl)
    x = torch.randn(B, T, C, Fq, generator=gen)
    w = torch.randn(C, C, 1, 3, generator=gen) * 0.2
    xr = _nchw(x).requires_grad_(True); wr = w.clone().requires_grad_(True)
    y_ref = F.conv2d(xr, wr, None, padding=(0, 1))                                  # cruse_net.py:143 (R5)
    dy = torch.randn(B, T, C, Fq, generator=gen)
    y_ref.backward(_nchw(dy))
    base = torch.randn(B, T, C, Fq, generator=gen)
    for prec, tol in PRECS:
        y = ops.conv_gather(x.cuda(), w.cuda(), None, B, T, C, Fq, C, Fq, KT=1, S=1, pad=1, prec=prec)
        assert rel_l2(_nchw(y), y_ref) < tol, prec
        dx = base.clone().cuda()
        ops.conv_gather(dy.cuda(), w.cuda(), None, B, T, C, Fq, C, Fq, KT=1, S=1, pad=1, w_layout=1, out=dx, accum=True,
                        prec=prec)
        assert rel_l2(_nchw(dx.cpu() - base), xr.grad) < tol, prec
    for prec, tol in WPRECS:
        dw = torch.zeros_like(w).cuda()
        ops.conv_wgrad(dy.cuda(), x.cuda(), dw, B, T, C, Fq, C, Fq, KT=1, S=1, pad=1, prec=prec)
        assert rel_l2(dw, wr.grad) < tol, prec


@pytest.mark.parametrize("lvl", range(4))
def test_decoder_convT_fwd_bwd(ops, lvl):
    Cout, Fo, Cin, Fg = LEVELS[lvl]            # convT k: ch[k] x F_k -> ch[k-1] x 2F_k
    B, T = 2, 11
    gen = torch.Generator().manual_seed(30 + lvl)
    u = torch.randn(B, T, Cin, Fg, generator=gen)
    w = torch.randn(Cin, Cout, 1, 3, generator=gen) * 0.2
    b = torch.randn(Cout, generator=gen)
    ur = _nchw(u).requires_grad_(True); wr = w.clone().requires_grad_(True); br = b.clone().requires_grad_(True)
    v_ref = F.conv_transpose2d(ur, wr, br, stride=(1, 2))[..., :-1]                 # cruse_net.py:140,161 (R2)
    assert v_ref.shape[-1] == Fo
    dv = torch.randn(B, T, Cout, Fo, generator=gen)
    v_ref.backward(_nchw(dv))
    for prec, tol in PRECS:
        v = ops.conv_scatter2(u.cuda(), w.cuda(), b.cuda(), B, T, Cin, Fg, Cout, KT=1, pad=0, prec=prec)
        assert rel_l2(_nchw(v), v_ref) < tol, prec
        vs = ops.conv_scatter2(u.cuda(), w.cuda(), b.cuda(), B, T, Cin, Fg, Cout, KT=1, pad=0, act=1, prec=prec)
        assert rel_l2(_nchw(vs), torch.sigmoid(v_ref)) < tol, prec
        du = ops.conv_gather(dv.cuda(), w.cuda(), None, B, T, Cout, Fo, Cin, Fg, KT=1, S=2, pad=0, prec=prec)
        assert rel_l2(_nchw(du), ur.grad) < tol, prec
    for prec, tol in WPRECS:
        dw = torch.zeros_like(w).cuda()
        ops.conv_wgrad(u.cuda(), dv.cuda(), dw, B, T, Cin, Fg, Cout, Fo, KT=1, S=2, pad=0, prec=prec)
        assert rel_l2(dw, wr.grad) < tol, prec


def test_conv_golden_and_causality(ops, golden):
    g = golden("g2_conv.npz")       # cust_conv.Conv2dNormAct causal pad == R4 crop, run by the reference
    x = torch.from_numpy(g["x"])    # [2,1,9,160] NCHW, C=1 -> same memory as frame-major
    y = ops.conv_gather(t(x).view(2, 9, 1, 160), t(g["w"]), t(g["b"]), 2, 9, 1, 160, 8, 80, KT=2, S=2, pad=1)
    assert rel_l2(_nchw(y), torch.from_numpy(g["y"])) < 1e-5
    # causality: output frame t must not depend on frames > t, nor on other clips
    xa = torch.randn(2, 9, 1, 160); xb = xa.clone(); xb[:, 5:] += 1.0
    ya = ops.conv_gather(xa.cuda(), t(g["w"]), t(g["b"]), 2, 9, 1, 160, 8, 80, KT=2, S=2, pad=1)
    yb = ops.conv_gather(xb.cuda(), t(g["w"]), t(g["b"]), 2, 9, 1, 160, 8, 80, KT=2, S=2, pad=1)
    assert torch.equal(ya[:, :5], yb[:, :5]) and not torch.equal(ya[:, 5:], yb[:, 5:])


@pytest.mark.parametrize("lvl", range(4))
@pytest.mark.parametrize("prec", [None, "f32", "bf16x3", "bf16"])
def test_conv_bnstats_epilogue(ops, lvl, prec):
    """cruse_conv_*_bnstats: same y as the plain conv (bit-exact), sums == f64 sums of that y; T = 21 leaves a ragged
    8-frame tile, B*T tiles exceed nothing; the arena-less call clears the sums itself."""
    Cin, Fin, Cout, Fout = LEVELS[lvl]
    B, T = 3, 21
    gen = torch.Generator().manual_seed(70 + lvl)
    x = (torch.randn(B, T, Cin, Fin, generator=gen) + 0.3).cuda()
    w = (torch.randn(Cout, Cin, 2, 3, generator=gen) * 0.2).cuda()
    b = torch.randn(Cout, generator=gen).cuda()
    y0 = ops.conv_gather(x, w, b, B, T, Cin, Fin, Cout, Fout, KT=2, S=2, pad=1, prec=prec)
    y, sums = ops.conv_gather_bnstats(x, w, b, B, T, Cin, Fin, Cout, Fout, KT=2, S=2, pad=1, prec=prec)
    assert torch.equal(y, y0)
    ref = torch.cat([y.double().sum(dim=(0, 1, 3)), (y.double() ** 2).sum(dim=(0, 1, 3))])
    assert sums.numel() == ops.BN_STAT_REPLICAS * 2 * Cout           # [replica][2][Cout]: the statistic is the sum over replicas
    assert rel_l2(sums.view(-1, 2 * Cout).sum(0), ref) < 1e-6
    # decoder form: convT k maps ch[k] x F_k -> ch[k-1] x 2 F_k
    u = (torch.randn(B, T, Cout, Fout, generator=gen) - 0.2).cuda()
    wt = (torch.randn(Cout, Cin, 1, 3, generator=gen) * 0.2).cuda()
    bt = torch.randn(Cin, generator=gen).cuda()
    v0 = ops.conv_scatter2(u, wt, bt, B, T, Cout, Fout, Cin, KT=1, pad=0, prec=prec)
    v, sums = ops.conv_scatter2_bnstats(u, wt, bt, B, T, Cout, Fout, Cin, KT=1, pad=0, prec=prec)
    assert torch.equal(v, v0)
    ref = torch.cat([v.double().sum(dim=(0, 1, 3)), (v.double() ** 2).sum(dim=(0, 1, 3))])
    assert rel_l2(sums.view(-1, 2 * Cin).sum(0), ref) < 1e-6
    # the consumer folds the replicas: same mean / rstd as from a separate bn_stats pass
    gam = torch.ones(Cin).cuda(); bet = torch.zeros(Cin).cuda()
    _, m_a, r_a = ops.bn_finalize_act_fwd(v, sums, B * T * Fin, 1e-5, 0.1, gam, bet, None, B * T, Cin, Fin)
    _, m_b, r_b = ops.bn_finalize_act_fwd(v, ops.bn_stats(v, B * T, Cin, Fin), B * T * Fin, 1e-5, 0.1, gam, bet, None, B * T, Cin, Fin)
    assert rel_l2(m_a, m_b) < 1e-6 and rel_l2(r_a, r_b) < 1e-6
    with ops.ARENA.step(x.device):      # pre-zeroed arena slices, as the training step uses them
        _, s1 = ops.conv_gather_bnstats(x, w, b, B, T, Cin, Fin, Cout, Fout, KT=2, S=2, pad=1, prec=prec)
        _, s2 = ops.conv_scatter2_bnstats(u, wt, bt, B, T, Cout, Fout, Cin, KT=1, pad=0, prec=prec)
        assert rel_l2(s2.view(-1, 2 * Cin).sum(0), ref) < 1e-6 and s1.data_ptr() != s2.data_ptr()


def test_conv_shape_errors(ops):
    x = torch.zeros(1, 4, 1, 160).cuda(); w = torch.zeros(8, 1, 2, 3).cuda()
    with pytest.raises(RuntimeError, match="Fout"):
        ops.conv_gather(x, w, None, 1, 4, 1, 160, 8, 90, KT=2, S=2, pad=1)


# ------------------------------------------------------------------ BatchNorm / LayerNorm
@pytest.mark.parametrize("C,Fq", [(8, 80), (64, 10), (1, 160)])
def test_batchnorm_relu_skip_fwd_bwd(ops, C, Fq):
    B, T = 3, 17
    rows = B * T
    gen = torch.Generator().manual_seed(40 + C)
    y = torch.randn(B, T, C, Fq, generator=gen) * 2 + 0.7
    skip = torch.randn(B, T, C, Fq, generator=gen)
    bn = torch.nn.BatchNorm2d(C)
    with torch.no_grad():
        bn.weight.copy_(1 + 0.3 * torch.randn(C, generator=gen)); bn.bias.copy_(0.2 * torch.randn(C, generator=gen))
    bn.train()
    yr = _nchw(y).requires_grad_(True)
    out_ref = torch.relu(bn(yr)) + _nchw(skip)
    rm = torch.zeros(C).cuda(); rv = torch.ones(C).cuda()
    sums = ops.bn_stats(y.cuda(), rows, C, Fq)
    mean, rstd = ops.bn_finalize(sums, rows * Fq, C, 1e-5, 0.1, rm, rv)
    assert max_abs(rm, bn.running_mean) < 1e-6 and max_abs(rv, bn.running_var) < 1e-5
    out = ops.bn_act_fwd(y.cuda(), mean, rstd, bn.weight.detach().cuda(), bn.bias.detach().cuda(), skip.cuda(), rows, C, Fq)
    assert rel_l2(_nchw(out), out_ref) < 1e-5
    dout = torch.randn(B, T, C, Fq, generator=gen)
    out_ref.backward(_nchw(dout))
    dg = torch.zeros(C).cuda(); db = torch.zeros(C).cuda()
    dy = ops.bn_act_bwd(dout.cuda(), y.cuda(), mean, rstd, bn.weight.detach().cuda(), bn.bias.detach().cuda(), rows, C, Fq,
                        True, True, dg, db)
    assert rel_l2(_nchw(dy), yr.grad) < 2e-5
    assert rel_l2(dg, bn.weight.grad) < 2e-5 and rel_l2(db, bn.bias.grad) < 2e-5
    # with batch statistics the conv-bias gradient sum(dy) is exactly 0 (the mean subtraction cancels the bias): the
    # library reports the closed form, the data sum is rounding noise
    dbz = torch.zeros(C).cuda()
    ops.bn_act_bwd(dout.cuda(), y.cuda(), mean, rstd, bn.weight.detach().cuda(), bn.bias.detach().cuda(), rows, C, Fq,
                   True, True, torch.zeros(C).cuda(), torch.zeros(C).cuda(), dbias=dbz)
    assert float(dbz.abs().max()) == 0.0
    assert float(dy.double().sum(dim=(0, 1, 3)).abs().max()) < 1e-4 * float(dy.double().abs().sum(dim=(0, 1, 3)).max())
    # eval mode uses the running statistics
    bn.eval()
    m2, r2 = ops.bn_eval_stats(rm, rv, 1e-5)
    out_e = ops.bn_act_fwd(y.cuda(), m2, r2, bn.weight.detach().cuda(), bn.bias.detach().cuda(), None, rows, C, Fq)
    assert rel_l2(_nchw(out_e), torch.relu(bn(_nchw(y)))) < 1e-5
    # the fused conv-bias gradient is the per-channel sum of dy (non-trivial in eval mode)
    dbias = torch.zeros(C).cuda()
    dy_e = ops.bn_act_bwd(dout.cuda(), y.cuda(), m2, r2, bn.weight.detach().cuda(), bn.bias.detach().cuda(), rows, C, Fq,
                          True, False, None, None, dbias=dbias)
    assert rel_l2(dbias, dy_e.double().sum(dim=(0, 1, 3)).float()) < 1e-5


@pytest.mark.parametrize("H,g", [(640, 1), (640, 4), (1024, 2), (96, 3), (1024, 1), (100, 1)])
def test_layernorm_interleave_fwd_bwd(ops, H, g):
    rows = 37
    gen = torch.Generator().manual_seed(50 + g)
    x = torch.randn(rows, H, generator=gen) * 1.5 + 0.3
    res = torch.randn(rows, H, generator=gen)
    ln = torch.nn.LayerNorm(H)
    with torch.no_grad():
        ln.weight.copy_(1 + 0.3 * torch.randn(H, generator=gen)); ln.bias.copy_(0.2 * torch.randn(H, generator=gen))
    xr = x.clone().requires_grad_(True)
    inter = torch.stack(torch.chunk(xr, g, dim=-1), dim=-1).flatten(-2, -1)      # cruse_net.py:43-45
    y_ref = ln(inter) + res
    y, mean, rstd = ops.ln_fwd(x.cuda(), ln.weight.detach().cuda(), ln.bias.detach().cuda(), res.cuda(), rows, H, g)
    assert rel_l2(y, y_ref) < 1e-5
    dy = torch.randn(rows, H, generator=gen)
    y_ref.backward(dy)
    dgm = torch.zeros(H).cuda(); dbt = torch.zeros(H).cuda()
    dx = ops.ln_bwd(dy.cuda(), x.cuda(), mean, rstd, ln.weight.detach().cuda(), rows, H, g, dgm, dbt)
    assert rel_l2(dx, xr.grad) < 2e-5
    assert rel_l2(dgm, ln.weight.grad) < 2e-5 and rel_l2(dbt, ln.bias.grad) < 2e-5


# ------------------------------------------------------------------ GEMM
@pytest.mark.parametrize("prec,tol", [("f32", 2e-6), ("bf16x3", 3e-5), ("bf16", 1e-2)])
def test_gemm_all_layouts(ops, prec, tol):
    gen = torch.Generator().manual_seed(60)
    for (M, N, K) in ((200, 130, 96), (128, 128, 32), (37, 5, 7), (300, 1920, 640)):
        A = torch.randn(M, K, generator=gen); Bm = torch.randn(K, N, generator=gen); bias = torch.randn(N, generator=gen)
        ref = A.double() @ Bm.double()
        for ta in (False, True):
            for tb in (False, True):
                Ad = (A.t().contiguous() if ta else A).cuda(); Bd = (Bm.t().contiguous() if tb else Bm).cuda()
                C = torch.empty(M, N).cuda()
                ops.gemm(ta, tb, M, N, K, Ad, 0, Ad.shape[1], Bd, 0, Bd.shape[1], C, 0, N, bias=bias.cuda(), prec=prec)
                assert rel_l2(C, ref + bias.double()) < tol, (M, N, K, ta, tb)
    # asymmetric identity check (transpose-detecting): A = I, B asymmetric
    I = torch.eye(64); Bm = torch.arange(64 * 48, dtype=torch.float32).view(64, 48) / 100
    C = torch.empty(64, 48).cuda()
    ops.gemm(False, False, 64, 48, 64, I.cuda(), 0, 64, Bm.cuda(), 0, 48, C, 0, 48, prec="f32")
    assert max_abs(C, Bm) < 1e-6


def test_gemm_splitk_accumulate_shift_and_strides(ops):
    gen = torch.Generator().manual_seed(61)
    T, B, Hh, G3 = 7, 3, 64, 96
    rows = B * T
    dgh = torch.randn(rows, 2 * G3, generator=gen)       # two groups side by side
    h = torch.randn(rows, 2 * Hh, generator=gen)
    hprev = torch.zeros_like(h)
    hv = h.view(B, T, -1); hprev.view(B, T, -1)[:, 1:] = hv[:, :-1]
    for grp in range(2):
        init = torch.randn(G3, Hh, generator=gen)
        C = init.clone().cuda()
        ops.gemm(True, False, G3, Hh, rows, dgh.cuda(), grp * G3, 2 * G3, h.cuda(), grp * Hh, 2 * Hh, C, 0, Hh,
                 accumulate=True, splitk=4, b_shift_T=T, prec="f32")
        ref = init.double() + dgh[:, grp * G3:(grp + 1) * G3].double().t() @ hprev[:, grp * Hh:(grp + 1) * Hh].double()
        assert rel_l2(C, ref) < 1e-5, grp
    out = torch.zeros(2 * G3).cuda()
    ops.col_sum(dgh.cuda(), G3, rows, G3, 2 * G3, out[G3:])
    assert rel_l2(out[G3:], dgh[:, G3:].sum(0)) < 1e-5 and float(out[:G3].abs().max()) == 0.0


# ------------------------------------------------------------------ GRU recurrence
def _gru_ref(x, grus):
    outs = [m(c)[0] for m, c in zip(grus, torch.chunk(x, len(grus), dim=-1))]
    return torch.cat(outs, dim=-1)


@pytest.mark.parametrize("H,g,B,T,prec,tol", [
    (640, 1, 2, 21, "f32", 2e-5), (640, 4, 5, 9, "f32", 2e-5), (640, 2, 3, 7, "bf16x3", 1e-4),
    (640, 1, 9, 12, "bf16", 3e-2), (1024, 2, 2, 5, "f32", 2e-5), (640, 1, 20, 6, "f32", 2e-5)])
def test_gru_sequence_fwd_bwd(ops, H, g, B, T, prec, tol):
    Hg = H // g
    rows = B * T
    torch.manual_seed(70 + g)
    grus = [torch.nn.GRU(Hg, Hg, 1, batch_first=True) for _ in range(g)]
    x = torch.randn(B, T, H)
    xr = x.clone().requires_grad_(True)
    y_ref = _gru_ref(xr, grus)
    gi = torch.empty(B, T, g * 3 * Hg).cuda()
    dev = lambda p: p.detach().cuda().contiguous()
    for i, m in enumerate(grus):
        ops.gemm(False, True, rows, 3 * Hg, Hg, x.cuda(), i * Hg, H, dev(m.weight_ih_l0), 0, Hg, gi, i * 3 * Hg, 3 * H,
                 bias=dev(m.bias_ih_l0), prec=prec)
    w_hh = [dev(m.weight_hh_l0) for m in grus]; b_hh = [dev(m.bias_hh_l0) for m in grus]
    h, coef, an, z = ops.gru_seq_fwd(gi, w_hh, b_hh, B, T, g, Hg, prec)
    torch.cuda.synchronize()
    assert ops.gru_status() == 0, "recurrence hand-off timed out"
    assert rel_l2(h, y_ref) < tol
    dout = torch.randn(B, T, H)
    y_ref.backward(dout)
    dh = ops.gru_seq_bwd(dout.cuda(), w_hh, coef, z, B, T, g, Hg, prec)
    dgi, dgh = ops.gru_gate_grads(dh, coef, an, rows, g, Hg, prec)
    torch.cuda.synchronize()
    assert ops.gru_status() == 0
    for i, m in enumerate(grus):
        dwhh = torch.zeros(3 * Hg, Hg).cuda(); dwih = torch.zeros(3 * Hg, Hg).cuda()
        ops.gemm(True, False, 3 * Hg, Hg, rows, dgh, i * 3 * Hg, 3 * H, h, i * Hg, H, dwhh, 0, Hg, accumulate=True,
                 splitk=2, b_shift_T=T, prec=prec)
        ops.gemm(True, False, 3 * Hg, Hg, rows, dgi, i * 3 * Hg, 3 * H, x.cuda(), i * Hg, H, dwih, 0, Hg, accumulate=True,
                 splitk=2, prec=prec)
        assert rel_l2(dwhh, m.weight_hh_l0.grad) < 5 * tol, i
        assert rel_l2(dwih, m.weight_ih_l0.grad) < 5 * tol, i
        db = torch.zeros(3 * Hg).cuda()
        ops.col_sum(dgh, i * 3 * Hg, rows, 3 * Hg, 3 * H, db)
        assert rel_l2(db, m.bias_hh_l0.grad) < 5 * tol
        dx = torch.empty(B, T, Hg).cuda()
        ops.gemm(False, False, rows, Hg, 3 * Hg, dgi, i * 3 * Hg, 3 * H, dev(m.weight_ih_l0), 0, Hg, dx, 0, Hg, prec=prec)
        assert rel_l2(dx, xr.grad[..., i * Hg:(i + 1) * Hg]) < 5 * tol


def test_gru_full_length_status(ops):
    """T = 401, B = 64 (BASELINE config 2 shape): finishes, no hand-off time-out, outputs bounded."""
    B, T, H = 64, 401, 640
    torch.manual_seed(71)
    gi = (0.5 * torch.randn(B, T, 3 * H)).cuda()
    w = (torch.randn(3 * H, H) / 25).cuda(); b = torch.zeros(3 * H).cuda()
    h, coef, an, z = ops.gru_seq_fwd(gi, [w], [b], B, T, 1, H, "bf16")
    torch.cuda.synchronize()
    assert ops.gru_status() == 0
    assert torch.isfinite(h).all() and float(h.abs().max()) <= 1.0
    # independence of chains: clip 3 alone gives the same rows (bit-exact: same tiles, same order)
    h1, *_ = ops.gru_seq_fwd(gi[3:4].contiguous(), [w], [b], 1, T, 1, H, "bf16")
    assert rel_l2(h1[0], h[3]) < 1e-6
    # backward at full length: finishes, finite, and the chains are independent as well
    dout = (0.1 * torch.randn(B, T, H)).cuda()
    dh = ops.gru_seq_bwd(dout, [w], coef, z, B, T, 1, H, "bf16")
    torch.cuda.synchronize()
    assert ops.gru_status() == 0 and torch.isfinite(dh).all()
    dh5 = ops.gru_seq_bwd(dout[5:6].contiguous(), [w], coef[5:6].contiguous(), z[5:6].contiguous(), 1, T, 1, H, "bf16")
    assert rel_l2(dh5[0], dh[5]) < 1e-6


# ------------------------------------------------------------------ mask + loss, Adam
def test_mask_loss_against_reference_wo_male(ops, golden):
    from oracle import cruse_oracle as O
    gen = torch.Generator().manual_seed(80)
    B, T, Fn, Fs = 2, 21, 160, 161
    mask = torch.rand(B, 1, T, Fn, generator=gen).requires_grad_(True)
    nre = torch.randn(B, 1, T, Fs, generator=gen) * 0.3; nim = torch.randn(B, 1, T, Fs, generator=gen) * 0.3
    cre = torch.randn(B, 1, T, Fs, generator=gen) * 0.2; cim = torch.randn(B, 1, T, Fs, generator=gen) * 0.2
    est = O.masking(mask, nre, nim).permute(0, 3, 1, 2)
    loss_ref = O.wo_male(torch.cat([cre, cim], 1), est, torch.cat([nre, nim], 1))
    loss_ref.backward()
    cmag = torch.sqrt(cre ** 2 + cim ** 2)
    ls, dmask, dlogit, er, ei = ops.mask_loss(mask.detach().cuda().view(B * T, Fn), nre.cuda().view(B * T, Fs),
                                              nim.cuda().view(B * T, Fs), cmag.cuda().view(B * T, Fs), B * T, Fn, Fs,
                                              want_dmask=True, want_dlogit=True, want_est=True)
    assert abs(float(ls) / (B * T * Fs) - float(loss_ref)) < 1e-6 * max(1.0, abs(float(loss_ref)))
    assert rel_l2(dmask.view(B, 1, T, Fn), mask.grad) < 1e-5
    m = mask.detach()
    assert rel_l2(dlogit.view(B, 1, T, Fn), mask.grad * m * (1 - m)) < 1e-5
    assert float(er.view(B, T, Fs)[..., 160].abs().max()) == 0.0       # R8: bin 160 := 0
    assert rel_l2(er.view(B, T, Fs)[..., :160], (m * nre[..., :160]).squeeze(1)) < 1e-6


def test_sdnr_loss_against_oracle(ops):
    """sdnr (loss_func/loss.py:151-175, vad == 1): alpha curve of test/test_loss.py:33-51 at several SNRs."""
    from oracle import cruse_oracle as O
    gen = torch.Generator().manual_seed(81)
    B, T, Fn, Fs = 3, 11, 160, 161
    cre = torch.randn(B, T, Fs, generator=gen) * 0.2; cim = torch.randn(B, T, Fs, generator=gen) * 0.2
    vre = torch.randn(B, T, Fs, generator=gen) * 0.3; vim = torch.randn(B, T, Fs, generator=gen) * 0.3
    for snr in (-5.0, 5.0, 20.0):
        mask = torch.rand(B, T, Fn, generator=gen).requires_grad_(True)
        g = torch.nn.functional.pad(mask, (0, Fs - Fn)).unsqueeze(1)                       # [B,1,T,Fs] gain
        clean = torch.stack([cre, cim], 1); noise = torch.stack([vre, vim], 1)                # [B,2,T,Fs]
        ref = O.sdnr(clean, g, noise, snr, beta=20.0)
        ref.backward()
        ls, dmask, _ = ops.mask_sdnr(mask.detach().cuda().view(B * T, Fn), cre.cuda().view(B * T, Fs), cim.cuda().view(B * T, Fs),
                                     (cre + vre).cuda().view(B * T, Fs), (cim + vim).cuda().view(B * T, Fs), B * T, Fn, Fs, B, snr,
                                     20.0, want_dmask=True)
        assert abs(float(ls) / (B * Fs) - float(ref)) <= 1e-5 * abs(float(ref)), snr
        assert rel_l2(dmask.view(B, T, Fn), mask.grad) < 1e-5, snr


def test_adam_matches_torch(ops):
    gen = torch.Generator().manual_seed(90)
    p0 = torch.randn(1000, generator=gen)
    p = torch.nn.Parameter(p0.clone())
    opt = torch.optim.Adam([p], lr=1e-2, betas=(0.9, 0.99), eps=1e-8)
    pd = p0.clone().cuda(); m = torch.zeros(1000).cuda(); v = torch.zeros(1000).cuda()
    for step in range(1, 4):
        g = torch.randn(1000, generator=gen)
        p.grad = g.clone(); opt.step()
        ops.adam_step(pd, (2 * g).cuda(), m, v, 1e-2, 0.9, 0.99, 1e-8, 0.0, step, grad_scale=0.5)
    assert max_abs(pd, p.detach()) < 1e-6


# ------------------------------------------------------------------ bf16-operand GEMM path (CRUSE_PREC_BF16)
@pytest.mark.parametrize("M,N,K,sk", [(300, 200, 128, 1), (128, 128, 64, 1), (1920, 640, 1088, 4), (65, 33, 256, 2)])
def test_gemm_bf16_nt(ops, M, N, K, sk):
    gen = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(M, K + 64, generator=gen).cuda().to(torch.bfloat16)          # lda > K
    Bm = torch.randn(N, K, generator=gen).cuda().to(torch.bfloat16)
    bias = torch.randn(N, generator=gen).cuda()
    ref = A[:, :K].double() @ Bm.double().t()
    C = torch.full((M, N + 4), 7.0).cuda()                                      # ldc > N: the margin must survive
    ops.gemm_bf16_nt(M, N, K, A, 0, K + 64, Bm, 0, K, C, 0, N + 4, bias=bias)
    assert rel_l2(C[:, :N], ref + bias.double()) < 1e-6 and (C[:, N:] == 7.0).all()
    C0 = torch.randn(M, N + 4, generator=gen).cuda()
    C = C0.clone()
    ops.gemm_bf16_nt(M, N, K, A, 0, K + 64, Bm, 0, K, C, 0, N + 4, accumulate=True, splitk=sk)
    assert rel_l2(C[:, :N], ref + C0[:, :N].double()) < 1e-6 and torch.equal(C[:, N:], C0[:, N:])


def test_bf16_layout_kernels(ops):
    B, T, H = 3, 17, 96
    rows = B * T
    gen = torch.Generator().manual_seed(5)
    x = torch.randn(rows, H, generator=gen).cuda()
    assert torch.equal(ops.cast_bf16(x), x.to(torch.bfloat16))
    untile = lambda y: y.permute(1, 0, 2).reshape(y.shape[1], -1)          # [kt, lines, 64] -> [lines, ldT]
    xt = ops.transpose_bf16(x, rows, H)
    assert xt.shape == (1, H, 64)
    xt = untile(xt)
    assert torch.equal(xt[:, :rows], x.to(torch.bfloat16).t()) and xt[:, rows:].abs().max() == 0
    xs = untile(ops.transpose_bf16(x, rows, H, shift_T=T))
    ref = torch.zeros(B, T, H).cuda(); ref[:, 1:] = x.view(B, T, H)[:, :-1]
    assert torch.equal(xs[:, :rows], ref.view(rows, H).to(torch.bfloat16).t()) and xs[:, rows:].abs().max() == 0
    # several k-tiles + a GEMM on the K-tiled operands: C = x^T x
    x2 = torch.randn(200, 160, generator=gen).cuda()
    t2 = ops.transpose_bf16(x2, 200, 160)
    assert t2.shape == (4, 160, 64) and torch.equal(untile(t2)[:, :200], x2.to(torch.bfloat16).t())
    C = torch.zeros(160, 160).cuda()
    ops.gemm_bf16_nt(160, 160, 256, t2, 0, 64, t2, 0, 64, C, 0, 160, accumulate=True, splitk=2, a_kstride=160 * 64,
                     b_kstride=160 * 64)
    xb = x2.to(torch.bfloat16).double()
    assert rel_l2(C, xb.t() @ xb) < 1e-6


@pytest.mark.parametrize("G,Hg", [(1, 128), (2, 96), (4, 160)])
def test_gru_gate_grads_bf16_matches_f32_form(ops, G, Hg):
    rows, H = 3 * 37, G * Hg
    gen = torch.Generator().manual_seed(G * 100 + Hg)
    dh = torch.randn(rows, H, generator=gen).cuda()
    an = torch.rand(rows, H, generator=gen).cuda()
    coef = torch.randn(rows, G, 3, Hg, generator=gen).cuda().to(torch.bfloat16)
    dgi_ref, dgh_ref = ops.gru_gate_grads(dh, coef, an, rows, G, Hg, "bf16")   # f32 [rows,G,3,Hg]
    db_ih = [torch.ones(3 * Hg).cuda() for _ in range(G)]
    db_hh = [torch.ones(3 * Hg).cuda() for _ in range(G)]
    dgi, dgT, ldT = ops.gru_gate_grads_bf16(dh, coef, an, rows, G, Hg, db_ih, db_hh)
    assert ldT == 128 and torch.equal(dgi, dgi_ref.view(rows, G, 3, Hg).to(torch.bfloat16))
    slabs = torch.cat([dgi_ref.view(rows, G, 3, Hg), dgh_ref.view(rows, G, 3, Hg)[:, :, 2:]], dim=2)   # r, z, n_i, n_h
    dgT = dgT.permute(1, 2, 3, 0, 4).reshape(G, 4, Hg, ldT)                 # [kt,G,4,Hg,64] -> [G,4,Hg,ldT]
    assert torch.equal(dgT[..., :rows], slabs.permute(1, 2, 3, 0).to(torch.bfloat16)) and dgT[..., rows:].abs().max() == 0
    for g in range(G):
        assert rel_l2(db_ih[g] - 1, dgi_ref.view(rows, G, 3 * Hg)[:, g].double().sum(0)) < 1e-5
        assert rel_l2(db_hh[g] - 1, dgh_ref.view(rows, G, 3 * Hg)[:, g].double().sum(0)) < 1e-5


@pytest.mark.parametrize("H,B,T", [(640, 9, 12), (128, 8, 7), (256, 3, 5), (384, 16, 6), (512, 1, 4), (160, 10, 8),
                                   (96, 4, 6), (32, 8, 5), (288, 7, 5),
                                   (640, 8, 1), (640, 5, 2), (640, 8, 3), (320, 3, 1), (320, 11, 2)])   # loader / helper wave prologues
def test_gru_bwd_reduce_scatter_matches_all_gather_form(ops, H, B, T):
    """bf16 mode has two backward recurrence kernels (gru.hip): both must give the same dh from the same inputs."""
    import os
    torch.manual_seed(H + B)
    gi = (0.5 * torch.randn(B, T, 3 * H)).cuda()
    w = [(torch.randn(3 * H, H) / H ** 0.5).cuda()]; b = [torch.zeros(3 * H).cuda()]
    h, coef, an, z = ops.gru_seq_fwd(gi, w, b, B, T, 1, H, "bf16")
    dout = torch.randn(B, T, H).cuda()
    dh_rs = ops.gru_seq_bwd(dout, w, coef, z, B, T, 1, H, "bf16")
    with ops.options(gru_bwd_rs=0):
        dh_ag = ops.gru_seq_bwd(dout, w, coef, z, B, T, 1, H, "bf16")
    torch.cuda.synchronize()
    assert ops.gru_status() == 0
    assert torch.isfinite(dh_rs).all() and rel_l2(dh_rs, dh_ag) < 5e-3
    # and against the f32-precision reference of the same recurrence (autograd of the closed form)
    c = coef.float().view(B, T, 3, H)
    wd = w[0].double()
    ref = torch.zeros(B, T, H, dtype=torch.float64, device="cuda")
    nxt = torch.zeros(B, H, dtype=torch.float64, device="cuda")
    for s in range(T - 1, -1, -1):
        cur = dout[:, s].double()
        if s < T - 1:
            dgh = (nxt.unsqueeze(1) * c[:, s + 1].double()).reshape(B, 3 * H)
            cur = cur + z[:, s + 1].double() * nxt + dgh @ wd
        ref[:, s] = cur
        nxt = cur
    assert rel_l2(dh_rs, ref) < 1e-2 and rel_l2(dh_ag, ref) < 1e-2


@pytest.mark.parametrize("H,B,T", [(640, 9, 12), (128, 8, 7), (256, 3, 5), (384, 16, 6), (512, 1, 4), (160, 10, 8),
                                   (96, 4, 6), (32, 8, 5), (288, 7, 5),
                                   (640, 8, 1), (640, 5, 2), (640, 8, 3), (320, 3, 1), (320, 11, 2)])   # loader / helper wave prologues
def test_gru_fwd_lean_matches_generic_kernel(ops, H, B, T):
    """bf16 mode has two forward recurrence kernels (gru.hip): same algorithm, so the same h / coefficients."""
    import os
    torch.manual_seed(H + B + 1)
    gi = (0.5 * torch.randn(B, T, 3 * H)).cuda()
    w = [(torch.randn(3 * H, H) / H ** 0.5).cuda()]; b = [(0.1 * torch.randn(3 * H)).cuda()]
    with ops.options(gru_wlo=0):                   # same algorithm == without the lean kernel's W_hh low-plane pass
        lean = ops.gru_seq_fwd(gi, w, b, B, T, 1, H, "bf16")
        h_nosave = ops.gru_seq_fwd(gi, w, b, B, T, 1, H, "bf16", save=False)[0]
        with ops.options(gru_fwd_lean=0):
            gen = ops.gru_seq_fwd(gi, w, b, B, T, 1, H, "bf16")
    torch.cuda.synchronize()
    assert ops.gru_status() == 0
    for x, y, name in zip(lean, gen, ("h", "coef", "an", "z")):
        # same algorithm, but the lean kernel's gates use bare v_rcp/v_exp (1 ulp) and h travels as bf16: a last-bit
        # difference can flip one bf16 rounding of the exchanged state, so "equal" means well inside bf16 resolution
        tol = 2e-3 if x.dtype == torch.bfloat16 else 1e-4
        assert torch.isfinite(x.float()).all() and rel_l2(x.float(), y.float()) < tol, name
    assert rel_l2(h_nosave, lean[0]) < 1e-6                # same kernel, saves off: identical


@pytest.mark.parametrize("form,Cin,Cout,Fin,accum", [("gather", 16, 32, 40, False), ("gather", 1, 8, 161, False), ("gather", 32, 64, 20, False),
                                                     ("scatter2", 64, 32, 10, True), ("scatter2", 16, 8, 40, True), ("scatter2", 32, 16, 20, False)])
def test_conv_dgrad_accumulates_the_batchnorm_backward_sums(ops, form, Cin, Cout, Fin, accum):
    """cruse_conv_gather_bnbwd / cruse_conv_scatter2_bnbwd: the data-gradient conv's epilogue accumulates the backward sums of the
    BatchNorm(+ReLU) its output is the incoming gradient of -- bn_act_bwd with those sums == bn_act_bwd with its own reduce
    pass over (dout, y); same conv output.  (Cin = 1: the VALU conv + the reduce pass inside the library.)"""
    torch.manual_seed(Cin * 7 + Cout)
    B, T = 3, 21
    prec = "bf16"
    if form == "gather":                                     # decoder data gradient: KT = 1, stride 2 (Fout = Fin // 2)
        Fout = Fin // 2
        x = torch.randn(B, T, Cin, Fin).cuda()
        w = (0.2 * torch.randn(Cout, Cin, 1, 3)).cuda()
        run = lambda bn, out: ops.conv_gather(x, w, None, B, T, Cin, Fin, Cout, Fout, KT=1, S=2, pad=0, out=out, accum=accum, prec=prec,
                                              bn_bwd=bn)
        C, F = Cout, Fout
    else:
        g = torch.randn(B, T, Cin, Fin).cuda()
        w = (0.2 * torch.randn(Cin, Cout, 2, 3)).cuda()
        C, F = Cout, 2 * Fin
        run = lambda bn, out: ops.conv_scatter2(g, w, None, B, T, Cin, Fin, Cout, KT=2, pad=1, out=out, accum=accum, prec=prec, bn_bwd=bn)
    rows = B * T
    y = torch.randn(B, T, C, F).cuda()                       # pre-BN tensor of the BatchNorm behind the conv output
    mean = (0.1 * torch.randn(C)).cuda(); rstd = (1.0 + 0.2 * torch.rand(C)).cuda()
    gamma = (1.0 + 0.3 * torch.randn(C)).cuda(); beta = (0.2 * torch.randn(C)).cuda()
    base = torch.randn(B, T, C, F).cuda()
    out0 = run(None, base.clone() if accum else None)
    out1, sums = run((y, mean, rstd, gamma, beta, True), base.clone() if accum else None)
    assert torch.equal(out0, out1)
    res = []
    for sm in (None, sums):
        dg = torch.zeros(C).cuda(); db = torch.zeros(C).cuda()
        dy = ops.bn_act_bwd(out0, y, mean, rstd, gamma, beta, rows, C, F, True, True, dg, db, sums=sm)
        res.append((dy, dg, db))
    for a, b, name in zip(res[1], res[0], ("dy", "dgamma", "dbeta")):
        assert rel_l2(a, b) < 2e-6, name


@pytest.mark.parametrize("H,B,T,G", [(640, 64, 9, 1), (640, 24, 7, 1), (128, 16, 5, 1), (384, 5, 6, 1), (256, 33, 4, 2), (512, 17, 1, 1)])
def test_gru_fwd_wide_chains_match_lean_kernel_bit_for_bit(ops, H, B, T, G):
    """cruse_gru_seq_fwd_ex(chain_clips = 16): chains of 16 clips (half the workgroups per clip; gru_fwd_w16_kernel, gru_w16.hip) -- the
    lean kernel's arithmetic in the lean kernel's summation order, so h and the saved coefficient rows are identical; run as time
    chunks too (a continuation takes its state from the h rows the previous launch wrote)."""
    torch.manual_seed(H + B)
    Hg = H // G
    gi = (0.5 * torch.randn(B, T, 3 * H)).cuda()
    w = [(torch.randn(3 * Hg, Hg) / Hg ** 0.5).cuda() for _ in range(G)]; b = [(0.1 * torch.randn(3 * Hg)).cuda() for _ in range(G)]
    # (gru_wlo = 0: the lean kernel's W_hh low-plane pass for Hg <= 320 has no wide-chain form)
    with ops.options(gru_wlo=0):
        lean = ops.gru_seq_fwd(gi, w, b, B, T, G, Hg, "bf16")
        wide = ops.gru_seq_fwd(gi, w, b, B, T, G, Hg, "bf16", wide=True)
        for x, y, name in zip(wide, lean, ("h", "coef", "an", "z")):
            assert torch.equal(x, y), name
        nosave = ops.gru_seq_fwd(gi, w, b, B, T, G, Hg, "bf16", wide=True, save=False)
        assert torch.equal(nosave[0], lean[0])
        if T >= 4:
            cuts = [(0, T // 2), (T // 2, T - T // 2)] if T < 7 else [(0, 2), (2, 1), (3, T - 5), (T - 2, 2)]
            out = None
            for c in cuts:                                   # every chunk clears its own scratch
                out = ops.gru_seq_fwd(gi, w, b, B, T, G, Hg, "bf16", out=out, chunk=c, wide=True)
            for x, y, name in zip(out, lean, ("h", "coef", "an", "z")):
                assert torch.equal(x, y), ("chunked", name)
    with pytest.raises(RuntimeError):
        ops.gru_seq_fwd(gi, w, b, B, T, G, Hg, "bf16", h0=torch.zeros(B, H).cuda(), wide=True)
    assert ops.gru_status() == 0


@pytest.mark.parametrize("H,B,T,G,slabs", [(640, 64, 9, 1, 4), (640, 24, 7, 1, 3), (128, 16, 5, 1, 4), (384, 5, 6, 1, 3), (256, 33, 4, 2, 4),
                                            (512, 17, 1, 1, 3)])
def test_gru_bwd_wide_chains_match_the_chains_of_8(ops, H, B, T, G, slabs):
    """cruse_gru_seq_bwd_ex(chain_clips = 16): chains of 16 clips (gru_bwd_w16_kernel: the all-gather step on eight compute waves) --
    dh and the in-kernel gate gradients (3 or 4 slabs) equal those of the chains of 8 up to the f32 summation order / one bf16 rounding
    of the exchanged products, and the exact-f32 recurrence within the bf16 mode's error; run as time chunks (last chunk first, the
    gradient carried across the cut): bit-identical to the one launch."""
    torch.manual_seed(H + B)
    Hg = H // G
    gi = (0.5 * torch.randn(B, T, 3 * H)).cuda()
    w = [(torch.randn(3 * Hg, Hg) / Hg ** 0.5).cuda() for _ in range(G)]; b = [(0.1 * torch.randn(3 * Hg)).cuda() for _ in range(G)]
    dout = torch.randn(B, T, H).cuda()
    h, coef, an, z = ops.gru_seq_fwd(gi, w, b, B, T, G, Hg, "bf16")
    ref = ops.gru_seq_bwd(dout, w, coef, z, B, T, G, Hg, "bf16", an=an, want_dgi=True, dg_slabs=slabs)
    wide = ops.gru_seq_bwd(dout, w, coef, z, B, T, G, Hg, "bf16", an=an, want_dgi=True, dg_slabs=slabs, wide=True)
    assert rel_l2(wide[0], ref[0]) < 3e-3, "dh"
    assert rel_l2(wide[1].float(), ref[1].float()) < 4e-3, "dgi"
    exact = ops.gru_seq_bwd(dout, w, coef.float(), z, B, T, G, Hg, "f32")
    assert rel_l2(wide[0], exact) < 4e-3
    only_dh = ops.gru_seq_bwd(dout, w, coef, z, B, T, G, Hg, "bf16", wide=True)
    assert torch.equal(only_dh, wide[0])
    if T >= 4:
        cuts = [(0, T // 2), (T // 2, T - T // 2)] if T < 7 else [(0, 2), (2, 1), (3, T - 5), (T - 2, 2)]
        out = (torch.zeros_like(ref[0]), torch.zeros_like(ref[1]))
        for c in reversed(cuts):
            ops.gru_seq_bwd(dout, w, coef, z, B, T, G, Hg, "bf16", an=an, want_dgi=True, dg_slabs=slabs, out=out, chunk=c, wide=True)
        assert torch.equal(out[0], wide[0]), "chunked dh"
        assert torch.equal(out[1].view(torch.int16), wide[1].view(torch.int16)), "chunked dgi"
    assert ops.gru_status() == 0


@pytest.mark.parametrize("M,N,K", [(25664, 1920, 640), (2500, 200, 160), (4133, 640, 640), (130, 96, 64)])
def test_gemm_f16_single_pass_gate_projection(ops, M, N, K):
    """cruse_gemm_f16_nt + cruse_ktile_f16 (ABI 9): gi = x W_ih^T + b_ih (cruse_net.py:23-31,44,50) in one pass on IEEE-f16 operands --
    exact (f32 accumulation order aside) on the f16-rounded operands, 8 x closer to the f32 product than a bf16 pass; K not a
    multiple of 64 (zero-padded K tile, the A rows read on into their neighbours), ragged M / N tiles."""
    torch.manual_seed(M + N + K)
    kp = (K + 63) // 64 * 64
    A = torch.randn(M, K).cuda(); W = (torch.randn(N, K) / K ** 0.5).cuda(); bias = torch.randn(N).cuda()
    ld = kp + 64
    Ah = torch.full((M, ld), 3.0, dtype=torch.float16).cuda()                  # (finite garbage beside the K columns, as a neighbouring group's)
    Ah[:, :K] = A.half()
    Wt = ops.ktile_f16(W, N, K)
    assert Wt.shape == (kp // 64, N, 64)
    want_t = torch.zeros(N, kp, dtype=torch.float16).cuda(); want_t[:, :K] = W.half()
    assert torch.equal(Wt.permute(1, 0, 2).reshape(N, kp), want_t)
    C = torch.full((M, N), 7.0).cuda()
    ops.gemm_f16_nt(M, N, kp, Ah, 0, ld, Wt, 0, 64, C, 0, N, bias=bias, b_kstride=N * 64)
    ref16 = A.half().double() @ W.half().double().t() + bias.double()
    exact = A.double() @ W.double().t() + bias.double()
    assert rel_l2(C, ref16) < 2e-6
    e16 = rel_l2(C, exact)
    Cb = torch.empty(M, N).cuda()
    ops.gemm_bf16_nt(M, N, kp, Ah.to(torch.bfloat16), 0, ld, ops.ktile_bf16(W, N, K), 0, 64, Cb, 0, N, bias=bias, b_kstride=N * 64)
    assert e16 < 5e-4 and e16 < 0.2 * rel_l2(Cb, exact)
    with pytest.raises(RuntimeError):
        ops.gemm_f16_nt(M, N, kp, Ah.to(torch.bfloat16), 0, ld, Wt, 0, 64, C, 0, N)


def test_f16_operand_copies_from_the_producers(ops):
    """cruse_ln_fwd_c / cruse_bn_finalize_act_fwd_c (ABI 9): the 2-byte operand copy the next gate projection reads takes the element
    type asked for -- f16 copy == out.half(), bf16 copy == out.bfloat16(), the f32 outputs identical either way."""
    torch.manual_seed(3)
    rows, H = 777, 640
    x = torch.randn(rows, H).cuda(); gam = (1 + 0.1 * torch.randn(H)).cuda(); bet = (0.1 * torch.randn(H)).cuda()
    outs = {}
    for dt in (torch.float16, torch.bfloat16):
        cp = torch.empty(rows * H, dtype=dt).cuda()
        y, _, _ = ops.ln_fwd(x, gam, bet, None, rows, H, 1, out_bf16=cp)
        assert torch.equal(cp.view(rows, H), y.to(dt))
        outs[dt] = y
    assert torch.equal(outs[torch.float16], outs[torch.bfloat16])
    with pytest.raises(RuntimeError):                                          # the interleaving form (g > 1) has no fused f16 copy
        ops.ln_fwd(x, gam, bet, None, rows, H, 4, out_bf16=torch.empty(rows * H, dtype=torch.float16).cuda())
    C, F = 64, 10
    yb = torch.randn(rows, C, F).cuda(); g2 = (1 + 0.1 * torch.randn(C)).cuda(); b2 = (0.1 * torch.randn(C)).cuda()
    sums = torch.stack([yb.double().sum((0, 2)), (yb.double() ** 2).sum((0, 2))]).flatten().contiguous()
    outs = {}
    for dt in (torch.float16, torch.bfloat16):
        cp = torch.empty(rows * C * F, dtype=dt).cuda()
        e, _, _ = ops.bn_finalize_act_fwd(yb, sums, rows * F, 1e-5, 0.1, g2, b2, None, rows, C, F, out_bf16=cp)
        assert torch.equal(cp.view(rows, C, F), e.to(dt))
        outs[dt] = e
    assert torch.equal(outs[torch.float16], outs[torch.bfloat16])


@pytest.mark.parametrize("M,N,K,splitk", [(1920, 640, 25664, -8), (640, 640, 25664, -8), (200, 96, 4096, 6), (130, 100, 1024, -3)])
def test_gemm_split_k_slabs_match_atomics_and_are_reproducible(ops, M, N, K, splitk):
    """cruse_gemm_bf16_nt_slabs: the k-slices' partial sums go to slabs that one kernel adds to C in slice order -- the same sums as
    the atomic split-K form (up to the order of the f32 additions), on top of what C held, bit-identical from run to run; k-slices
    pinned to XCDs and dealt round-robin, ragged tiles."""
    torch.manual_seed(M + K)
    A = torch.randn(M, K).cuda().to(torch.bfloat16); Bm = torch.randn(N, K).cuda().to(torch.bfloat16)
    base = torch.randn(M, N).cuda()
    ref = base.clone()
    ops.gemm_bf16_nt(M, N, K, A, 0, K, Bm, 0, K, ref, 0, N, accumulate=True, splitk=splitk)
    outs = []
    for _ in range(3):
        c = base.clone()
        ops.gemm_bf16_nt(M, N, K, A, 0, K, Bm, 0, K, c, 0, N, accumulate=True, splitk=splitk, slabs=True)
        outs.append(c)
    assert rel_l2(outs[0], ref) < 2e-6
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    exact = base.double() + A.double() @ Bm.double().t()
    assert rel_l2(outs[0], exact) < 2e-6


@pytest.mark.parametrize("Hg,K,Ms", [(640, 25664, None), (128, 4096, None), (128, 1100, (256, 128, 70)), (160, 6000, None), (96, 2000, (70, 200, 33))])
def test_gemm_concatenated_weight_gradient_products(ops, Hg, K, Ms):
    """cruse_gemm_bf16_nt_slabs_cat (ABI 9): the three weight-gradient products of a GRU layer -- (r, z, n_i)^T x, (r, z)^T h_{t-1},
    n_h^T h_{t-1} (autograd of nn.GRU, cruse_net.py:23-31) -- as ONE launch on the concatenated output [dW_ih ; dW_hh]: bit-identical to the
    three slab launches it replaces, and right against f64; products that are not whole row tiles (the grouped GRUs: Hg = 160) and K not a
    multiple of 64 (zero frames) as well; k-slices pinned to XCDs and dealt round-robin."""
    torch.manual_seed(Hg + K)
    ldT = (K + 63) // 64 * 64
    dgT = (torch.randn(ldT // 64, 1, 4, Hg, 64) * 0.1).cuda().to(torch.bfloat16)
    xT = torch.randn(ldT // 64, Hg, 64).cuda().to(torch.bfloat16)
    hT = torch.randn(ldT // 64, Hg, 64).cuda().to(torch.bfloat16)
    if ldT != K:                                                                   # frames K <= r < ldT are zero in the time-major operands
        for t_ in (dgT.view(ldT // 64, -1, 64), xT, hT):
            t_[-1, :, K % 64:] = 0
    ka, kb = 4 * Hg * 64, Hg * 64
    Ms = Ms or (3 * Hg, 2 * Hg, Hg)
    a_rows = (0, 0, 3 * Hg)
    Bs = (xT, hT, hT)
    base = torch.randn(sum(Ms), Hg).cuda()
    for sk in (-8, 5):
        c3 = base.clone()
        off = 0
        for M, ar, Bm in zip(Ms, a_rows, Bs):
            ops.gemm_bf16_nt(M, Hg, ldT, dgT, ar * 64, 64, Bm, 0, 64, c3, off * Hg, Hg, accumulate=True, splitk=sk, slabs=True, a_kstride=ka, b_kstride=kb)
            off += M
        cc = base.clone()
        ops.gemm_bf16_nt_cat(list(Ms), Hg, ldT, dgT, list(a_rows), 64, list(Bs), 0, 64, cc, 0, Hg, sk, a_kstride=ka, b_kstride=kb)
        assert torch.equal(cc, c3), sk
    A = dgT.float().permute(0, 4, 1, 2, 3).reshape(ldT, 4 * Hg).double()          # [frame][slab * Hg + unit]
    ref, off = base.double().clone(), 0
    for M, ar, Bm in zip(Ms, a_rows, Bs):
        ref[off:off + M] += A[:, ar:ar + M].t() @ Bm.float().permute(0, 2, 1).reshape(ldT, Hg).double()
        off += M
    assert rel_l2(cc, ref) < 2e-6


@pytest.mark.parametrize("rows,G,Hg,x3", [(25664, 4, 160, 2), (25664, 4, 160, 1), (1000, 2, 320, 0), (333, 3, 96, 2)])
def test_gemm_all_groups_in_one_launch(ops, rows, G, Hg, x3):
    """cruse_gemm_bf16_nt_groups (ABI 9): the forward gate projections gi_q = x_q W_ih,q^T + b_ih,q and the input gradients dx_q = dgi_q W_ih,q of
    the G groups of a GGRU layer (cruse_net.py:14-55, rnn_groups > 1) as ONE launch each -- bit-identical to the G launches they replace;
    x3: 0 plain bf16, 1 W_ih hi / lo planes, 2 x split as well; K rounded up to 64 (the operand reads on into the next group), accumulate."""
    torch.manual_seed(rows + Hg)
    H, kp = G * Hg, (Hg + 63) // 64 * 64
    x = torch.randn(rows, H).cuda()
    Ws = [(torch.randn(3 * Hg, Hg) / Hg ** 0.5).cuda() for _ in range(G)]
    bias_all = torch.randn(G, 3 * Hg + 8).cuda()                      # (uniformly strided per-group biases, as in the flat parameter buffer)
    biases = [bias_all[q, :3 * Hg] for q in range(G)]
    assert ops.uniform_stride(biases) == 3 * Hg + 8
    x_hi, x_lo = ops.cast_bf16_padded(x, pad=64, split=True)
    if x3 < 2:
        x_lo = None
    W_hi = torch.empty(G, kp // 64, 3 * Hg, 64, dtype=torch.bfloat16).cuda()
    W_lo = torch.empty_like(W_hi) if x3 else None
    for q in range(G):
        ops.ktile_bf16(Ws[q], 3 * Hg, Hg, split=bool(x3), out=(W_hi[q], W_lo[q] if x3 else None))
    gi0 = torch.full((rows, 3 * H), 7.0).cuda(); gi1 = gi0.clone()
    for q in range(G):
        if x3:
            ops.gemm_bf16x3_nt(rows, 3 * Hg, kp, x_hi, x_lo, q * Hg, H, W_hi[q], W_lo[q], 0, 64, gi0, q * 3 * Hg, 3 * H, bias=biases[q], b_kstride=3 * Hg * 64)
        else:
            ops.gemm_bf16_nt(rows, 3 * Hg, kp, x_hi, q * Hg, H, W_hi[q], 0, 64, gi0, q * 3 * Hg, 3 * H, bias=biases[q], b_kstride=3 * Hg * 64)
    ops.gemm_bf16_nt_groups(rows, 3 * Hg, kp, G, x_hi, x_lo, H, Hg, W_hi, W_lo, 64, W_hi[0].numel(), gi1, 3 * H, 3 * Hg, bias=biases[0],
                            bias_gstep=3 * Hg + 8, b_kstride=3 * Hg * 64)
    assert torch.equal(gi0, gi1)
    ref = torch.cat([x[:, q * Hg:(q + 1) * Hg].double() @ Ws[q].double().t() + biases[q].double() for q in range(G)], 1)
    assert rel_l2(gi1, ref) < (1e-2 if x3 == 0 else 3e-3 if x3 == 1 else 3e-5)
    # input gradients: dgi [rows, G, 3, Hg] bf16 (padded buffer) against the stacked K-tiled transposes of W_ih
    dgi = ops.dgi_buffer(rows, G, Hg, "cuda")
    dgi.view(-1)[:rows * 3 * H] = (torch.randn(rows * 3 * H) * 0.1).cuda().to(torch.bfloat16)
    stack = torch.empty(G, (3 * Hg + 63) // 64, Hg, 64, dtype=torch.bfloat16).cuda()
    for q in range(G):
        ops.transpose_bf16(Ws[q], 3 * Hg, Hg, out=stack[q])
    for acc in (False, True):
        base = torch.randn(rows, H).cuda()
        d0, d1 = base.clone(), base.clone()
        for q in range(G):
            ops.gemm_bf16_nt(rows, Hg, stack.shape[1] * 64, dgi, q * 3 * Hg, 3 * H, stack[q], 0, 64, d0, q * Hg, H, accumulate=acc, b_kstride=Hg * 64)
        ops.gemm_bf16_nt_groups(rows, Hg, stack.shape[1] * 64, G, dgi, None, 3 * H, 3 * Hg, stack, None, 64, stack[0].numel(), d1, H, Hg,
                                accumulate=acc, b_kstride=Hg * 64)
        assert torch.equal(d0, d1), acc


def test_gru_wide_chains_at_the_bench_length(ops):
    """T = 401, B = 64, Hg = 640: the wide-chain forward launch (4 chains of 16 on 80 CUs) against the lean one (8 chains of 8 on
    160) over the whole sequence -- 401 dependent hand-offs per chain -- bit for bit; two wide launches side by side on the two
    XCD halves (slots 0 / 1, xcd_rot 0 / 4) as well."""
    torch.manual_seed(11)
    B, T, H = 64, 401, 640
    gi = (0.5 * torch.randn(B, T, 3 * H)).cuda()
    w = [(torch.randn(3 * H, H) / H ** 0.5).cuda()]; b = [(0.1 * torch.randn(3 * H)).cuda()]
    lean = ops.gru_seq_fwd(gi, w, b, B, T, 1, H, "bf16")           # (Hg = 640: the lean kernel on the tag-free hand-off -- same sums)
    wide = ops.gru_seq_fwd(gi, w, b, B, T, 1, H, "bf16", wide=True)
    for x, y, name in zip(wide, lean, ("h", "coef", "an", "z")):
        assert torch.equal(x, y), name
    s2 = torch.cuda.Stream()
    s2.wait_stream(torch.cuda.current_stream())
    a1 = ops.gru_seq_fwd(gi, w, b, B, T, 1, H, "bf16", wide=True, slot=0, xcd_rot=0)
    with torch.cuda.stream(s2):
        a2 = ops.gru_seq_fwd(gi, w, b, B, T, 1, H, "bf16", wide=True, slot=1, xcd_rot=4)
    torch.cuda.current_stream().wait_stream(s2)
    torch.cuda.synchronize()
    assert torch.equal(a1[0], lean[0]) and torch.equal(a2[0], lean[0])
    assert ops.gru_status() == 0


@pytest.mark.parametrize("B", [136, 104])
def test_gru_batch_beyond_the_cu_count(ops, B):
    """Hg = 640, B > 96: the chains of 8 x 20 workgroups do not fit the 256 CUs; make_plan then takes WIDE chains (gru_w16.hip): B = 136 --
    two launches of 8 + 1 chains of 16 (the last chain with 8 clips) instead of three on chains of 8; B = 104 -- one wide launch each
    way.  Same results as the batch run in two parts on chains of 8: forward bit for bit (same sums in the same order), backward up to
    the f32 summation order."""
    torch.manual_seed(3)
    T, H = 6, 640
    gi = (0.5 * torch.randn(B, T, 3 * H)).cuda()
    w = [(torch.randn(3 * H, H) / H ** 0.5).cuda()]; b = [(0.1 * torch.randn(3 * H)).cuda()]
    dout = torch.randn(B, T, H).cuda()
    full = ops.gru_seq_fwd(gi, w, b, B, T, 1, H, "bf16")
    dh, dgi = ops.gru_seq_bwd(dout, w, full[1], full[3], B, T, 1, H, "bf16", an=full[2], want_dgi=True)
    for lo, hi in ((0, 64), (64, B)):
        part = ops.gru_seq_fwd(gi[lo:hi].contiguous(), w, b, hi - lo, T, 1, H, "bf16")
        for x, y, name in zip(part, full, ("h", "coef", "an", "z")):
            assert torch.equal(x, y[lo:hi]), name
        dh_p, dgi_p = ops.gru_seq_bwd(dout[lo:hi].contiguous(), w, part[1], part[3], hi - lo, T, 1, H, "bf16", an=part[2], want_dgi=True)
        assert rel_l2(dh_p, dh[lo:hi]) < 3e-3 and rel_l2(dgi_p.float(), dgi[lo * T:hi * T].view(dgi_p.shape).float()) < 4e-3
    assert ops.gru_status() == 0


def test_gru_lean_kernels_with_groups(ops):
    """Grouped GRU (G = 2, Hg = 256) through the bf16-mode lean forward / reduce-scatter backward kernels:
    both must agree with the generic kernels on the same inputs (chains = batch groups x GRU groups)."""
    import os
    _lean_grouped_case(ops, 11, 9, 2, 256)
    _lean_grouped_case(ops, 9, 7, 4, 160)          # BASELINE config 3 geometry: 4 groups of 160


def _lean_grouped_case(ops, B, T, G, Hg):
    import os
    torch.manual_seed(5)
    gi = (0.5 * torch.randn(B, T, G * 3 * Hg)).cuda()
    w = [(torch.randn(3 * Hg, Hg) / Hg ** 0.5).cuda() for _ in range(G)]
    b = [(0.1 * torch.randn(3 * Hg)).cuda() for _ in range(G)]
    dout = torch.randn(B, T, G * Hg).cuda()
    with ops.options(gru_wlo=0):                   # same algorithm == without the lean kernel's W_hh low-plane pass
        lean = ops.gru_seq_fwd(gi, w, b, B, T, G, Hg, "bf16")
        dh_rs = ops.gru_seq_bwd(dout, w, lean[1], lean[3], B, T, G, Hg, "bf16")
        with ops.options(gru_fwd_lean=0, gru_bwd_rs=0):
            gen = ops.gru_seq_fwd(gi, w, b, B, T, G, Hg, "bf16")
            dh_ag = ops.gru_seq_bwd(dout, w, lean[1], lean[3], B, T, G, Hg, "bf16")
    torch.cuda.synchronize()
    assert ops.gru_status() == 0
    assert rel_l2(lean[0], gen[0]) < 1e-4 and rel_l2(lean[1].float(), gen[1].float()) < 2e-3
    assert rel_l2(dh_rs, dh_ag) < 5e-3


@pytest.mark.parametrize("Hg,G", [(160, 4), (320, 2), (640, 1)])
def test_gru_fwd_w_hh_low_plane(ops, Hg, G):
    """The lean forward recurrence with W_hh as hi+lo bf16 planes (default for Hg <= 320): closer to the exact-f32
    recurrence than the single-plane form, on a long sequence where the weight rounding accumulates."""
    import os
    B, T = 8, 201
    torch.manual_seed(3)
    gi = (0.5 * torch.randn(B, T, G * 3 * Hg)).cuda()
    w = [(torch.randn(3 * Hg, Hg) / Hg ** 0.5).cuda() for _ in range(G)]
    b = [(0.1 * torch.randn(3 * Hg)).cuda() for _ in range(G)]
    ref = ops.gru_seq_fwd(gi, w, b, B, T, G, Hg, "f32", save=False)[0]
    err = {}
    for knob in ("0", "1"):
        with ops.options(gru_wlo=int(knob)):
            err[knob] = rel_l2(ops.gru_seq_fwd(gi, w, b, B, T, G, Hg, "bf16", save=False)[0], ref)
    default = rel_l2(ops.gru_seq_fwd(gi, w, b, B, T, G, Hg, "bf16", save=False)[0], ref)
    print(f"[gru W_hh planes Hg={Hg}] rel-L2 vs f32 recurrence: 1 plane {err['0']:.2e}, 2 planes {err['1']:.2e}, default {default:.2e}")
    assert ops.gru_status() == 0
    assert err["1"] < 0.7 * err["0"]
    assert default == pytest.approx(err["1"] if Hg <= 320 else err["0"], rel=0.2)


def test_ktile_bf16_and_padded_k_gemm(ops):
    """K not a multiple of 64 (Hg = 160): the weight operand is K-tiled with zero padding, the activation operand is
    read past its 160 columns into finite neighbours -- the product must still be x[:, :160] @ W^T."""
    rows, Hg = 300, 160
    gen = torch.Generator().manual_seed(11)
    x = torch.randn(rows, 4 * Hg, generator=gen).cuda()                  # 4 groups side by side
    w = torch.randn(3 * Hg, Hg, generator=gen).cuda()
    wk = ops.ktile_bf16(w, 3 * Hg, Hg)
    assert wk.shape == (3, 3 * Hg, 64)
    flat = wk.permute(1, 0, 2).reshape(3 * Hg, 192)
    assert torch.equal(flat[:, :Hg], w.to(torch.bfloat16)) and flat[:, Hg:].abs().max() == 0
    xb = ops.cast_bf16_padded(x, pad=64)
    for i in (0, 3):                                                     # first and LAST group (reads past the row end)
        C = torch.zeros(rows, 3 * Hg).cuda()
        ops.gemm_bf16_nt(rows, 3 * Hg, 192, xb, i * Hg, 4 * Hg, wk, 0, 64, C, 0, 3 * Hg, b_kstride=3 * Hg * 64)
        ref = x[:, i * Hg:(i + 1) * Hg].to(torch.bfloat16).double() @ w.to(torch.bfloat16).double().t()
        assert rel_l2(C, ref) < 1e-6, i


def test_gemm_bf16x3_forward_projection(ops):
    """Split-bf16 x3 form (the forward gate projection of the bf16 mode): ~f32 operand accuracy from bf16 MFMAs."""
    rows, Hg = 400, 160
    gen = torch.Generator().manual_seed(21)
    x = torch.randn(rows, 2 * Hg, generator=gen).cuda()
    w = torch.randn(3 * Hg, Hg, generator=gen).cuda(); b = torch.randn(3 * Hg, generator=gen).cuda()
    x_hi, x_lo = ops.cast_bf16_padded(x, pad=64, split=True)
    assert rel_l2((x_hi[:x.numel()].float() + x_lo[:x.numel()].float()).view_as(x), x) < 2e-5
    w_hi, w_lo = ops.ktile_bf16(w, 3 * Hg, Hg, split=True)
    for i in (0, 1):
        C = torch.zeros(rows, 3 * Hg).cuda()
        ops.gemm_bf16x3_nt(rows, 3 * Hg, 192, x_hi, x_lo, i * Hg, 2 * Hg, w_hi, w_lo, 0, 64, C, 0, 3 * Hg, bias=b,
                           b_kstride=3 * Hg * 64)
        ref = x[:, i * Hg:(i + 1) * Hg].double() @ w.double().t() + b.double()
        assert rel_l2(C, ref) < 2e-5, i
        C1 = torch.zeros(rows, 3 * Hg).cuda()
        ops.gemm_bf16_nt(rows, 3 * Hg, 192, x_hi, i * Hg, 2 * Hg, w_hi, 0, 64, C1, 0, 3 * Hg, bias=b, b_kstride=3 * Hg * 64)
        assert rel_l2(C1, ref) > 50 * rel_l2(C, ref)                # the plain bf16 product is two orders coarser


def test_bf16_operand_copies_from_bn_and_ln(ops):
    """bn_finalize_act_fwd / ln_fwd also write the bf16 copy the next gate GEMM reads: identical to a cast of their f32
    output (vectorised LayerNorm in-kernel, the group-interleaved form through the cast kernel)."""
    gen = torch.Generator().manual_seed(81)
    rows, C, Fq = 37, 64, 10
    y = torch.randn(rows, C, Fq, generator=gen).cuda()
    gam = (torch.rand(C, generator=gen) + 0.5).cuda(); bet = torch.randn(C, generator=gen).cuda()
    sums = ops.bn_stats(y, rows, C, Fq)
    bf = torch.empty(rows * C * Fq, dtype=torch.bfloat16).cuda()
    out, _, _ = ops.bn_finalize_act_fwd(y, sums, rows * Fq, 1e-5, 0.1, gam, bet, None, rows, C, Fq, out_bf16=bf)
    assert torch.equal(bf.view_as(out), out.to(torch.bfloat16))
    for H, g in ((640, 1), (640, 4)):
        x = torch.randn(rows, H, generator=gen).cuda()
        w = (torch.rand(H, generator=gen) + 0.5).cuda(); b = torch.randn(H, generator=gen).cuda()
        bf = torch.empty(rows * H, dtype=torch.bfloat16).cuda()
        yl, _, _ = ops.ln_fwd(x, w, b, None, rows, H, g, out_bf16=bf)
        y0, _, _ = ops.ln_fwd(x, w, b, None, rows, H, g)
        assert torch.equal(yl, y0) and torch.equal(bf.view_as(yl), yl.to(torch.bfloat16))


@pytest.mark.parametrize("H,G,B,T", [(640, 1, 9, 12), (640, 4, 3, 7), (320, 1, 8, 1), (640, 1, 5, 2), (512, 1, 16, 5), (288, 1, 7, 5)])
def test_gru_bwd_writes_gate_gradients_itself(ops, H, G, B, T):
    """cruse_gru_seq_bwd_on(dgi): the recurrence's loader wave writes dgi = dh * (c_r, c_z, a_n); the same dh as without it
    and bit-identical to the gate-gradient pass on that dh; cruse_gru_gate_grads_bf16(dgi = NULL) still makes dgT / biases."""
    Hg = H // G
    torch.manual_seed(H + B + T)
    gi = (0.5 * torch.randn(B, T, 3 * H)).cuda()
    w = [(torch.randn(3 * Hg, Hg) / Hg ** 0.5).cuda() for _ in range(G)]
    b = [torch.zeros(3 * Hg).cuda() for _ in range(G)]
    h, coef, an, z = ops.gru_seq_fwd(gi, w, b, B, T, G, Hg, "bf16")
    dout = torch.randn(B, T, H).cuda()
    dh0 = ops.gru_seq_bwd(dout, w, coef, z, B, T, G, Hg, "bf16")
    dh1, dgi1 = ops.gru_seq_bwd(dout, w, coef, z, B, T, G, Hg, "bf16", an=an, want_dgi=True)
    torch.cuda.synchronize()
    assert ops.gru_status() == 0
    assert torch.equal(dh0, dh1)
    rows = B * T
    db_i = [torch.zeros(3 * Hg).cuda() for _ in range(G)]; db_h = [torch.zeros(3 * Hg).cuda() for _ in range(G)]
    dgi_ref, dgT_ref, ldT = ops.gru_gate_grads_bf16(dh1, coef, an, rows, G, Hg, db_i, db_h)
    assert torch.equal(dgi1.view(-1), dgi_ref.view(-1))
    db_i2 = [torch.zeros(3 * Hg).cuda() for _ in range(G)]; db_h2 = [torch.zeros(3 * Hg).cuda() for _ in range(G)]
    none, dgT2, _ = ops.gru_gate_grads_bf16(dh1, coef, an, rows, G, Hg, db_i2, db_h2, want_dgi=False)
    assert none is None and torch.equal(dgT2, dgT_ref)
    for a_, b_ in zip(db_i + db_h, db_i2 + db_h2):
        assert rel_l2(a_, b_) < 1e-5


@pytest.mark.parametrize("form,dout_bf16", [("gather", True), ("scatter", True), ("scatter_mt4", True), ("gather", False)])
def test_conv_data_gradient_with_fused_batchnorm_backward_input(ops, form, dout_bf16):
    """cruse_conv_*_bnbwd_in (ABI 8): one call == cruse_bn_act_bwd_apply(dout -> dy bf16, parameter gradients) + the data-gradient conv on
    that dy, with the output-side BatchNorm sums.  In the bf16 data-gradient mode with a bf16 dout the MFMA kernel forms dy while staging:
    the dy it writes must be the separate pass's bits (same arithmetic, one rounding), hence the same conv output and sums; a shape the fused
    kernel declines (64-row tiles) and an f32 dout run the two calls."""
    torch.manual_seed(3)
    B, T = 3, 21
    if form == "gather":                                   # decoder data gradient: dv [B,T,16,40] -> du [B,T,32,20]
        Ci, Fi, Co, Fo = 16, 40, 32, 20
        w = (0.2 * torch.randn(Co, Ci, 1, 3)).cuda()
    elif form == "scatter":                                # encoder data gradient: dy [B,T,32,20] -> de [B,T,16,40], accumulating
        Ci, Fi, Co, Fo = 32, 20, 16, 40
        w = (0.2 * torch.randn(Ci, Co, 2, 3)).cuda()
    else:                                                  # dy [B,T,16,40] -> de [B,T,... 64 output channels: declined by the fused kernel
        Ci, Fi, Co, Fo = 16, 40, 64, 80
        w = (0.2 * torch.randn(Ci, Co, 2, 3)).cuda()
    rows = B * T
    dout32 = torch.randn(B, T, Ci, Fi).cuda()
    dout = dout32.bfloat16() if dout_bf16 else dout32
    y_in = torch.randn(B, T, Ci, Fi).cuda()
    mean = y_in.mean(dim=(0, 1, 3)).contiguous(); rstd = (1.0 / (y_in.var(dim=(0, 1, 3), unbiased=False) + 1e-5).sqrt()).contiguous()
    gamma = (torch.rand(Ci) + 0.5).cuda(); beta = (0.1 * torch.randn(Ci)).cuda()
    # backward sums of the input BatchNorm: replica 0 holds them (as after cruse_bn_act_bwd_reduce)
    xh = (y_in - mean.view(1, 1, -1, 1)) * rstd.view(1, 1, -1, 1)
    g = dout.float() * ((xh * gamma.view(1, 1, -1, 1) + beta.view(1, 1, -1, 1)) > 0)
    sums = torch.zeros(ops.BN_STAT_REPLICAS, 2 * Ci, dtype=torch.float64).cuda()
    sums[0, :Ci] = g.double().sum(dim=(0, 1, 3)); sums[0, Ci:] = (g.double() * xh.double()).sum(dim=(0, 1, 3))
    # output-side BatchNorm (the level the data gradient feeds)
    by = torch.randn(B, T, Co, Fo).cuda()
    om = by.mean(dim=(0, 1, 3)).contiguous(); orr = (1.0 / (by.var(dim=(0, 1, 3), unbiased=False) + 1e-5).sqrt()).contiguous()
    og = (torch.rand(Co) + 0.5).cuda(); ob = (0.1 * torch.randn(Co)).cuda()
    bnb = (by, om, orr, og, ob, True)
    base = torch.randn(B, T, Co, Fo).cuda()
    res = {}
    for fused in (False, True):
        dg = torch.zeros(Ci).cuda(); db = torch.zeros(Ci).cuda()
        bn_in = (y_in, mean, rstd, gamma, beta, sums, True, True, dg, db, None)
        if fused:
            if form == "gather":
                out, osums, dy = ops.conv_gather_bwd_in(dout, bn_in, w, B, T, Ci, Fi, Co, Fo, KT=1, S=2, pad=0, prec=ops.PREC_BF16, bn_bwd=bnb)
            else:
                out, osums, dy = ops.conv_scatter2_bwd_in(dout, bn_in, w, B, T, Ci, Fi, Co, KT=2, pad=1, out=base.clone(), accum=True,
                                                          prec=ops.PREC_BF16, bn_bwd=bnb)
        else:
            dy = ops.bn_act_bwd(dout, y_in, mean, rstd, gamma, beta, rows, Ci, Fi, True, True, dg, db, sums=sums, out_bf16=True)
            if form == "gather":
                out, osums = ops.conv_gather(dy, w, None, B, T, Ci, Fi, Co, Fo, KT=1, S=2, pad=0, prec=ops.PREC_BF16, bn_bwd=bnb)
            else:
                out, osums = ops.conv_scatter2(dy, w, None, B, T, Ci, Fi, Co, KT=2, pad=1, out=base.clone(), accum=True, prec=ops.PREC_BF16,
                                               bn_bwd=bnb)
        torch.cuda.synchronize()
        res[fused] = (dy.float().clone(), out.clone(), osums.view(ops.BN_STAT_REPLICAS, -1).sum(0).clone(), dg.clone(), db.clone())
    assert torch.equal(res[True][0], res[False][0]), "dy written while staging differs from the separate pass"
    assert torch.equal(res[True][1], res[False][1])
    assert rel_l2(res[True][2], res[False][2]) < 1e-6
    assert torch.equal(res[True][3], res[False][3]) and torch.equal(res[True][4], res[False][4])
    # and against the closed form of the BatchNorm backward
    cnt = rows * Fi
    want = gamma.view(1, 1, -1, 1) * rstd.view(1, 1, -1, 1) * (g - (sums[0, :Ci] / cnt).float().view(1, 1, -1, 1)
                                                                 - xh * (sums[0, Ci:] / cnt).float().view(1, 1, -1, 1))
    assert rel_l2(res[True][0], want) < 5e-3               # (bf16 storage)


@pytest.mark.parametrize("G,B,T", [(4, 9, 12), (4, 3, 1), (4, 5, 2), (2, 20, 37), (2, 8, 3), (4, 64, 60)])
def test_gru_all_gather_backward_on_grouped_widths(ops, G, B, T):
    """The all-gather backward kernel with the register-direct sweep at Hg = 160 / 320 (3 Hg / 32 k-steps do not divide over the four
    waves: the missing ones re-read the last k-step against zero weights): against the reduce-scatter kernel and the exact-f32 kernels,
    with the gate gradients written by the loader wave."""
    H = 640; Hg = H // G
    torch.manual_seed(G * 1000 + B + T)
    gi = (0.5 * torch.randn(B, T, 3 * H)).cuda()
    w = [(torch.randn(3 * Hg, Hg) / Hg ** 0.5).cuda() for _ in range(G)]; b = [(0.1 * torch.randn(3 * Hg)).cuda() for _ in range(G)]
    dout = (2.0 * torch.randn(B, T, H)).cuda()
    f32 = ops.gru_seq_fwd(gi, w, b, B, T, G, Hg, "f32")
    ref = ops.gru_seq_bwd(dout, w, f32[1], f32[3], B, T, G, Hg, "f32")
    f = ops.gru_seq_fwd(gi, w, b, B, T, G, Hg, "bf16")
    out = {}
    for ag, opts in ((0, dict(gru_tf=0)), (2, {})):                  # 0: the tagged reduce-scatter kernel, 2: the default all-gather kernel
        with ops.options(**opts):
            out[ag] = ops.gru_seq_bwd(dout, w, f[1], f[3], B, T, G, Hg, "bf16", an=f[2], want_dgi=True)
            plain = ops.gru_seq_bwd(dout, w, f[1], f[3], B, T, G, Hg, "bf16")
            torch.cuda.synchronize()
            assert ops.gru_status() == 0 and torch.equal(plain, out[ag][0])
    assert torch.isfinite(out[2][0]).all() and rel_l2(out[2][0], out[0][0]) < 5e-3
    if T > 1:
        assert rel_l2(out[2][0], ref) < 1e-2 and rel_l2(out[2][0], ref) <= 1.05 * rel_l2(out[0][0], ref) + 1e-6
    rows = B * T
    db_i = [torch.zeros(3 * Hg).cuda() for _ in range(G)]; db_h = [torch.zeros(3 * Hg).cuda() for _ in range(G)]
    dgi_ref, _, _ = ops.gru_gate_grads_bf16(out[2][0], f[1], f[2], rows, G, Hg, db_i, db_h)
    assert torch.equal(out[2][1].view(-1), dgi_ref.view(-1))


@pytest.mark.parametrize("B,T", [(9, 12), (3, 1), (5, 2), (8, 3), (20, 37), (64, 60)])
def test_gru_register_direct_sweeps_and_all_gather_backward(ops, B, T):
    """Round-4 recurrence kernels at Hg = 640 (bf16 mode).  Forward: the tag-free register-direct sweep (default) gives the bits of the
    tagged lean kernel (gru_tf = 0: same sums, same bf16 hand-off values).  Backward: the all-gather kernel and the tagged reduce-scatter
    kernel agree with the f64 recurrence on the saved coefficients; the gate gradients written by the loader wave are those of the
    separate pass on the same dh."""
    H = 640
    torch.manual_seed(B * 100 + T)
    gi = (0.5 * torch.randn(B, T, 3 * H)).cuda()
    w = [(torch.randn(3 * H, H) / H ** 0.5).cuda()]; b = [(0.1 * torch.randn(3 * H)).cuda()]
    with ops.options(gru_tf=0):
        f0 = ops.gru_seq_fwd(gi, w, b, B, T, 1, H, "bf16")
    f1 = ops.gru_seq_fwd(gi, w, b, B, T, 1, H, "bf16")
    torch.cuda.synchronize()
    assert ops.gru_status() == 0
    for x, y in zip(f0, f1):
        assert torch.equal(x, y)
    h, coef, an, z = f1
    dout = (3.0 * torch.randn(B, T, H)).cuda()
    out = {}
    for ag, opts in ((0, dict(gru_tf=0)), (2, {})):
        with ops.options(**opts):
            out[ag] = ops.gru_seq_bwd(dout, w, coef, z, B, T, 1, H, "bf16", an=an, want_dgi=True)
            plain = ops.gru_seq_bwd(dout, w, coef, z, B, T, 1, H, "bf16")
            torch.cuda.synchronize()
            assert ops.gru_status() == 0 and torch.equal(plain, out[ag][0])
    c = coef.float().view(B, T, 3, H)
    wd = w[0].double()
    ref = torch.zeros(B, T, H, dtype=torch.float64, device="cuda")
    nxt = torch.zeros(B, H, dtype=torch.float64, device="cuda")
    for s_ in range(T - 1, -1, -1):
        cur = dout[:, s_].double()
        if s_ < T - 1:
            dgh = (nxt.unsqueeze(1) * c[:, s_ + 1].double()).reshape(B, 3 * H)
            cur = cur + z[:, s_ + 1].double() * nxt + dgh @ wd
        ref[:, s_] = cur
        nxt = cur
    for ag in (0, 2):
        assert torch.isfinite(out[ag][0]).all() and rel_l2(out[ag][0], ref) < 1e-2, (ag, rel_l2(out[ag][0], ref))
    # (the all-gather form accumulates the whole K in f32; the reduce-scatter form exchanges bf16 partial sums)
    assert rel_l2(out[2][0], ref) <= 1.05 * rel_l2(out[0][0], ref) + 1e-6
    rows = B * T
    db_i = [torch.zeros(3 * H).cuda()]; db_h = [torch.zeros(3 * H).cuda()]
    dgi_ref, _, _ = ops.gru_gate_grads_bf16(out[2][0], coef, an, rows, 1, H, db_i, db_h)
    assert torch.equal(out[2][1].view(-1), dgi_ref.view(-1))




def _wgrad_ref(a, bt, KT, S, pad):
    """dw[ca][cb][kt][kf] = sum a[b,t,ca,fa] * bt[b, t-(KT-1)+kt, cb, fa*S - pad + kf] in f64 (include/cruse_hip.h, cruse_conv_wgrad)"""
    B, T, Ca, Fa = a.shape
    _, _, Cb, Fb = bt.shape
    a64, b64 = a.double(), bt.double()
    dw = torch.zeros(Ca, Cb, KT, 3, dtype=torch.float64)
    for kt in range(KT):
        sh = KT - 1 - kt                                      # source frame t - sh
        src = torch.zeros_like(b64)
        src[:, sh:] = b64[:, :T - sh] if sh else b64
        for kf in range(3):
            idx = torch.arange(Fa) * S - pad + kf
            ok = (idx >= 0) & (idx < Fb)
            g = torch.zeros(B, T, Cb, Fa, dtype=torch.float64)
            g[..., ok] = src[..., idx[ok]]
            dw[:, :, kt, kf] = torch.einsum("btaf,btcf->ac", a64, g)
    return dw


@pytest.mark.parametrize("shape", [
    # (B, T, Ca, Fa, Cb, KT, S, pad): the three conv forms at ragged sizes -- rows of 8 / 12 / 20 / 10 positions (one window; an end-aligned
    # last window that overlaps the one before it by 4 / 4 / 6 positions), channel counts that leave padding rows / columns in the
    # 16 x 16 tiles and partial 16-channel source tiles, one- and two-frame clips (frame t - 1 of every clip's first frame is zero)
    (2, 5, 8, 8, 1, 2, 2, 1), (3, 1, 16, 12, 8, 2, 2, 1), (1, 2, 32, 20, 16, 2, 2, 1), (2, 7, 64, 10, 32, 2, 2, 1), (2, 9, 24, 10, 12, 2, 2, 1),
    (2, 5, 8, 40, 1, 1, 2, 0), (3, 3, 16, 12, 8, 1, 2, 0), (2, 4, 64, 10, 32, 1, 2, 0), (1, 6, 40, 20, 24, 1, 2, 0),
    (2, 5, 8, 16, 8, 1, 1, 1), (3, 3, 32, 20, 32, 1, 1, 1), (2, 4, 64, 10, 64, 1, 1, 1), (1, 11, 12, 12, 40, 1, 1, 1),
    (5, 21, 16, 40, 8, 2, 2, 1)])
def test_conv_wgrad_register_direct_kernel(ops, shape):
    """wgrad_rd.hip (bf16 mode): fragments loaded straight from the two tensors.  Against the f64 contraction of the bf16-rounded operands
    (only the f32 accumulation order differs: 2e-5) and -- same operand bits, same products -- the LDS-staged kernel; every bf16 / f32
    storage combination of the two operands gives the same bits."""
    B, T, Ca, Fa, Cb, KT, S, pad = shape
    Fb = S * Fa
    gen = torch.Generator().manual_seed(sum(shape))
    a = torch.randn(B, T, Ca, Fa, generator=gen).to(torch.bfloat16)
    bt = torch.randn(B, T, Cb, Fb, generator=gen).to(torch.bfloat16)
    ref = _wgrad_ref(a.float(), bt.float(), KT, S, pad)
    outs = {}
    for rd in (1, 0):
        ops.set_option("wg_rd", rd)
        try:
            for da, db in ((torch.bfloat16, torch.bfloat16), (torch.float32, torch.bfloat16), (torch.bfloat16, torch.float32), (torch.float32, torch.float32)):
                dw = torch.zeros(Ca, Cb, KT, 3).cuda()
                ops.conv_wgrad(a.to(da).cuda(), bt.to(db).cuda(), dw, B, T, Ca, Fa, Cb, Fb, KT=KT, S=S, pad=pad, prec="bf16")
                outs[(rd, da, db)] = dw.cpu()
        finally:
            ops.set_option("wg_rd", None)
    first = outs[(1, torch.bfloat16, torch.bfloat16)]
    assert rel_l2(first, ref) < 2e-5
    for key, dw in outs.items():
        if key[0] == 1:
            assert torch.equal(dw, first), key
        else:
            assert rel_l2(dw, ref) < 2e-5, key
    # accumulation into an existing gradient
    dw = torch.full((Ca, Cb, KT, 3), 0.5).cuda()
    ops.conv_wgrad(a.cuda(), bt.cuda(), dw, B, T, Ca, Fa, Cb, Fb, KT=KT, S=S, pad=pad, prec="bf16")
    assert torch.allclose(dw.cpu(), first + 0.5, rtol=1e-6, atol=1e-5)


@pytest.mark.parametrize("form", ["gather", "scatter", "wgrad_a", "wgrad_bt"])
def test_backward_only_tensors_stored_as_bf16_give_the_same_bits(ops, form):
    """EngineConfig.bf16_dy (round 4): the BatchNorm-backward output is stored as bf16 -- the data-gradient convs (plain bf16
    operands) and the weight gradient (bf16 mode) round it to bf16 anyway, so a bf16 tensor and the f32 tensor holding the same
    rounded values must give identical results; cruse_bn_act_bwd_apply(dy_dtype = bf16) == the f32 result rounded once.  The
    VALU fall-backs refuse a bf16 input loudly."""
    torch.manual_seed(7)
    B, T = 3, 21
    if form == "gather":                                   # decoder data gradient: dv [B,T,8,80] -> du [B,T,16,40]
        x = torch.randn(B, T, 8, 80).cuda(); w = (0.2 * torch.randn(16, 8, 1, 3)).cuda()
        run = lambda t_: ops.conv_gather(t_, w, None, B, T, 8, 80, 16, 40, KT=1, S=2, pad=0, prec=ops.PREC_BF16)   # (plain bf16 operands: the data-gradient mode)
    elif form == "scatter":                                # encoder data gradient: dy [B,T,32,20] -> de [B,T,16,40], accumulating
        x = torch.randn(B, T, 32, 20).cuda(); w = (0.2 * torch.randn(32, 16, 2, 3)).cuda()
        base = torch.randn(B, T, 16, 40).cuda()
        run = lambda t_: ops.conv_scatter2(t_, w, None, B, T, 32, 20, 16, KT=2, pad=1, out=base.clone(), accum=True, prec=ops.PREC_BF16)
    else:
        a = torch.randn(B, T, 16, 40).cuda(); bt = torch.randn(B, T, 8, 80).cuda()

        def run(t_):
            dw = torch.zeros(16, 8, 2, 3).cuda()
            aa, bb = (t_, bt) if form == "wgrad_a" else (a, t_)
            ops.conv_wgrad(aa, bb, dw, B, T, 16, 40, 8, 80, KT=2, S=2, pad=1, prec="bf16")
            return dw
        x = a if form == "wgrad_a" else bt
    xb = x.to(torch.bfloat16)
    assert torch.equal(run(xb), run(xb.float()))
    if form == "gather":
        with pytest.raises(RuntimeError, match="bf16"):    # Cin = 1: no MFMA form
            ops.conv_gather(torch.randn(B, T, 1, 160).cuda().to(torch.bfloat16), (0.2 * torch.randn(8, 1, 1, 3)).cuda(), None,
                            B, T, 1, 160, 8, 80, KT=1, S=2, pad=0, prec=ops.PREC_BF16)
        rows, C, F = B * T, 16, 40
        y = torch.randn(B, T, C, F).cuda(); dout = torch.randn(B, T, C, F).cuda()
        mean = (0.1 * torch.randn(C)).cuda(); rstd = (1.0 + 0.2 * torch.rand(C)).cuda()
        gamma = (1.0 + 0.3 * torch.randn(C)).cuda(); beta = (0.2 * torch.randn(C)).cuda()
        outs = []
        for bf in (False, True):
            dg = torch.zeros(C).cuda(); db = torch.zeros(C).cuda()
            outs.append(ops.bn_act_bwd(dout, y, mean, rstd, gamma, beta, rows, C, F, True, True, dg, db, out_bf16=bf))
        assert outs[1].dtype == torch.bfloat16 and torch.equal(outs[1], outs[0].to(torch.bfloat16))


@pytest.mark.parametrize("form", ["gather", "gather_bnbwd", "scatter_accum"])
def test_data_gradients_stored_as_bf16(ops, form):
    """EngineConfig.bf16_de: a data-gradient conv of the bf16 mode writes its output as bf16 -- the f32 result rounded once; when
    it accumulates, (old bf16 + conv) in f32, rounded once; the BatchNorm-backward sums of its epilogue are those of the f32 values."""
    torch.manual_seed(8)
    B, T = 3, 21
    if form.startswith("gather"):
        x = torch.randn(B, T, 16, 40).cuda().to(torch.bfloat16); w = (0.2 * torch.randn(32, 16, 1, 3)).cuda()
        C, F = 32, 20
        run = lambda bf, bn: ops.conv_gather(x, w, None, B, T, 16, 40, 32, 20, KT=1, S=2, pad=0, prec=ops.PREC_BF16, bn_bwd=bn, out_bf16=bf)
    else:
        x = torch.randn(B, T, 32, 20).cuda().to(torch.bfloat16); w = (0.2 * torch.randn(32, 16, 2, 3)).cuda()
        C, F = 16, 40
        base = torch.randn(B, T, C, F).cuda().to(torch.bfloat16)
        run = lambda bf, bn: ops.conv_scatter2(x, w, None, B, T, 32, 20, 16, KT=2, pad=1, out=(base.clone() if bf else base.float()), accum=True,
                                               prec=ops.PREC_BF16, bn_bwd=bn)
    bn = None
    if form != "gather":
        y = torch.randn(B, T, C, F).cuda()
        bn = (y, (0.1 * torch.randn(C)).cuda(), (1.0 + 0.2 * torch.rand(C)).cuda(), (1.0 + 0.3 * torch.randn(C)).cuda(), (0.2 * torch.randn(C)).cuda(), True)
    r32, rbf = run(False, bn), run(True, bn)
    if bn is not None:
        (r32, s32), (rbf, sbf) = r32, rbf
        assert rel_l2(sbf, s32) < 1e-6
    assert rbf.dtype == torch.bfloat16 and torch.equal(rbf, r32.to(torch.bfloat16))


# ---------------------------------------------------------------------------------------------------------------- round 6: 2-byte gi rows
@pytest.mark.parametrize("M,N,K", [(25664, 1920, 640), (4133, 640, 640), (130, 96, 64), (77, 50, 128)])
def test_gate_projection_with_two_byte_result_rows(ops, M, N, K):
    """cruse_gemm_nt_out16 (ABI 13): the forward gate projections storing f16 / bf16 rows -- the f32 accumulation of the entry point with the
    f32 result, rounded once at the store: BIT-IDENTICAL to rounding that entry point's result, for every operand form the step uses
    (bf16 one plane, bf16 x . (W hi + lo), f16 one plane, f16 x . (W hi + lo)); ragged tiles, a row stride that is no multiple of 4 (scalar stores)."""
    torch.manual_seed(M + K)
    A = torch.randn(M, K).cuda(); W = (torch.randn(N, K) / K ** 0.5).cuda(); bias = torch.randn(N).cuda()
    Ab, Ah = A.to(torch.bfloat16), A.half()
    Wb, Wbl = ops.ktile_bf16(W, N, K, split=True)
    Wh, Whl = ops.ktile_f16(W, N, K, split=True)
    ks = N * 64
    forms = {"bf16": lambda C: ops.gemm_bf16_nt(M, N, K, Ab, 0, K, Wb, 0, 64, C, 0, C.shape[1], bias=bias, b_kstride=ks),
             "bf16 x . (W hi + lo)": lambda C: ops.gemm_bf16x3_nt(M, N, K, Ab, None, 0, K, Wb, Wbl, 0, 64, C, 0, C.shape[1], bias=bias, b_kstride=ks),
             "f16": lambda C: ops.gemm_f16_nt(M, N, K, Ah, 0, K, Wh, 0, 64, C, 0, C.shape[1], bias=bias, b_kstride=ks),
             "f16 x . (W hi + lo)": lambda C: ops.gemm_f16_nt(M, N, K, Ah, 0, K, Wh, 0, 64, C, 0, C.shape[1], bias=bias, b_kstride=ks, B_lo=Whl)}
    for name, fn in forms.items():
        C32 = torch.empty(M, N).cuda()
        fn(C32)
        for dt in (torch.float16, torch.bfloat16):
            for ldc in (N, N + 3):
                C16 = torch.full((M, ldc), 5.0, dtype=dt).cuda()
                fn(C16)
                assert torch.equal(C16[:, :N], C32.to(dt)), (name, dt, ldc)
                assert ldc == N or bool((C16[:, N:] == 5.0).all()), "columns beyond N are not touched"
    with pytest.raises(RuntimeError):                     # a 2-byte result is stored, not accumulated
        ops.gemm_bf16x3_nt(M, N, K, Ab, None, 0, K, Wb, Wbl, 0, 64, torch.zeros(M, N, dtype=torch.float16).cuda(), 0, N, accumulate=True, b_kstride=ks)


@pytest.mark.parametrize("B,T", [(8, 9), (64, 33), (13, 1), (96, 5)])
def test_forward_recurrence_on_f16_gi_rows(ops, B, T):
    """cruse_gru_seq_fwd_gi16 (ABI 13): the bench step's forward recurrence (Hg = 640, chains of 8, tag-free register-direct hand-off) reading gi as
    IEEE f16 rows -- the helper wave widens them exactly, so h / coefficients / a_n / z are BIT-IDENTICAL to the f32-row kernel on the f16-rounded
    values; refused (CRUSE_E_SHAPE) where no kernel serves them: other widths, wide chains (B > 96), the f32 modes."""
    H = 640
    torch.manual_seed(B * 7 + T)
    gi = (0.7 * torch.randn(B, T, 3 * H)).cuda()
    w = [(torch.randn(3 * H, H) / H ** 0.5).cuda()]; b = [(0.1 * torch.randn(3 * H)).cuda()]
    g16 = gi.half()
    a = ops.gru_seq_fwd(g16, w, b, B, T, 1, H, "bf16")
    r = ops.gru_seq_fwd(g16.float(), w, b, B, T, 1, H, "bf16")
    torch.cuda.synchronize()
    assert ops.gru_status() == 0
    for x, y, name in zip(a, r, ("h", "coef", "an", "z")):
        assert torch.equal(x.view(torch.int16) if x.dtype == torch.bfloat16 else x, y.view(torch.int16) if y.dtype == torch.bfloat16 else y), name
    h_only = ops.gru_seq_fwd(g16, w, b, B, T, 1, H, "bf16", save=False)[0]
    assert torch.equal(h_only, a[0])
    assert ops.gru_plan(B, 1, H)["clips_per_chain"] == 8
    for bad in (lambda: ops.gru_seq_fwd(gi[:, :, :3 * 320].contiguous().half(), [w[0][:960, :320].contiguous()], [b[0][:960].contiguous()], B, T, 1, 320, "bf16"),
                lambda: ops.gru_seq_fwd(torch.zeros(128, 2, 3 * H, dtype=torch.float16).cuda(), w, b, 128, 2, 1, H, "bf16"),
                lambda: ops.gru_seq_fwd(g16, w, b, B, T, 1, H, "f32"),
                lambda: ops.gru_seq_fwd(g16, w, b, B, T, 1, H, "bf16", h0=torch.zeros(B, H).cuda())):
        with pytest.raises(RuntimeError):
            bad()
    assert ops.gru_status() == 0


@pytest.mark.parametrize("rows,Hg,accumulate", [(25664, 640, False), (25664, 640, True), (1000, 128, False), (77, 64, True)])
def test_input_gradient_gemm_from_time_major_gate_gradients(ops, rows, Hg, accumulate):
    """cruse_gemm_bf16_nt_atr (round 4, back in round 6 as EngineConfig.dx_atr): dX = dgi . W_ih read from the TIME-MAJOR K-tiled gate gradients dgT
    [ceil(rows / 64)][4][Hg][64] -- the operand the weight-gradient GEMMs consume -- through transposing LDS reads: BIT-IDENTICAL to the row-major
    product on the same bf16 values (same k order, same accumulation), ragged last row block included."""
    torch.manual_seed(rows + Hg)
    dh = torch.randn(rows, Hg).cuda() * 1e-3
    coef = torch.randn(rows, 3 * Hg).cuda().to(torch.bfloat16)
    an = torch.rand(rows, Hg).cuda()
    db_ih, db_hh = [torch.zeros(3 * Hg).cuda()], [torch.zeros(3 * Hg).cuda()]
    dgi, dgT, ldT = ops.gru_gate_grads_bf16(dh, coef, an, rows, 1, Hg, db_ih, db_hh)
    W = (torch.randn(3 * Hg, Hg) / Hg ** 0.5).cuda()
    w_t = ops.transpose_bf16(W, 3 * Hg, Hg)                       # K-tiled [ceil(3 Hg / 64), Hg, 64]
    init = torch.randn(rows, Hg).cuda()
    a, b = init.clone(), init.clone()
    ops.gemm_bf16_nt(rows, Hg, w_t.shape[0] * 64, dgi, 0, 3 * Hg, w_t, 0, 64, a, 0, Hg, accumulate=accumulate, b_kstride=Hg * 64)
    ops.gemm_bf16_nt_atr(rows, Hg, 3 * Hg, dgT, 0, 4 * Hg * 64, ldT // 64, w_t, 0, 64, b, 0, Hg, accumulate=accumulate, b_kstride=Hg * 64)
    assert torch.equal(a, b)
    want = dgi.view(rows, 3 * Hg).double() @ W.to(torch.bfloat16).double() + (init.double() if accumulate else 0.0)
    assert rel_l2(b, want) < 1e-5
