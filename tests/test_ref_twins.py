"""Pins oracle/cruse_ref.c -- the plain-C CPU twins of the C ABI's core entry points -- before anything is compared
with it: every twin against the torch op the reference calls at the cited site (torch.stft / istft, nn.Conv2d, nn.ConvTranspose2d,
nn.BatchNorm2d, nn.LayerNorm, nn.GRU, torch.optim.Adam; values and autograd gradients, all in float64), and against the committed
golden vectors that hold exactly that boundary (G1 stft, G7 istft, G2 conv, G8 DeepFilter, G5 WO-MALE through the pinned oracle).
tests/test_gpu_abi_ref.py then compares the HIP entry points with the twins through the C ABI."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ref_lib as R  # noqa: E402
from ref_lib import LL, call  # noqa: E402

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def ref():
    return R.load()


def f32(x):
    return np.ascontiguousarray(np.asarray(x, dtype=np.float32))


def close(a, b, tol):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    err = np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)
    assert err <= tol, err


def test_twins_mirror_the_header():
    """every twin is an entry point of include/cruse_hip.h with the same parameter list (types and order)"""
    names = R.twin_names()
    assert len(names) >= 33
    for n in names:
        hp, sp = R.header_params(n), R.source_params(n)
        assert len(hp) == len(sp), (n, hp, sp)
        for a, b in zip(hp, sp):
            ta, tb = a.rsplit(" ", 1)[0].replace(" *", "*"), b.rsplit(" ", 1)[0].replace(" *", "*")
            assert ta == tb, (n, a, b)


def test_stft_istft_twins_vs_golden_and_torch(ref):
    g1 = np.load(os.path.join(GOLD, "g1_stft.npz"))
    for L in (3200, 3199, 3201):
        x = f32(g1[f"x_{L}"])
        X = g1[f"X_{L}"]                                           # [B, F, T, 2] (the reference's layout)
        B, Fb, T = X.shape[:3]
        re, im, mag = (np.zeros((B, T, Fb), np.float32) for _ in range(3))
        call(ref, "cruse_stft_fwd", x, B, L, 320, 160, re, im, mag, Fb, 0.0, None)
        close(re, X[..., 0].transpose(0, 2, 1), 2e-6)
        close(im, X[..., 1].transpose(0, 2, 1), 2e-6)
        close(mag, np.sqrt(X[..., 0] ** 2 + X[..., 1] ** 2).transpose(0, 2, 1), 2e-6)
    # another frame size against torch.stft directly
    x = torch.randn(2, 1000, dtype=torch.float64)
    Xt = torch.stft(x, 64, 16, 64, torch.hann_window(64, dtype=torch.float64), center=True, return_complex=True)   # [B, F, T]
    T = Xt.shape[2]
    re, im = np.zeros((2, T, 33), np.float32), np.zeros((2, T, 33), np.float32)
    call(ref, "cruse_stft_fwd", f32(x), 2, 1000, 64, 16, re, im, None, 0, 0.0, None)
    close(re, Xt.real.permute(0, 2, 1), 2e-6)
    close(im, Xt.imag.permute(0, 2, 1), 2e-6)
    g7 = np.load(os.path.join(GOLD, "g7_istft.npz"))
    X, m = g7["X"], g7["m"]
    for Xi, want in ((X, g7["y_rt"]), (X * m[..., None], g7["y_masked"])):
        B, Fb, T = Xi.shape[:3]
        re, im = f32(Xi[..., 0].transpose(0, 2, 1)), f32(Xi[..., 1].transpose(0, 2, 1))
        y = np.zeros((B, 3200), np.float32)
        call(ref, "cruse_istft_fwd", re, im, B, T, 320, 160, 3200, y, None)
        close(y, want, 5e-6)


def _conv_ref_torch(x, w, b, KT, S, pad):
    """frame-major [B,T,C,F] -> Conv2d((KT,3), stride (1,S)) with KT-1 zero frames in front and `pad` bins either side"""
    xt = x.permute(0, 2, 1, 3)                                     # [B, C, T, F]
    xt = F.pad(xt, (pad, pad, KT - 1, 0))
    return F.conv2d(xt, w, b, stride=(1, S)).permute(0, 2, 1, 3)


def test_conv_twins_vs_torch_and_golden(ref):
    g2 = np.load(os.path.join(GOLD, "g2_conv.npz"))
    x = f32(g2["x"].transpose(0, 2, 1, 3))                          # [B,1,T,F] -> [B,T,1,F]
    B, T, Cin, Fin = x.shape
    Cout, Fout = g2["y"].shape[1], g2["y"].shape[3]
    y = np.zeros((B, T, Cout, Fout), np.float32)
    call(ref, "cruse_conv_gather", x, f32(g2["w"]), f32(g2["b"]), y, B, T, Cin, Fin, Cout, Fout, 2, 2, 1, 0, 0, 0, -1, 0, 0, None)
    close(y.transpose(0, 2, 1, 3), g2["y"], 2e-6)
    torch.manual_seed(0)
    B, T, Cin, Fin, Cout = 2, 5, 3, 12, 4
    for (KT, S, pad) in ((2, 2, 1), (1, 1, 1), (2, 1, 1)):
        xt = torch.randn(B, T, Cin, Fin, dtype=torch.float64, requires_grad=True)
        w = torch.randn(Cout, Cin, KT, 3, dtype=torch.float64, requires_grad=True)
        b = torch.randn(Cout, dtype=torch.float64)
        yt = _conv_ref_torch(xt, w, b, KT, S, pad)
        Fout = yt.shape[3]
        y = np.zeros((B, T, Cout, Fout), np.float32)
        call(ref, "cruse_conv_gather", f32(xt.detach()), f32(w.detach()), f32(b), y, B, T, Cin, Fin, Cout, Fout, KT, S, pad, 0, 0, 0, -1, 0, 0, None)
        close(y, yt.detach(), 2e-6)
        ys = np.zeros_like(y)
        call(ref, "cruse_conv_gather", f32(xt.detach()), f32(w.detach()), f32(b), ys, B, T, Cin, Fin, Cout, Fout, KT, S, pad, 0, 1, 0, -1, 0, 0, None)
        close(ys, torch.sigmoid(yt).detach(), 2e-6)
        dy = torch.randn_like(yt)
        dxt, dwt = torch.autograd.grad(yt, (xt, w), dy)
        # weight gradient: a = dy, bt = x
        dw = np.full((Cout, Cin, KT, 3), 0.5, np.float32)
        call(ref, "cruse_conv_wgrad", f32(dy), f32(xt.detach()), dw, B, T, Cout, Fout, Cin, Fin, KT, S, pad, -1, 0, 0, None, None)
        close(dw - 0.5, dwt, 3e-6)
        # data gradient: stride 2 -> the scatter form (KT, pad as the forward conv); stride 1 -> the gather form on the flipped weight
        dx = np.zeros((B, T, Cin, Fin), np.float32)
        if S == 2:
            call(ref, "cruse_conv_scatter2", f32(dy), f32(w.detach()), None, dx, B, T, Cout, Fout, Cin, Fin, KT, pad, 0, 0, -1, 0, 0, None)
            close(dx, dxt, 3e-6)
        elif KT == 1:
            call(ref, "cruse_conv_gather", f32(dy), f32(w.detach()), None, dx, B, T, Cout, Fout, Cin, Fin, 1, 1, 1, 1, 0, 0, -1, 0, 0, None)
            close(dx, dxt, 3e-6)
    # ConvTranspose2d((1,3), stride (1,2)) + the [..., :-1] crop (cruse_net.py:161-164)
    Cs, Fg = 4, 7
    g = torch.randn(B, T, Cs, Fg, dtype=torch.float64)
    w = torch.randn(Cs, Cout, 1, 3, dtype=torch.float64)
    b = torch.randn(Cout, dtype=torch.float64)
    yt = F.conv_transpose2d(g.permute(0, 2, 1, 3), w, b, stride=(1, 2))[..., :-1].permute(0, 2, 1, 3)
    assert yt.shape[3] == 2 * Fg
    y = np.zeros((B, T, Cout, 2 * Fg), np.float32)
    call(ref, "cruse_conv_scatter2", f32(g), f32(w), f32(b), y, B, T, Cs, Fg, Cout, 2 * Fg, 1, 0, 0, 0, -1, 0, 0, None)
    close(y, yt, 2e-6)
    acc = np.ones_like(y)
    call(ref, "cruse_conv_scatter2", f32(g), f32(w), f32(b), acc, B, T, Cs, Fg, Cout, 2 * Fg, 1, 0, 0, 1, -1, 0, 0, None)
    close(acc - 1.0, yt, 3e-6)
    s = np.full(Cout, 2.0, np.float32)
    call(ref, "cruse_channel_sum", y, LL(B * T), Cout, 2 * Fg, s, None)
    close(s - 2.0, y.astype(np.float64).sum((0, 1, 3)), 2e-6)


def test_batchnorm_twins_vs_torch(ref):
    torch.manual_seed(1)
    B, T, C, Fq = 3, 4, 5, 6
    rows = B * T
    for relu in (1, 0):
        y = (torch.randn(B, T, C, Fq, dtype=torch.float64) * 1.5 + 0.3).requires_grad_(True)
        gamma = (1 + 0.2 * torch.randn(C, dtype=torch.float64)).requires_grad_(True)
        beta = (0.1 * torch.randn(C, dtype=torch.float64)).requires_grad_(True)
        skip = torch.randn(B, T, C, Fq, dtype=torch.float64)
        rm, rv = torch.randn(C, dtype=torch.float64), torch.rand(C, dtype=torch.float64) + 0.5
        rm_t, rv_t = rm.clone(), rv.clone()
        o = F.batch_norm(y.permute(0, 2, 1, 3), rm_t, rv_t, gamma, beta, True, 0.1, 1e-5)
        o = (F.relu(o) if relu else o).permute(0, 2, 1, 3) + skip
        sums = np.full(2 * C, 7.0)
        call(ref, "cruse_bn_stats", f32(y.detach()), LL(rows), C, Fq, sums, 0, None)
        mean, rstd = np.zeros(C, np.float32), np.zeros(C, np.float32)
        rm_n, rv_n = f32(rm), f32(rv)
        call(ref, "cruse_bn_finalize", sums, LL(rows * Fq), C, 1e-5, 0.1, mean, rstd, rm_n, rv_n, None)
        close(rm_n, rm_t, 2e-6)
        close(rv_n, rv_t, 2e-6)
        out = np.zeros((B, T, C, Fq), np.float32)
        call(ref, "cruse_bn_act_fwd", f32(y.detach()), mean, rstd, f32(gamma.detach()), f32(beta.detach()), f32(skip), out, LL(rows), C, Fq, relu, None)
        close(out, o.detach(), 3e-6)
        do = torch.randn_like(o)
        dy_t, dg_t, db_t = torch.autograd.grad(o, (y, gamma, beta), do)
        bsums = np.zeros(2 * C)
        call(ref, "cruse_bn_act_bwd_reduce", f32(do), f32(y.detach()), mean, rstd, f32(gamma.detach()), f32(beta.detach()), LL(rows), C, Fq, relu,
             bsums, 1, None)
        dy, dg, db, dbias = np.zeros((B, T, C, Fq), np.float32), np.ones(C, np.float32), np.ones(C, np.float32), np.ones(C, np.float32)
        call(ref, "cruse_bn_act_bwd_apply", f32(do), f32(y.detach()), mean, rstd, f32(gamma.detach()), f32(beta.detach()), bsums, 1, LL(rows), C, Fq,
             relu, 1, 0, dy, 0, dg, db, dbias, None)
        close(dy, dy_t, 1e-5)
        close(dg - 1.0, dg_t, 1e-5)
        close(db - 1.0, db_t, 1e-5)
        assert np.all(dbias == 1.0)                               # batch statistics cancel the bias of the conv in front exactly


def test_layernorm_twins_vs_torch(ref):
    torch.manual_seed(2)
    rows, H = 7, 24
    for g in (1, 2, 4):
        Hg = H // g
        x = torch.randn(rows, H, dtype=torch.float64, requires_grad=True)
        gamma = (1 + 0.3 * torch.randn(H, dtype=torch.float64)).requires_grad_(True)
        beta = (0.2 * torch.randn(H, dtype=torch.float64)).requires_grad_(True)
        res = torch.randn(rows, H, dtype=torch.float64)
        # cruse_net.py:43-45: the groups' outputs are stacked on a new LAST axis and flattened, then normalised
        xi = x.view(rows, g, Hg).permute(0, 2, 1).reshape(rows, H)
        yt = F.layer_norm(xi, (H,), gamma, beta, 1e-5) + res
        y, mean, rstd = np.zeros((rows, H), np.float32), np.zeros(rows, np.float32), np.zeros(rows, np.float32)
        yb = np.zeros((rows, H), np.uint16)
        call(ref, "cruse_ln_fwd", f32(x.detach()), f32(gamma.detach()), f32(beta.detach()), f32(res), y, yb, mean, rstd, LL(rows), H, g, 1e-5, 0,
             LL(0), LL(0), None)
        close(y, yt.detach(), 3e-6)
        close(torch.from_numpy(yb.astype(np.int32) << 16).view(torch.float32), torch.from_numpy(y).to(torch.bfloat16).float(), 0.0)
        dy = torch.randn_like(yt)
        dx_t, dg_t, db_t = torch.autograd.grad(yt, (x, gamma, beta), dy)
        dx, dg, db = np.zeros((rows, H), np.float32), np.ones(H, np.float32), np.ones(H, np.float32)
        call(ref, "cruse_ln_bwd", f32(dy), f32(x.detach()), mean, rstd, f32(gamma.detach()), LL(rows), H, g, dx, dg, db, None)
        close(dx, dx_t, 1e-5)
        close(dg - 1.0, dg_t, 1e-5)
        close(db - 1.0, db_t, 1e-5)


def test_gemm_twin_vs_matmul(ref):
    torch.manual_seed(3)
    M, N, K = 5, 7, 9
    A, Bm, bias = torch.randn(M, K, dtype=torch.float64), torch.randn(K, N, dtype=torch.float64), torch.randn(N, dtype=torch.float64)
    for tA in (0, 1):
        for tB in (0, 1):
            a = f32(A.t() if tA else A)
            b = f32(Bm.t() if tB else Bm)
            C = np.ones((M, N), np.float32)
            call(ref, "cruse_gemm", tA, tB, M, N, K, a, a.shape[1], b, b.shape[1], C, N, f32(bias), 1, 1, 0, 0, None)
            close(C - 1.0, A @ Bm + bias, 3e-6)
    # the h_{t-1} operand of dW_hh: row k of B is row k - 1, zero at the first frame of each clip
    Tq = 3
    Bs = torch.cat([torch.zeros(1, N, dtype=torch.float64), Bm[:-1]]).clone()
    Bs[::Tq] = 0
    C = np.zeros((M, N), np.float32)
    call(ref, "cruse_gemm", 0, 0, M, N, K, f32(A), K, f32(Bm), N, C, N, None, 0, 1, Tq, 0, None)
    close(C, A @ Bs, 3e-6)
    # bf16 mode: RNE-rounded operands
    C = np.zeros((M, N), np.float32)
    call(ref, "cruse_gemm", 0, 0, M, N, K, f32(A), K, f32(Bm), N, C, N, None, 0, 1, 0, 2, None)
    close(C, A.float().bfloat16().double() @ Bm.float().bfloat16().double(), 3e-6)


def test_gru_twins_vs_torch_gru(ref):
    """nn.GRU (cruse_net.py:23-31,44,50) forward, and the whole backward assembled from the twins: dh -> (dgi, dgh) -> dx, dW_ih, dW_hh, biases"""
    torch.manual_seed(4)
    B, T, G, Hg, I = 2, 6, 2, 8, 5
    grus = [torch.nn.GRU(I, Hg, batch_first=True).double() for _ in range(G)]
    xs = [torch.randn(B, T, I, dtype=torch.float64, requires_grad=True) for _ in range(G)]
    hs = [gru(x)[0] for gru, x in zip(grus, xs)]
    h_t = torch.cat(hs, -1)                                        # [B, T, G*Hg] ("cat" layout)
    gi = torch.stack([x @ gru.weight_ih_l0.t() + gru.bias_ih_l0 for gru, x in zip(grus, xs)], 2)   # [B, T, G, 3*Hg]
    w_hh = [f32(gru.weight_hh_l0.detach()) for gru in grus]
    b_hh = [f32(gru.bias_hh_l0.detach()) for gru in grus]
    H = G * Hg
    h, an, z = (np.zeros((B, T, H), np.float32) for _ in range(3))
    coef = np.zeros((B, T, G, 3 * Hg), np.float32)
    call(ref, "cruse_gru_seq_fwd", f32(gi.detach()), R.ptr_array(w_hh), R.ptr_array(b_hh), h, coef, an, z, B, T, G, Hg, 0, None, None)
    close(h, h_t.detach(), 3e-6)
    h2 = np.zeros_like(h)
    call(ref, "cruse_gru_seq_fwd", f32(gi.detach()), R.ptr_array(w_hh), R.ptr_array(b_hh), h2, None, None, None, B, T, G, Hg, 0, None, None)
    assert np.array_equal(h, h2)
    dout = torch.randn_like(h_t)
    params = [p for gru in grus for p in (gru.weight_ih_l0, gru.weight_hh_l0, gru.bias_ih_l0, gru.bias_hh_l0)]
    grads = torch.autograd.grad(h_t, xs + params, dout)
    dx_t, dp_t = grads[:G], grads[G:]
    dh = np.zeros((B, T, H), np.float32)
    call(ref, "cruse_gru_seq_bwd", f32(dout), R.ptr_array(w_hh), coef, z, dh, B, T, G, Hg, 0, None, None)
    dgi, dgh = np.zeros((B * T, G, 3 * Hg), np.float32), np.zeros((B * T, G, 3 * Hg), np.float32)
    call(ref, "cruse_gru_gate_grads", dh, coef, an, dgi, dgh, LL(B * T), G, Hg, 0, None)
    hprev = np.concatenate([np.zeros((B, 1, H), np.float32), h[:, :-1]], 1).reshape(B * T, G, Hg).astype(np.float64)
    for g in range(G):
        gru = grus[g]
        d_i, d_h = dgi[:, g].astype(np.float64), dgh[:, g].astype(np.float64)
        close(d_i @ gru.weight_ih_l0.detach().numpy(), dx_t[g].reshape(B * T, I), 2e-5)
        close(d_i.T @ xs[g].detach().numpy().reshape(B * T, I), dp_t[4 * g], 2e-5)
        close(d_h.T @ hprev[:, g], dp_t[4 * g + 1], 2e-5)
        close(d_i.sum(0), dp_t[4 * g + 2], 2e-5)
        close(d_h.sum(0), dp_t[4 * g + 3], 2e-5)
    # bf16 mode: the coefficients are stored as bf16
    cb = np.zeros((B, T, G, 3 * Hg), np.uint16)
    call(ref, "cruse_gru_seq_fwd", f32(gi.detach()), R.ptr_array(w_hh), R.ptr_array(b_hh), h2, cb, an, z, B, T, G, Hg, 2, None, None)
    close(torch.from_numpy(cb.astype(np.int32) << 16).view(torch.float32), torch.from_numpy(coef).to(torch.bfloat16).float(), 0.0)


def test_mask_loss_twin_vs_pinned_oracle_and_autograd(ref):
    from oracle import cruse_oracle as O
    g5 = np.load(os.path.join(GOLD, "g5_loss.npz"))
    # the oracle's wo_male is pinned by G5 ...
    v = O.wo_male(torch.from_numpy(g5["ref"]), torch.from_numpy(g5["est"]), torch.from_numpy(g5["unproc"]))
    assert abs(float(v) - float(g5["wo_male"])) <= 1e-6 * abs(float(g5["wo_male"]))
    # ... and the twin is the oracle's masking + wo_male on the same noisy / clean spectra
    unproc = torch.from_numpy(g5["unproc"]).double()               # [B, 2, T, F]
    clean = torch.from_numpy(g5["ref"]).double()
    B, _, T, Fs = unproc.shape
    Fn = Fs - 1
    torch.manual_seed(5)
    mask = torch.rand(B, T, Fn, dtype=torch.float64, requires_grad=True)
    mfull = F.pad(mask, (0, Fs - Fn))
    est = torch.stack([mfull * unproc[:, 0], mfull * unproc[:, 1]], 1)
    loss = O.wo_male(clean, est, unproc)
    dm_t, = torch.autograd.grad(loss, mask)
    rows = B * T
    cmag = torch.sqrt(clean[:, 0] ** 2 + clean[:, 1] ** 2)
    ls = np.zeros(1)
    dmask, dlogit = np.zeros((rows, Fn), np.float32), np.zeros((rows, Fn), np.float32)
    er, ei = np.zeros((rows, Fs), np.float32), np.zeros((rows, Fs), np.float32)
    call(ref, "cruse_mask_loss_fwd", f32(mask.detach()), f32(unproc[:, 0]), f32(unproc[:, 1]), f32(cmag), LL(rows), Fn, Fs, 2.0, 1.0, ls, dmask,
         dlogit, er, ei, None)
    assert abs(ls[0] / (rows * Fs) - float(loss)) <= 2e-6 * abs(float(loss))
    close(dmask.reshape(B, T, Fn), dm_t, 1e-5)
    close(dlogit.reshape(B, T, Fn), dm_t * mask.detach() * (1 - mask.detach()), 1e-5)
    close(er.reshape(B, T, Fs), est[:, 0].detach(), 2e-6)
    er2, ei2 = np.zeros_like(er), np.zeros_like(ei)
    call(ref, "cruse_mask_apply", f32(mask.detach()), f32(unproc[:, 0]), f32(unproc[:, 1]), LL(rows), Fn, Fs, er2, ei2, None)
    close(er2, er, 1e-6)
    close(ei2, ei, 1e-6)
    dl = np.zeros((rows, Fn), np.float32)
    call(ref, "cruse_sigmoid_bwd", dmask, f32(mask.detach()).reshape(rows, Fn), dl, LL(rows * Fn), None)
    close(dl, dlogit, 1e-6)


def test_deepfilter_twin_vs_golden(ref):
    g8 = np.load(os.path.join(GOLD, "g8_deepfilter.npz"))
    B, Fq, T = g8["xr"].shape
    o_r, o_i = np.zeros((B, Fq, T), np.float32), np.zeros((B, Fq, T), np.float32)
    call(ref, "cruse_deepfilter_fwd", f32(g8["xr"]), f32(g8["xi"]), f32(g8["hr"]), f32(g8["hi"]), B, Fq, T, 5, 1, o_r, o_i, None)        # DeepFilter(t_dim = 1, f_dim = 5)
    close(np.concatenate([o_r, o_i], 1), g8["y"], 3e-6)


def test_adam_twin_vs_torch_adam(ref):
    torch.manual_seed(6)
    n = 50
    for wd in (0.0, 0.01):
        p_t = torch.randn(n, dtype=torch.float64, requires_grad=True)
        opt = torch.optim.Adam([p_t], lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=wd)
        p, m, v = f32(p_t.detach()), np.zeros(n, np.float32), np.zeros(n, np.float32)
        for step in range(1, 5):
            g = torch.randn(n, dtype=torch.float64)
            p_t.grad = g.clone()
            opt.step()
            call(ref, "cruse_adam_step", p, f32(g * 4.0), m, v, LL(n), 1e-3, 0.9, 0.999, 1e-8, wd, step, 0.25, None)
        close(p, p_t.detach(), 2e-6)


def test_adjoint_and_time_domain_twins_vs_torch(ref):
    from oracle import cruse_oracle as O
    torch.manual_seed(7)
    # istft backward = autograd of torch.istft
    B, T, Fb, L = 2, 9, 161, 1280
    X = torch.randn(B, Fb, T, dtype=torch.complex128, requires_grad=True)
    y = torch.istft(X, 320, 160, 320, torch.hann_window(320, dtype=torch.float64), center=True, length=L)
    dw = torch.randn(B, L, dtype=torch.float64)
    gX, = torch.autograd.grad(y, X, dw)
    dre, dim = np.zeros((B, T, Fb), np.float32), np.zeros((B, T, Fb), np.float32)
    call(ref, "cruse_istft_bwd", f32(dw), B, T, 320, 160, L, dre, dim, None)
    # (torch hands back the gradient of a complex leaf as dre + i dim; bins 0 and 160 carry no imaginary gradient)
    close(dre, gX.real.permute(0, 2, 1), 5e-6)
    gi = gX.imag.permute(0, 2, 1).clone()
    gi[..., 0] = 0; gi[..., -1] = 0
    close(dim, gi, 5e-6)
    # mask_apply backward
    rows, Fn, Fs = 6, 8, 9
    mask = torch.rand(rows, Fn, dtype=torch.float64, requires_grad=True)
    logit = torch.randn(rows, Fn, dtype=torch.float64, requires_grad=True)
    nre, nim = torch.randn(rows, Fs, dtype=torch.float64), torch.randn(rows, Fs, dtype=torch.float64)
    dre_t, dim_t = torch.randn(rows, Fs, dtype=torch.float64), torch.randn(rows, Fs, dtype=torch.float64)
    for through in (0, 1):
        m = torch.sigmoid(logit) if through else mask
        mf = F.pad(m, (0, Fs - Fn))
        g, = torch.autograd.grad(((mf * nre) * dre_t + (mf * nim) * dim_t).sum(), logit if through else mask)
        out = np.zeros((rows, Fn), np.float32)
        call(ref, "cruse_mask_apply_bwd", f32(dre_t), f32(dim_t), f32(nre), f32(nim), f32(m.detach()), LL(rows), Fn, Fs, through, out, None)
        close(out, g, 3e-6)
    # si_snr_loss (train_base/loss.py:7-25) through the pinned oracle
    Bq, Lq = 3, 400
    x = torch.randn(Bq, Lq, dtype=torch.float64, requires_grad=True)
    s = torch.randn(Bq, Lq, dtype=torch.float64)
    loss_t = O.si_snr_loss(x, s)
    gx, = torch.autograd.grad(loss_t, x)
    mom, loss, coef = np.zeros((Bq, 5)), np.zeros(1), np.zeros((Bq, 4), np.float32)
    call(ref, "cruse_sisnr_fwd", f32(x.detach()), f32(s), Bq, Lq, 1e-8, mom, loss, coef, None)
    assert abs(loss[0] - float(loss_t)) <= 1e-5 * abs(float(loss_t))
    dx = np.zeros((Bq, Lq), np.float32)
    call(ref, "cruse_sisnr_bwd", f32(x.detach()), f32(s), coef, Bq, Lq, 0.5, dx, None)
    close(dx, 0.5 * gx, 2e-5)
    # L1 / MSE on waveforms
    est = torch.randn(1000, dtype=torch.float64, requires_grad=True)
    refw = torch.randn(1000, dtype=torch.float64)
    for mse, fn in ((0, F.l1_loss), (1, F.mse_loss)):
        lt = fn(est, refw, reduction="sum")
        g, = torch.autograd.grad(lt, est)
        ls, d = np.zeros(1), np.zeros(1000, np.float32)
        call(ref, "cruse_wave_l1_mse", f32(est.detach()), f32(refw), LL(1000), mse, 0.001, ls, d, None)
        assert abs(ls[0] - float(lt)) <= 2e-6 * abs(float(lt))
        close(d, 0.001 * g, 2e-6)
    # DeepFilter backward through the oracle's module (pinned by G8)
    ts = [torch.randn(2, 9, 11, dtype=torch.float64, requires_grad=True) for _ in range(4)]
    w = torch.randn(2, 18, 11, dtype=torch.float64)
    (O.DeepFilter(1, 2).double()(ts[:2], ts[2:]) * w).sum().backward()
    outs = [np.zeros((2, 9, 11), np.float32) for _ in range(4)]
    call(ref, "cruse_deepfilter_bwd", f32(w[:, :9]), f32(w[:, 9:]), *[f32(t.detach()) for t in ts], 2, 9, 11, 2, 1, *outs, None)
    for o, t in zip(outs, ts):
        close(o, t.grad, 3e-6)


def test_bookkeeping_twins(ref):
    g = np.random.default_rng(8)
    C, rows, Fq = 5, 12, 7
    rm, rv = f32(g.standard_normal(C)), f32(np.abs(g.standard_normal(C)) + 0.5)
    mean, rstd = np.zeros(C, np.float32), np.zeros(C, np.float32)
    call(ref, "cruse_bn_eval_stats", rm, rv, C, 1e-5, mean, rstd, None)
    close(mean, rm, 0.0)
    close(rstd, 1.0 / np.sqrt(rv.astype(np.float64) + 1e-5), 1e-6)
    # the fused forward == finalize + act on the folded replicas
    y = f32(g.standard_normal((rows, C, Fq)) * 1.3 + 0.2)
    gamma, beta, skip = f32(g.standard_normal(C)), f32(g.standard_normal(C)), f32(g.standard_normal((rows, C, Fq)))
    s1 = np.zeros(2 * C)
    call(ref, "cruse_bn_stats", y, LL(rows), C, Fq, s1, 0, None)
    reps = np.ascontiguousarray(np.stack([s1 * 0.5, s1 * 0.25, s1 * 0.25, s1 * 0.0]))
    m1, r1, m2, r2 = (np.zeros(C, np.float32) for _ in range(4))
    ra, rb, rc, rd = rm.copy(), rv.copy(), rm.copy(), rv.copy()
    o1, o2 = np.zeros((rows, C, Fq), np.float32), np.zeros((rows, C, Fq), np.float32)
    call(ref, "cruse_bn_finalize", s1, LL(rows * Fq), C, 1e-5, 0.1, m1, r1, ra, rb, None)
    call(ref, "cruse_bn_act_fwd", y, m1, r1, gamma, beta, skip, o1, LL(rows), C, Fq, 1, None)
    ob = np.zeros((rows, C, Fq), np.uint16)
    call(ref, "cruse_bn_finalize_act_fwd", y, reps, 4, LL(rows * Fq), 1e-5, 0.1, gamma, beta, skip, o2, ob, m2, r2, rc, rd, LL(rows), C, Fq, 1, None)
    for a, b in ((o1, o2), (m1, m2), (r1, r2), (ra, rc), (rb, rd)):
        assert np.array_equal(a, b)
    close(torch.from_numpy(ob.astype(np.int32) << 16).view(torch.float32), torch.from_numpy(o2).to(torch.bfloat16).float(), 0.0)
    x = f32(g.standard_normal((rows, 20)))
    out = np.full(6, 3.0, np.float32)
    call(ref, "cruse_col_sum", np.ascontiguousarray(x.reshape(-1)[4:]), LL(rows - 1), 6, 20, out, None)     # a column slice: ld 20, first column 4
    close(out - 3.0, x[:rows - 1, 4:10].astype(np.float64).sum(0), 2e-6)
    a, b = f32(g.standard_normal(100)), f32(g.standard_normal(100))
    o = np.zeros(100, np.float32)
    call(ref, "cruse_axpby", o, a, b, 0.5, -2.0, LL(100), None)
    close(o, 0.5 * a.astype(np.float64) - 2.0 * b, 1e-6)
    call(ref, "cruse_axpby", o, a, None, 3.0, 1.0, LL(100), None)
    close(o, 3.0 * a.astype(np.float64), 1e-6)
    xs = f32(np.concatenate([g.standard_normal(1000), [1.00390625, 1.01171875, -1.00390625, 3.3895314e38, 1e-40, 0.0]]))   # ties, overflow edge, subnormal
    yb = np.zeros(xs.size, np.uint16)
    call(ref, "cruse_cast_bf16", xs, yb, LL(xs.size), None)
    want = torch.from_numpy(xs).to(torch.bfloat16).view(torch.int16).numpy().view(np.uint16)
    assert np.array_equal(yb, want)
