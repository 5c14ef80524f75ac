"""ABI-level parity: every core entry point of libcruse_hip.so against its plain-C CPU twin (oracle/cruse_ref.c, pinned
by tests/test_ref_twins.py) -- the SAME argument list goes to both, host arrays to the twin and their device copies to the HIP entry
point, and every output buffer is compared.  Exact modes (VALU kernels, v_mfma_f32_16x16x4_f32) to f32 rounding, the bf16 MFMA modes to
the rounding of their operands."""
import ctypes
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import ref_lib as R  # noqa: E402
from ref_lib import LL  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ref():
    return R.load()


@pytest.fixture(scope="module")
def hip():
    from cruse_amd._lib import lib
    return lib


class Side:
    """an argument that differs between the two sides (scratch buffers, host arrays of device / host pointers)"""
    def __init__(self, hip, ref):
        self.hip, self.ref = hip, ref


def PLACE(a):
    """device mirror of a host array (tools/abi_guard_sweep.py swaps this for a placement at the very end of an allocator segment, so that a
    read or write past a tensor leaves the mapping and faults)"""
    return torch.from_numpy(a).cuda()


def rng(seed):
    return np.random.default_rng(seed)


def rnd(g, *shape, scale=1.0, shift=0.0):
    return np.ascontiguousarray((g.standard_normal(shape) * scale + shift).astype(np.float32))


def both(hip, ref, name, args, outs, tol, l2=False):
    """call entry point `name` and its twin with `args` (numpy arrays are mirrored to the device); compare the arrays at the
    positions `outs`.  tol: max |a - b| / max |b| (or rel-L2 with l2=True); a dict {position: tol} sets it per output."""
    from cruse_amd._lib import check
    dev = {}
    hargs, rargs = [], []
    for i, a in enumerate(args):
        if isinstance(a, Side):
            hargs.append(a.hip); rargs.append(a.ref)
        elif isinstance(a, np.ndarray):
            t = PLACE(a)
            dev[i] = t
            hargs.append(t.data_ptr()); rargs.append(a)
        elif isinstance(a, LL):
            hargs.append(a.value); rargs.append(a)
        else:
            hargs.append(a); rargs.append(a)
    hargs[-1] = torch.cuda.current_stream().cuda_stream
    check(getattr(hip, name)(*hargs))
    torch.cuda.synchronize()
    R.call(ref, name, *rargs)
    for i in outs:
        got = dev[i].cpu().numpy()
        want = args[i]
        if got.dtype == np.uint16:                                # bf16 payloads
            got = (got.astype(np.uint32) << 16).view(np.float32)
            want = (want.astype(np.uint32) << 16).view(np.float32)
        got, want = got.astype(np.float64), want.astype(np.float64)
        assert np.isfinite(got).all(), (name, i)
        tl = tol[i] if isinstance(tol, dict) else tol
        if l2:
            err = np.linalg.norm(got - want) / max(np.linalg.norm(want), 1e-30)
        else:
            err = np.abs(got - want).max() / max(np.abs(want).max(), 1e-30)
        assert err <= tl, (name, i, err, tl)


def test_stft_and_istft(hip, ref):
    g = rng(0)
    for (B, L, n_fft, hop) in ((2, 3200, 320, 160), (2, 3199, 320, 160), (2, 1000, 64, 16)):
        T, Fb = 1 + L // hop, n_fft // 2 + 1
        x = rnd(g, B, L, scale=0.1)
        re, im, mag = (np.zeros((B, T, Fb), np.float32) for _ in range(3))
        both(hip, ref, "cruse_stft_fwd", [x, B, L, n_fft, hop, re, im, mag, Fb, 1e-12, None], (5, 6, 7), 3e-5)
        if n_fft != 320:
            continue                                               # (cruse_istft_fwd is the 320-point form; other sizes: cruse_istft_framed)
        L2 = (T - 1) * hop
        wave = np.zeros((B, L2), np.float32)
        both(hip, ref, "cruse_istft_fwd", [rnd(g, B, T, Fb), rnd(g, B, T, Fb), B, T, n_fft, hop, L2, wave, None], (7,), 3e-5)


def test_conv_forms(hip, ref):
    g = rng(1)
    B, T = 2, 5
    # (Cin, Fin, Cout, Fout, KT, S, pad, w_layout): the encoder's first conv (VALU only), an MFMA-eligible encoder conv, the stride-1 data gradient
    for (Cin, Fin, Cout, Fout, KT, S, pad, wl) in ((1, 160, 16, 80, 2, 2, 1, 0), (16, 80, 32, 40, 2, 2, 1, 0), (16, 40, 16, 40, 1, 1, 1, 1),
                                                    (16, 40, 16, 40, 1, 1, 1, 0)):
        x = rnd(g, B, T, Cin, Fin)
        w = rnd(g, *((Cin, Cout, 1, 3) if wl else (Cout, Cin, KT, 3)), scale=0.3)
        bias = rnd(g, Cout)
        for prec, tol in ((-1, 2e-5), (0, 2e-5), (1, 1e-4), (2, 2e-2)):
            y = np.zeros((B, T, Cout, Fout), np.float32)
            both(hip, ref, "cruse_conv_gather", [x, w, bias, y, B, T, Cin, Fin, Cout, Fout, KT, S, pad, wl, 0, 0, prec, 0, 0, None], (3,), tol,
                 l2=prec > 0)
        y = rnd(g, B, T, Cout, Fout)
        both(hip, ref, "cruse_conv_gather", [x, w, None, y, B, T, Cin, Fin, Cout, Fout, KT, S, pad, wl, 0, 1, -1, 0, 0, None], (3,), 2e-5)   # accumulate
        y = np.zeros((B, T, Cout, Fout), np.float32)
        both(hip, ref, "cruse_conv_gather", [x, w, bias, y, B, T, Cin, Fin, Cout, Fout, KT, S, pad, wl, 1, 0, -1, 0, 0, None], (3,), 2e-5)   # sigmoid
    # scatter form: the decoder's ConvTranspose2d (KT 1, pad 0) and the data gradient of the encoder conv (KT 2, pad 1)
    for (Cs, Fg, Cout, KT, pad) in ((32, 20, 16, 1, 0), (32, 20, 16, 2, 1), (16, 80, 1, 1, 0)):
        gq = rnd(g, B, T, Cs, Fg)
        w = rnd(g, Cs, Cout, KT, 3, scale=0.3)
        bias = rnd(g, Cout)
        for prec, tol in ((-1, 2e-5), (0, 2e-5), (2, 2e-2)):
            y = np.zeros((B, T, Cout, 2 * Fg), np.float32)
            both(hip, ref, "cruse_conv_scatter2", [gq, w, bias, y, B, T, Cs, Fg, Cout, 2 * Fg, KT, pad, 0, 0, prec, 0, 0, None], (3,), tol, l2=prec > 0)
        y = rnd(g, B, T, Cout, 2 * Fg)
        both(hip, ref, "cruse_conv_scatter2", [gq, w, None, y, B, T, Cs, Fg, Cout, 2 * Fg, KT, pad, 0, 1, -1, 0, 0, None], (3,), 2e-5)


def test_conv_weight_gradient_and_channel_sum(hip, ref):
    g = rng(2)
    B, T = 2, 6
    for (Ca, Fa, Cb, Fb, KT, S, pad) in ((32, 40, 16, 80, 2, 2, 1), (16, 80, 16, 80, 1, 1, 1), (16, 80, 1, 160, 2, 2, 1)):
        a, bt = rnd(g, B, T, Ca, Fa), rnd(g, B, T, Cb, Fb)
        nws = hip.cruse_conv_wgrad_ws_bytes(Ca, Cb, KT)
        ws = torch.zeros(max(nws, 16), dtype=torch.uint8, device="cuda")
        for prec, tol in ((-1, 3e-5), (0, 3e-5), (2, 2e-2)):
            dw = rnd(g, Ca, Cb, KT, 3)                              # "+=": the gradient buffer already holds something
            both(hip, ref, "cruse_conv_wgrad", [a, bt, dw, B, T, Ca, Fa, Cb, Fb, KT, S, pad, prec, 0, 0, Side(ws.data_ptr(), None), None], (2,), tol,
                 l2=prec > 0)
    x = rnd(g, B * T, 24, 33)
    out = rnd(g, 24)
    both(hip, ref, "cruse_channel_sum", [x, LL(B * T), 24, 33, out, None], (4,), 2e-5)


def test_batchnorm_family(hip, ref):
    g = rng(3)
    rows, C, Fq = 42, 16, 40
    y = rnd(g, rows, C, Fq, scale=1.5, shift=0.3)
    gamma, beta = rnd(g, C, scale=0.2, shift=1.0), rnd(g, C, scale=0.1)
    sums = np.zeros(2 * C)
    both(hip, ref, "cruse_bn_stats", [y, LL(rows), C, Fq, sums, 0, None], (4,), 1e-6)
    R.call(ref, "cruse_bn_stats", y, LL(rows), C, Fq, sums, 0, None)
    mean, rstd = np.zeros(C, np.float32), np.zeros(C, np.float32)
    rm, rv = rnd(g, C), np.abs(rnd(g, C)) + 0.5
    both(hip, ref, "cruse_bn_finalize", [sums, LL(rows * Fq), C, 1e-5, 0.1, mean, rstd, rm, rv, None], (5, 6, 7, 8), 2e-6)
    R.call(ref, "cruse_bn_finalize", sums, LL(rows * Fq), C, 1e-5, 0.1, mean, rstd, None, None, None)
    skip, dout = rnd(g, rows, C, Fq), rnd(g, rows, C, Fq)
    for relu in (1, 0):
        out = np.zeros((rows, C, Fq), np.float32)
        both(hip, ref, "cruse_bn_act_fwd", [y, mean, rstd, gamma, beta, skip, out, LL(rows), C, Fq, relu, None], (6,), 2e-6)
        bs = np.zeros(2 * C)
        both(hip, ref, "cruse_bn_act_bwd_reduce", [dout, y, mean, rstd, gamma, beta, LL(rows), C, Fq, relu, bs, 0, None], (10,), 1e-5)
        R.call(ref, "cruse_bn_act_bwd_reduce", dout, y, mean, rstd, gamma, beta, LL(rows), C, Fq, relu, bs, 0, None)
        for training in (1, 0):
            dy = np.zeros((rows, C, Fq), np.float32)
            dg, db, dbias = rnd(g, C), rnd(g, C), rnd(g, C)
            both(hip, ref, "cruse_bn_act_bwd_apply", [dout, y, mean, rstd, gamma, beta, bs, 1, LL(rows), C, Fq, relu, training, 0, dy, 0, dg, db, dbias,
                                                     None], (14, 16, 17, 18), 1e-5)


def test_layernorm(hip, ref):
    g = rng(4)
    rows = 42
    for (H, ig) in ((640, 1), (640, 4), (128, 2)):
        x, res = rnd(g, rows, H), rnd(g, rows, H)
        gamma, beta = rnd(g, H, scale=0.3, shift=1.0), rnd(g, H, scale=0.2)
        y, mean, rstd = np.zeros((rows, H), np.float32), np.zeros(rows, np.float32), np.zeros(rows, np.float32)
        yb = np.zeros((rows, H), np.uint16)
        both(hip, ref, "cruse_ln_fwd", [x, gamma, beta, res, y, yb, mean, rstd, LL(rows), H, ig, 1e-5, 0, LL(0), LL(0), None], (4, 5, 6, 7),
             {4: 3e-6, 5: 8e-3, 6: 3e-6, 7: 3e-6})               # (the bf16 copy: one rounding step of 2^-8 where the f32 values differ in the last place)
        R.call(ref, "cruse_ln_fwd", x, gamma, beta, res, y, None, mean, rstd, LL(rows), H, ig, 1e-5, 0, LL(0), LL(0), None)
        dy = rnd(g, rows, H)
        dx, dgm, dbt = np.zeros((rows, H), np.float32), rnd(g, H), rnd(g, H)
        both(hip, ref, "cruse_ln_bwd", [dy, x, mean, rstd, gamma, LL(rows), H, ig, dx, dgm, dbt, None], (8, 9, 10), 2e-5)


def test_gemm(hip, ref):
    g = rng(5)
    M, N, K = 136, 200, 96
    A, Bm, bias = rnd(g, M, K), rnd(g, K, N), rnd(g, N)
    for tA in (0, 1):
        for tB in (0, 1):
            a = np.ascontiguousarray(A.T) if tA else A
            b = np.ascontiguousarray(Bm.T) if tB else Bm
            for prec, tol in ((0, 2e-5), (2, 1e-5)):                # (bf16: the twin rounds the operands the same way; the products are then exact in f32)
                C = rnd(g, M, N)
                both(hip, ref, "cruse_gemm", [tA, tB, M, N, K, a, a.shape[1], b, b.shape[1], C, N, bias, 1, 1, 0, prec, None], (9,), tol)
    C = np.zeros((M, N), np.float32)
    both(hip, ref, "cruse_gemm", [0, 0, M, N, K, A, K, Bm, N, C, N, None, 0, 1, 8, 0, None], (9,), 2e-5)      # the shifted h_{t-1} operand


def test_gru_recurrence_and_gate_gradients(hip, ref):
    g = rng(6)
    for (B, T, G, Hg) in ((3, 9, 2, 32), (9, 7, 1, 64)):
        H = G * Hg
        gi = rnd(g, B, T, G, 3 * Hg)
        w = [rnd(g, 3 * Hg, Hg, scale=1.0 / np.sqrt(Hg)) for _ in range(G)]
        bh = [rnd(g, 3 * Hg, scale=0.1) for _ in range(G)]
        wd, bd = [PLACE(a) for a in w], [PLACE(a) for a in bh]
        wa = Side(ctypes.cast((ctypes.c_void_p * G)(*[t.data_ptr() for t in wd]), ctypes.c_void_p), R.ptr_array(w))
        ba = Side(ctypes.cast((ctypes.c_void_p * G)(*[t.data_ptr() for t in bd]), ctypes.c_void_p), R.ptr_array(bh))
        ws = torch.zeros(hip.cruse_gru_ws_bytes(B, G, Hg), dtype=torch.uint8, device="cuda")
        wsa = Side(ws.data_ptr(), None)
        dout = rnd(g, B, T, H)
        for prec, tol in ((0, 2e-5), (2, 2e-2)):
            h, an, z = (np.zeros((B, T, H), np.float32) for _ in range(3))
            coef = np.zeros((B, T, G, 3 * Hg), np.uint16 if prec == 2 else np.float32)
            both(hip, ref, "cruse_gru_seq_fwd", [gi, wa, ba, h, coef, an, z, B, T, G, Hg, prec, wsa, None], (3, 4, 5, 6), tol, l2=prec == 2)
            R.call(ref, "cruse_gru_seq_fwd", gi, wa.ref, ba.ref, h, coef, an, z, B, T, G, Hg, prec, None, None)
            dh = np.zeros((B, T, H), np.float32)
            both(hip, ref, "cruse_gru_seq_bwd", [dout, wa, coef, z, dh, B, T, G, Hg, prec, wsa, None], (4,), tol, l2=prec == 2)
            R.call(ref, "cruse_gru_seq_bwd", dout, wa.ref, coef, z, dh, B, T, G, Hg, prec, None, None)
            dgi, dgh = np.zeros((B * T, G, 3 * Hg), np.float32), np.zeros((B * T, G, 3 * Hg), np.float32)
            both(hip, ref, "cruse_gru_gate_grads", [dh, coef, an, dgi, dgh, LL(B * T), G, Hg, prec, None], (3, 4), 2e-6)
        assert int(ws[:4].view(torch.int32)[0]) == 0             # no hand-off time-out


def test_mask_loss_deepfilter_adam(hip, ref):
    g = rng(7)
    rows, Fn, Fs = 42, 160, 161
    mask = np.ascontiguousarray(g.uniform(0.05, 0.95, (rows, Fn)).astype(np.float32))
    nre, nim = rnd(g, rows, Fs), rnd(g, rows, Fs)
    cmag = np.abs(rnd(g, rows, Fs))
    ls = np.zeros(1)
    dmask, dlogit = np.zeros((rows, Fn), np.float32), np.zeros((rows, Fn), np.float32)
    er, ei = np.zeros((rows, Fs), np.float32), np.zeros((rows, Fs), np.float32)
    both(hip, ref, "cruse_mask_loss_fwd", [mask, nre, nim, cmag, LL(rows), Fn, Fs, 2.0, 1.0, ls, dmask, dlogit, er, ei, None], (9, 10, 11, 12, 13),
         {9: 1e-5, 10: 5e-5, 11: 5e-5, 12: 1e-6, 13: 1e-6})
    er2, ei2 = np.zeros_like(er), np.zeros_like(ei)
    both(hip, ref, "cruse_mask_apply", [mask, nre, nim, LL(rows), Fn, Fs, er2, ei2, None], (6, 7), 1e-6)
    dl = np.zeros((rows, Fn), np.float32)
    both(hip, ref, "cruse_sigmoid_bwd", [rnd(g, rows, Fn), mask, dl, LL(rows * Fn), None], (2,), 1e-6)
    B, Fq, T = 2, 161, 50
    xs = [rnd(g, B, Fq, T) for _ in range(4)]
    o_r, o_i = np.zeros((B, Fq, T), np.float32), np.zeros((B, Fq, T), np.float32)
    both(hip, ref, "cruse_deepfilter_fwd", xs + [B, Fq, T, 5, 1, o_r, o_i, None], (9, 10), 1e-5)
    n = 5000
    p, gr = rnd(g, n), rnd(g, n)
    m, v = rnd(g, n, scale=0.1), np.abs(rnd(g, n, scale=0.1))
    for wd in (0.0, 0.01):
        both(hip, ref, "cruse_adam_step", [p, gr, m, v, LL(n), 1e-3, 0.9, 0.999, 1e-8, wd, 3, 0.5, None], (0, 2, 3), 2e-6)


def test_adjoints_time_domain_losses_and_bookkeeping(hip, ref):
    g = rng(8)
    B, T, Fb, L = 2, 21, 161, 3200
    dre, dim = np.zeros((B, T, Fb), np.float32), np.zeros((B, T, Fb), np.float32)
    both(hip, ref, "cruse_istft_bwd", [rnd(g, B, L), B, T, 320, 160, L, dre, dim, None], (6, 7), 3e-5)
    rows, Fn, Fs = 42, 160, 161
    mask = np.ascontiguousarray(g.uniform(0.05, 0.95, (rows, Fn)).astype(np.float32))
    for through in (0, 1):
        out = np.zeros((rows, Fn), np.float32)
        both(hip, ref, "cruse_mask_apply_bwd", [rnd(g, rows, Fs), rnd(g, rows, Fs), rnd(g, rows, Fs), rnd(g, rows, Fs), mask, LL(rows), Fn, Fs, through, out,
                                               None], (9,), 3e-6)
    Bq, Lq = 5, 16000
    x, s = rnd(g, Bq, Lq), rnd(g, Bq, Lq)
    x += 0.5 * s
    mom, loss, coef = np.zeros((Bq, 5)), np.zeros(1), np.zeros((Bq, 4), np.float32)
    both(hip, ref, "cruse_sisnr_fwd", [x, s, Bq, Lq, 1e-8, mom, loss, coef, None], (5, 6, 7), {5: 1e-6, 6: 1e-6, 7: 2e-5})
    dx = np.zeros((Bq, Lq), np.float32)
    both(hip, ref, "cruse_sisnr_bwd", [x, s, coef, Bq, Lq, 0.25, dx, None], (6,), 2e-6)
    n = 50000
    est, refw = rnd(g, n), rnd(g, n)
    for mse in (0, 1):
        ls, d = np.zeros(1), np.zeros(n, np.float32)
        both(hip, ref, "cruse_wave_l1_mse", [est, refw, LL(n), mse, 1.0 / n, ls, d, None], (5, 6), 2e-6)
    Bd, Fq, Td = 2, 161, 50
    ts = [rnd(g, Bd, Fq, Td) for _ in range(6)]
    outs = [np.zeros((Bd, Fq, Td), np.float32) for _ in range(4)]
    both(hip, ref, "cruse_deepfilter_bwd", ts + [Bd, Fq, Td, 5, 1] + outs + [None], (11, 12, 13, 14), 1e-5)
    C, Fc = 16, 40
    rm, rv = rnd(g, C), np.abs(rnd(g, C)) + 0.5
    mean, rstd = np.zeros(C, np.float32), np.zeros(C, np.float32)
    both(hip, ref, "cruse_bn_eval_stats", [rm, rv, C, 1e-5, mean, rstd, None], (4, 5), 2e-6)
    y = rnd(g, rows, C, Fc, scale=1.4, shift=0.2)
    s1 = np.zeros(2 * C)
    R.call(ref, "cruse_bn_stats", y, LL(rows), C, Fc, s1, 0, None)
    reps = np.ascontiguousarray(np.stack([s1 * w for w in (0.5, 0.25, 0.125, 0.125)]))
    gamma, beta, skip = rnd(g, C, scale=0.2, shift=1.0), rnd(g, C, scale=0.1), rnd(g, rows, C, Fc)
    out, ob = np.zeros((rows, C, Fc), np.float32), np.zeros((rows, C, Fc), np.uint16)
    both(hip, ref, "cruse_bn_finalize_act_fwd", [y, reps, 4, LL(rows * Fc), 1e-5, 0.1, gamma, beta, skip, out, ob, mean, rstd, rm, rv, LL(rows), C, Fc, 1,
                                                None], (9, 10, 11, 12, 13, 14), {9: 3e-6, 10: 8e-3, 11: 2e-6, 12: 2e-6, 13: 2e-6, 14: 2e-6})
    xg = rnd(g, rows * 20)
    o = rnd(g, 6)
    both(hip, ref, "cruse_col_sum", [xg, LL(rows - 1), 6, 20, o, None], (4,), 3e-6)
    a, b = rnd(g, 10000), rnd(g, 10000)
    o = np.zeros(10000, np.float32)
    both(hip, ref, "cruse_axpby", [o, a, b, 0.5, -2.0, LL(10000), None], (0,), 1e-6)
    both(hip, ref, "cruse_axpby", [o, a, None, 3.0, 0.0, LL(10000), None], (0,), 1e-6)
    xs = np.ascontiguousarray(np.concatenate([rnd(g, 4090), np.array([1.00390625, 1.01171875, -1.00390625, 3.3895314e38, 0.0, -0.0], np.float32)]))
    yb = np.zeros(xs.size, np.uint16)
    both(hip, ref, "cruse_cast_bf16", [xs, yb, LL(xs.size), None], (1,), 0.0)


def test_edge_shapes(hip, ref):
    """the smallest and the ragged cases: one clip, one frame, one channel, odd widths, channel counts off the MFMA tiles' multiples"""
    g = rng(9)
    refused = []

    def both_(name, args, outs, tol, l2=False):
        """both(), or a LOUD refusal: an entry point may decline a shape outside its documented domain (CRUSE_E_SHAPE + message), never answer wrongly"""
        try:
            both(hip, ref, name, args, outs, tol, l2)
        except RuntimeError as ex:
            assert "cruse_hip error -1: " in str(ex) and len(str(ex)) > 30, ex
            refused.append((name, str(ex)[20:90]))
    # convolutions: (B, T, Cin, Fin, Cout, Fout, KT, S, pad)
    for (B, T, Cin, Fin, Cout, Fout, KT, S, pad) in ((1, 1, 1, 5, 1, 5, 1, 1, 1), (1, 1, 8, 10, 8, 5, 2, 2, 1), (3, 2, 24, 20, 40, 10, 2, 2, 1),
                                                     (1, 7, 8, 7, 64, 7, 1, 1, 1), (2, 3, 64, 6, 8, 3, 2, 2, 1), (1, 2, 12, 9, 20, 9, 1, 1, 1)):
        x, w, bias = rnd(g, B, T, Cin, Fin), rnd(g, Cout, Cin, KT, 3, scale=0.3), rnd(g, Cout)
        for prec, tol in ((-1, 2e-5), (0, 3e-5), (2, 2e-2)):
            y = np.zeros((B, T, Cout, Fout), np.float32)
            both_("cruse_conv_gather", [x, w, bias, y, B, T, Cin, Fin, Cout, Fout, KT, S, pad, 0, 0, 0, prec, 0, 0, None], (3,), tol, l2=prec > 0)
            if Cout % 4 or (Cin % 4 and Cin != 1):
                continue                                           # (cruse_conv_wgrad: Ca a multiple of 4, Cb a multiple of 4 or 1 -- refused otherwise, below)
            dw = rnd(g, Cout, Cin, KT, 3)
            nws = hip.cruse_conv_wgrad_ws_bytes(Cout, Cin, KT)
            ws = torch.zeros(max(nws, 16), dtype=torch.uint8, device="cuda")
            both_("cruse_conv_wgrad", [rnd(g, B, T, Cout, Fout), x, dw, B, T, Cout, Fout, Cin, Fin, KT, S, pad, prec, 0, 0,
                                       Side(ws.data_ptr(), None), None], (2,), 3e-5 if prec <= 0 else 2e-2, l2=prec > 0)
    for (B, T, Cs, Fg, Cout, KT, pad) in ((1, 1, 1, 1, 1, 1, 0), (1, 1, 8, 3, 8, 2, 1), (2, 3, 40, 5, 24, 1, 0), (1, 2, 64, 4, 64, 2, 1)):
        gq, w, bias = rnd(g, B, T, Cs, Fg), rnd(g, Cs, Cout, KT, 3, scale=0.3), rnd(g, Cout)
        for prec, tol in ((-1, 2e-5), (0, 3e-5), (2, 2e-2)):
            y = np.zeros((B, T, Cout, 2 * Fg), np.float32)
            both_("cruse_conv_scatter2", [gq, w, bias, y, B, T, Cs, Fg, Cout, 2 * Fg, KT, pad, 0, 0, prec, 0, 0, None], (3,), tol, l2=prec > 0)
    # normalisations on one row / one position
    for (rows, C, Fq) in ((1, 1, 1), (1, 3, 7), (5, 130, 3)):
        y = rnd(g, rows, C, Fq, shift=0.5)
        sums = np.zeros(2 * C)
        both_("cruse_bn_stats", [y, LL(rows), C, Fq, sums, 0, None], (4,), 1e-6)
        o = rnd(g, C)
        both_("cruse_channel_sum", [y, LL(rows), C, Fq, o, None], (4,), 2e-5)
    for (rows, H, ig) in ((1, 32, 1), (3, 96, 3), (2, 1024, 4)):
        x, res = rnd(g, rows, H), rnd(g, rows, H)
        gamma, beta = rnd(g, H, scale=0.3, shift=1.0), rnd(g, H, scale=0.2)
        y, mean, rstd = np.zeros((rows, H), np.float32), np.zeros(rows, np.float32), np.zeros(rows, np.float32)
        both_("cruse_ln_fwd", [x, gamma, beta, res, y, None, mean, rstd, LL(rows), H, ig, 1e-5, 0, LL(0), LL(0), None], (4, 6, 7), 5e-6)
        dx, dgm, dbt = np.zeros((rows, H), np.float32), rnd(g, H), rnd(g, H)
        both_("cruse_ln_bwd", [rnd(g, rows, H), x, mean, rstd, gamma, LL(rows), H, ig, dx, dgm, dbt, None], (8, 9, 10), 2e-5)
    # the recurrence on one clip / one frame / the largest unit count, and a batch that is not a multiple of a chain
    for (B, T, G, Hg) in ((1, 1, 1, 32), (1, 5, 1, 1024), (13, 3, 4, 32), (17, 2, 1, 160)):
        H = G * Hg
        gi = rnd(g, B, T, G, 3 * Hg)
        w = [rnd(g, 3 * Hg, Hg, scale=1.0 / np.sqrt(Hg)) for _ in range(G)]
        bh = [rnd(g, 3 * Hg, scale=0.1) for _ in range(G)]
        wd, bd = [PLACE(a) for a in w], [PLACE(a) for a in bh]
        wa = Side(ctypes.cast((ctypes.c_void_p * G)(*[t.data_ptr() for t in wd]), ctypes.c_void_p), R.ptr_array(w))
        ba = Side(ctypes.cast((ctypes.c_void_p * G)(*[t.data_ptr() for t in bd]), ctypes.c_void_p), R.ptr_array(bh))
        ws = torch.zeros(hip.cruse_gru_ws_bytes(B, G, Hg), dtype=torch.uint8, device="cuda")
        wsa = Side(ws.data_ptr(), None)
        for prec, tol in ((0, 3e-5), (2, 3e-2)):
            h, an, z = (np.zeros((B, T, H), np.float32) for _ in range(3))
            coef = np.zeros((B, T, G, 3 * Hg), np.uint16 if prec == 2 else np.float32)
            both_("cruse_gru_seq_fwd", [gi, wa, ba, h, coef, an, z, B, T, G, Hg, prec, wsa, None], (3, 4, 5, 6), tol, l2=prec == 2)
            dh = np.zeros((B, T, H), np.float32)
            both_("cruse_gru_seq_bwd", [rnd(g, B, T, H), wa, coef, z, dh, B, T, G, Hg, prec, wsa, None], (4,), tol, l2=prec == 2)
        assert int(ws[:4].view(torch.int32)[0]) == 0
    # a shape outside an entry point's domain is refused with CRUSE_E_SHAPE and a message, never computed wrongly
    a1, b1, d1 = (torch.zeros(8, device="cuda") for _ in range(3))
    assert hip.cruse_conv_wgrad(a1.data_ptr(), b1.data_ptr(), d1.data_ptr(), 1, 1, 1, 5, 1, 5, 1, 1, 1, -1, 0, 0, None, None) == -1
    assert b"multiple of 4" in hip.cruse_last_error()
    # STFT of clips shorter than a frame and a half (reflect padding reaches back over most of the clip; at L = 161, the shortest torch
    # admits, both frames are mirror-symmetric and the imaginary part is identically 0 -- compared on the real part there)
    for L, outs in ((161, (5,)), (200, (5, 6)), (321, (5, 6))):
        T = 1 + L // 160
        x = rnd(g, 1, L, scale=0.1)
        re, im = np.zeros((1, T, 161), np.float32), np.zeros((1, T, 161), np.float32)
        both_("cruse_stft_fwd", [x, 1, L, 320, 160, re, im, None, 0, 0.0, None], outs, 3e-5)
    print("refused (outside the documented domain):", sorted(set(refused)))
    # the convolutions, the forward recurrence and the STFT take every shape above; the documented limits that do refuse: bn_stats rows of C*F % 4 != 0 or
    # > 768 floats, weight-gradient tile counts the exact VALU kernel cannot split over 256 threads, the f32 backward recurrence beyond Hg = 800 (LDS)
    assert not any(n in ("cruse_conv_gather", "cruse_conv_scatter2", "cruse_gru_seq_fwd", "cruse_stft_fwd", "cruse_ln_fwd", "cruse_ln_bwd") for n, _ in refused)
    assert all("Hg=1024" in m for n, m in refused if n == "cruse_gru_seq_bwd")
    assert set(n for n, _ in refused) <= {"cruse_bn_stats", "cruse_conv_wgrad", "cruse_gru_seq_bwd"}


def test_conv_wgrad_sweep_over_small_and_odd_widths(hip, ref):
    """every width 1..13 (odd ones used to reach the LDS-staged MFMA kernel, which pairs the bins of a row: wrong sums, found by this sweep in r5 --
    they now take the exact VALU kernel), T of 1 and 5, both strides and tap counts, in the f32-MFMA and bf16 modes: the result is the twin's or the
    call is refused"""
    g = rng(10)
    checked = 0
    for prec, tol in ((0, 1e-4), (2, 3e-2)):
        for KT, S in ((1, 1), (2, 2), (1, 2), (2, 1)):
            for Ca, Cb in ((8, 8), (64, 8), (20, 12), (8, 64)):
                for Fa in range(1, 14):
                    for T in (1, 5):
                        B, Fb = 2, Fa * S
                        a, x, dw = rnd(g, B, T, Ca, Fa), rnd(g, B, T, Cb, Fb), rnd(g, Ca, Cb, KT, 3)
                        want = dw.copy()
                        R.call(ref, "cruse_conv_wgrad", a, x, want, B, T, Ca, Fa, Cb, Fb, KT, S, 1, prec, 0, 0, None, None)
                        ws = torch.zeros(max(hip.cruse_conv_wgrad_ws_bytes(Ca, Cb, KT), 16), dtype=torch.uint8, device="cuda")
                        ad, xd, dd = PLACE(a), PLACE(x), PLACE(dw)
                        rc = hip.cruse_conv_wgrad(ad.data_ptr(), xd.data_ptr(), dd.data_ptr(), B, T, Ca, Fa, Cb, Fb, KT, S, 1, prec, 0, 0, ws.data_ptr(),
                                                  torch.cuda.current_stream().cuda_stream)
                        if rc != 0:
                            assert rc == -1 and hip.cruse_last_error(), (prec, KT, S, Ca, Cb, Fa, T)
                            continue
                        got = dd.cpu().numpy().astype(np.float64)
                        err = np.abs(got - want).max() / np.abs(want).max()
                        assert err <= tol, (prec, KT, S, Ca, Cb, Fa, T, err)
                        checked += 1
    assert checked > 600


def test_conv_sweep_over_small_and_odd_widths(hip, ref):
    """gather and scatter forms over widths 1 ... 40 (odd ones too), one and three frames, every tap / stride / pad / weight-layout form and channel
    pairs on and off the MFMA tiles, in the f32-MFMA, split-bf16 and bf16 modes: the twin's result, or a refusal"""
    g = rng(11)
    st = torch.cuda.current_stream().cuda_stream
    checked = 0
    for prec, tol in ((0, 1e-4), (1, 2e-4), (2, 3e-2)):
        for KT, S, pad, wl in ((2, 2, 1, 0), (1, 1, 1, 1), (2, 1, 1, 0), (1, 2, 0, 0)):
            for Cin, Cout in ((8, 8), (64, 16), (24, 40), (8, 1), (1, 8)):
                for Fin in (1, 3, 5, 8, 9, 17, 40):
                    for T in (1, 3):
                        B, Fout = 2, (Fin + 2 * pad - 3) // S + 1
                        if Fout <= 0:
                            continue
                        x, bias = rnd(g, B, T, Cin, Fin), rnd(g, Cout)
                        w = rnd(g, *((Cin, Cout, 1, 3) if wl else (Cout, Cin, KT, 3)), scale=0.3)
                        want = np.zeros((B, T, Cout, Fout), np.float32)
                        R.call(ref, "cruse_conv_gather", x, w, bias, want, B, T, Cin, Fin, Cout, Fout, KT, S, pad, wl, 0, 0, prec, 0, 0, None)
                        xd, wd, bd = PLACE(x), PLACE(w), PLACE(bias)
                        yd = PLACE(np.zeros((B, T, Cout, Fout), np.float32))
                        rc = hip.cruse_conv_gather(xd.data_ptr(), wd.data_ptr(), bd.data_ptr(), yd.data_ptr(), B, T, Cin, Fin, Cout, Fout, KT, S, pad, wl, 0, 0,
                                                   prec, 0, 0, st)
                        if rc != 0:
                            assert rc == -1 and hip.cruse_last_error()
                            continue
                        got = yd.cpu().numpy().astype(np.float64)
                        err = np.linalg.norm(got - want) / max(np.linalg.norm(want), 1e-30)
                        assert err <= tol, ("gather", prec, KT, S, pad, wl, Cin, Cout, Fin, T, err)
                        checked += 1
        for KT, pad in ((1, 0), (2, 1)):
            for Cs, Cout in ((8, 8), (64, 64), (40, 24), (16, 1)):
                for Fg in (1, 3, 5, 8, 13, 40):
                    for T in (1, 3):
                        B = 2
                        gq, w, bias = rnd(g, B, T, Cs, Fg), rnd(g, Cs, Cout, KT, 3, scale=0.3), rnd(g, Cout)
                        want = np.zeros((B, T, Cout, 2 * Fg), np.float32)
                        R.call(ref, "cruse_conv_scatter2", gq, w, bias, want, B, T, Cs, Fg, Cout, 2 * Fg, KT, pad, 0, 0, prec, 0, 0, None)
                        gd, wd, bd = PLACE(gq), PLACE(w), PLACE(bias)
                        yd = PLACE(np.zeros((B, T, Cout, 2 * Fg), np.float32))
                        rc = hip.cruse_conv_scatter2(gd.data_ptr(), wd.data_ptr(), bd.data_ptr(), yd.data_ptr(), B, T, Cs, Fg, Cout, 2 * Fg, KT, pad, 0, 0, prec, 0, 0, st)
                        if rc != 0:
                            assert rc == -1 and hip.cruse_last_error()
                            continue
                        got = yd.cpu().numpy().astype(np.float64)
                        err = np.linalg.norm(got - want) / max(np.linalg.norm(want), 1e-30)
                        assert err <= tol, ("scatter2", prec, KT, pad, Cs, Cout, Fg, T, err)
                        checked += 1
    assert checked > 900
