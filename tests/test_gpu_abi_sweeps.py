"""Sweeps over small and odd shapes for the entry points WITHOUT a twin of their own, checked against compositions of the twins (fused convolutions:
conv + BatchNorm batch sums / BatchNorm-backward sums), against numpy on the bf16 values (the bf16 / f16 GEMM family and its layout kernels) and against
torch (the general NCHW convolutions, f32 and f16 storage: forward, the _ex forms, data gradient, weight gradient, bias gradient).  tools/abi_guard_sweep.py
--sweeps runs the same functions with every tensor at the end of an allocator segment."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import ref_lib as R  # noqa: E402
from ref_lib import LL  # noqa: E402

pytestmark = pytest.mark.gpu


def PLACE(a):
    return torch.from_numpy(a).cuda()


def D(a):
    return PLACE(a)


def Z(*shape, dtype=np.float32):
    return D(np.zeros(shape, dtype))


def st():
    return torch.cuda.current_stream().cuda_stream


@pytest.fixture(scope="module")
def ref():
    return R.load()


@pytest.fixture(scope="module")
def hip():
    from cruse_amd._lib import lib
    return lib


def test_fused_convolutions_equal_the_composed_twins(hip, ref):
    """cruse_conv_{gather,scatter2}_bnstats / _bnbwd: y is the twin convolution, the 16 replicas of sums add up to the twin statistics of that y"""
    g = np.random.default_rng(4)
    rnd = lambda *s: np.ascontiguousarray(g.standard_normal(s).astype(np.float32))
    NREP = 16

    def rel(got, want, l2=True):
        got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
        return np.linalg.norm(got - want) / max(np.linalg.norm(want), 1e-30)
    bad, n, refused = [], 0, 0
    for prec, tol in ((-1, 1e-5), (0, 1e-4), (1, 2e-4), (2, 3e-2)):
        # gather + bnstats / bnbwd
        for KT, S, pad, wl in ((2, 2, 1, 0), (1, 1, 1, 1), (1, 1, 1, 0)):
            for Cin, Cout in ((8, 8), (16, 32), (64, 16), (24, 40), (1, 16)):
                for Fin in (2, 5, 8, 10, 17, 40):
                    for T in (1, 4):
                        pass
                        B = 2
                        Fout = (Fin + 2 * pad - 3) // S + 1
                        if Fout <= 0: continue
                        x = rnd(B, T, Cin, Fin); w = rnd(*((Cin, Cout, 1, 3) if wl else (Cout, Cin, KT, 3))) * 0.3; bias = rnd(Cout)
                        ytw = np.zeros((B, T, Cout, Fout), np.float32)
                        R.call(ref, "cruse_conv_gather", x, w, bias if not wl else None, ytw, B, T, Cin, Fin, Cout, Fout, KT, S, pad, wl, 0, 0, prec, 0, 0, None)
                        if not wl:
                            s_tw = np.zeros(2 * Cout); R.call(ref, "cruse_bn_stats", ytw, LL(B * T), Cout, Fout, s_tw, 0, None)
                            xd, wd, bd = D(x), D(w), D(bias); yd = Z(B, T, Cout, Fout); sd = Z(NREP, 2 * Cout, dtype=np.float64)
                            rc = hip.cruse_conv_gather_bnstats(xd.data_ptr(), wd.data_ptr(), bd.data_ptr(), yd.data_ptr(), B, T, Cin, Fin, Cout, Fout, KT, S, pad, prec, sd.data_ptr(), 1, st())
                            torch.cuda.synchronize()
                            if rc: refused += 1
                            else:
                                n += 1
                                e1, e2 = rel(yd.cpu().numpy(), ytw), rel(sd.sum(0).cpu().numpy(), s_tw)
                                if not (e1 <= tol and e2 <= max(tol, 1e-5)): bad.append(("gather_bnstats", prec, KT, S, pad, Cin, Cout, Fin, T, e1, e2))
                        # bnbwd: y = conv (no bias), sums of g = y*[bn(bn_y)>0], g*xhat
                        ynb = np.zeros((B, T, Cout, Fout), np.float32)
                        R.call(ref, "cruse_conv_gather", x, w, None, ynb, B, T, Cin, Fin, Cout, Fout, KT, S, pad, wl, 0, 0, prec, 0, 0, None)
                        bn_y = rnd(B, T, Cout, Fout); mean, rstd = rnd(Cout) * 0.2, np.abs(rnd(Cout)) + 0.5; gamma, beta = rnd(Cout) * 0.3 + 1, rnd(Cout) * 0.2
                        s_tw = np.zeros(2 * Cout)
                        R.call(ref, "cruse_bn_act_bwd_reduce", ynb, bn_y, mean, rstd, gamma, beta, LL(B * T), Cout, Fout, 1, s_tw, 0, None)
                        xd, wd = D(x), D(w); yd = Z(B, T, Cout, Fout); sd = Z(NREP, 2 * Cout, dtype=np.float64)
                        keep = [D(bn_y), D(mean), D(rstd), D(gamma), D(beta)]
                        rc = hip.cruse_conv_gather_bnbwd(xd.data_ptr(), wd.data_ptr(), yd.data_ptr(), B, T, Cin, Fin, Cout, Fout, KT, S, pad, wl, 0, prec, *[k.data_ptr() for k in keep], 1, sd.data_ptr(), 1, 0, 0, st())
                        torch.cuda.synchronize()
                        if rc: refused += 1
                        else:
                            n += 1
                            e1, e2 = rel(yd.cpu().numpy(), ynb), rel(sd.sum(0).cpu().numpy(), s_tw)
                            # (a ReLU decision may flip where bn(bn_y) ~ 0 only through y's rounding: not here, bn_y is an input)
                            if not (e1 <= tol and e2 <= max(10 * tol, 1e-5)): bad.append(("gather_bnbwd", prec, KT, S, pad, wl, Cin, Cout, Fin, T, e1, e2))
        for KT, pad in ((1, 0), (2, 1)):
            for Cs, Cout in ((8, 8), (32, 16), (40, 24), (16, 1)):
                for Fg in (1, 3, 5, 8, 10, 40):
                    for T in (1, 4):
                        pass
                        B = 2
                        gq, w, bias = rnd(B, T, Cs, Fg), rnd(Cs, Cout, KT, 3) * 0.3, rnd(Cout)
                        ytw = np.zeros((B, T, Cout, 2 * Fg), np.float32)
                        R.call(ref, "cruse_conv_scatter2", gq, w, bias, ytw, B, T, Cs, Fg, Cout, 2 * Fg, KT, pad, 0, 0, prec, 0, 0, None)
                        s_tw = np.zeros(2 * Cout); R.call(ref, "cruse_bn_stats", ytw, LL(B * T), Cout, 2 * Fg, s_tw, 0, None)
                        gd, wd, bd = D(gq), D(w), D(bias); yd = Z(B, T, Cout, 2 * Fg); sd = Z(NREP, 2 * Cout, dtype=np.float64)
                        rc = hip.cruse_conv_scatter2_bnstats(gd.data_ptr(), wd.data_ptr(), bd.data_ptr(), yd.data_ptr(), B, T, Cs, Fg, Cout, 2 * Fg, KT, pad, prec, sd.data_ptr(), 1, st())
                        torch.cuda.synchronize()
                        if rc: refused += 1
                        else:
                            n += 1
                            e1, e2 = rel(yd.cpu().numpy(), ytw), rel(sd.sum(0).cpu().numpy(), s_tw)
                            if not (e1 <= tol and e2 <= max(tol, 1e-5)): bad.append(("scatter2_bnstats", prec, KT, pad, Cs, Cout, Fg, T, e1, e2))
                        ynb = np.zeros((B, T, Cout, 2 * Fg), np.float32)
                        R.call(ref, "cruse_conv_scatter2", gq, w, None, ynb, B, T, Cs, Fg, Cout, 2 * Fg, KT, pad, 0, 0, prec, 0, 0, None)
                        bn_y = rnd(B, T, Cout, 2 * Fg); mean, rstd = rnd(Cout) * 0.2, np.abs(rnd(Cout)) + 0.5; gamma, beta = rnd(Cout) * 0.3 + 1, rnd(Cout) * 0.2
                        s_tw = np.zeros(2 * Cout)
                        R.call(ref, "cruse_bn_act_bwd_reduce", ynb, bn_y, mean, rstd, gamma, beta, LL(B * T), Cout, 2 * Fg, 1, s_tw, 0, None)
                        yd = Z(B, T, Cout, 2 * Fg); sd = Z(NREP, 2 * Cout, dtype=np.float64)
                        keep = [D(bn_y), D(mean), D(rstd), D(gamma), D(beta)]
                        rc = hip.cruse_conv_scatter2_bnbwd(gd.data_ptr(), wd.data_ptr(), yd.data_ptr(), B, T, Cs, Fg, Cout, 2 * Fg, KT, pad, 0, prec, *[k.data_ptr() for k in keep], 1, sd.data_ptr(), 1, 0, 0, st())
                        torch.cuda.synchronize()
                        if rc: refused += 1
                        else:
                            n += 1
                            e1, e2 = rel(yd.cpu().numpy(), ynb), rel(sd.sum(0).cpu().numpy(), s_tw)
                            if not (e1 <= tol and e2 <= max(10 * tol, 1e-5)): bad.append(("scatter2_bnbwd", prec, KT, pad, Cs, Cout, Fg, T, e1, e2))
    assert not bad, bad[:10]
    assert n > 1500

def test_bf16_gemm_family_and_layout_kernels(hip):
    """cruse_gemm_bf16_nt (split-K, accumulate), _bf16x3_nt, cruse_gemm_f16_nt on ragged M / N against numpy products of the SAME 16-bit values;
    cruse_transpose_bf16, cruse_ktile_bf16 (hi + lo planes), cruse_cast_bf16_split bit for bit"""
    g = np.random.default_rng(6)
    rnd = lambda *s: np.ascontiguousarray(g.standard_normal(s).astype(np.float32))
    n = 0
    bad = []
    def bf16(a):      # RNE to bf16, as uint16
        u = np.ascontiguousarray(a, np.float32).view(np.uint32)
        r = (u + 0x7fff + ((u >> 16) & 1)) >> 16
        return r.astype(np.uint16)
    def f32_of(b): return (b.astype(np.uint32) << 16).view(np.float32)
    def chk(tag, got, want, tol):
        nonlocal n; n += 1
        got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
        e = np.linalg.norm(got - want) / max(np.linalg.norm(want), 1e-30)
        if not e <= tol or not np.isfinite(got).all(): bad.append((tag, float(e)))
    # row-major NT gemm, ragged M / N, K multiple of 64, splitk, accumulate
    for (M, N, K) in ((1, 1, 64), (3, 5, 64), (129, 127, 128), (200, 96, 192), (64, 1920, 640), (5, 640, 1920), (17, 33, 320)):
        A, Bm, bias = bf16(rnd(M, K)), bf16(rnd(N, K)), rnd(N)
        want = f32_of(A).astype(np.float64) @ f32_of(Bm).astype(np.float64).T
        for acc, sk in ((0, 1), (1, 1), (1, 3)):
            C0 = rnd(M, N)
            Cd = D(C0.copy()); Ad, Bd, bd = D(A), D(Bm), D(bias)
            rc = hip.cruse_gemm_bf16_nt(M, N, K, Ad.data_ptr(), K, 64, Bd.data_ptr(), K, 64, Cd.data_ptr(), N, bd.data_ptr() if sk == 1 else None, acc, sk, st())
            torch.cuda.synchronize()
            if rc: print("refused gemm_bf16_nt", M, N, K, acc, sk, hip.cruse_last_error()); continue
            chk(("gemm_bf16_nt", M, N, K, acc, sk), Cd.cpu().numpy(), want + (bias if sk == 1 else 0) + (C0 if acc else 0), 2e-5)
        # x3 form: A_lo / B_lo planes
        Af, Bf = rnd(M, K), rnd(N, K)
        Ah, Bh = bf16(Af), bf16(Bf); Al, Bl = bf16(Af - f32_of(Ah)), bf16(Bf - f32_of(Bh))
        want3 = (f32_of(Ah).astype(np.float64) @ (f32_of(Bh).astype(np.float64) + f32_of(Bl)).T) + f32_of(Al).astype(np.float64) @ f32_of(Bh).astype(np.float64).T
        Cd = Z(M, N)
        ks = [D(Ah), D(Al), D(Bh), D(Bl)]
        rc = hip.cruse_gemm_bf16x3_nt(M, N, K, ks[0].data_ptr(), ks[1].data_ptr(), K, 64, ks[2].data_ptr(), ks[3].data_ptr(), K, 64, Cd.data_ptr(), N, None, 0, st())
        torch.cuda.synchronize()
        if rc: print("refused x3", M, N, K, hip.cruse_last_error())
        else: chk(("gemm_bf16x3_nt", M, N, K), Cd.cpu().numpy(), want3, 2e-5)
        # f16 form
        A16, B16 = rnd(M, K).astype(np.float16), rnd(N, K).astype(np.float16)
        Cd = Z(M, N); a16, b16 = D(A16.view(np.uint16)), D(B16.view(np.uint16)); bd = D(bias)
        rc = hip.cruse_gemm_f16_nt(M, N, K, a16.data_ptr(), K, 64, b16.data_ptr(), K, 64, Cd.data_ptr(), N, bd.data_ptr(), st())
        torch.cuda.synchronize()
        if rc: print("refused f16", M, N, K, hip.cruse_last_error())
        else: chk(("gemm_f16_nt", M, N, K), Cd.cpu().numpy(), A16.astype(np.float64) @ B16.astype(np.float64).T + bias, 2e-5)
        # two f16 planes of B (cruse_ktile_f16_split makes them from f32 weights; here row-major planes built the same way)
        Bw = rnd(N, K) * 0.05
        Bhi = Bw.astype(np.float16); Blo = (Bw - Bhi.astype(np.float32)).astype(np.float16)
        Cd = Z(M, N); bh, bl = D(Bhi.view(np.uint16)), D(Blo.view(np.uint16))
        rc = hip.cruse_gemm_f16x2_nt(M, N, K, a16.data_ptr(), K, 64, bh.data_ptr(), bl.data_ptr(), K, 64, Cd.data_ptr(), N, bd.data_ptr(), st())
        torch.cuda.synchronize()
        if rc: bad.append(("refused f16x2", M, N, K))
        else:
            chk(("gemm_f16x2_nt", M, N, K), Cd.cpu().numpy(), A16.astype(np.float64) @ (Bhi.astype(np.float64) + Blo.astype(np.float64)).T + bias, 2e-5)
            # ... and the two planes carry the f32 weights to ~2^-20
            chk(("f16x2 vs f32 weights", M, N, K), Cd.cpu().numpy(), A16.astype(np.float64) @ Bw.astype(np.float64).T + bias, 3e-5)
    # layout kernels: transpose (K-tiled time-major), ktile, cast split
    for (rows, cols) in ((1, 32), (3, 64), (65, 160), (130, 640), (401, 96), (64, 1920)):
        x = rnd(rows, cols)
        ldT = (rows + 63) // 64 * 64
        for shift in sorted({0, rows, rows // 2 if rows % 2 == 0 else rows}):          # (shift_T: frames per clip -- rows are whole clips)
            yT = Z(ldT // 64, cols, 64, dtype=np.uint16); xd = D(x)
            rc = hip.cruse_transpose_bf16(xd.data_ptr(), rows, cols, cols, yT.data_ptr(), ldT, shift, st())
            torch.cuda.synchronize()
            if rc: print("refused transpose", rows, cols, shift, hip.cruse_last_error()); continue
            want = np.zeros((ldT, cols), np.float32)
            if shift: 
                want[1:rows] = x[:rows - 1]; want[0:rows:shift] = 0
            else: want[:rows] = x
            wantT = f32_of(bf16(want)).reshape(ldT // 64, 64, cols).transpose(0, 2, 1)
            chk(("transpose_bf16", rows, cols, shift), f32_of(yT.cpu().numpy()), wantT, 0.0)
        kp = (cols + 63) // 64 * 64
        y, ylo = Z(kp // 64, rows, 64, dtype=np.uint16), Z(kp // 64, rows, 64, dtype=np.uint16); xd = D(x)
        rc = hip.cruse_ktile_bf16(xd.data_ptr(), rows, cols, cols, y.data_ptr(), ylo.data_ptr(), st())
        torch.cuda.synchronize()
        y16, y16lo = Z(kp // 64, rows, 64, dtype=np.uint16), Z(kp // 64, rows, 64, dtype=np.uint16)
        rc2 = hip.cruse_ktile_f16_split(xd.data_ptr(), rows, cols, cols, y16.data_ptr(), y16lo.data_ptr(), st())
        torch.cuda.synchronize()
        if rc2: bad.append(("refused ktile_f16_split", rows, cols))
        else:
            xp16 = np.zeros((rows, kp), np.float32); xp16[:, :cols] = x
            h16 = xp16.astype(np.float16); l16 = (xp16 - h16.astype(np.float32)).astype(np.float16)
            chk(("ktile_f16 hi", rows, cols), y16.cpu().numpy().view(np.float16).astype(np.float32), h16.astype(np.float32).reshape(rows, kp // 64, 64).transpose(1, 0, 2), 0.0)
            chk(("ktile_f16 lo", rows, cols), y16lo.cpu().numpy().view(np.float16).astype(np.float32), l16.astype(np.float32).reshape(rows, kp // 64, 64).transpose(1, 0, 2), 0.0)
        if rc: print("refused ktile", rows, cols, hip.cruse_last_error())
        else:
            xp = np.zeros((rows, kp), np.float32); xp[:, :cols] = x
            hi = bf16(xp); lo = bf16(xp - f32_of(hi))
            chk(("ktile hi", rows, cols), f32_of(y.cpu().numpy()), f32_of(hi).reshape(rows, kp // 64, 64).transpose(1, 0, 2), 0.0)
            chk(("ktile lo", rows, cols), f32_of(ylo.cpu().numpy()), f32_of(lo).reshape(rows, kp // 64, 64).transpose(1, 0, 2), 0.0)
        if (rows * cols) % 4 == 0:
            yh, yl = Z(rows, cols, dtype=np.uint16), Z(rows, cols, dtype=np.uint16)
            rc = hip.cruse_cast_bf16_split(xd.data_ptr(), yh.data_ptr(), yl.data_ptr(), rows * cols, st())
            torch.cuda.synchronize()
            if rc: print("refused cast split", rows, cols, hip.cruse_last_error())
            else:
                hi = bf16(x); chk(("cast hi", rows, cols), f32_of(yh.cpu().numpy()), f32_of(hi), 0.0); chk(("cast lo", rows, cols), f32_of(yl.cpu().numpy()), f32_of(bf16(x - f32_of(hi))), 0.0)
    assert not bad, bad[:10]
    assert n >= 60

def test_general_nchw_convolutions_over_small_and_odd_shapes():
    """cruse_conv2d_nchw / _ex / _wgrad_ex through cruse_amd.nn_generic's raw wrappers against torch in float64: pointwise, depthwise (dilated, causal
    pad), grouped and full 3x3 / 2x3 kernels, channel counts on and off the MFMA tiles, planes from 1x1 to 161x9, f32 and f16 storage"""
    from cruse_amd import nn_generic as G
    g = np.random.default_rng(7)
    rnd = lambda *s: np.ascontiguousarray(g.standard_normal(s).astype(np.float32))

    def rel(a, b):
        a, b = a.double().cpu().flatten(), b.double().cpu().flatten()
        return float((a - b).norm() / b.norm().clamp_min(1e-30))
    bad, n = [], 0
    for dt, tol in ((np.float32, 2e-5), (np.float16, 3e-3)):
        for (Cin, Cout, KH, KW, groups, dil) in ((3, 5, 1, 1, 1, (1, 1)), (24, 24, 1, 1, 1, (1, 1)), (8, 40, 1, 1, 1, (1, 1)), (64, 64, 1, 1, 1, (1, 1)),
                                                 (24, 24, 3, 3, 24, (1, 2)), (8, 8, 3, 3, 8, (1, 8)), (5, 5, 3, 3, 5, (1, 1)), (4, 6, 3, 3, 1, (1, 1)), (6, 4, 2, 3, 2, (1, 1))):
            for (B, H, W) in ((1, 1, 1), (1, 3, 5), (2, 7, 33), (1, 161, 9), (3, 5, 130)):
                pt, pl = (KH - 1) * dil[0] // 2, (KW - 1) * dil[1]
                x = rnd(B, Cin, H, W).astype(dt); w = rnd(Cout, Cin // groups, KH, KW) * 0.3; bias = rnd(Cout)
                xt = torch.from_numpy(x.astype(np.float32))
                xp = F.pad(xt, (pl, 0, pt, (KH - 1) * dil[0] - pt))
                want = F.conv2d(xp.double(), torch.from_numpy(w).double(), torch.from_numpy(bias).double(), dilation=dil, groups=groups)
                Ho, Wo = want.shape[2], want.shape[3]
                if Ho != H or Wo != W: continue
                xd, wd, bd = D(x), D(w), D(bias)
                out = D(np.zeros((B, Cout, H, W), dt))
                try:
                    y = G._conv_raw(xd, wd, bd, (H, W), KH, KW, (1, 1), dil, pt, pl, groups, 1, False, Cout, out=out)
                    torch.cuda.synchronize()
                except Exception as ex:
                    bad.append(("EXC fwd", dt.__name__, Cin, Cout, KH, KW, groups, dil, B, H, W, repr(ex)[:100])); continue
                n += 1
                e = rel(y, want)
                if not e <= tol: bad.append(("fwd", dt.__name__, Cin, Cout, KH, KW, groups, dil, B, H, W, e))
                # with batch sums, and with a residual (cruse_conv2d_nchw_ex; the two are not offered together)
                res = rnd(B, Cout, H, W).astype(dt); rd = D(res)
                sums = D(np.zeros((G.BN_STAT_REPLICAS, 2 * Cout), np.float64)); out2 = D(np.zeros((B, Cout, H, W), dt)); out3 = D(np.zeros((B, Cout, H, W), dt))
                try:
                    y2 = G._conv_raw(xd, wd, bd, (H, W), KH, KW, (1, 1), dil, pt, pl, groups, 1, False, Cout, out=out2, bn_sums=sums)
                    y3 = G._conv_raw(xd, wd, bd, (H, W), KH, KW, (1, 1), dil, pt, pl, groups, 1, False, Cout, out=out3, residual=rd)
                    torch.cuda.synchronize()
                    n += 1
                    yv = y2.double().cpu()
                    s_want = torch.cat([yv.sum((0, 2, 3)), (yv * yv).sum((0, 2, 3))])
                    e, e2, e3 = rel(y2, want), rel(sums.sum(0), s_want), rel(y3, want + torch.from_numpy(res.astype(np.float32)).double())
                    if not (e <= tol and e2 <= 1e-4 and e3 <= tol * 2): bad.append(("ex", dt.__name__, Cin, Cout, KH, KW, groups, dil, B, H, W, e, e2, e3))
                except Exception as ex:
                    bad.append(("EXC ex", dt.__name__, Cin, Cout, KH, KW, groups, dil, B, H, W, repr(ex)[:100]))
                # data gradient (transposed form) and weight gradient
                dy = rnd(B, Cout, H, W).astype(dt); dyd = D(dy)
                xt2 = xt.double().requires_grad_(True); wt2 = torch.from_numpy(w).double().requires_grad_(True)
                yy = F.conv2d(F.pad(xt2, (pl, 0, pt, (KH - 1) * dil[0] - pt)), wt2, None, dilation=dil, groups=groups)
                dxw, dww = torch.autograd.grad(yy, (xt2, wt2), torch.from_numpy(dy.astype(np.float32)).double())
                outdx = D(np.zeros((B, Cin, H, W), dt))
                try:
                    dx = G._conv_raw(dyd, wd, None, (H, W), KH, KW, (1, 1), dil, pt, pl, groups, 1, True, Cin, out=outdx)
                    dw = D(np.zeros(w.shape, np.float32)); db = D(np.zeros(Cout, np.float32))
                    G._wgrad_raw(dyd, xd, dw, KH, KW, (1, 1), dil, pt, pl, groups, 1, db=db)
                    torch.cuda.synchronize()
                except Exception as ex:
                    bad.append(("EXC bwd", dt.__name__, Cin, Cout, KH, KW, groups, dil, B, H, W, repr(ex)[:100])); continue
                n += 1
                e, e2 = rel(dx, dxw), rel(dw, dww)
                e3 = rel(db, torch.from_numpy(dy.astype(np.float32)).double().sum((0, 2, 3)))
                if not (e <= tol * 2 and e2 <= max(tol * 2, 2e-4) and e3 <= max(tol, 1e-4)): bad.append(("bwd", dt.__name__, Cin, Cout, KH, KW, groups, dil, B, H, W, e, e2, e3))
    assert not bad, bad[:10]
    assert n >= 250
