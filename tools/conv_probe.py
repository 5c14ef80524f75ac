"""Where the time of an MFMA convolution goes (library option cm_dbg = 1: s_memtime stamps of workgroup 0 / wave 0) for the
U-Net's layers at the bench shape, forward (split-bf16 x3 + BatchNorm sums) and data gradient (plain bf16)."""
import ctypes, sys, torch
sys.path.insert(0, '.')
from cruse_amd import ops
from cruse_amd._lib import lib
B, T = 64, 401
ch, Fk = (1, 8, 16, 32, 64), (161, 80, 40, 20, 10)


def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def stamps():
    out = (ctypes.c_ulonglong * 8)()
    lib.cruse_conv_mfma_stamps(ctypes.cast(out, ctypes.c_void_p))
    return list(out)


def report(name, fn, mbytes):
    ops.set_option("cm_dbg", 0)
    us = timeit(fn)
    ops.set_option("cm_dbg", 1)
    fn(); torch.cuda.synchronize()
    st = stamps()
    ops.set_option("cm_dbg", 0)
    tiles, nts, tot = max(st[4], 1), max(st[5], 1), max(st[6], 1)
    print(f"{name:34s} {us:6.1f} us ({mbytes / us:4.2f} TB/s) | workgroup 0: {tot} cycles = prologue {st[0]} + {tiles} tiles x "
          f"(staging {st[1] / tiles:.0f} + {nts / tiles:.1f} N-tiles x (k-loop {st[2] / nts:.0f} + epilogue {st[3] / nts:.0f}))", flush=True)


torch.manual_seed(0)
for k in (2, 3, 4):                                    # encoder forward conv k: ch[k-1] -> ch[k], (2,3) kernel, stride (1,2)
    x = torch.randn(B, T, ch[k - 1], Fk[k - 1]).cuda(); w = (0.1 * torch.randn(ch[k], ch[k - 1], 2, 3)).cuda()
    report(f"fwd conv{k} {ch[k-1]}->{ch[k]} (x3 + BN sums)", lambda: ops.conv_gather_bnstats(x, w, None, B, T, ch[k - 1], Fk[k - 1], ch[k], Fk[k], KT=2, S=2, pad=1, prec="bf16x3"),
           (x.numel() + B * T * ch[k] * Fk[k]) * 4 / 1e6)
for k in (4, 3, 2):                                    # its data gradient: conv_scatter2 of dy [ch[k], Fk[k]] -> [ch[k-1], Fk[k-1]]
    dy = torch.randn(B, T, ch[k], Fk[k]).cuda(); w = (0.1 * torch.randn(ch[k], ch[k - 1], 2, 3)).cuda()
    out = torch.zeros(B, T, ch[k - 1], Fk[k - 1]).cuda()
    report(f"dgrad conv{k} {ch[k]}->{ch[k-1]} (bf16, accum)", lambda: ops.conv_scatter2(dy, w, None, B, T, ch[k], Fk[k], ch[k - 1], KT=2, pad=1, out=out, accum=True, prec="bf16"),
           (dy.numel() + 2 * out.numel()) * 4 / 1e6)
for k in (4, 3):                                       # decoder forward convT k: ch[k] -> ch[k-1] (1,3), stride (1,2): scatter form
    u = torch.randn(B, T, ch[k], Fk[k]).cuda(); w = (0.1 * torch.randn(ch[k], ch[k - 1], 1, 3)).cuda()
    report(f"fwd convT{k} {ch[k]}->{ch[k-1]} (x3 + BN sums)", lambda: ops.conv_scatter2_bnstats(u, w, None, B, T, ch[k], Fk[k], ch[k - 1], KT=1, pad=0, prec="bf16x3"),
           (u.numel() + B * T * ch[k - 1] * Fk[k - 1]) * 4 / 1e6)
