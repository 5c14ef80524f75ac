"""Profiling aid: the conv / convT / data-gradient launches of one training step, timed in isolation."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cruse_amd import ops
B, T = 64, 401
ch, F = [1, 8, 16, 32, 64], [160, 80, 40, 20, 10]
PREC = os.environ.get("PREC", "bf16")
PREC = int(PREC) if PREC.isdigit() else PREC


def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    tot = 0.0
    R = lambda *s: torch.randn(*s, device="cuda")
    for k in range(1, 5):
        cases = {}
        x0, e, u = R(B, T, ch[k - 1], F[k - 1]), R(B, T, ch[k], F[k]), R(B, T, ch[k], F[k])
        w, b = R(ch[k], ch[k - 1], 2, 3), R(ch[k])
        ws = R(ch[k], ch[k], 1, 3)
        wt, bt = R(ch[k], ch[k - 1], 1, 3), R(ch[k - 1])
        y, s_, v, de = torch.empty_like(e), torch.empty_like(e), torch.empty_like(x0), torch.zeros_like(e)
        cases["enc fwd"] = lambda: ops.conv_gather(x0, w, b, B, T, ch[k - 1], F[k - 1], ch[k], F[k], KT=2, S=2, pad=1, out=y, prec=PREC)
        cases["skip fwd"] = lambda: ops.conv_gather(e, ws, None, B, T, ch[k], F[k], ch[k], F[k], KT=1, S=1, pad=1, out=s_, prec=PREC)
        cases["dec fwd"] = lambda: ops.conv_scatter2(u, wt, bt, B, T, ch[k], F[k], ch[k - 1], KT=1, pad=0, prec=PREC)
        cases["dec dgrad"] = lambda: ops.conv_gather(x0, wt, None, B, T, ch[k - 1], F[k - 1], ch[k], F[k], KT=1, S=2, pad=0, prec=PREC)
        cases["skip dgrad"] = lambda: ops.conv_gather(e, ws, None, B, T, ch[k], F[k], ch[k], F[k], KT=1, S=1, pad=1, w_layout=1, out=de, accum=True, prec=PREC)
        if k > 1:
            cases["enc dgrad"] = lambda: ops.conv_scatter2(e, w, None, B, T, ch[k], F[k], ch[k - 1], KT=2, pad=1, prec=PREC)
        for name, fn in cases.items():
            us = timeit(fn)
            tot += us
            print(f"L{k} {name:10s} {ch[k-1]:2d}x{F[k-1]:3d} <-> {ch[k]:2d}x{F[k]:3d}: {us:7.1f} us  {2 * B * T * 640 * 4 / 1e6 / us:5.2f} TB/s")
    print(f"total {tot:.0f} us")


if __name__ == "__main__":
    main()
