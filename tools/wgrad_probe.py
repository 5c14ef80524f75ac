"""Profiling aid: the 12 conv weight-gradient launches of one training step, timed in isolation."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cruse_amd import ops
B, T = 64, 401
ch, F = [1, 8, 16, 32, 64], [160, 80, 40, 20, 10]


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
    for k in range(1, 5):
        for name, (Ca, Fa, Cb, Fb, KT, S, pad) in {
            "enc": (ch[k], F[k], ch[k - 1], F[k - 1], 2, 2, 1), "skip": (ch[k], F[k], ch[k], F[k], 1, 1, 1),
            "dec": (ch[k], F[k], ch[k - 1], F[k - 1], 1, 2, 0)}.items():
            a = torch.randn(B, T, Ca, Fa, device="cuda"); bt = torch.randn(B, T, Cb, Fb, device="cuda")
            if name == "dec":
                dw = torch.zeros(Ca, Cb, 1, 3, device="cuda")
            else:
                dw = torch.zeros(Ca, Cb, KT, 3, device="cuda")
            us = timeit(lambda: ops.conv_wgrad(a, bt, dw, B, T, Ca, Fa, Cb, Fb, KT=KT, S=S, pad=pad, prec="bf16"))
            tot += us
            mb = (a.numel() + bt.numel()) * 4 / 1e6
            print(f"L{k} {name:4s} Ca={Ca:2d} Fa={Fa:3d} Cb={Cb:2d} Fb={Fb:3d} KT={KT}: {us:7.1f} us  {mb / us:5.2f} TB/s")
    print(f"total {tot:.0f} us")


if __name__ == "__main__":
    main()
