"""phase ablation of wgrad_mfma (library option wg_dbg: 1 skip patch build, 2 skip MFMA loop, 4 skip loads, 8 skip A / raw image)"""
import sys, torch
sys.path.insert(0, '.')
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


for k, name in ((2, "enc"), (3, "enc"), (4, "enc"), (2, "skip"), (4, "skip"), (3, "dec")):
    Ca, Fa, Cb, Fb, KT, S, pad = {"enc": (ch[k], F[k], ch[k - 1], F[k - 1], 2, 2, 1), "skip": (ch[k], F[k], ch[k], F[k], 1, 1, 1),
                                  "dec": (ch[k], F[k], ch[k - 1], F[k - 1], 1, 2, 0)}[name]
    a = torch.randn(B, T, Ca, Fa, device="cuda").to(torch.bfloat16); bt = torch.randn(B, T, Cb, Fb, device="cuda")
    if name == "enc":
        bt = bt.to(torch.bfloat16)
    dw = torch.zeros(Ca, Cb, KT, 3, device="cuda")
    res = []
    for dbg in (0, 1, 2, 3, 8, 11, 15):
        with ops.options(wg_dbg=dbg):
            res.append((dbg, timeit(lambda: ops.conv_wgrad(a, bt, dw, B, T, Ca, Fa, Cb, Fb, KT=KT, S=S, pad=pad, prec="bf16"))))
    print(f"L{k} {name:4s} Ca={Ca} Fa={Fa} Cb={Cb} Fb={Fb} KT={KT}: " + "  ".join(f"dbg{d}={u:.0f}" for d, u in res))
