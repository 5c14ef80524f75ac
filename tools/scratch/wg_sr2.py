import sys, torch
sys.path.insert(0, '.')
from cruse_amd import ops
def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
ch, F = [1, 8, 16, 32, 64], [160, 80, 40, 20, 10]
B, T = 64, 401
for k in range(1, 5):
    for name, (Ca, Fa, Cb, Fb, KT, S, pad) in {"enc": (ch[k], F[k], ch[k - 1], F[k - 1], 2, 2, 1), "skip": (ch[k], F[k], ch[k], F[k], 1, 1, 1),
                                               "dec": (ch[k], F[k], ch[k - 1], F[k - 1], 1, 2, 0)}.items():
        a = torch.randn(B, T, Ca, Fa, device="cuda").bfloat16(); bt = torch.randn(B, T, Cb, Fb, device="cuda").bfloat16()
        dw = torch.zeros(Ca, Cb, KT, 3, device="cuda")
        row = []
        for sr in (0, 1):
            for tfw in (4, 8):
                for grid in (256, 512):
                    try:
                        with ops.options(wg_sr=sr, wg_tfw=tfw, wg_grid=grid):
                            us = timeit(lambda: ops.conv_wgrad(a, bt, dw, B, T, Ca, Fa, Cb, Fb, KT=KT, S=S, pad=pad, prec="bf16"))
                        row.append(f"sr{sr} tfw{tfw} g{grid}: {us:.1f}")
                    except Exception as e:
                        row.append(f"sr{sr} tfw{tfw} g{grid}: n/a")
        print(f"L{k} {name:4s}: " + " | ".join(row), flush=True)
