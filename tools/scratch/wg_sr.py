"""Shifted-row operand of the MFMA weight gradient (library option wg_sr) against the patch-matrix form: results and time, the step's 12 shapes,
f32 and bf16 operand tensors."""
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
def rel(a, b): return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))
ch, F = [1, 8, 16, 32, 64], [160, 80, 40, 20, 10]
tot = {0: 0.0, 1: 0.0}
for (B, T) in ((3, 21), (64, 401)):
    for k in range(1, 5):
        for name, (Ca, Fa, Cb, Fb, KT, S, pad) in {"enc": (ch[k], F[k], ch[k - 1], F[k - 1], 2, 2, 1), "skip": (ch[k], F[k], ch[k], F[k], 1, 1, 1),
                                                   "dec": (ch[k], F[k], ch[k - 1], F[k - 1], 1, 2, 0)}.items():
            torch.manual_seed(k)
            a = torch.randn(B, T, Ca, Fa, device="cuda"); bt = torch.randn(B, T, Cb, Fb, device="cuda")
            for bf in (False, True):
                aa = a.bfloat16() if bf else a; bb = bt.bfloat16() if bf else bt
                out = {}
                for sr in (0, 1):
                    with ops.options(wg_sr=sr):
                        dw = torch.zeros(Ca, Cb, KT, 3, device="cuda")
                        ops.conv_wgrad(aa, bb, dw, B, T, Ca, Fa, Cb, Fb, KT=KT, S=S, pad=pad, prec="bf16")
                        torch.cuda.synchronize()
                        out[sr] = dw
                        if B == 64 and bf:
                            us = timeit(lambda: ops.conv_wgrad(aa, bb, dw, B, T, Ca, Fa, Cb, Fb, KT=KT, S=S, pad=pad, prec="bf16")); tot[sr] += us
                            out[(sr, 't')] = us
                line = f"B={B} L{k} {name:4s} Ca={Ca:2d} Fa={Fa:3d} Cb={Cb:2d} Fb={Fb:3d} KT={KT} bf16-tensors={bf}: sr vs patch {rel(out[1], out[0]):.2e} equal {bool(torch.equal(out[1], out[0]))}"
                if B == 64 and bf: line += f" | {out[(0, 't')]:.1f} -> {out[(1, 't')]:.1f} us"
                print(line, flush=True)
print("total", tot)
