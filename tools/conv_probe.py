"""Timing probe for the conv / wgrad kernels (profiling aid)."""
import sys, torch
sys.path.insert(0, '.')
from cruse_amd import ops
T = 401
def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
LEVELS = [(8, 80, 16, 40), (16, 40, 32, 20), (32, 20, 64, 10)]
for prec in ("bf16x3", "bf16"):
    for (Cin, Fin, Cout, Fout) in LEVELS:
        for B in (8, 64):
            x = torch.randn(B, T, Cin, Fin).cuda(); w = torch.randn(Cout, Cin, 2, 3).cuda(); b = torch.randn(Cout).cuda()
            y = torch.empty(B, T, Cout, Fout).cuda()
            dy = torch.randn(B, T, Cout, Fout).cuda(); dw = torch.zeros_like(w)
            t_f = timeit(lambda: ops.conv_gather(x, w, b, B, T, Cin, Fin, Cout, Fout, KT=2, S=2, pad=1, out=y, prec=prec))
            t_w = timeit(lambda: ops.conv_wgrad(dy, x, dw, B, T, Cout, Fout, Cin, Fin, KT=2, S=2, pad=1, prec=prec))
            mb = B * T * 640 * 4 * 2 / 1e6
            print(f"{prec:7s} enc {Cin:2d}->{Cout:2d} B={B:2d}: fwd {t_f:7.1f} us ({mb / t_f * 1e6 / 1e6:6.0f} GB/s)  wgrad {t_w:7.1f} us")
