"""Profiling aid: every convolution launch of the bench training step (B = 64 x 401 frames, 640 floats per frame at every
level), timed alone, with the HBM rate against the algorithmic bytes (input + output, + output again when accumulating)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cruse_amd import ops


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    dev = "cuda"
    B, T = 64, 401
    ch = (1, 8, 16, 32, 64); Fk = [160, 80, 40, 20, 10]
    torch.manual_seed(0)
    fprec = os.environ.get("FPREC", "bf16")       # the engine's mode: forward convs split (x3), backward-data plain bf16
    mb = B * T * 640 * 4 / 1e6

    def rep(name, us, nbuf):
        print(f"{name:44s} {us:7.1f} us   {nbuf * mb / us:5.2f} TB/s")
    for k in range(1, 5):
        x = torch.randn(B, T, ch[k - 1], Fk[k - 1], device=dev)
        w = torch.randn(ch[k], ch[k - 1], 2, 3, device=dev) * 0.1
        b = torch.zeros(ch[k], device=dev)
        with ops.ARENA.step(x.device):
            rep(f"enc{k} fwd conv(2,3)/s2 {ch[k-1]}->{ch[k]} +stats", timeit(lambda: ops.conv_gather_bnstats(x, w, b, B, T, ch[k - 1], Fk[k - 1], ch[k], Fk[k], KT=2, S=2, pad=1, prec=fprec)), 2 if k > 1 else 1.25)
        rep(f"enc{k} fwd conv(2,3)/s2 {ch[k-1]}->{ch[k]}", timeit(lambda: ops.conv_gather(x, w, b, B, T, ch[k - 1], Fk[k - 1], ch[k], Fk[k], KT=2, S=2, pad=1, prec=fprec)), 2 if k > 1 else 1.25)
        dy = torch.randn(B, T, ch[k], Fk[k], device=dev)
        if k > 1:
            de = torch.zeros(B, T, ch[k - 1], Fk[k - 1], device=dev)
            rep(f"enc{k} bwd-data scatter2 KT=2 accum {ch[k]}->{ch[k-1]}", timeit(lambda: ops.conv_scatter2(dy, w, None, B, T, ch[k], Fk[k], ch[k - 1], KT=2, pad=1, out=de, accum=True, prec="bf16")), 3)
        dw = torch.zeros_like(w)
        rep(f"enc{k} wgrad", timeit(lambda: ops.conv_wgrad(dy, x, dw, B, T, ch[k], Fk[k], ch[k - 1], Fk[k - 1], KT=2, S=2, pad=1, prec="bf16")), 2 if k > 1 else 1.25)
        e = torch.randn(B, T, ch[k], Fk[k], device=dev)
        ws = torch.randn(ch[k], ch[k], 1, 3, device=dev) * 0.1
        rep(f"skip{k} fwd conv(1,3) {ch[k]}->{ch[k]}", timeit(lambda: ops.conv_gather(e, ws, None, B, T, ch[k], Fk[k], ch[k], Fk[k], KT=1, S=1, pad=1, prec=fprec)), 2)
        o = torch.empty_like(e)
        rep(f"skip{k} bwd-data (w_layout 1)", timeit(lambda: ops.conv_gather(dy, ws, None, B, T, ch[k], Fk[k], ch[k], Fk[k], KT=1, S=1, pad=1, w_layout=1, out=o, prec="bf16")), 2)
        dws = torch.zeros_like(ws)
        rep(f"skip{k} wgrad", timeit(lambda: ops.conv_wgrad(dy, e, dws, B, T, ch[k], Fk[k], ch[k], Fk[k], KT=1, S=1, pad=1, prec="bf16")), 2)
        wt = torch.randn(ch[k], ch[k - 1], 1, 3, device=dev) * 0.1
        bt = torch.zeros(ch[k - 1], device=dev)
        u = torch.randn(B, T, ch[k], Fk[k], device=dev)
        with ops.ARENA.step(x.device):
            rep(f"dec{k} fwd convT(1,3)/s2 {ch[k]}->{ch[k-1]} (+stats)", timeit(lambda: ops.conv_scatter2_bnstats(u, wt, bt, B, T, ch[k], Fk[k], ch[k - 1], KT=1, pad=0, prec=fprec)), 2 if k > 1 else 1.25)
        dv = torch.randn(B, T, ch[k - 1], Fk[k - 1], device=dev)
        rep(f"dec{k} bwd-data gather S=2 {ch[k-1]}->{ch[k]}", timeit(lambda: ops.conv_gather(dv, wt, None, B, T, ch[k - 1], Fk[k - 1], ch[k], Fk[k], KT=1, S=2, pad=0, prec="bf16")), 2 if k > 1 else 1.25)
        dwt = torch.zeros_like(wt)
        rep(f"dec{k} wgrad", timeit(lambda: ops.conv_wgrad(u, dv, dwt, B, T, ch[k], Fk[k], ch[k - 1], Fk[k - 1], KT=1, S=2, pad=0, prec="bf16")), 2 if k > 1 else 1.25)


if __name__ == "__main__":
    main()
