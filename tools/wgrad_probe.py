"""Profiling aid: the 12 conv weight-gradient launches of one training step, timed in isolation at the bench shape -- the register-direct
stream (wgrad_rd.hip, option wg_rd = 1) against the LDS-staged kernel (wgrad_mfma.hip, wg_rd = 0), for bf16 / f32 operand tensors, with the
relative difference of the two results.  usage: python tools/wgrad_probe.py [grid ...]"""
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
    grids = [int(v) for v in sys.argv[1:]] or [0]
    tot = {}
    for k in range(1, 5):
        for name, (Ca, Fa, Cb, Fb, KT, S, pad) in {
            "enc": (ch[k], F[k], ch[k - 1], F[k - 1], 2, 2, 1), "skip": (ch[k], F[k], ch[k], F[k], 1, 1, 1),
            "dec": (ch[k], F[k], ch[k - 1], F[k - 1], 1, 2, 0)}.items():
            a32 = torch.randn(B, T, Ca, Fa, device="cuda").to(torch.bfloat16).float(); b32 = torch.randn(B, T, Cb, Fb, device="cuda").to(torch.bfloat16).float()
            line = f"L{k} {name:4s} Ca={Ca:2d} Fa={Fa:3d} Cb={Cb:2d} Fb={Fb:3d} KT={KT}:"
            for dt in ("bf16", "f32"):
                a, bt = (a32.to(torch.bfloat16), b32.to(torch.bfloat16)) if dt == "bf16" else (a32, b32)
                mb = (a.numel() * a.element_size() + bt.numel() * bt.element_size()) / 1e6
                res = {}
                for rd in (0, 1):
                    for gr in (grids if rd else [0]):
                        ops.set_option("wg_rd", rd); ops.set_option("wg_grid", gr or None)
                        dw = torch.zeros(Ca, Cb, KT, 3, device="cuda")
                        ops.conv_wgrad(a, bt, dw, B, T, Ca, Fa, Cb, Fb, KT=KT, S=S, pad=pad, prec="bf16")
                        us = timeit(lambda: ops.conv_wgrad(a, bt, dw, B, T, Ca, Fa, Cb, Fb, KT=KT, S=S, pad=pad, prec="bf16"))
                        dw.zero_(); ops.conv_wgrad(a, bt, dw, B, T, Ca, Fa, Cb, Fb, KT=KT, S=S, pad=pad, prec="bf16"); torch.cuda.synchronize()
                        res[(rd, gr)] = (us, dw.clone())
                        tot[(dt, rd, gr)] = tot.get((dt, rd, gr), 0.0) + us
                ref = res[(0, 0)][1]
                line += f"  {dt}: staged {res[(0, 0)][0]:6.1f} us"
                for gr in grids:
                    us, dw = res[(1, gr)]
                    line += f" | rd{'' if not gr else gr} {us:6.1f} us {mb / us:5.2f} TB/s diff {float((dw - ref).norm() / ref.norm()):.1e}"
            print(line, flush=True)
    ops.set_option("wg_rd", None); ops.set_option("wg_grid", None)
    print("totals (us):", {f"{dt} {'rd' if rd else 'staged'}{gr or ''}": round(v) for (dt, rd, gr), v in tot.items()})


if __name__ == "__main__":
    main()
