"""Profiling aid: isolated timings of the row-wise kernels at the bench shape (B=64, T=401, H=640)."""
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
    B, T, H = 64, 401, 640
    rows = B * T
    x = torch.randn(rows, H, device=dev); res = torch.randn(rows, H, device=dev)
    gm = torch.randn(H, device=dev); bt = torch.randn(H, device=dev)
    y, m, s = ops.ln_fwd(x, gm, bt, res, rows, H, 1)
    mb = rows * H * 4 / 1e6
    us = timeit(lambda: ops.ln_fwd(x, gm, bt, res, rows, H, 1)); print(f"ln_fwd+res   {us:7.1f} us  {3 * mb / us:6.2f} TB/s(alg)")
    us = timeit(lambda: ops.ln_fwd(x, gm, bt, None, rows, H, 1)); print(f"ln_fwd       {us:7.1f} us  {2 * mb / us:6.2f} TB/s(alg)")
    dg = torch.zeros(H, device=dev); db = torch.zeros(H, device=dev)
    us = timeit(lambda: ops.ln_bwd(res, x, m, s, gm, rows, H, 1, dg, db)); print(f"ln_bwd       {us:7.1f} us  {3 * mb / us:6.2f} TB/s(alg)")
    coef = torch.randn(rows, 1, 3, H, device=dev).to(torch.bfloat16)
    an = torch.rand(rows, H, device=dev)
    dbi = [torch.zeros(3 * H, device=dev)]; dbh = [torch.zeros(3 * H, device=dev)]
    us = timeit(lambda: ops.gru_gate_grads_bf16(x, coef, an, rows, 1, H, dbi, dbh))
    print(f"gate_grads_bf16 {us:7.1f} us  {(2 * mb + 2 * 1.5 * mb + 2 * mb) / us:6.2f} TB/s(alg)")
    us = timeit(lambda: ops.transpose_bf16(x, rows, H)); print(f"transpose_bf16 {us:7.1f} us  {1.5 * mb / us:6.2f} TB/s(alg)")
    us = timeit(lambda: ops.cast_bf16(x)); print(f"cast_bf16    {us:7.1f} us  {1.5 * mb / us:6.2f} TB/s(alg)")
    for C, F in [(16, 40), (32, 20), (64, 10)]:
        yy = x.view(rows, C, F)
        sums = ops.bn_stats(yy, rows, C, F)
        mean, rstd = ops.bn_finalize(sums, rows * F, C, 1e-5, 0.1, None, None)
        g2 = torch.randn(C, device=dev); b2 = torch.randn(C, device=dev)
        d1 = torch.zeros(C, device=dev); d2 = torch.zeros(C, device=dev); d3 = torch.zeros(C, device=dev)
        us0 = timeit(lambda: ops.bn_stats(yy, rows, C, F))
        us1 = timeit(lambda: ops.bn_act_fwd(yy, mean, rstd, g2, b2, None, rows, C, F))
        us2 = timeit(lambda: ops.bn_act_bwd(res, yy, mean, rstd, g2, b2, rows, C, F, True, True, d1, d2, dbias=d3))
        print(f"C={C:3d} F={F:3d}: bn_stats {us0:6.1f} us ({mb / us0:5.2f} TB/s)  bn_act_fwd {us1:6.1f} us ({2 * mb / us1:5.2f})  "
              f"bn_act_bwd {us2:6.1f} us ({5 * mb / us2:5.2f})")


if __name__ == "__main__":
    main()
