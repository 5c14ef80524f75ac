"""Profiling aid: the BatchNorm / LayerNorm streaming kernels of the training step, timed alone at the four U-Net level
shapes (rows = 64 x 401 frames, C x F = 640 columns), with the HBM rate each one reaches."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cruse_amd import ops


def timeit(fn, n=30):
    for _ in range(5):
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
    rows = B * T
    torch.manual_seed(0)
    for C, F in [(8, 80), (16, 40), (32, 20), (64, 10)]:
        y = torch.randn(rows, C, F, device=dev)
        dout = torch.randn(rows, C, F, device=dev)
        skip = torch.randn(rows, C, F, device=dev)
        gamma = torch.rand(C, device=dev) + 0.5; beta = torch.randn(C, device=dev) * 0.1
        dg = torch.zeros(C, device=dev); db = torch.zeros(C, device=dev); dbias = torch.zeros(C, device=dev)
        mb = y.numel() * 4 / 1e6
        sums = ops.bn_stats(y, rows, C, F)
        out, mean, rstd = ops.bn_finalize_act_fwd(y, sums, rows * F, 1e-5, 0.1, gamma, beta, skip, rows, C, F)
        t_stats = timeit(lambda: ops.bn_stats(y, rows, C, F))
        t_fwd = timeit(lambda: ops.bn_finalize_act_fwd(y, sums, rows * F, 1e-5, 0.1, gamma, beta, skip, rows, C, F))
        t_bwd = timeit(lambda: ops.bn_act_bwd(dout, y, mean, rstd, gamma, beta, rows, C, F, True, True, dg, db, dbias))
        from cruse_amd.ops import lib, _p, _stream, check
        sm = torch.zeros(2 * C, device=dev, dtype=torch.float64); dyo = torch.empty_like(y)
        t_red = timeit(lambda: check(lib.cruse_bn_act_bwd_reduce(_p(dout), _p(y), _p(mean), _p(rstd), _p(gamma), _p(beta), rows, C, F,
                                                                 1, _p(sm), 1, _stream())))
        t_app = timeit(lambda: check(lib.cruse_bn_act_bwd_apply(_p(dout), _p(y), _p(mean), _p(rstd), _p(gamma), _p(beta), _p(sm), 1, rows,
                                                                C, F, 1, 1, 0, _p(dyo), 0, _p(dg), _p(db), _p(dbias), _stream())))
        print(f"            reduce alone {t_red:6.1f} us ({2 * mb / t_red:5.2f} TB/s)   apply alone {t_app:6.1f} us ({3 * mb / t_app:5.2f} TB/s)")
        print(f"C={C:3d} F={F:3d}: bn_stats {t_stats:6.1f} us ({mb / t_stats:5.2f} TB/s)  fin_act_fwd {t_fwd:6.1f} us "
              f"({3 * mb / t_fwd:5.2f} TB/s)  bn_act_bwd reduce+apply {t_bwd:6.1f} us ({5 * mb / t_bwd:5.2f} TB/s)")
    H = 640
    x = torch.randn(rows, H, device=dev); dy = torch.randn(rows, H, device=dev)
    g = torch.rand(H, device=dev) + 0.5; b = torch.zeros(H, device=dev)
    yy, mean, rstd = ops.ln_fwd(x, g, b, None, rows, H)
    dg = torch.zeros(H, device=dev); db = torch.zeros(H, device=dev)
    mb = x.numel() * 4 / 1e6
    t = timeit(lambda: ops.ln_fwd(x, g, b, None, rows, H))
    print(f"ln_fwd {t:6.1f} us ({2 * mb / t:5.2f} TB/s)")
    t = timeit(lambda: ops.ln_bwd(dy, x, mean, rstd, g, rows, H, 1, dg, db))
    print(f"ln_bwd {t:6.1f} us ({3 * mb / t:5.2f} TB/s)")
    from cruse_amd.ops import lib, _p, _stream, check
    dx = torch.empty_like(x)
    t = timeit(lambda: check(lib.cruse_ln_bwd(_p(dy), _p(x), _p(mean), _p(rstd), _p(g), rows, H, 1, _p(dx), None, None, _stream())))
    print(f"ln_bwd without dgamma/dbeta {t:6.1f} us")


if __name__ == "__main__":
    main()
