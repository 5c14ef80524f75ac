import sys, torch
sys.path.insert(0, '.')
from cruse_amd import ops
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
B, T, H = 64, 401, 640
rows = B * T
torch.manual_seed(0)
for (M, N, K, name) in [(rows, 3 * H, H, "gi"), (rows, H, 3 * H, "dX"), (2500, 200, 192, "ragged"), (4096 + 37, 640, 640, "ragged M")]:
    A = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    Bm = torch.randn(N, K, device="cuda").to(torch.bfloat16)
    bias = torch.randn(N, device="cuda")
    ref = (A.double() @ Bm.double().t() + bias.double())
    for big in (0, 1):
        with ops.options(gb_bm256=big):
            C = torch.zeros(M, N, device="cuda")
            ops.gemm_bf16_nt(M, N, K, A, 0, K, Bm, 0, K, C, 0, N, bias=bias)
            err = float((C.double() - ref).norm() / ref.norm())
            C2 = torch.ones(M, N, device="cuda")
            ops.gemm_bf16_nt(M, N, K, A, 0, K, Bm, 0, K, C2, 0, N, accumulate=True)
            err2 = float((C2.double() - (ref - bias.double() + 1)).norm() / ref.norm())
            us = timeit(lambda: ops.gemm_bf16_nt(M, N, K, A, 0, K, Bm, 0, K, C, 0, N, bias=bias))
        print(f"{name} M={M} N={N} K={K} bm256={big}: {us:.1f} us {2 * M * N * K / us / 1e6:.0f} TFLOP/s  err {err:.2e} accum err {err2:.2e}")
