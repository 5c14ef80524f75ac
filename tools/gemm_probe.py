"""Profiling aid: time + check the bf16 NT GEMM and its layout kernels on the step's gate-projection shapes."""
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
    torch.manual_seed(0)
    for (M, N, K, sk, name) in [(rows, 3 * H, H, 1, "gi (NT)"), (rows, H, 3 * H, 1, "dX"), (3 * H, H, rows, 7, "dW split7"),
                                (3 * H, H, rows, 4, "dW split4"), (3 * H, H, rows, 14, "dW split14"), (200, 96, 128, 1, "ragged")]:
        A = torch.randn(M, K, device=dev).to(torch.bfloat16)
        Bm = torch.randn(N, K, device=dev).to(torch.bfloat16)
        bias = torch.randn(N, device=dev)
        C = torch.zeros(M, N, device=dev)
        ops.gemm_bf16_nt(M, N, K, A, 0, K, Bm, 0, K, C, 0, N, bias=bias if sk == 1 else None, accumulate=sk > 1, splitk=sk)
        ref = A.float() @ Bm.float().t() + (bias if sk == 1 else 0)
        err = ((C - ref).norm() / ref.norm()).item()
        us = timeit(lambda: ops.gemm_bf16_nt(M, N, K, A, 0, K, Bm, 0, K, C, 0, N, bias=None, accumulate=sk > 1, splitk=sk))
        print(f"{name:12s} M={M} N={N} K={K} sk={sk}: rel_err {err:.2e}  {us:8.1f} us  {2.0*M*N*K/us/1e6:7.1f} TF/s")
        if sk == 1:
            # old f32-input kernel on the same product
            Af, Bf = A.float(), Bm.float()
            us0 = timeit(lambda: ops.gemm(False, True, M, N, K, Af, 0, K, Bf, 0, K, C, 0, N, prec="bf16"))
            print(f"{'':12s} f32-input kernel: {us0:8.1f} us")
    # weight-gradient shape on K-tiled time-major operands
    x = torch.randn(rows, H, device=dev)
    g3 = torch.randn(rows, 4 * H, device=dev)
    xT = ops.transpose_bf16(x, rows, H); gT = ops.transpose_bf16(g3, rows, 4 * H)
    ldT = xT.shape[0] * 64
    C = torch.zeros(3 * H, H, device=dev)
    for sk in (3, 6, -8, -16, -24):
        us = timeit(lambda: ops.gemm_bf16_nt(3 * H, H, ldT, gT, 0, 64, xT, 0, 64, C, 0, H, accumulate=True, splitk=sk,
                                             a_kstride=4 * H * 64, b_kstride=H * 64))
        print(f"dW K-tiled sk={sk}: {us:8.1f} us  {2.0*3*H*H*ldT/us/1e6:7.1f} TF/s")
    ref = g3[:, :3 * H].to(torch.bfloat16).float().t() @ x.to(torch.bfloat16).float()
    for sk in (6, -8, -16):
        C.zero_()
        ops.gemm_bf16_nt(3 * H, H, ldT, gT, 0, 64, xT, 0, 64, C, 0, H, accumulate=True, splitk=sk, a_kstride=4 * H * 64, b_kstride=H * 64)
        print(f"dW K-tiled sk={sk} rel_err", ((C - ref).norm() / ref.norm()).item())
    # the three weight-gradient products of one GRU layer, as the step issues them
    hT = ops.transpose_bf16(x, rows, H, shift_T=T)
    G2 = torch.zeros(3 * H, H, device=dev)
    ka, kb = 4 * H * 64, H * 64
    for sk in (0, -8, -16):
        def layer(sk=sk):
            s1 = sk or 3; s2 = sk or 5; s3 = sk or 10
            ops.gemm_bf16_nt(3 * H, H, ldT, gT, 0, 64, xT, 0, 64, C, 0, H, accumulate=True, splitk=s1, a_kstride=ka, b_kstride=kb)
            ops.gemm_bf16_nt(2 * H, H, ldT, gT, 0, 64, hT, 0, 64, G2, 0, H, accumulate=True, splitk=s2, a_kstride=ka, b_kstride=kb)
            ops.gemm_bf16_nt(H, H, ldT, gT, 3 * H * 64, 64, hT, 0, 64, G2, 2 * H * H, H, accumulate=True, splitk=s3, a_kstride=ka, b_kstride=kb)
        print(f"one layer's dW_ih + dW_hh (3 launches), splitk {sk or 'r01 (3,5,10)'}: {timeit(layer):8.1f} us")
    print(f"cast_bf16 [{rows},{H}]: {timeit(lambda: ops.cast_bf16(x)):.1f} us; transpose_bf16: {timeit(lambda: ops.transpose_bf16(x, rows, H)):.1f} us; "
          f"shifted {timeit(lambda: ops.transpose_bf16(x, rows, H, shift_T=T)):.1f} us")


if __name__ == "__main__":
    main()
