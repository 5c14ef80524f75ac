"""Profiling aid: the TN weight-gradient GEMM (cruse_gemm_bf16_tn) against the NT form on time-major copies, alone, on one GRU
layer's three products at the bench shape (K = 64 x 401 frames)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cruse_amd import ops  # noqa: E402
from tools.gemm_probe import timeit  # noqa: E402


def main():
    dev = "cuda"
    B, T, H = 64, 401, 640
    rows = B * T
    torch.manual_seed(0)
    dg4 = torch.randn(rows, 4 * H, device=dev).to(torch.bfloat16)
    x_bf = torch.randn(rows, H, device=dev).to(torch.bfloat16)
    h = torch.randn(rows, H, device=dev)
    C = torch.zeros(3 * H, H, device=dev)
    for sk in (0, 4, 6, 8, 16):
        for (M, a_off, Bop, shift, name) in ((3 * H, 0, x_bf, 0, "dW_ih  bf16 B"), (2 * H, 0, h, T, "dW_hh rz f32 B"),
                                             (H, 3 * H, h, T, "dW_hh n  f32 B"), (3 * H, 0, h, 0, "dW_ih  f32 B")):
            us = timeit(lambda: ops.gemm_bf16_tn(M, H, rows, dg4, a_off, 4 * H, Bop, 0, H, C, 0, H, b_shift_T=shift, splitk=sk))
            print(f"TN {name} M={M} splitk={sk}: {us:8.1f} us  {2.0 * M * H * rows / us / 1e6:7.1f} TF/s")
    gT = ops.transpose_bf16(dg4.float(), rows, 4 * H)
    xT = ops.transpose_bf16(x_bf.float(), rows, H)
    ldT = xT.shape[0] * 64
    for sk in (6, -8):
        us = timeit(lambda: ops.gemm_bf16_nt(3 * H, H, ldT, gT, 0, 64, xT, 0, 64, C, 0, H, accumulate=True, splitk=sk,
                                             a_kstride=4 * H * 64, b_kstride=H * 64))
        print(f"NT dW_ih on time-major copies splitk={sk}: {us:8.1f} us  {2.0 * 3 * H * H * ldT / us / 1e6:7.1f} TF/s")


if __name__ == "__main__":
    main()
