"""Profiling aid: what the vendor GEMM (hipBLASLt behind torch.mm) reaches on the step's plain bf16 GEMM shapes --
the yardstick for gemm_bf16_nt (tools/gemm_probe.py)."""
import sys, os, torch


def timeit(fn, n=20):
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
    rows, H = 64 * 401, 640
    torch.manual_seed(0)
    x = torch.randn(rows, H, device=dev).bfloat16()
    w = torch.randn(3 * H, H, device=dev).bfloat16()
    dg = torch.randn(rows, 3 * H, device=dev).bfloat16()
    for name, fn, flops in [
        ("gi  = x W^T      [25664x640]x[640x1920]", lambda: torch.mm(x, w.t()), 2.0 * rows * H * 3 * H),
        ("dX  = dg W       [25664x1920]x[1920x640]", lambda: torch.mm(dg, w), 2.0 * rows * H * 3 * H),
        ("dW  = dg^T x     [1920x25664]x[25664x640]", lambda: torch.mm(dg.t(), x), 2.0 * rows * H * 3 * H),
    ]:
        us = timeit(fn)
        print(f"{name:48s} bf16 out: {us:7.1f} us  {flops / us / 1e6:7.1f} TF/s")
    try:
        us = timeit(lambda: torch.mm(dg.t(), x, out_dtype=torch.float32))
        print(f"dW with f32 output (out_dtype): {us:7.1f} us  {2.0 * rows * H * 3 * H / us / 1e6:7.1f} TF/s")
        us = timeit(lambda: torch.mm(x, w.t(), out_dtype=torch.float32))
        print(f"gi with f32 output (out_dtype): {us:7.1f} us")
    except Exception as e:
        print("out_dtype not supported:", str(e)[:200])


if __name__ == "__main__":
    main()
