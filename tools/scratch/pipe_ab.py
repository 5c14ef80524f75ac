import sys, torch
sys.path.insert(0, '.')
from cruse_amd import ops
def timeit(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
rows, H = 25664, 640
torch.manual_seed(0)
res = {}
for rep in range(3):
    for (M, N, K, name) in [(rows, 3 * H, H, "gi"), (rows, H, 3 * H, "dX")]:
        A = torch.randn(M, K, device="cuda").to(torch.bfloat16); Bm = torch.randn(N, K, device="cuda").to(torch.bfloat16)
        C = torch.zeros(M, N, device="cuda")
        for pipe in (0, 1):
            with ops.options(gb_pipe=pipe):
                us = timeit(lambda: ops.gemm_bf16_nt(M, N, K, A, 0, K, Bm, 0, K, C, 0, N))
            res.setdefault((name, pipe), []).append(round(us, 1))
    # split-K weight gradient (deep variant)
    Hg, K = 640, 25664
    ldT = (K + 63) // 64 * 64
    dgT = (torch.randn(ldT // 64, 1, 4, Hg, 64, device="cuda") * 0.1).to(torch.bfloat16)
    xT = torch.randn(ldT // 64, Hg, 64, device="cuda").to(torch.bfloat16); hT = torch.randn(ldT // 64, Hg, 64, device="cuda").to(torch.bfloat16)
    Cc = torch.zeros(2, 3 * Hg, Hg, device="cuda")
    for pipe in (0, 1):
        with ops.options(gb_pipe=pipe):
            us = timeit(lambda: ops.gemm_bf16_nt_cat([3 * Hg, 2 * Hg, Hg], Hg, ldT, dgT, [0, 0, 3 * Hg], 64, [xT, hT, hT], 0, 64, Cc, 0, Hg, -8, a_kstride=4 * Hg * 64, b_kstride=Hg * 64))
        res.setdefault(("dW cat", pipe), []).append(round(us, 1))
for k, v in res.items(): print(k, v)
