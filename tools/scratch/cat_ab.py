import sys, torch
sys.path.insert(0, '.')
from cruse_amd import ops
torch.manual_seed(0)
Hg, K = 640, 25664
ldT = (K + 63) // 64 * 64
dgT = (torch.randn(ldT // 64, 1, 4, Hg, 64, device="cuda") * 0.1).to(torch.bfloat16)
xT = torch.randn(ldT // 64, Hg, 64, device="cuda").to(torch.bfloat16)
hT = torch.randn(ldT // 64, Hg, 64, device="cuda").to(torch.bfloat16)
ka, kb = 4 * Hg * 64, Hg * 64
base = torch.randn(2, 3 * Hg, Hg, device="cuda")
def three(C):
    ops.gemm_bf16_nt(3 * Hg, Hg, ldT, dgT, 0, 64, xT, 0, 64, C[0], 0, Hg, accumulate=True, splitk=-8, slabs=True, a_kstride=ka, b_kstride=kb)
    ops.gemm_bf16_nt(2 * Hg, Hg, ldT, dgT, 0, 64, hT, 0, 64, C[1], 0, Hg, accumulate=True, splitk=-8, slabs=True, a_kstride=ka, b_kstride=kb)
    ops.gemm_bf16_nt(Hg, Hg, ldT, dgT, 3 * Hg * 64, 64, hT, 0, 64, C[1], 2 * Hg * Hg, Hg, accumulate=True, splitk=-8, slabs=True, a_kstride=ka, b_kstride=kb)
def cat(C):
    ops.gemm_bf16_nt_cat([3 * Hg, 2 * Hg, Hg], Hg, ldT, dgT, [0, 0, 3 * Hg], 64, [xT, hT, hT], 0, 64, C, 0, Hg, -8, a_kstride=ka, b_kstride=kb)
c3 = base.clone(); three(c3)
cc = base.clone(); cat(cc)
torch.cuda.synchronize()
print("equal", torch.equal(c3, cc), float((c3 - cc).abs().max()))
A = dgT.float().permute(0, 4, 1, 2, 3).reshape(ldT, 4 * Hg).double()      # [k][slab*Hg+u]
X = xT.float().permute(0, 2, 1).reshape(ldT, Hg).double(); Hm = hT.float().permute(0, 2, 1).reshape(ldT, Hg).double()
ref = base.double().clone()
ref[0] += A[:, :3 * Hg].t() @ X
ref[1][:2 * Hg] += A[:, :2 * Hg].t() @ Hm
ref[1][2 * Hg:] += A[:, 3 * Hg:].t() @ Hm
print("vs f64", float((cc.double() - ref).norm() / ref.norm()))
for name, fn in (("three", three), ("cat", cat)):
    C = base.clone()
    for _ in range(3): fn(C)
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fn(C)
    e1.record(); torch.cuda.synchronize()
    print(name, f"{e0.elapsed_time(e1) / 20 * 1e3:.1f} us")
