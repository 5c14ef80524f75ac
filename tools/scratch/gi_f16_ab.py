"""gi_f16 (single-pass f16 forward gate projection): kernel check vs f64, forward parity vs the CPU oracle next to the gi_x3 = 3 default."""
import sys, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tools')
from cruse_amd import ops
from cruse_amd.config import EngineConfig
from cruse_amd.engine import TrainEngine
from cruse_amd.model.cruse_net import unet2_forward
from oracle import cruse_oracle as O
import parity_probe as P

torch.manual_seed(0)
for (M, N, K) in ((25664, 1920, 640), (2500, 200, 192), (4133, 640, 640)):
    A = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda") * 0.04; b = torch.randn(N, device="cuda")
    Ah = A.half()
    Wt = ops.ktile_f16(W, N, K)
    kp = (K + 63) // 64 * 64
    if kp != K:
        Ap = torch.zeros(M, kp, device="cuda", dtype=torch.float16); Ap[:, :K] = Ah; Ah = Ap
    C = torch.empty(M, N, device="cuda")
    ops.gemm_f16_nt(M, N, kp, Ah, 0, kp, Wt, 0, 64, C, 0, N, bias=b, b_kstride=N * 64)
    ref = Ah[:, :K].double() @ W.half().double().t() + b.double()
    ref32 = A.double() @ W.double().t() + b.double()
    print(f"gemm_f16_nt M={M} N={N} K={K}: vs f64 on the f16 operands {float((C.double() - ref).norm() / ref.norm()):.2e}, vs exact f32 operands {float((C.double() - ref32).norm() / ref32.norm()):.2e}", flush=True)
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    for _ in range(3): ops.gemm_f16_nt(M, N, kp, Ah, 0, kp, Wt, 0, 64, C, 0, N, bias=b, b_kstride=N * 64)
    e0.record()
    for _ in range(20): ops.gemm_f16_nt(M, N, kp, Ah, 0, kp, Wt, 0, 64, C, 0, N, bias=b, b_kstride=N * 64)
    e1.record(); torch.cuda.synchronize()
    print(f"   {e0.elapsed_time(e1) / 20 * 1e3:.1f} us", flush=True)

for (B, T, init, G) in ((2, 21, "closed", 1), (8, 401, "closed", 1), (8, 401, "random", 1), (8, 401, "closed", 4), (8, 401, "random", 4), (8, 401, "closed", 2)):
    L = (T - 1) * 160
    noisy, clean = O.synth_pair(B, L, seed=11)
    for f16 in (False, True):
        o, m = P.build_pair(G, init, "bf16")
        with torch.no_grad():
            feats = O.pre_stft(noisy, 320, 160, 320, f_net=160)
            mask_o = o(feats["mag_net"])
        eng = TrainEngine(m, use_graph=False, config=EngineConfig(gi_f16=f16))
        re_, im_, mag = ops.stft(noisy.cuda(), 320, 160, mag_bins=160, mag_eps=1e-8)
        from cruse_amd import config
        from cruse_amd.model import cruse_net as Mn
        for training in (True, False):
            with config.use(eng.cfg), Mn.use_scheduler(eng.side):
                mask, ctx = unet2_forward(mag.view(B, 1, T, 160), eng.flat.P, eng.Bf, m.ch, m.rnn_groups, "bf16", training=training, save=training, update_running=False)
            torch.cuda.synchronize()
            if not training:
                continue
            mo = mask_o.reshape(B, T, 160)
            mk = mask.reshape(B, T, 160)
            spec = torch.stack([re_[..., :160], im_[..., :160]], -1).cpu()
            est = mk.cpu().unsqueeze(-1) * spec
            est_o = mo.unsqueeze(-1) * spec
            print(f"g={G} B={B} T={T} {init} gi_f16={f16}: mask rel-L2 {P.rel(mk, mo):.3e}  enhanced spectrum rel-L2 {P.rel(est, est_o):.3e}", flush=True)
