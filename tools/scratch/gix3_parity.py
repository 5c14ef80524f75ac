"""Forward parity of the bf16 mode (enhanced spectrum / mask vs the CPU oracle) for the gi_x3 settings; T = 401 and fixture-sized cases."""
import sys, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tools')
from cruse_amd import ops
from cruse_amd.config import EngineConfig
from cruse_amd.engine import TrainEngine
from cruse_amd.model.cruse_net import unet2_forward
from oracle import cruse_oracle as O
import parity_probe as P
for (B, T, init) in ((2, 21, "closed"), (8, 401, "closed"), (8, 401, "random")):
    L = (T - 1) * 160
    noisy, clean = O.synth_pair(B, L, seed=11)
    for x3 in (3, 2, 1, 0):
        o, m = P.build_pair(1, init, "bf16")
        with torch.no_grad():
            feats = O.pre_stft(noisy, 320, 160, 320, f_net=160)
            mask_o = o(feats["mag_net"])
        eng = TrainEngine(m, use_graph=False, config=EngineConfig(gi_x3=x3))
        re_, im_, mag = ops.stft(noisy.cuda(), 320, 160, mag_bins=160, mag_eps=1e-8)
        from cruse_amd import config
        from cruse_amd.model import cruse_net as M
        with config.use(eng.cfg), M.use_scheduler(eng.side):
            mask, ctx = unet2_forward(mag.view(B, 1, T, 160), eng.flat.P, eng.Bf, m.ch, m.rnn_groups, "bf16", training=True, save=True, update_running=False)
        torch.cuda.synchronize()
        mo = mask_o.reshape(B, T, 160) if mask_o.dim() == 4 else mask_o
        mk = mask.reshape(B, T, 160)
        # enhanced spectrum = mask * noisy spectrum on bins 0..159
        spec = torch.stack([re_[..., :160], im_[..., :160]], -1).cpu()
        est = mk.cpu().unsqueeze(-1) * spec
        est_o = mo.reshape(B, T, 160).unsqueeze(-1) * spec
        print(f"B={B} T={T} {init} gi_x3={x3}: mask rel-L2 {P.rel(mk, mo.reshape(B, T, 160)):.3e}  enhanced spectrum rel-L2 {P.rel(est, est_o):.3e}", flush=True)
