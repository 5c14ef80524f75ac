import sys, numpy as np, torch
sys.path.insert(0, '.')
from cruse_amd.engine import TrainEngine
from cruse_amd.config import EngineConfig
from cruse_amd.model import cruse_net as M
from oracle import cruse_oracle as O
def rel_l2(a, b): return float((a.double().cpu() - b.double().cpu()).norm() / b.double().norm())
g = np.load("tests/golden/g16_df_step_g1.npz")
for f16 in (0, 1, 2, 3):
    o = O.unet_2(rnn_groups=1); O.closed_form_init(o)
    m = M.unet_2(rnn_groups=1, precision="bf16")
    m.load_state_dict(o.state_dict(), strict=True)
    eng = TrainEngine(m.cuda(), use_graph=False, loss="wo_male_df", config=EngineConfig(gi_f16=f16))
    ls = eng._fwd_bwd(torch.from_numpy(g["noisy"]).cuda(), torch.from_numpy(g["clean"]).cuda())
    est = eng._last_est.permute(1, 0, 2, 3)
    print("f16", f16, "loss rel", abs(eng.loss_value(ls) - float(g["loss"])) / abs(float(g["loss"])), "est", rel_l2(est, torch.from_numpy(g["est"])))
    for name in eng.flat.names:
        if "gn/" + name not in g.files: continue
        gn, got = float(g["gn/" + name]), float(eng.flat.G[name].norm())
        r = abs(got - gn) / (gn + 1e-30)
        if r > 0.05 and gn > 1e-6: print("   ", name, got, gn, f"{r:.3f}")

# gradients vs ORACLE autograd at T = 401 (random and closed-form init), per setting
sys.path.insert(0, 'tests')
B, T = 8, 401
for init in ("random", "closed"):
    for f16 in (0, 3):
        if init == "closed":
            o = O.unet_2(rnn_groups=1); O.closed_form_init(o)
        else:
            torch.manual_seed(7); o = O.unet_2(rnn_groups=1)
        m = M.unet_2(rnn_groups=1, precision="bf16"); m.load_state_dict(o.state_dict(), strict=True); o.train(); m.train()
        noisy, clean = O.synth_pair(B, (T - 1) * 160, seed=11)
        loss_o, _ = O.train_step_loss(o, noisy, clean); loss_o.backward()
        eng = TrainEngine(m.cuda(), use_graph=False, config=EngineConfig(gi_f16=f16))
        ls = eng._fwd_bwd(noisy.cuda(), clean.cuda())
        allg, allo, worst = [], [], {}
        for n, p in o.named_parameters():
            if n not in eng.flat.G or p.grad is None or (n.endswith(".bias") and n.startswith("conv") and n != "conv1_t.bias"): continue
            worst[n] = rel_l2(eng.flat.G[n], p.grad)
            allg.append(eng.flat.G[n].detach().double().cpu().flatten()); allo.append(p.grad.double().flatten())
        tot = float((torch.cat(allg) - torch.cat(allo)).norm() / torch.cat(allo).norm())
        top = sorted(worst.items(), key=lambda kv: -kv[1])[:4]
        print(f"T=401 {init} gi_f16={f16}: loss rel {abs(eng.loss_value(ls) - float(loss_o)) / abs(float(loss_o)):.2e} all-grad rel-L2 {tot:.3e} worst {[(k, round(v, 4)) for k, v in top]}", flush=True)
