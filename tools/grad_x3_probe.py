import sys, torch
sys.path.insert(0, '.')
from cruse_amd.config import EngineConfig
from cruse_amd.engine import TrainEngine
from cruse_amd.model import cruse_net as M
from oracle import cruse_oracle as O
from tests.util import rel_l2
B, T = 8, 401
for init in ("random", "closed"):
    if init == "closed":
        o = O.unet_2(rnn_groups=1); O.closed_form_init(o)
    else:
        torch.manual_seed(7); o = O.unet_2(rnn_groups=1)
    o.train()
    noisy, clean = O.synth_pair(B, (T - 1) * 160, seed=11)
    o.zero_grad()
    loss_o, _ = O.train_step_loss(o, noisy, clean); loss_o.backward()
    for x3 in (False, True):
        m = M.unet_2(rnn_groups=1, precision="bf16"); m.load_state_dict(o.state_dict()); m = m.cuda().train()
        eng = TrainEngine(m, use_graph=False, config=EngineConfig(conv_bwd_x3=x3))
        eng._fwd_bwd(noisy.cuda(), clean.cuda()); torch.cuda.synchronize()
        errs = {n: rel_l2(eng.flat.G[n], p.grad) for n, p in o.named_parameters() if n in eng.flat.G and p.grad is not None and not (n.endswith(".bias") and n.startswith("conv") and n != "conv1_t.bias")}
        convs = {k: v for k, v in errs.items() if k.startswith("conv") and k.endswith("weight")}
        print(f"[{init} conv_bwd_x3={x3}] conv weight-gradient rel-L2:", {k: f"{v:.1e}" for k, v in sorted(convs.items())})
