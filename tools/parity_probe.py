"""Forward / gradient parity of a precision mode against the CPU oracle at arbitrary (B, T, g, init).

    python tools/parity_probe.py [--prec bf16] [--cases "8x401:1:closed,8x401:4:random"] [--grads]

Prints one line per case: enhanced-spectrum and mask rel-L2 vs the oracle (and, with --grads, the worst
per-tensor gradient rel-L2).  Used to establish the numbers the T = 401 parity tests gate on."""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def build_pair(grp, init, prec, seed=7):
    from cruse_amd.model import cruse_net as M
    from oracle import cruse_oracle as O
    if init == "closed":
        o = O.unet_2(rnn_groups=grp)
        O.closed_form_init(o)
    else:
        torch.manual_seed(seed)
        o = O.unet_2(rnn_groups=grp)            # torch default init (Kaiming-uniform convs, U(+-1/sqrt(H)) GRU)
    m = M.unet_2(rnn_groups=grp, precision=prec)
    m.load_state_dict(o.state_dict(), strict=True)
    o.train(); m.train()
    return o, m.cuda()


def rel(a, b):
    a = a.detach().double().cpu().flatten(); b = b.detach().double().cpu().flatten()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def oracle_ggru_stages(o, e4):
    """h1 / l1 / h2 in the product's layouts (cat over groups), following oracle GGRU.forward."""
    g = o.gru
    x = e4.transpose(1, 2).contiguous()
    x = x.view(x.size(0), x.size(1), -1)
    xs = torch.chunk(x, g.groups, dim=-1)
    h1s = [g.gru_list1[i](xs[i])[0] for i in range(g.groups)]
    l1 = g.ln1(torch.flatten(torch.stack(h1s, dim=-1), start_dim=-2, end_dim=-1))
    ls = torch.chunk(l1, g.groups, dim=-1)
    h2 = torch.cat([g.gru_list2[i](ls[i])[0] for i in range(g.groups)], dim=-1)
    return dict(x=x, h1=torch.cat(h1s, dim=-1), l1=l1, h2=h2, out=g.ln2(h2))


def run_stages(B, T, grp, init, prec, rec_prec):
    from cruse_amd import ops
    from cruse_amd.engine import TrainEngine
    from cruse_amd.model.cruse_net import unet2_forward
    from oracle import cruse_oracle as O
    L = (T - 1) * 160
    o, m = build_pair(grp, init, prec)
    noisy, clean = O.synth_pair(B, L, seed=11)
    with torch.no_grad():
        feats = O.pre_stft(noisy, 320, 160, 320, f_net=160)
        mask_o, im = o(feats["mag_net"], return_intermediates=True)
        st = oracle_ggru_stages(o, im["e4"])
    eng = TrainEngine(m, use_graph=False)
    _, _, mag = ops.stft(noisy.cuda(), 320, 160, mag_bins=160, mag_eps=1e-8)
    if rec_prec:
        real = ops.gru_seq_fwd

        def patched(gi, w_hh, b_hh, B_, T_, G_, Hg_, prec_, **kw):
            return real(gi, w_hh, b_hh, B_, T_, G_, Hg_, rec_prec, **kw)
        ops.gru_seq_fwd = patched
    mask, ctx = unet2_forward(mag.view(B, 1, T, 160), eng.flat.P, eng.Bf, m.ch, m.rnn_groups, prec, training=True,
                              save=True, update_running=False)
    if rec_prec:
        ops.gru_seq_fwd = real
    g = ctx["gctx"]
    e4 = ctx["es"][4].view(B, T, -1)
    out = [f"e4 {rel(e4, st['x']):.2e}", f"h1 {rel(g['h1'], st['h1']):.2e}", f"l1 {rel(g['l1'], st['l1']):.2e}",
           f"h2 {rel(g['h2'], st['h2']):.2e}", f"mask {rel(mask, mask_o):.2e}"]
    sd = lambda t: float(t.std(dim=-1).mean())
    out.append(f"| per-frame std: h1 {sd(st['h1']):.3g} h2 {sd(st['h2']):.3g}  |h1| {float(st['h1'].abs().mean()):.3g}")
    print(f"[stages {prec} rec={rec_prec or prec} GI_X3={os.environ.get('CRUSE_GI_X3','3')} B={B} T={T} g={grp} {init}] " + "  ".join(out), flush=True)


def run_case(B, T, grp, init, prec, grads):
    from cruse_amd import ops
    from cruse_amd.engine import TrainEngine
    from oracle import cruse_oracle as O
    L = (T - 1) * 160
    o, m = build_pair(grp, init, prec)
    noisy, clean = O.synth_pair(B, L, seed=11)
    t0 = time.perf_counter()
    if grads:
        loss_o, aux = O.train_step_loss(o, noisy, clean)
        loss_o.backward()
    else:
        with torch.no_grad():
            loss_o, aux = O.train_step_loss(o, noisy, clean)
    t_or = time.perf_counter() - t0
    eng = TrainEngine(m, use_graph=False)
    nre, nim, mag = ops.stft(noisy.cuda(), 320, 160, mag_bins=160, mag_eps=1e-8)
    if grads:
        ls = eng._fwd_bwd(noisy.cuda(), clean.cuda())
        mask = eng._last_mask
    else:
        from cruse_amd.model.cruse_net import unet2_forward
        mask, _ = unet2_forward(mag.view(B, 1, T, 160), eng.flat.P, eng.Bf, m.ch, m.rnn_groups, prec, training=True,
                                save=False, update_running=False)
    er, ei = ops.mask_apply(mask.contiguous().view(B * T, 160), nre, nim, B * T, 160, 161)
    est = torch.stack([er.view(B, T, 161), ei.view(B, T, 161)], dim=-1)
    e_est, e_mask = rel(est, aux["est"]), rel(mask.view(B, 1, T, 160), aux["mask"])
    line = f"[parity {prec} B={B} T={T} g={grp} {init:6s}] est {e_est:.3e} mask {e_mask:.3e} (oracle {t_or:.1f}s)"
    if grads:
        lv = eng.loss_value(ls)
        worst, wname = 0.0, ""
        allg, allo = [], []
        for n, p in o.named_parameters():
            dead = n.endswith(".bias") and n.startswith("conv") and n != "conv1_t.bias"   # bias before BN: d/db == 0
            if n in eng.flat.G and p.grad is not None and not dead:
                r = rel(eng.flat.G[n], p.grad)
                allg.append(eng.flat.G[n].detach().double().cpu().flatten()); allo.append(p.grad.double().flatten())
                if r > worst:
                    worst, wname = r, n
        tot = float((torch.cat(allg) - torch.cat(allo)).norm() / torch.cat(allo).norm())
        line += f" loss {lv:.6f}/{float(loss_o):.6f} grad rel-L2: all {tot:.3e} worst {worst:.3e} ({wname})"
    line += f" gru_status {ops.gru_status()}"
    print(line, flush=True)
    return e_est


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--prec", default="bf16")
    ap.add_argument("--cases", default="2x21:1:closed,8x101:1:closed,8x401:1:closed,8x401:1:random,8x401:4:closed,8x401:4:random")
    ap.add_argument("--grads", action="store_true")
    ap.add_argument("--stages", action="store_true", help="per-stage errors through the GGRU")
    ap.add_argument("--rec-prec", default="", help="(--stages) run the forward recurrences in this precision")
    a = ap.parse_args()
    torch.set_num_threads(min(os.cpu_count() or 1, 16))
    for c in a.cases.split(","):
        bt, g, init = c.split(":")
        B, T = map(int, bt.split("x"))
        if a.stages:
            run_stages(B, T, int(g), init, a.prec, a.rec_prec)
        else:
            run_case(B, T, int(g), init, a.prec, a.grads)


if __name__ == "__main__":
    main()
