"""GPU: parity of the BENCH precision mode (bf16 MFMA operands) at the bench's sequence length.

VERDICT r1 "weak" 1-2: the 3.3 M frames/s mode was parity-gated at T = 21 only, and its gradients were never
compared with the oracle at model level.  Here: forward at T = 401 (B = 8; g = 1 and g = 4; closed-form and seeded
torch-default init) against the CPU oracle with the north_star bar  enhanced-spectrum rel-L2 <= 1e-3 ; and the full
training step's gradients (fixture G6 at T = 21, and the oracle at T = 401) with stated tolerances."""
import pytest
import torch

from tests.util import rel_l2, t

pytestmark = pytest.mark.gpu

FWD_TOL = 1e-3               # north_star: "enhanced-spectrogram output matches the reference ... to <= 1e-3 rel"
# bf16-mode gradient tolerances (backward GEMMs run on plain bf16 operands, f32 accumulate):
GRAD_TOL_ALL = 1.5e-2        # all trained tensors as one vector, rel-L2 (measured r5: <= 1.15e-2 over g = 1 / 4, both inits)
GRAD_TOL_TENSOR = 0.07       # any weight matrix / conv kernel, rel-L2 (measured: <= 5.1e-2, conv4.weight)
GRAD_TOL_SUMS = 0.22         # bias / norm-affine vectors: sums over all frames of signed terms (cancellation; measured: <= 0.149 at T = 401,
                             # 0.187 on the T = 21 fixture G6) -- tightened in r5 from 2e-2 / 0.10 / 0.30; the figures repeat to four digits run to run


def _pair(grp, init, prec, seed=7):
    from cruse_amd.model import cruse_net as M
    from oracle import cruse_oracle as O
    if init == "closed":
        o = O.unet_2(rnn_groups=grp); O.closed_form_init(o)
    else:
        torch.manual_seed(seed); o = O.unet_2(rnn_groups=grp)
    m = M.unet_2(rnn_groups=grp, precision=prec)
    m.load_state_dict(o.state_dict(), strict=True)
    o.train(); m.train()
    return o, m.cuda()


def _dead_bias(n):
    """a conv bias that feeds a BatchNorm: its true gradient is exactly 0 (pure rounding noise on both sides)."""
    return n.endswith(".bias") and n.startswith("conv") and n != "conv1_t.bias"


def _grad_report(eng, o):
    allg, allo, worst = [], [], {}
    for n, p in o.named_parameters():
        if n not in eng.flat.G or p.grad is None or _dead_bias(n):
            continue
        r = rel_l2(eng.flat.G[n], p.grad)
        worst[n] = r
        allg.append(eng.flat.G[n].detach().double().cpu().flatten()); allo.append(p.grad.double().flatten())
    tot = float((torch.cat(allg) - torch.cat(allo)).norm() / torch.cat(allo).norm())
    return tot, worst


def _check_grads(tot, worst, tag):
    wmat = max((v, k) for k, v in worst.items() if k.endswith("weight") and "bn" not in k and ".ln" not in k)
    wsum = max((v, k) for k, v in worst.items() if not (k.endswith("weight") and "bn" not in k and ".ln" not in k))
    print(f"[grad parity {tag}] all {tot:.3e}  worst matrix {wmat[0]:.3e} ({wmat[1]})  worst bias/affine {wsum[0]:.3e} ({wsum[1]})")
    assert tot <= GRAD_TOL_ALL, (tag, tot)
    assert wmat[0] <= GRAD_TOL_TENSOR, (tag, wmat)
    assert wsum[0] <= GRAD_TOL_SUMS, (tag, wsum)


@pytest.mark.parametrize("grp,init", [(1, "closed"), (1, "random"), (4, "closed"), (4, "random")])
def test_bf16_forward_parity_at_bench_length(grp, init):
    """T = 401 (4 s clips), B = 8: 401 dependent recurrence steps per layer with h exchanged as bf16."""
    from cruse_amd import ops
    from cruse_amd.engine import TrainEngine
    from cruse_amd.model.cruse_net import unet2_forward
    from oracle import cruse_oracle as O
    B, T = 8, 401
    o, m = _pair(grp, init, "bf16")
    noisy, clean = O.synth_pair(B, (T - 1) * 160, seed=11)
    with torch.no_grad():
        mask_o, est_o, _ = O.enhanced_spectrum(o, noisy)
    eng = TrainEngine(m, use_graph=False)
    nre, nim, mag = ops.stft(noisy.cuda(), 320, 160, mag_bins=160, mag_eps=1e-8)
    assert mag.shape == (B, T, 160)
    mask, _ = unet2_forward(mag.view(B, 1, T, 160), eng.flat.P, eng.Bf, m.ch, m.rnn_groups, "bf16", training=True,
                            save=False, update_running=False)
    er, ei = ops.mask_apply(mask.contiguous().view(B * T, 160), nre, nim, B * T, 160, 161)
    est = torch.stack([er.view(B, T, 161), ei.view(B, T, 161)], dim=-1)
    e_est, e_mask = rel_l2(est, est_o), rel_l2(mask.view(B, 1, T, 160), mask_o)
    print(f"[parity bf16 T=401 B=8 g={grp} {init}] enhanced-spectrum rel-L2 {e_est:.3e}  mask {e_mask:.3e}")
    assert ops.gru_status() == 0
    assert e_est <= FWD_TOL and e_mask <= FWD_TOL
    assert float(est[..., 160, :].abs().max()) == 0.0          # R8: bin 160 of the enhanced spectrum is zero


@pytest.mark.parametrize("init", ["closed", "random"])
def test_f16_gate_projection_masks_at_bench_length(init):
    """EngineConfig.gi_f16 (round 4): the forward gate projections of GGRU layer 1 / 2 (bit 0 / 1) as one pass on f16 operands
    (cruse_gemm_f16_nt) instead of bf16 x with W_ih hi / lo planes.  Every mask meets the north_star bar at T = 401; the default (2) is not
    further from the oracle than the split-bf16 form (0) it replaced; mask 3 (layer 1 on f16 x against two f16 planes of W_ih) is the closest."""
    from cruse_amd import ops
    from cruse_amd.config import EngineConfig
    from cruse_amd.engine import TrainEngine
    from cruse_amd.model.cruse_net import unet2_forward
    from cruse_amd import config
    from cruse_amd.model import cruse_net as M
    from oracle import cruse_oracle as O
    B, T = 8, 401
    noisy, _ = O.synth_pair(B, (T - 1) * 160, seed=11)
    err = {}
    for mask_bits in (0, 1, 2, 3):
        o, m = _pair(1, init, "bf16")
        with torch.no_grad():
            mask_o, est_o, _ = O.enhanced_spectrum(o, noisy)
        eng = TrainEngine(m, use_graph=False, config=EngineConfig(gi_f16=mask_bits))
        nre, nim, mag = ops.stft(noisy.cuda(), 320, 160, mag_bins=160, mag_eps=1e-8)
        with config.use(eng.cfg), M.use_scheduler(eng.side):
            mask, _ = unet2_forward(mag.view(B, 1, T, 160), eng.flat.P, eng.Bf, m.ch, m.rnn_groups, "bf16", training=True,
                                    save=False, update_running=False)
        er, ei = ops.mask_apply(mask.contiguous().view(B * T, 160), nre, nim, B * T, 160, 161)
        est = torch.stack([er.view(B, T, 161), ei.view(B, T, 161)], dim=-1)
        err[mask_bits] = rel_l2(est, est_o)
        assert err[mask_bits] <= FWD_TOL
    print(f"[parity bf16 T=401 g=1 {init}] enhanced-spectrum rel-L2 by gi_f16 mask: " + ", ".join(f"{k}: {v:.3e}" for k, v in err.items()))
    assert EngineConfig().gi_f16 == 2                     # (layer 2: one f16 pass; mask 3 adds layer 1 on f16 x against W_ih hi + lo planes)
    assert err[2] <= 1.05 * err[0] and err[3] <= err[2]


@pytest.mark.parametrize("grp", [1, 4])
def test_bf16_forward_parity_at_the_full_bench_batch(grp):
    """the BENCH batch itself -- B = 64 x T = 401, i.e. 8 (g = 1) / 32 (g = 4) concurrent GRU chains
    placed by the per-XCD ticket (gru.hip claim_chain) -- against the CPU oracle, every clip on its own: a chain that
    picked up another chain's panel, or a clip routed to the wrong batch group, shows as ONE clip far off the rest."""
    from cruse_amd import ops
    from cruse_amd.engine import TrainEngine
    from cruse_amd.model.cruse_net import unet2_forward
    from oracle import cruse_oracle as O
    B, T = 64, 401
    o, m = _pair(grp, "random", "bf16")
    noisy, _ = O.synth_pair(B, (T - 1) * 160, seed=21)
    torch.set_num_threads(min(16, torch.get_num_threads()))
    with torch.no_grad():
        mask_o, est_o, _ = O.enhanced_spectrum(o, noisy)
    eng = TrainEngine(m, use_graph=False)
    nre, nim, mag = ops.stft(noisy.cuda(), 320, 160, mag_bins=160, mag_eps=1e-8)
    mask, _ = unet2_forward(mag.view(B, 1, T, 160), eng.flat.P, eng.Bf, m.ch, m.rnn_groups, "bf16", training=True,
                            save=False, update_running=False)
    er, ei = ops.mask_apply(mask.contiguous().view(B * T, 160), nre, nim, B * T, 160, 161)
    est = torch.stack([er.view(B, T, 161), ei.view(B, T, 161)], dim=-1).cpu()
    assert ops.gru_status() == 0
    e_all = rel_l2(est, est_o)
    per_clip = torch.tensor([rel_l2(est[b], est_o[b]) for b in range(B)])
    print(f"[parity bf16 B=64 T=401 g={grp}] enhanced-spectrum rel-L2 {e_all:.3e}; per clip max {float(per_clip.max()):.3e} "
          f"median {float(per_clip.median()):.3e}")
    assert e_all <= FWD_TOL
    assert float(per_clip.max()) <= 2.5 * FWD_TOL and float(per_clip.max()) <= 6 * float(per_clip.median())
    # BatchNorm statistics are batch-wide: the B = 8 and B = 64 runs are different computations, both pinned by the oracle


@pytest.mark.parametrize("grp", [1, 4])
def test_bf16_train_step_vs_golden_g6(golden, grp):
    """fixture G6 (B = 2, T = 21) in the bench mode: loss, enhanced spectrum, every gradient norm vs the fixture."""
    from cruse_amd.engine import TrainEngine
    from cruse_amd import ops
    g = golden(f"g6_step_g{grp}.npz")
    o, m = _pair(grp, "closed", "bf16")
    eng = TrainEngine(m, use_graph=False)
    noisy, clean = t(g["noisy"]), t(g["clean"])
    ls = eng._fwd_bwd(noisy, clean)
    assert abs(eng.loss_value(ls) - float(g["loss"])) <= 2e-4 * abs(float(g["loss"]))
    assert rel_l2(eng._last_mask.view(2, 1, 21, 160), torch.from_numpy(g["mask"])) <= FWD_TOL
    worst = 0.0
    for name in eng.flat.names:
        if "gn/" + name not in g.files or _dead_bias(name):
            continue
        gn, got = float(g["gn/" + name]), float(eng.flat.G[name].norm())
        is_mat = name.endswith("weight") and "bn" not in name and ".ln" not in name
        tol = GRAD_TOL_TENSOR if is_mat else GRAD_TOL_SUMS
        assert abs(got - gn) <= tol * gn + 2e-6, (name, got, gn)
        worst = max(worst, abs(got - gn) / max(gn, 1e-9))
    print(f"[grad parity bf16 G6 g={grp}] worst gradient-norm deviation {worst:.3e}")
    assert ops.gru_status() == 0


@pytest.mark.parametrize("grp,init", [(1, "closed"), (1, "random"), (4, "closed"), (4, "random")])
def test_bf16_gradients_vs_oracle_at_bench_length(grp, init):
    """The kernels the bench runs (lean forward recurrence, reduce-scatter backward recurrence, bf16-operand GEMMs)
    against ORACLE autograd -- not against the generic kernels -- at T = 401."""
    from cruse_amd.engine import TrainEngine
    from cruse_amd import ops
    from oracle import cruse_oracle as O
    B, T = 8, 401
    o, m = _pair(grp, init, "bf16")
    noisy, clean = O.synth_pair(B, (T - 1) * 160, seed=11)
    loss_o, _ = O.train_step_loss(o, noisy, clean)
    loss_o.backward()
    eng = TrainEngine(m, use_graph=False)
    ls = eng._fwd_bwd(noisy.cuda(), clean.cuda())
    assert abs(eng.loss_value(ls) - float(loss_o.detach())) <= 1e-4 * abs(float(loss_o.detach()))
    tot, worst = _grad_report(eng, o)
    _check_grads(tot, worst, f"T=401 g={grp} {init}")
    assert ops.gru_status() == 0


def test_f32_gate_mode_at_bench_length():
    """the parity-gate mode (exact-f32 MFMA) at T = 401: forward and gradients."""
    from cruse_amd.engine import TrainEngine
    from oracle import cruse_oracle as O
    B, T = 4, 401
    o, m = _pair(1, "random", "f32")
    noisy, clean = O.synth_pair(B, (T - 1) * 160, seed=12)
    loss_o, aux = O.train_step_loss(o, noisy, clean)
    loss_o.backward()
    eng = TrainEngine(m, use_graph=False)
    ls = eng._fwd_bwd(noisy.cuda(), clean.cuda())
    assert rel_l2(eng._last_mask.view(B, 1, T, 160), aux["mask"]) <= 1e-4
    assert abs(eng.loss_value(ls) - float(loss_o.detach())) <= 1e-5 * abs(float(loss_o.detach()))
    tot, worst = _grad_report(eng, o)
    print(f"[grad parity f32 T=401] all {tot:.3e} worst {max(worst.values()):.3e}")
    assert tot <= 1e-3 and max(worst.values()) <= 5e-3


def test_wide_chain_plan_at_b128_vs_oracle():
    """make_plan (csrc/gru.hip) switches to the 16-clip-chain kernels (gru_w16.hip) when B > 96 at Hg = 640, and until
    round 6 only kernel-vs-kernel tests covered them.  unet_2(rnn_groups = 1) at B = 128 x T = 401 in the bench mode against the CPU
    oracle: loss, enhanced spectrum of every clip, and all gradients against oracle autograd at the bench-length tolerances."""
    from cruse_amd import ops
    from cruse_amd.engine import TrainEngine
    from oracle import cruse_oracle as O
    B, T = 128, 401
    o, m = _pair(1, "random", "bf16")
    noisy, clean = O.synth_pair(B, (T - 1) * 160, seed=31)
    torch.set_num_threads(min(16, torch.get_num_threads()))
    with torch.no_grad():
        _, est_o, _ = O.enhanced_spectrum(o, noisy)
    loss_o, _ = O.train_step_loss(o, noisy, clean)
    loss_o.backward()
    eng = TrainEngine(m, use_graph=False)
    nz = noisy.cuda()
    ls = eng._fwd_bwd(nz, clean.cuda())
    assert ops.gru_status() == 0
    assert abs(eng.loss_value(ls) - float(loss_o.detach())) <= 1e-4 * abs(float(loss_o.detach()))
    nre, nim, _ = ops.stft(nz, 320, 160, mag_bins=160, mag_eps=1e-8)
    er, ei = ops.mask_apply(eng._last_mask.contiguous().view(B * T, 160), nre, nim, B * T, 160, 161)
    est = torch.stack([er.view(B, T, 161), ei.view(B, T, 161)], dim=-1).cpu()
    per_clip = torch.tensor([rel_l2(est[b], est_o[b]) for b in range(B)])
    e_all = rel_l2(est, est_o)
    print(f"[parity bf16 B=128 T=401 g=1, wide chains] enhanced-spectrum rel-L2 {e_all:.3e}; per clip max {float(per_clip.max()):.3e}")
    assert e_all <= FWD_TOL and float(per_clip.max()) <= 2.5 * FWD_TOL
    tot, worst = _grad_report(eng, o)
    _check_grads(tot, worst, "B=128 T=401 g=1 wide chains")
    # the plan really is the wide one at this batch (and the chains of 8 below it)
    assert ops.gru_plan(B, 1, 640)["clips_per_chain"] == 16 and ops.gru_plan(64, 1, 640)["clips_per_chain"] == 8


@pytest.mark.parametrize("init", ["closed", "random"])
def test_f16_stored_gate_preactivations_hold_the_oracle_bars(init, golden):
    """EngineConfig.gi_store_f16 (round 6, opt-in): gi rows of both GGRU layers stored as IEEE f16.  Against the CPU oracle at T = 401, B = 8: enhanced
    spectrum inside the north_star bar and no further from the oracle than 1.1 x the f32-row default; gradients inside the bench-mode tolerances; and the
    closed-form step fixture G6 holds."""
    from cruse_amd import config, ops
    from cruse_amd.config import EngineConfig
    from cruse_amd.engine import TrainEngine
    from cruse_amd.model import cruse_net as M
    from oracle import cruse_oracle as O
    B, T = 8, 401
    noisy, clean = O.synth_pair(B, (T - 1) * 160, seed=11)
    err = {}
    for on in (False, True):
        o, m = _pair(1, init, "bf16")
        if on:
            loss_o, _ = O.train_step_loss(o, noisy, clean)
            loss_o.backward()
        with torch.no_grad():
            _, est_o, _ = O.enhanced_spectrum(o, noisy)
        eng = TrainEngine(m, use_graph=False, config=EngineConfig(gi_store_f16=on))
        ls = eng._fwd_bwd(noisy.cuda(), clean.cuda())
        nre, nim, _ = ops.stft(noisy.cuda(), 320, 160, mag_bins=160, mag_eps=1e-8)
        er, ei = ops.mask_apply(eng._last_mask.contiguous().view(B * T, 160), nre, nim, B * T, 160, 161)
        est = torch.stack([er.view(B, T, 161), ei.view(B, T, 161)], dim=-1)
        err[on] = rel_l2(est, est_o)
        if on:
            tot, worst = _grad_report(eng, o)
            _check_grads(tot, worst, f"gi_store_f16 T=401 g=1 {init}")
    print(f"[parity bf16 T=401 g=1 {init}] enhanced-spectrum rel-L2 with f32 / f16 gi rows: {err[False]:.3e} / {err[True]:.3e}")
    assert err[True] <= FWD_TOL and err[True] <= 1.1 * err[False]
    if init == "closed":
        g = golden("g6_step_g1.npz")
        o, m = _pair(1, "closed", "bf16")
        eng = TrainEngine(m, use_graph=False, config=EngineConfig(gi_store_f16=True))
        ls = eng._fwd_bwd(t(g["noisy"]), t(g["clean"]))
        assert abs(eng.loss_value(ls) - float(g["loss"])) <= 2e-4 * abs(float(g["loss"]))
        assert rel_l2(eng._last_mask.view(2, 1, 21, 160), torch.from_numpy(g["mask"])) <= FWD_TOL
    assert ops.gru_status() == 0
