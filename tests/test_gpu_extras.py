"""GPU parity of the rows either side of the hot path (8f items 2-3;
BASELINE config 4) against the round-2 fixtures generated from the reference's own code (tests/golden/make_golden_r2.py)
and against oracle autograd for the gradients."""
import numpy as np
import pytest
import torch

from tests.util import max_abs, rel_l2, t

pytestmark = pytest.mark.gpu


def _load_like(product, oracle_mod, scale=1.0):
    from oracle import cruse_oracle as O
    O.closed_form_init(oracle_mod, scale)
    product.load_state_dict(oracle_mod.state_dict(), strict=True)
    return product.cuda()


def _grad_check(product, oracle_mod, x_cpu, tol=2e-4, wtol=None, list_input=False):
    """same random cotangent through both; input and parameter gradients."""
    wtol = wtol or 5 * tol
    gen = torch.Generator().manual_seed(99)
    if list_input:
        xo = [a.clone().requires_grad_(True) for a in x_cpu]
        xp = [a.clone().cuda().requires_grad_(True) for a in x_cpu]
    else:
        xo = x_cpu.clone().requires_grad_(True)
        xp = x_cpu.clone().cuda().requires_grad_(True)
    yo = oracle_mod(xo)
    yo = yo[0] if isinstance(yo, tuple) else yo
    w = torch.randn(yo.shape, generator=gen)
    (yo * w).sum().backward()
    yp = product(xp)
    yp = yp[0] if isinstance(yp, tuple) else yp
    (yp * w.cuda()).sum().backward()
    for a, b in zip(xp if list_input else [xp], xo if list_input else [xo]):
        assert rel_l2(a.grad, b.grad) < tol
    for (n, po), (_, pp) in zip(oracle_mod.named_parameters(), product.named_parameters()):
        if po.grad is None:
            continue
        gn = float(po.grad.norm())
        if gn < 1e-3:                                                 # e.g. a conv bias in front of a BatchNorm: exactly 0 in
            assert float(pp.grad.norm()) < 5e-3, n                    # exact arithmetic, rounding noise on both sides
            continue
        assert rel_l2(pp.grad, po.grad) < wtol, n


# ---------------------------------------------------------------------------------------------------------------- a13
def test_rmse_c_rmse_sisnr_wo_male_vs_reference(golden):
    from cruse_amd import loss_func as L
    from oracle import cruse_oracle as O
    from oracle import cruse_oracle_ext as X
    g = golden("g10_losses.npz")
    ref, est = torch.from_numpy(g["ref"]), torch.from_numpy(g["est"])
    for name, fn, ofn, tol in (("rmse", L.rmse, X.rmse, 1e-5), ("c_rmse", L.c_rmse, X.c_rmse, 2e-5)):
        e = est.clone().cuda().requires_grad_(True)
        v = fn(ref.cuda(), e)
        assert abs(float(v) - float(g[name])) <= tol * abs(float(g[name])), name
        v.backward()
        eo = est.clone().requires_grad_(True)
        ofn(ref, eo).backward()
        assert rel_l2(e.grad, eo.grad) < 1e-4, name
    g5 = golden("g5_loss.npz")
    s1, s2 = torch.from_numpy(g5["s1"]), torch.from_numpy(g5["s2"])
    a = s1.clone().cuda().requires_grad_(True)
    v = L.sisnr(a, s2.cuda())
    assert abs(float(v) - float(g5["sisnr"])) <= 1e-5 * abs(float(g5["sisnr"]))          # the reference's own run (G5)
    v.backward()
    ao = s1.clone().requires_grad_(True)
    O.sisnr(ao, s2).backward()
    assert rel_l2(a.grad, ao.grad) < 1e-4
    # loss_func selector class (loss_func/loss.py:15-35) incl. wo_male on explicit spectra (value pinned in G5)
    r5, e5, u5 = (torch.from_numpy(g5[k]) for k in ("ref", "est", "unproc"))
    lf = L.loss_func("WO_MALE")
    e = e5.clone().cuda().requires_grad_(True)
    v = lf.loss(e, r5.cuda(), u5.cuda())
    assert abs(float(v) - float(g5["wo_male"])) <= 1e-5 * abs(float(g5["wo_male"]))
    v.backward()
    eo = e5.clone().requires_grad_(True)
    O.wo_male(r5, eo, u5).backward()
    assert rel_l2(e.grad, eo.grad) < 1e-4
    assert abs(float(L.loss_func("SI-SNR").loss(s1.cuda(), s2.cuda())) + float(g5["sisnr"])) <= 1e-5 * abs(float(g5["sisnr"]))
    assert abs(float(L.loss_func("MSE").loss(est.cuda(), ref.cuda())) - float(g["rmse"])) <= 1e-5 * float(g["rmse"])
    with pytest.raises(RuntimeError, match="Dimension mismatch"):
        L.rmse(ref.cuda(), est[:, :, :5].cuda())


# ---------------------------------------------------------------------------------------------------------------- a14
def test_mask_py_vs_reference(golden):
    from cruse_amd.acoustics import mask as M
    g = golden("g9_mask.npz")
    a, b, c, d = (t(g[k]) for k in "abcd")
    r, i = M.complex_mul(a, b, c, d)
    assert max_abs(r, torch.from_numpy(g["cm_r"])) < 1e-6 and max_abs(i, torch.from_numpy(g["cm_i"])) < 1e-6
    irm = M.build_ideal_ratio_mask(a.abs(), c.abs())
    assert irm.shape == (2, 5, 7, 1) and rel_l2(irm, torch.from_numpy(g["irm"])) < 1e-5
    cirm = M.build_complex_ideal_ratio_mask(torch.complex(a, b), torch.complex(c, d))
    assert cirm.shape == (2, 5, 7, 2) and rel_l2(cirm, torch.from_numpy(g["cirm"])) < 1e-5
    dec = M.decompress_cIRM(t(g["cirm"]))
    assert rel_l2(dec, torch.from_numpy(g["decomp"])) < 1e-5
    x = torch.linspace(-150, 50, 64).cuda()                                   # the -100 clamp branch of compress (mask.py:47)
    from oracle import cruse_oracle_ext as X
    assert rel_l2(M.compress_cIRM(x), X.compress_cIRM(x.cpu())) < 1e-5
    # full-size round trip (B=8, F=161, T=401): decompress(compress(m)) == m inside the limit
    m = (torch.rand(8, 161, 401, 2) * 8 - 4).cuda()
    assert rel_l2(M.decompress_cIRM(M.compress_cIRM(m)), m) < 1e-4


def test_elementwise_ops_carry_gradients():
    """mask.py / PreProcess.masking / CustomSTFT m,p / CustomISTFT are plain (differentiable) torch in the
    reference and sit between the network and the loss -- the HIP versions must carry the same gradients."""
    from cruse_amd.acoustics import mask as M
    from cruse_amd.acoustics.feature import mag_phase, polar_to_rect
    from cruse_amd.acoustics.preprocess import _pair_op
    from oracle import cruse_oracle_ext as X
    torch.manual_seed(3)
    shp = (2, 5, 7)
    a, b, c, d = (torch.randn(shp) for _ in range(4))
    w1, w2 = torch.randn(shp), torch.randn(shp)

    def grads(fn, inputs, dev):
        xs = [x.clone().to(dev).requires_grad_(True) for x in inputs]
        outs = fn(*xs)
        outs = outs if isinstance(outs, tuple) else (outs,)
        sum((o * w.to(dev)).sum() for o, w in zip(outs, (w1, w2))).backward()
        return [x.grad.cpu() for x in xs]

    cases = [
        ("complex_mul", M.complex_mul, lambda nr, ni, mr, mi: (nr * mr - ni * mi, nr * mi + ni * mr), (a, b, c, d)),   # mask.py:61-64
        ("compress", M.compress_cIRM, X.compress_cIRM, (3 * a,)),
        ("decompress", M.decompress_cIRM, X.decompress_cIRM, (6 * a,)),                 # incl. elements beyond +-9.9 (clamped: 0)
        ("pair", lambda p, q, r, s_: _pair_op(5, p, q, r, s_), lambda p, q, r, s_: (p * r, q * s_), (a, b, c, d)),
        ("mag_phase", mag_phase, lambda r, i: (torch.sqrt(r * r + i * i), torch.atan2(i, r)), (a, b)),
        ("polar_to_rect", polar_to_rect, lambda m, p: (m * torch.cos(p), m * torch.sin(p)), (a.abs(), b)),
    ]
    for name, fn, ref, inputs in cases:
        got, want = grads(fn, inputs, "cuda"), grads(ref, inputs, "cpu")
        for k, (g_, w_) in enumerate(zip(got, want)):
            assert rel_l2(g_, w_) < 1e-5, (name, k, rel_l2(g_, w_))
    # "mag_mapping" passes the SAME mask twice: both products' gradients must add up in it
    m = c.clone().cuda().requires_grad_(True)
    o1, o2 = _pair_op(5, a.cuda(), b.cuda(), m, m)
    (o1 * w1.cuda() + o2 * w2.cuda()).sum().backward()
    assert rel_l2(m.grad.cpu(), a * w1 + b * w2) < 1e-6


def test_sdnr_wrapper_and_noncontiguous_estimates(golden):
    """loss_func.sdnr (loss_func/loss.py:151-175) through the dotted path: fixture G18 values, gradient vs oracle autograd;
    rmse / c_rmse / wo_male with a NON-contiguous estimate (`est.contiguous()` lost requires_grad)."""
    from loss_func.loss import c_rmse, rmse, sdnr, wo_male
    from oracle import cruse_oracle as O
    from oracle import cruse_oracle_ext as X
    g = golden("g18_sdnr.npz")
    clean, noise, gain = (torch.from_numpy(g[k]) for k in ("clean", "noise", "gain"))
    for k, snr in enumerate(g["snr"]):
        gg = gain.clone().cuda().requires_grad_(True)
        v = sdnr(clean.cuda(), gg, noise.cuda(), snr=float(snr))
        want = float(g["value"][k])
        assert abs(float(v) - want) <= 2e-5 * abs(want)
        v.backward()
        go = gain.clone().requires_grad_(True)
        O.sdnr(clean, go, noise, float(snr), beta=20.0).backward()
        assert rel_l2(gg.grad, go.grad) < 1e-4
    g10 = golden("g10_losses.npz")
    ref, est = torch.from_numpy(g10["ref"]), torch.from_numpy(g10["est"])
    for fn, ofn in ((rmse, X.rmse), (c_rmse, X.c_rmse), (lambda r, e: wo_male(r, e, r + 0.5 * e.detach()), None)):
        e_t = est.transpose(2, 3).contiguous().cuda().requires_grad_(True)       # the network's output, then permuted
        v = fn(ref.cuda(), e_t.transpose(2, 3))
        v.backward()
        assert e_t.grad is not None and torch.isfinite(e_t.grad).all()
        if ofn is not None:
            eo = est.clone().requires_grad_(True)
            ofn(ref, eo).backward()
            assert rel_l2(e_t.grad.transpose(2, 3), eo.grad) < 1e-4


# ---------------------------------------------------------------------------------------------------------------- a9
CONV_CASES = {
    "cna": lambda m: m.Conv2dNormAct(1, 16, (2, 3), fstride=2),
    "cna_sep": lambda m: m.Conv2dNormAct(16, 32, (2, 3), fstride=2, separable=True),
    "cna_dil": lambda m: m.Conv2dNormAct(16, 16, (3, 3), fstride=1, dilation=2, norm_layer=None),
    "ctna": lambda m: m.ConvTranspose2dNormAct(16, 8, (2, 3), fstride=2),
    "ctna_sep": lambda m: m.ConvTranspose2dNormAct(16, 8, (1, 3), fstride=2, separable=True),
    "kxf_normal": lambda m: m.convkxf(16, 32, k=2, f=3, fstride=2, batch_norm=True),
    "kxf_transposed": lambda m: m.convkxf(16, 8, k=2, f=3, fstride=2, mode="transposed", batch_norm=True),
    "kxf_upsample": lambda m: m.convkxf(16, 8, k=2, f=3, fstride=2, mode="upsample", batch_norm=True),
    "kxf_full": lambda m: m.convkxf(16, 8, k=1, f=3, fstride=2, mode="upsample", depthwise=False),
}


@pytest.mark.parametrize("case", sorted(CONV_CASES))
def test_cust_conv_blocks_vs_reference(golden, case):
    """Conv2dNormAct / ConvTranspose2dNormAct / convkxf (normal, transposed, upsample; depthwise + 1x1): outputs of the
    reference's imported modules in train and eval mode; gradients vs oracle autograd."""
    from cruse_amd.model.based_model import cust_conv as P
    from oracle import cruse_oracle_ext as X
    g = golden("g11_convblocks.npz")
    o = CONV_CASES[case](X)
    p = _load_like(CONV_CASES[case](P), o)
    assert list(p.state_dict().keys()) == list(o.state_dict().keys())
    x = torch.from_numpy(g[f"{case}/x"])
    o.train(); p.train()
    y = p(x.cuda())
    want = torch.from_numpy(g[f"{case}/y_train"])
    assert y.shape == want.shape and rel_l2(y, want) < 2e-5, case
    p.eval()
    with torch.no_grad():
        assert rel_l2(p(x.cuda()), torch.from_numpy(g[f"{case}/y_eval"])) < 2e-5, case
    # gradients (fresh modules: the train-mode call above moved the running statistics)
    o2 = CONV_CASES[case](X)
    p2 = _load_like(CONV_CASES[case](P), o2)
    o2.train(); p2.train()
    _grad_check(p2, o2, x, tol=5e-4)
    for k, v in o2.state_dict().items():
        if "running" in k or "num_batches" in k:
            assert max_abs(p2.state_dict()[k].float(), v.float()) < 1e-4, (case, k)


def test_freq_upsample_standalone(golden):
    from cruse_amd.model.based_model.cust_conv import FreqUpsample
    g = golden("g11_convblocks.npz")
    y = FreqUpsample(2)(t(g["upsample/x"]))
    assert torch.equal(y.cpu(), torch.from_numpy(g["upsample/y"]))


@pytest.mark.parametrize("grp", [1, 4])
def test_cruse4_mag_add_skip_upsample_model(golden, grp):
    """model.cruse.CRUSE4MagAddSkipUpsample (model/cruse.py:14): encoder Conv2dNormAct + additive conv skips +
    GGRU + the nearest-upsample decoder convkxf(mode="upsample"), against fixture G21 (the composition run on the
    reference's own blocks) -- mask in train and eval mode, every parameter-gradient norm -- and oracle autograd; then the
    bench clip length."""
    from model.cruse import CRUSE4MagAddSkipUpsample
    from oracle import cruse_oracle_ext as X
    g = golden("g21_cruse_upsample.npz")
    o = X.CRUSE4MagAddSkipUpsample(rnn_groups=grp)
    p = _load_like(CRUSE4MagAddSkipUpsample(rnn_groups=grp, precision="f32"), o)
    assert list(p.state_dict().keys()) == list(o.state_dict().keys())
    x, w = torch.from_numpy(g["x"]), torch.from_numpy(g["w"])
    o.train(); p.train()
    y = p(x.cuda())
    assert rel_l2(y, torch.from_numpy(g[f"g{grp}/mask_train"])) < 1e-5
    (y * w.cuda()).sum().backward()
    for n, q in p.named_parameters():
        want = float(g[f"g{grp}/gn/{n}"])
        if want < 1e-6 or (n.startswith("enc") and n.endswith(".1.bias")):
            continue                                                   # conv biases in front of a BatchNorm: exactly 0 (rounding noise)
        assert abs(float(q.grad.norm()) - want) <= 2e-3 * want + 1e-7, (n, float(q.grad.norm()), want)
    p.eval()
    with torch.no_grad():
        assert rel_l2(p(x.cuda()), torch.from_numpy(g[f"g{grp}/mask_eval"])) < 1e-5
    p.train()
    p.zero_grad(set_to_none=True)
    _grad_check(p, o, x, tol=1e-3, wtol=5e-3)
    with pytest.raises(RuntimeError, match="expects"):
        p(torch.rand(2, 1, 11, 161).cuda())
    # bench clip length, bf16 mode against the f32 mode of the same weights: mask within 1e-3, gradients below
    torch.manual_seed(5)
    big = CRUSE4MagAddSkipUpsample(rnn_groups=grp, precision="bf16").cuda()
    ref = CRUSE4MagAddSkipUpsample(rnn_groups=grp, precision="f32").cuda()
    ref.load_state_dict(big.state_dict())
    xb, wb = torch.rand(8, 1, 401, 160).cuda() + 0.05, torch.randn(8, 1, 401, 160).cuda()
    m, mr = big(xb), ref(xb)
    assert m.shape == (8, 1, 401, 160) and float(m.min()) > 0.0 and float(m.max()) < 1.0
    assert rel_l2(m, mr) < 1e-3
    (m * wb).sum().backward(); (mr * wb).sum().backward()
    # (bf16 operands in every backward contraction and a white-noise cotangent: 5.9e-2 over all gradients, worst tensor 7.6e-2 --
    # unet_2 measures the same 5.9e-2 / 7.6e-2 under this probe, tools/scratch/bf16_grad_probe.py; the bars are 8e-2 / 0.15)
    pairs = [(n, a.grad, b.grad) for (n, a), (_, b) in zip(big.named_parameters(), ref.named_parameters()) if float(b.grad.norm()) > 1e-3]
    allg = rel_l2(torch.cat([a.flatten() for _, a, _ in pairs]), torch.cat([b.flatten() for _, _, b in pairs]))
    worst = max((rel_l2(a, b), n) for n, a, b in pairs)
    print(f"[cruse4 upsample g={grp}] bf16 vs f32 mode: mask {rel_l2(m, mr):.2e}, all gradients {allg:.2e}, worst tensor {worst[1]} {worst[0]:.2e}")
    assert allg < 8e-2 and worst[0] < 0.15, (allg, worst)


def test_upsample_w_and_its_gradient():
    from cruse_amd import ops
    torch.manual_seed(0)
    for rows, W in ((7, 10), (33, 5), (64 * 8, 80)):
        x = torch.randn(rows, W).cuda()
        xu = ops.upsample_w(x, rows, W, 2)
        assert torch.equal(xu, x.repeat_interleave(2, dim=1))
        g = torch.randn(rows, 2 * W).cuda()
        assert torch.equal(ops.downsum_w(g, rows, W, 2), g[:, 0::2] + g[:, 1::2])
    x3 = torch.randn(6, 4).cuda()
    assert torch.equal(ops.upsample_w(x3, 6, 4, 3), x3.repeat_interleave(3, dim=1))


# ---------------------------------------------------------------------------------------------------------------- a10
@pytest.mark.parametrize("name,kw", [("g2_l2", dict(num_layers=2, groups=2)), ("g4_l3_add", dict(num_layers=3, groups=4, add_outputs=True)),
                                     ("g2_noshuffle", dict(num_layers=2, groups=2, shuffle=False))])
def test_group_gru_with_shuffle_vs_reference(golden, name, kw):
    from cruse_amd.model.based_model.cust_conv import GroupGRU
    from oracle import cruse_oracle_ext as X
    g = golden("g12_groupgru.npz")
    o = X.GroupGRU(128, 128, **kw)
    p = _load_like(GroupGRU(128, 128, **kw), o, scale=2.0)
    x = torch.from_numpy(g["x"])
    y, s = p(x.cuda())
    assert rel_l2(y, torch.from_numpy(g[f"{name}/y"])) < 1e-5
    assert rel_l2(s, torch.from_numpy(g[f"{name}/state"])) < 1e-5
    y2, _ = p(x.cuda(), p.get_h0(2, device="cuda"))                              # explicit zero state: same result
    assert rel_l2(y2, y) < 1e-7
    with pytest.raises(RuntimeError, match="state has"):
        p(x.cuda(), torch.ones(1, 2, 128 // kw["groups"]).cuda())
    _grad_check(p, o, x, tol=5e-4, wtol=2e-3)


@pytest.mark.parametrize("name,kw", [("g2_l2", dict(num_layers=2, groups=2)), ("g4_l3_add", dict(num_layers=3, groups=4, add_outputs=True)),
                                     ("g1_l1", dict(num_layers=1, groups=1))])
def test_group_gru_nonzero_state_vs_reference(golden, name, kw):
    """GroupGRU.forward(x, state) with the explicit, NON-ZERO state the reference's working call path
    passes (cust_conv.py:392-416; fixture G19 = the reference's own run).  The state is detached there (:319), so the
    gradients wrt input and weights -- including the h0 term of dW_hh -- are compared with oracle autograd."""
    from cruse_amd.model.based_model.cust_conv import GroupGRU
    from oracle import cruse_oracle_ext as X
    g = golden("g19_groupgru_state.npz")
    o = X.GroupGRU(128, 128, **kw)
    p = _load_like(GroupGRU(128, 128, **kw), o, scale=2.0)
    x, st = torch.from_numpy(g["x"]), torch.from_numpy(g[f"{name}/state_in"])
    y, s = p(x.cuda(), st.cuda())
    assert rel_l2(y, torch.from_numpy(g[f"{name}/y"])) < 1e-5
    assert rel_l2(s, torch.from_numpy(g[f"{name}/state"])) < 1e-5
    xo = x.clone().requires_grad_(True); xp = x.clone().cuda().requires_grad_(True)
    torch.manual_seed(1)
    w = torch.randn(y.shape)
    (o(xo, st)[0] * w).sum().backward()
    (p(xp, st.cuda())[0] * w.cuda()).sum().backward()
    assert rel_l2(xp.grad, xo.grad) < 5e-4
    for (n, po), (_, pp) in zip(o.named_parameters(), p.named_parameters()):
        assert rel_l2(pp.grad, po.grad) < 2e-3, n


@pytest.mark.parametrize("name,grp", [("g2", 2), ("g1", 1)])
def test_grouped_gru_layer_bidirectional(golden, name, grp):
    """GroupedGRULayer(bidirectional=True, dropout=0.3) (cust_conv.py:250-325; fixture G22 from the reference's class as shipped):
    every group's nn.GRU runs both directions -- here the reverse one is the same persistent recurrence on the time-reversed
    rows with the *_reverse weights -- a group's output is [forward | reverse], its two final states h_{T-1} / h_0; dropout
    on a one-layer nn.GRU has no effect.  Outputs and states vs the fixture (with and without an initial state), gradients vs
    oracle autograd."""
    from cruse_amd.model.based_model.cust_conv import GroupedGRULayer
    from oracle import cruse_oracle_ext as X
    g = golden("g22_grouped_gru_bidirectional.npz")
    o = X.GroupedGRULayer(128, 128, grp, dropout=0.3, bidirectional=True).train()
    p = _load_like(GroupedGRULayer(128, 128, grp, dropout=0.3, bidirectional=True), o, scale=2.0).train()
    assert p.num_directions == 2 and tuple(p.get_h0(3).shape) == (grp * 2, 3, 128 // grp)
    x, st = torch.from_numpy(g["x"]), torch.from_numpy(g[f"{name}/state_in"])
    y, s = p(x.cuda(), st.cuda())
    assert tuple(y.shape) == (3, 9, 256)
    assert rel_l2(y, torch.from_numpy(g[f"{name}/y"])) < 1e-5 and rel_l2(s, torch.from_numpy(g[f"{name}/state"])) < 1e-5
    y0, s0 = p(x.cuda())
    assert rel_l2(y0, torch.from_numpy(g[f"{name}/y_zero_state"])) < 1e-5
    assert rel_l2(s0, torch.from_numpy(g[f"{name}/state_zero_state"])) < 1e-5
    xo = x.clone().requires_grad_(True); xp = x.clone().cuda().requires_grad_(True)
    torch.manual_seed(1)
    w = torch.randn(y.shape)
    (o(xo, st)[0] * w).sum().backward()
    (p(xp, st.cuda())[0] * w.cuda()).sum().backward()
    assert rel_l2(xp.grad, xo.grad) < 5e-4
    for (n, po), (_, pp) in zip(o.named_parameters(), p.named_parameters()):
        assert rel_l2(pp.grad, po.grad) < 2e-3, n


@pytest.mark.parametrize("name,kw", [("tb_g2", dict(batch_first=False)), ("tb_g2_nobias", dict(batch_first=False, bias=False)),
                                     ("bt_g2_nobias", dict(bias=False))])
def test_grouped_gru_layer_time_major_and_bias_free(golden, name, kw):
    """GroupedGRULayer(batch_first=False) / (bias=False) (cust_conv.py:259-277,303-325; fixture G23 = the reference's class as
    shipped): [T, B, I] input and output, state [G, B, H/g] either way; the bias-free layer has no bias parameters (state-dict
    keys as the reference's).  Outputs with and without an initial state vs the fixture, gradients vs oracle autograd."""
    from cruse_amd.model.based_model.cust_conv import GroupedGRULayer
    from oracle import cruse_oracle_ext as X
    g = golden("g23_grouped_gru_layouts.npz")
    o = X.GroupedGRULayer(128, 128, 2, **kw)
    p = _load_like(GroupedGRULayer(128, 128, 2, **kw), o, scale=2.0)
    assert list(p.state_dict().keys()) == list(o.state_dict().keys())
    assert ("bias" in " ".join(p.state_dict().keys())) == kw.get("bias", True)
    x_tb = torch.from_numpy(g["x_tb"])
    x = x_tb if not kw.get("batch_first", True) else x_tb.transpose(0, 1).contiguous()
    st = torch.from_numpy(g[f"{name}/state_in"])
    y, s = p(x.cuda(), st.cuda())
    assert y.shape == x.shape[:2] + (128,)
    assert rel_l2(y, torch.from_numpy(g[f"{name}/y"])) < 1e-5 and rel_l2(s, torch.from_numpy(g[f"{name}/state"])) < 1e-5
    y0, _ = p(x.cuda())
    assert rel_l2(y0, torch.from_numpy(g[f"{name}/y0"])) < 1e-5
    xo = x.clone().requires_grad_(True); xp = x.clone().cuda().requires_grad_(True)
    torch.manual_seed(2)
    w = torch.randn(y.shape)
    (o(xo, st)[0] * w).sum().backward()
    (p(xp, st.cuda())[0] * w.cuda()).sum().backward()
    assert rel_l2(xp.grad, xo.grad) < 5e-4
    for (n, po), (_, pp) in zip(o.named_parameters(), p.named_parameters()):
        assert rel_l2(pp.grad, po.grad) < 2e-3, n


@pytest.mark.parametrize("prec,Hg,G", [("f32", 128, 2), ("bf16", 640, 1), ("bf16", 160, 4), ("bf16x3", 96, 1)])
def test_recurrence_in_time_chunks_equals_one_launch(prec, Hg, G):
    """cruse_gru_seq_fwd_ex / _bwd_ex: a sequence run as consecutive time chunks (forward: the state carried through h;
    backward: the gradient carried through dh) gives the results of the single launch -- for the generic kernels and for
    the lean forward / reduce-scatter backward kernels of the bench mode."""
    from cruse_amd import ops
    torch.manual_seed(5)
    B, T, H = 11, 37, G * Hg
    gi = torch.randn(B, T, 3 * H).cuda()
    w = [(torch.randn(3 * Hg, Hg) / Hg ** 0.5).cuda() for _ in range(G)]
    b = [(0.1 * torch.randn(3 * Hg)).cuda() for _ in range(G)]
    h0 = (0.5 * torch.randn(B, H)).cuda()
    dout = torch.randn(B, T, H).cuda()
    want_dgi = prec == "bf16"
    h, coef, an, z = ops.gru_seq_fwd(gi, w, b, B, T, G, Hg, prec, h0=h0)
    ref_b = ops.gru_seq_bwd(dout, w, coef, z, B, T, G, Hg, prec, an=an if want_dgi else None, want_dgi=want_dgi)
    chunks = [(0, 10), (10, 1), (11, 17), (28, 9)]
    out = tuple(torch.full_like(t_, float("nan")) for t_ in (h, coef, an, z))
    for c in chunks:
        ops.gru_seq_fwd(gi, w, b, B, T, G, Hg, prec, h0=h0, out=out, chunk=c)
    for got, ref in zip(out, (h, coef, an, z)):
        assert torch.equal(got, ref)
    dh = torch.full_like(dout, float("nan"))
    bout = (dh, ops.dgi_buffer(B * T, G, Hg, dh.device)) if want_dgi else dh
    for c in reversed(chunks):
        ops.gru_seq_bwd(dout, w, coef, z, B, T, G, Hg, prec, an=an if want_dgi else None, want_dgi=want_dgi, out=bout, chunk=c)
    if want_dgi:
        assert torch.equal(bout[0], ref_b[0]) and torch.equal(bout[1], ref_b[1])
    else:
        assert torch.equal(bout, ref_b)
    assert ops.gru_status() == 0
    # and the initial state matters / is honoured: against a torch reference of step 0
    h_z, _, _, _ = ops.gru_seq_fwd(gi, w, b, B, T, G, Hg, prec, save=False)
    assert float((h_z[:, 0] - h[:, 0]).abs().max()) > 1e-3
    for gq in range(G):
        sl = slice(gq * Hg, (gq + 1) * Hg)
        gh = h0[:, sl] @ w[gq].t() + b[gq]
        g0 = gi[:, 0, gq * 3 * Hg:(gq + 1) * 3 * Hg]
        r = torch.sigmoid(g0[:, :Hg] + gh[:, :Hg]); zz = torch.sigmoid(g0[:, Hg:2 * Hg] + gh[:, Hg:2 * Hg])
        n = torch.tanh(g0[:, 2 * Hg:] + r * gh[:, 2 * Hg:])
        want = (1 - zz) * n + zz * h0[:, sl]
        assert rel_l2(h[:, 0, sl], want) < (2e-2 if prec == "bf16" else 1e-4 if prec == "bf16x3" else 1e-5)


@pytest.mark.parametrize("grp", [1, 2, 4])
def test_grouped_gru_layer_l1cat_fixture(golden, grp):
    """the `l1cat_g*` arrays of fixture G3 (cust_conv.GroupedGRULayer = GGRU's first layer before the interleave)."""
    from cruse_amd.model.based_model.cust_conv import GroupedGRULayer
    from oracle import cruse_oracle as O
    g = golden("g3_ggru.npz")
    mine = O.GGRU(hidden_size=640, groups=grp)
    O.closed_form_init(mine)
    lay = GroupedGRULayer(640, 640, grp)
    for i in range(grp):
        lay.layers[i].load_state_dict(mine.gru_list1[i].state_dict())
    lay = lay.cuda()
    seq = torch.from_numpy(g["x"]).transpose(1, 2).reshape(2, 21, 640)
    out, h = lay(seq.cuda(), lay.get_h0(2, device="cuda"))
    assert rel_l2(out, torch.from_numpy(g[f"l1cat_g{grp}"])) < 1e-5
    assert h.shape == (grp, 2, 640 // grp) and rel_l2(h.transpose(0, 1).reshape(2, 640), out[:, -1]) < 1e-7


# ---------------------------------------------------------------------------------------------------------------- a5, a6
@pytest.mark.parametrize("nfft,tag", [(None, "512"), (320, "320")])
def test_custom_stft_istft_vs_reference(golden, nfft, tag):
    from train_base.acoustics.feature import CustomISTFT, CustomSTFT
    g = golden("g13_stft_variants.npz")
    wav = t(g["wav"])
    st = CustomSTFT(320, 160, num_fft=nfft).cuda()
    ist = CustomISTFT(320, 160, num_fft=nfft).cuda()
    assert list(st.state_dict().keys()) == ["K"]
    m, p, r, i = st(wav)
    F = (512 if nfft is None else nfft) // 2 + 1
    assert m.shape == (2, F, 19)                                                  # T = (3200 - 320)//160 + 1, no centre padding
    assert rel_l2(r, torch.from_numpy(g[f"custom{tag}/r"])) < 1e-5 and rel_l2(i, torch.from_numpy(g[f"custom{tag}/i"])) < 1e-5
    assert rel_l2(m, torch.from_numpy(g[f"custom{tag}/m"])) < 1e-5
    y = ist(t(g[f"custom{tag}/m"]), t(g[f"custom{tag}/p"]))
    want = torch.from_numpy(g[f"custom{tag}/y"])
    assert y.shape == want.shape == (2, 1, 3200) and rel_l2(y, want) < 1e-5
    # the layer is differentiable wrt the waveform: the adjoint pair
    x = wav.clone().requires_grad_(True)
    _, _, r2, i2 = st(x)
    (r2.square().sum() + i2.square().sum()).backward()
    from oracle import cruse_oracle_ext as X
    xo = wav.cpu().clone().requires_grad_(True)
    _, _, ro, io = X.custom_stft(xo, X.init_stft_kernel(320, 160, num_fft=nfft), 160)
    (ro.square().sum() + io.square().sum()).backward()
    assert rel_l2(x.grad, xo.grad) < 1e-4
    with pytest.raises(RuntimeError, match="Expect 2D/3D"):
        st(torch.zeros(1, 1, 1, 400).cuda())


def test_conv_stft_hamming_vs_reference(golden):
    from train_base.acoustics.conv_stft import STFT
    g = golden("g13_stft_variants.npz")
    wav = t(g["wav"])
    cs = STFT(320, 160).cuda()
    assert set(cs.state_dict().keys()) == {"win", "fourier_basis_r", "fourier_basis_i", "idx"}
    sr, si, mag, pha = cs.stft(wav)
    assert sr.shape == (2, 21, 161)
    assert rel_l2(sr, torch.from_numpy(g["conv/spec_r"])) < 1e-5 and rel_l2(si, torch.from_numpy(g["conv/spec_i"])) < 1e-5
    assert rel_l2(mag, torch.from_numpy(g["conv/mag"])) < 1e-5
    rt = cs.istft(torch.stack([sr, si], dim=1))
    assert rt.shape == wav.shape and max_abs(rt, wav) < 5e-6                       # the (repaired) inverse inverts
    assert rel_l2(rt, torch.from_numpy(g["conv/roundtrip"])) < 1e-4
    # full size: 64 clips x 4 s
    big = torch.randn(64, 64000).cuda() * 0.1
    a, b, _, _ = cs.stft(big)
    assert a.shape == (64, 401, 161) and max_abs(cs.istft(torch.stack([a, b], dim=1)), big) < 2e-5


# ---------------------------------------------------------------------------------------------------------------- a16
def test_mtfaa_stft_and_blocks_vs_reference(golden):
    from model import mtfaa as P
    from oracle import cruse_oracle_ext as X
    g = golden("g14_mtfaa.npz")
    sig = t(g["sig"])
    for wt in ("hann", "hamm"):
        c = P.STFT(320, 160, 320, wt).transform(sig)
        want = torch.from_numpy(g[f"stft_{wt}"])
        assert c.shape == want.shape == (2, 2, 161, 26) and rel_l2(c, want) < 1e-5
    inv = P.STFT(320, 160, 320, "hann").inverse(t(g["stft_hann"][:, 0]), t(g["stft_hann"][:, 1]))
    assert rel_l2(inv, torch.from_numpy(g["stft_inv"])) < 1e-5 and max_abs(inv, sig) < 1e-5
    # ComplexConv2d
    o = X.ComplexConv2d(8, 12, (3, 3), padding=(1, 2))
    p = _load_like(P.ComplexConv2d(8, 12, (3, 3), padding=(1, 2)), o)
    xc = torch.from_numpy(g["cconv/x"])
    assert rel_l2(p(xc.cuda()), torch.from_numpy(g["cconv/y"])) < 1e-5
    _grad_check(p, o, xc, tol=5e-4)
    # PhaseEncoder
    o = X.PhaseEncoder(4, 2)
    p = _load_like(P.PhaseEncoder(4, 2), o, scale=3.0)
    cs = [torch.from_numpy(g["pe/x0"]), torch.from_numpy(g["pe/x1"])]
    y = p([c.cuda() for c in cs])
    assert rel_l2(y, torch.from_numpy(g["pe/y"])) < 1e-5
    _grad_check(p, o, cs, tol=1e-3, list_input=True)
    # TFCM_Block (dilations 1 and 4), train and eval mode
    xt = torch.from_numpy(g["tfcm/x"])
    for dila in (1, 4):
        o = X.TFCM_Block(24, (3, 3), dila)
        p = _load_like(P.TFCM_Block(24, (3, 3), dila), o, scale=2.0)
        assert list(p.state_dict().keys()) == list(o.state_dict().keys())
        o.train(); p.train()
        assert rel_l2(p(xt.cuda()), torch.from_numpy(g[f"tfcm_d{dila}/y_train"])) < 2e-5
        p.eval()
        with torch.no_grad():
            assert rel_l2(p(xt.cuda()), torch.from_numpy(g[f"tfcm_d{dila}/y_eval"])) < 2e-5
        o2 = X.TFCM_Block(24, (3, 3), dila)
        p2 = _load_like(P.TFCM_Block(24, (3, 3), dila), o2, scale=2.0)
        o2.train(); p2.train()
        _grad_check(p2, o2, xt, tol=1e-3, wtol=5e-3)
    # the 6-block stack of config 5 on its stated shape [B,24,161,T]: runs, finite, residual structure
    stack = P.TFCM(24, (3, 3), 6).cuda()
    xin = torch.randn(2, 24, 161, 101).cuda()
    out = stack(xin)
    assert out.shape == xin.shape and torch.isfinite(out).all()


def test_f32_casts_pass_tiny_gradients_through_unchanged(golden):
    """to_f32 on an already-f32 tensor is a forward no-op and must be a backward no-op too -- the
    gradient used to be rounded through f16 (1e-9 -> 0, 1e6 -> inf).  mtfaa.py:123-138 (ComplexLinearProjection) in an
    f32 model with a mean-normalised loss has gradients of 1e-7 and below."""
    from cruse_amd.nn_generic import to_f16, to_f32
    x = torch.randn(4096, device="cuda", requires_grad=True)
    g = torch.tensor([1e-9, 1e6, 1.2345678, -3e-8], device="cuda").repeat(1024)
    to_f32(x).backward(g)
    assert torch.equal(x.grad, g)                                      # bit-exact: no rounding at all
    h = x.detach().half().requires_grad_(True)
    to_f16(h).backward(g.half())
    assert h.grad.dtype == torch.float16 and torch.equal(h.grad, g.half())
    # a real cast: f32 -> f16 forward, the f16 gradient comes back widened to f32 (exactly)
    x2 = x.detach().clone().requires_grad_(True)
    y = to_f16(x2)
    assert y.dtype == torch.float16
    y.backward(torch.full_like(y, 0.5))
    assert x2.grad.dtype == torch.float32 and torch.equal(x2.grad, torch.full_like(x2, 0.5))
    # through the module the advice names, with a loss scaled so that every gradient is ~1e-7
    from model import mtfaa as P
    from oracle import cruse_oracle_ext as X
    o = X.PhaseEncoder(4, 2)
    p = _load_like(P.PhaseEncoder(4, 2), o, scale=3.0)
    g14 = golden("g14_mtfaa.npz")
    cs = [torch.from_numpy(g14["pe/x0"]), torch.from_numpy(g14["pe/x1"])]
    scale = 1e-7
    (o([c.clone() for c in cs]).sum() * scale).backward()
    (p([c.cuda() for c in cs]).sum() * scale).backward()
    for (n, a), (_, b) in zip(p.named_parameters(), o.named_parameters()):
        assert a.grad is not None and float(b.grad.abs().max()) < 1e-3
        assert rel_l2(a.grad, b.grad) < 1e-3, n


F16_FWD_TOL = 2e-3        # f16-storage run vs the F32 oracle, forward (measured 4.8e-4 / 7.8e-4)
F16_EMU_TOL = 1e-3        # f16-storage run vs the oracle's f16-storage model (oracle_ext.emulate_f16_storage), forward (1.0e-4 / 4.5e-4)
F16_GRAD_EMU_TOL = 1.5e-2 # ... every parameter gradient as ONE vector (measured 2.6e-3 / 8.5e-3)
F16_GRAD_TOL = 4e-2       # f16-storage gradients vs the F32 oracle, as one vector (1.7e-2 / 1.9e-2: what fp16 storage costs this stack;
                          # the oracle's own f16 model is as far from the f32 oracle: 1.7e-2 / 1.9e-2)


@pytest.mark.parametrize("init", ["closed", "random"])
def test_mtfaa_config5_fp16_on_its_stated_shape(init):
    """BASELINE config 5 "MTFAA ... fp16" (VERDICT r2 missing 1): the fp16 part of the stack tools/mtfaa_stress.py runs (the
    PhaseEncoder front end, mtfaa.py:141-163, stays f32 and is covered by test_mtfaa_stft_and_blocks_vs_reference):
    6 x TFCM_Block (dilations 1..32, :196-209) on [B,24,161,401] with f16 ACTIVATION STORAGE -- pointwise convolutions
    on v_mfma_f32_16x16x32_f16, depthwise dilated convolutions / BatchNorm / PReLU as f16-in f16-out streams with f32
    arithmetic, f32 parameters and statistics.
    Two references: (i) the f32 CPU oracle: forward within F16_FWD_TOL; (ii) the oracle's own MODEL of f16 storage (f32 math,
    every stored tensor and gradient rounded to f16): forward and every parameter gradient within F16_EMU_TOL -- this is what
    pins the kernels.  The distance between (i) and (ii) in the gradients is a property of fp16 training of this stack, not of
    the build: BatchNorm's backward subtracts two projections of the incoming gradient, and rounding it to 11 bits afterwards
    leaves 5e-2 of error on the weight gradients of the closed-form-init fixture (same figure on the CPU model and on the GPU)."""
    from model import mtfaa as P
    from cruse_amd.nn_generic import to_f16, to_f32
    from oracle import cruse_oracle as O
    from oracle import cruse_oracle_ext as X
    torch.manual_seed(0)
    B, T = 2, 401
    o_tf = X.TFCM(24, (3, 3), 6)                                      # "random": torch-default init, seeded
    if init == "closed":
        p_tf = _load_like(P.TFCM(24, (3, 3), 6), o_tf, scale=1.0)
    else:
        p_tf = P.TFCM(24, (3, 3), 6)
        p_tf.load_state_dict(o_tf.state_dict())
        p_tf = p_tf.cuda()
    o_tf.train(); p_tf.train()
    # a full-rank input: tools/mtfaa_stress.py feeds 12 copies of the PhaseEncoder's 2 channels, which makes some BatchNorm
    # channels nearly constant (variance ~ 0, rstd ~ 300) and the problem ill-conditioned in ANY precision -- f32 storage then
    # already differs from the oracle by 1e-2 in the gradients; the kernels are the same
    hin = torch.randn(B, 24, 161, T)
    w = torch.randn(B, 24, 161, T)
    names = [n for n, _ in o_tf.named_parameters()]

    def run(tf, x, cast_in, cast_out, ww):
        for p in tf.parameters():
            p.grad = None
        y = cast_out(tf(cast_in(x)))
        (y * ww).sum().backward()
        return y.detach().cpu(), [p.grad.detach().cpu().clone() for p in tf.parameters()]

    torch.set_num_threads(min(16, torch.get_num_threads()))
    y_o, g_o = run(o_tf, hin, lambda a: a, lambda a: a, w)
    hooks = X.emulate_f16_storage(o_tf)
    y_e, g_e = run(o_tf, hin, X.round_f16, lambda a: a, w)
    for h_ in hooks:
        h_.remove()
    y_h, g_h = run(p_tf, hin.cuda(), to_f16, to_f32, w.cuda())
    y_f, g_f = run(p_tf, hin.cuda(), lambda a: a, lambda a: a, w.cuda())

    def gdist(ga, gb):
        num = den = 0.0
        worst = (0.0, "")
        for n, a_, b_ in zip(names, ga, gb):
            if n.endswith("pconv1.0.bias") or n.endswith("dila_conv.1.bias"):
                continue                                               # (a conv bias in front of a BatchNorm: exactly 0)
            num += float((a_.double() - b_.double()).norm() ** 2); den += float(b_.double().norm() ** 2)
            worst = max(worst, (rel_l2(a_, b_), n))
        return (num / den) ** 0.5, worst

    f_f32, f_f16, f_emu = rel_l2(y_f, y_o), rel_l2(y_h, y_o), rel_l2(y_h, y_e)
    (g32, w32), (g16, w16), (gem, wem), (gmodel, _) = gdist(g_f, g_o), gdist(g_h, g_o), gdist(g_h, g_e), gdist(g_e, g_o)
    print(f"[config 5 fp16 {init} init, [{B},24,161,{T}]] forward rel-L2: f32 storage vs oracle {f_f32:.2e}; f16 storage vs oracle "
          f"{f_f16:.2e}, vs the oracle's f16-storage model {f_emu:.2e}.  parameter gradients (all as one vector / worst tensor): "
          f"f32 storage vs oracle {g32:.2e} / {w32[0]:.2e}; f16 storage vs oracle {g16:.2e} / {w16[0]:.2e} ({w16[1]}); "
          f"f16 storage vs f16 model {gem:.2e} / {wem[0]:.2e} ({wem[1]}); f16 model vs oracle {gmodel:.2e}")
    assert f_f32 < 5e-5 and g32 < 1e-3
    assert f_f16 < F16_FWD_TOL and g16 < F16_GRAD_TOL
    assert f_emu < F16_EMU_TOL and gem < F16_GRAD_EMU_TOL
    if init == "closed":                     # (random init leaves some BatchNorm-gamma gradients near 0: per-tensor ratios are noise there)
        assert wem[0] < 5e-2, wem
    assert torch.isfinite(y_h).all()


@pytest.mark.parametrize("Cin,Cout,transposed", [(24, 24, False), (24, 24, True), (72, 40, False), (8, 64, True), (128, 16, False)])
def test_pointwise_conv_f16_mfma_kernel(Cin, Cout, transposed):
    """gconv_pointwise_mfma_f16_kernel against the same contraction of the f16-rounded operands in f64; both weight layouts,
    bias, a position count that is not a multiple of 16, PReLU, accumulate."""
    from cruse_amd.nn_generic import _conv_raw
    torch.manual_seed(Cin + Cout)
    B, H, W = 3, 7, 13
    x = torch.randn(B, Cin, H, W).cuda().half()
    w = (torch.randn(Cin, Cout, 1, 1) if transposed else torch.randn(Cout, Cin, 1, 1)).cuda()
    bias = torch.randn(Cout).cuda()
    slope = torch.rand(Cout).cuda()
    wm = w[:, :, 0, 0].t() if transposed else w[:, :, 0, 0]                                # [Cout, Cin]
    want = torch.einsum("oc,bchw->bohw", wm.half().double(), x.double()) + bias.double().view(1, -1, 1, 1)
    y = _conv_raw(x, w, bias, (H, W), 1, 1, (1, 1), (1, 1), 0, 0, 1, 1, transposed, Cout)
    assert y.dtype == torch.float16 and rel_l2(y.double(), want) < 6e-4
    yp = _conv_raw(x, w, bias, (H, W), 1, 1, (1, 1), (1, 1), 0, 0, 1, 1, transposed, Cout, act=2, slope=slope)
    wantp = torch.where(want >= 0, want, slope.double().view(1, -1, 1, 1) * want)
    assert rel_l2(yp.double(), wantp) < 6e-4
    base = torch.randn(B, Cout, H, W).cuda().half()
    acc = _conv_raw(x, w, None, (H, W), 1, 1, (1, 1), (1, 1), 0, 0, 1, 1, transposed, Cout, out=base.clone(), accumulate=True)
    assert rel_l2(acc.double(), base.double() + want - bias.double().view(1, -1, 1, 1)) < 1e-3


@pytest.mark.parametrize("Cin,Cout,transposed", [(24, 24, False), (24, 24, True), (40, 8, False), (8, 32, True)])
def test_pointwise_conv_f16_transposed_through_lds_equals_the_gather_kernel(Cin, Cout, transposed):
    """gconv_pointwise_tr_f16_kernel (operand rows copied to LDS, MFMA tile by transposing LDS reads, swapped roles) computes the same
    products in the same order as the gather kernel (library option pw_valu = 2): identical bits -- on a plane whose size is neither a
    multiple of 64 nor of 8 (tails of a chunk, of a 16-byte group), with bias and PReLU."""
    from cruse_amd import ops
    from cruse_amd.nn_generic import _conv_raw
    torch.manual_seed(3 * Cin + Cout)
    B, H, W = 3, 11, 29
    x = torch.randn(B, Cin, H, W).cuda().half()
    w = (torch.randn(Cin, Cout, 1, 1) if transposed else torch.randn(Cout, Cin, 1, 1)).cuda()
    bias, slope = torch.randn(Cout).cuda(), torch.rand(Cout).cuda()
    outs = {}
    for v in (0, 2):
        ops.set_option("pw_valu", v)
        try:
            outs[v] = (_conv_raw(x, w, bias, (H, W), 1, 1, (1, 1), (1, 1), 0, 0, 1, 1, transposed, Cout),
                       _conv_raw(x, w, None, (H, W), 1, 1, (1, 1), (1, 1), 0, 0, 1, 1, transposed, Cout, act=2, slope=slope))
        finally:
            ops.set_option("pw_valu", None)
    assert torch.equal(outs[0][0], outs[2][0]) and torch.equal(outs[0][1], outs[2][1])


@pytest.mark.parametrize("kind", ["pointwise", "depthwise", "general"])
def test_conv_ex_delivers_batch_sums_and_residual(kind):
    """cruse_conv2d_nchw_ex: the BatchNorm batch sums are those of the STORED output (what cruse_bn_nchw_stats reads back), summed over the
    replicas; y = conv + residual is the convolution followed by the add (two roundings).  Pointwise / depthwise: in the kernels' epilogues;
    a 3x3 full convolution: the fallback passes."""
    from cruse_amd.nn_generic import BN_STAT_REPLICAS, _conv_raw, _zeros_f64
    torch.manual_seed(11)
    B, C, H, W = 3, 24, 13, 37
    x = torch.randn(B, C, H, W).cuda().half()
    if kind == "pointwise":
        w, geo = torch.randn(C, C, 1, 1).cuda() * 0.3, dict(KH=1, KW=1, dil=(1, 1), pt=0, pl=0, groups=1)
    elif kind == "depthwise":
        w, geo = torch.randn(C, 1, 3, 3).cuda() * 0.3, dict(KH=3, KW=3, dil=(1, 2), pt=1, pl=4, groups=C)
    else:
        w, geo = torch.randn(C, C, 3, 3).cuda() * 0.1, dict(KH=3, KW=3, dil=(1, 1), pt=1, pl=1, groups=1)
    Wout = W + (4 if kind == "depthwise" else (2 if kind == "general" else 0)) - geo["dil"][1] * (geo["KW"] - 1)
    bias = torch.randn(C).cuda()

    def run(**kw):
        return _conv_raw(x, w, bias, (H, Wout), geo["KH"], geo["KW"], (1, 1), geo["dil"], geo["pt"], geo["pl"], geo["groups"], 1, False, C, **kw)
    y = run()
    sums = _zeros_f64((BN_STAT_REPLICAS, 2 * C), x.device)
    y2 = run(bn_sums=sums)
    assert torch.equal(y, y2)
    tot = sums.sum(0).cpu()
    yd = y.double().cpu()
    assert torch.allclose(tot[:C], yd.sum((0, 2, 3)), rtol=1e-5, atol=1e-3) and torch.allclose(tot[C:], (yd * yd).sum((0, 2, 3)), rtol=1e-5, atol=1e-3)
    res = torch.randn(B, C, H, Wout).cuda().half()
    y3 = run(residual=res)
    assert torch.equal(y3, (y.float() + res.float()).half())


def test_batchnorm_nchw_fused_train_forward_and_backward_ex():
    """cruse_bn_nchw_fwd_train == cruse_bn_finalize + cruse_counters_add + cruse_bn_nchw_fwd (output bits, mean / rstd, running statistics, batch
    counter; sums given as 1 or 8 replicas); cruse_bn_nchw_bwd_ex == cruse_bn_nchw_bwd (dx bits, parameter gradients) and its dx_sum is the
    channel sum of dx up to the f16 rounding of the stored values (closed form: DESIGN 7c)."""
    from cruse_amd import ops
    from cruse_amd._lib import check, lib
    from cruse_amd.ops import _p, _stream
    torch.manual_seed(5)
    N, C, H, W = 4, 24, 9, 53
    HW = H * W
    x = (torch.randn(N, C, H, W) * 1.5 + 0.7).cuda().half()
    dy = torch.randn(N, C, H, W).cuda().half()
    gamma, beta, slope = (1 + 0.3 * torch.randn(C)).cuda(), (0.2 * torch.randn(C)).cuda(), torch.rand(C).cuda()
    sums = torch.zeros(2 * C, dtype=torch.float64).cuda()
    check(lib.cruse_bn_nchw_stats(_p(x), N, C, HW, _p(sums), 1, _stream()))
    rm0, rv0 = torch.randn(C).cuda(), torch.rand(C).cuda() + 0.5
    # separate passes
    rm, rv, nbt = rm0.clone(), rv0.clone(), torch.tensor(3, dtype=torch.int64).cuda()
    mean, rstd = ops.bn_finalize(sums, N * HW, C, 1e-5, 0.1, rm, rv)
    ops.counters_add([nbt], 1)
    y = torch.empty_like(x)
    check(lib.cruse_bn_nchw_fwd(_p(x), _p(mean), _p(rstd), _p(gamma), _p(beta), _p(slope), 2, N, C, HW, _p(y), 1, _stream()))
    for nrep in (1, 8):
        s_in = sums.clone() if nrep == 1 else torch.cat([sums.view(1, -1) * f for f in (0.5, 0.25, 0.125, 0.125, 0, 0, 0, 0)]).contiguous()
        rm2, rv2, nbt2 = rm0.clone(), rv0.clone(), torch.tensor(3, dtype=torch.int64).cuda()
        y2, mean2, rstd2 = torch.empty_like(x), torch.empty(C).cuda(), torch.empty(C).cuda()
        check(lib.cruse_bn_nchw_fwd_train(_p(x), _p(s_in), nrep, 1e-5, 0.1, _p(gamma), _p(beta), _p(slope), 2, N, C, HW, _p(y2), _p(mean2), _p(rstd2),
                                          _p(rm2), _p(rv2), _p(nbt2), 1, _stream()))
        assert torch.equal(y, y2) and torch.equal(mean, mean2) and torch.equal(rstd, rstd2), nrep
        assert torch.allclose(rm, rm2, rtol=1e-6, atol=1e-7) and torch.allclose(rv, rv2, rtol=1e-6, atol=1e-7) and int(nbt2) == int(nbt) == 4
    # backward
    out = {}
    for ex in (False, True):
        dx = torch.empty_like(x)
        scratch = torch.zeros(4 * C, dtype=torch.float64).cuda()
        dg, db, ds, dxs = (torch.zeros(C).cuda() for _ in range(4))
        if ex:
            check(lib.cruse_bn_nchw_bwd_ex(_p(dy), _p(x), _p(mean), _p(rstd), _p(gamma), _p(beta), _p(slope), 2, 1, N, C, HW, _p(scratch), 1, 0, _p(dx),
                                           _p(dg), _p(db), _p(ds), _p(dxs), 1, _stream()))
        else:
            check(lib.cruse_bn_nchw_bwd(_p(dy), _p(x), _p(mean), _p(rstd), _p(gamma), _p(beta), _p(slope), 2, 1, N, C, HW, _p(scratch), _p(dx),
                                        _p(dg), _p(db), _p(ds), 1, _stream()))
        out[ex] = (dx, dg, db, ds, dxs)
    assert torch.equal(out[False][0], out[True][0])
    for i in (1, 2, 3):
        assert torch.allclose(out[False][i], out[True][i], rtol=1e-5, atol=1e-5), i
    dxd = out[True][0].double()
    # the stored dx are f16: their sum scatters around the exact one by ~2^-11 |dx| sqrt(n); the closed form is the exact one
    noise = dxd.abs().sum((0, 2, 3)) * 2.0 ** -11 / (N * HW) ** 0.5 * 8 + 1e-4
    assert ((out[True][4].double() - dxd.sum((0, 2, 3))).abs() <= noise).all()


def test_data_gradient_conv_delivers_the_batchnorm_backward_sums():
    """cruse_conv2d_nchw_bnbwd: the data gradient of a pointwise / depthwise convolution that follows BatchNorm (+PReLU) equals the plain convolution's,
    and the four sums it leaves ([nrep][4][C]: sum d, sum d xh, sum d z[z<0], sum xh) make cruse_bn_nchw_bwd_ex (sums_replicas = nrep) produce
    the dx bits and parameter gradients of the pass that reduces them itself."""
    import ctypes
    from cruse_amd import ops
    from cruse_amd._lib import check, lib
    from cruse_amd.nn_generic import _conv_raw
    from cruse_amd.ops import _p, _stream
    torch.manual_seed(11)
    N, C, H, W = 3, 24, 10, 77
    HW = H * W
    NREP = 8
    for kind in ("pointwise", "depthwise"):
        xb = (torch.randn(N, C, H, W) * 1.2 + 0.3).cuda().half()           # the BatchNorm's input
        gamma, beta, slope = (1 + 0.3 * torch.randn(C)).cuda(), (0.2 * torch.randn(C)).cuda(), torch.rand(C).cuda()
        sums = torch.zeros(2 * C, dtype=torch.float64).cuda()
        check(lib.cruse_bn_nchw_stats(_p(xb), N, C, HW, _p(sums), 1, _stream()))
        mean, rstd = ops.bn_finalize(sums, N * HW, C, 1e-5, 0.1, torch.zeros(C).cuda(), torch.ones(C).cuda())
        if kind == "pointwise":
            Co, KH, KW, dil, pt, pl, groups = 40, 1, 1, (1, 1), 0, 0, 1
            w = (torch.randn(Co, C, 1, 1) * 0.2).cuda()
            Ho, Wo = H, W
        else:
            Co, KH, KW, dil, pt, pl, groups = C, 3, 3, (2, 1), 4, 1, C       # causal along H (pad 4 on top), same along W
            w = (torch.randn(C, 1, 3, 3) * 0.3).cuda()
            Ho, Wo = H, W
        dyc = torch.randn(N, Co, Ho, Wo).cuda().half()                      # gradient of the convolution's output
        # reference: plain data-gradient convolution, then the BatchNorm backward that reduces its own sums
        d_ref = _conv_raw(dyc, w, None, (H, W), KH, KW, (1, 1), dil, pt, pl, groups, 1, True, C)
        r = torch.zeros(NREP, 4, C, dtype=torch.float64).cuda()
        d_new = torch.empty_like(d_ref)
        got = ctypes.c_int(0)
        check(lib.cruse_conv2d_nchw_bnbwd(_p(dyc), _p(w), _p(d_new), N, Co, Ho, Wo, C, H, W, KH, KW, 1, 1, dil[0], dil[1], pt, pl, groups, 1,
                                          _p(xb), _p(mean), _p(rstd), _p(gamma), _p(beta), _p(slope), 2, _p(r), NREP, ctypes.addressof(got), 1,
                                          _stream()))
        assert got.value == 1, kind
        assert torch.equal(d_ref, d_new), kind
        out = {}
        for delivered in (False, True):
            dx = torch.empty_like(xb)
            scratch = r if delivered else torch.zeros(4 * C, dtype=torch.float64).cuda()
            dg, db, ds, dxs = (torch.zeros(C).cuda() for _ in range(4))
            check(lib.cruse_bn_nchw_bwd_ex(_p(d_ref), _p(xb), _p(mean), _p(rstd), _p(gamma), _p(beta), _p(slope), 2, 1, N, C, HW, _p(scratch), 1,
                                           NREP if delivered else 0, _p(dx), _p(dg), _p(db), _p(ds), _p(dxs), 1, _stream()))
            out[delivered] = (dx, dg, db, ds, dxs, scratch)
        rs = r.sum(0)
        ref4 = out[False][5].view(4, C)
        assert torch.allclose(rs, ref4, rtol=1e-5, atol=1e-4), (kind, (rs - ref4).abs().max())
        # the sums agree to f32 rounding of the partials, so dx may differ in the last f16 place of a few elements
        ddx = (out[False][0].float() - out[True][0].float()).abs()
        assert ddx.max() <= 2e-3 * out[False][0].float().abs().max() and (ddx > 0).float().mean() < 0.02, (kind, ddx.max())
        for i in (1, 2, 3, 4):
            assert torch.allclose(out[False][i], out[True][i], rtol=1e-4, atol=1e-3), (kind, i)


def test_pointwise_wgrad_ex_delivers_the_bias_gradient():
    """cruse_conv2d_nchw_wgrad_ex: dw as cruse_conv2d_nchw_wgrad, db = channel sums of dy -- inside the pointwise MFMA kernel (24 channels: a spare
    tile column carries the constant 1) and by the channel-sum pass where that kernel does not run (32 channels: no spare column; 3x3)."""
    from cruse_amd.nn_generic import _wgrad_raw
    torch.manual_seed(9)
    for C, K in ((24, 1), (32, 1), (8, 3)):
        B, H, W = 3, 9, 41
        x = torch.randn(B, C, H, W).cuda().half()
        dy = torch.randn(B, C, H, W).cuda().half()
        dw0, dw1, db = torch.zeros(C, C, K, K).cuda(), torch.zeros(C, C, K, K).cuda(), torch.zeros(C).cuda()
        p = K // 2
        _wgrad_raw(dy, x, dw0, K, K, (1, 1), (1, 1), p, p, 1, 1)
        _wgrad_raw(dy, x, dw1, K, K, (1, 1), (1, 1), p, p, 1, 1, db=db)
        assert torch.allclose(dw0, dw1, rtol=1e-5, atol=1e-4), (C, K)
        assert torch.allclose(db.double(), dy.double().sum((0, 2, 3)), rtol=1e-5, atol=1e-3), (C, K)


def test_gemm_f16_operand_mode():
    """CRUSE_PREC_F16: Frag<> on v_mfma_f32_16x16x32_f16 (operands rounded to f16, f32 accumulate), all four transpose forms,
    against the same product of f16-ROUNDED operands in f64 (exact up to the f32 accumulation order)."""
    from cruse_amd import ops
    torch.manual_seed(2)
    M, N, K = 200, 136, 328
    A, Bm = torch.randn(M, K).cuda(), torch.randn(N, K).cuda()
    want = (A.half().double() @ Bm.half().double().t()).float()
    for ta, tb in ((False, True), (False, False), (True, False), (True, True)):
        a = A.t().contiguous() if ta else A
        b = Bm if tb else Bm.t().contiguous()
        c = torch.zeros(M, N).cuda()
        ops.gemm(ta, tb, M, N, K, a, 0, a.shape[1], b, 0, b.shape[1], c, 0, N, prec="f16")
        assert rel_l2(c, want) < 2e-6, (ta, tb, rel_l2(c, want))
    assert rel_l2(want, A @ Bm.t()) > 1e-4                                   # (the f16 rounding is what is being modelled)


# ---------------------------------------------------------------------------------------------------------------- 8f.3
def test_snr_mix_vs_reference(golden):
    from dataset.dataset import SynDataset
    from cruse_amd.data import snr_mix
    g = golden("g15_snr_mix.npz")
    clean, noise, snr = t(g["clean"]), t(g["noise"]), t(g["snr"])
    noisy, c, n = snr_mix(clean, noise, snr, return_parts=True)
    assert rel_l2(noisy, torch.from_numpy(g["noisy"])) < 2e-6
    assert rel_l2(c, torch.from_numpy(g["clean_n"])) < 2e-6 and rel_l2(n, torch.from_numpy(g["noise_s"])) < 2e-6
    one = SynDataset.snr_mix(clean[1], noise[1], float(snr[1]))                   # single clip, the reference's call shape
    assert rel_l2(one, torch.from_numpy(g["noisy"][1])) < 2e-6
    # full size: 64 clips x 4 s; the mixture has the requested SNR and a unit-peak clean part
    B, L = 64, 64000
    gen = torch.Generator(device="cuda").manual_seed(0)
    cl = torch.randn(B, L, device="cuda", generator=gen) * 0.05
    nz = torch.randn(B, L, device="cuda", generator=gen) * 0.3
    want = torch.linspace(-5, 25, B).cuda()
    ny, c, n = snr_mix(cl, nz, want, return_parts=True)
    got = 10 * torch.log10(c.square().mean(-1) / n.square().mean(-1))
    assert max_abs(got, want) < 1e-3 and max_abs(c.abs().amax(-1), torch.ones(B)) < 1e-5
    assert rel_l2(ny, c + n) < 1e-7


def test_snr_mix_with_room_impulse_responses_vs_reference(golden):
    """the reference's full signature -- snr_mix(clean, noise, snr, target_dB_FS, floating, rir, rir_noise)
    -- with the RIR convolutions (scipy.signal.fftconvolve(.)[:L], dataset.py:245-248) on the GPU as a direct FIR kernel.
    Fixture G20 = the reference's own function run with two RIRs per clip."""
    from dataset.dataset import SynDataset
    from cruse_amd.data import fir_causal, snr_mix
    g = golden("g20_snr_mix_rir.npz")
    clean, noise, snr = t(g["clean"]), t(g["noise"]), t(g["snr"])
    noisy, c, n = snr_mix(clean, noise, snr, -25, 10, rir=t(g["rir"]), rir_noise=t(g["rir_noise"]), return_parts=True)
    assert rel_l2(noisy, torch.from_numpy(g["noisy"])) < 5e-6
    assert rel_l2(c, torch.from_numpy(g["clean_n"])) < 5e-6 and rel_l2(n, torch.from_numpy(g["noise_s"])) < 5e-6
    one = SynDataset.snr_mix(clean[1], noise[1], float(snr[1]), -25, 10, t(g["rir"])[0])        # positional, as the reference is called
    assert rel_l2(one, torch.from_numpy(g["noisy_clean_rir_only"])) < 5e-6
    # full size: 64 clips x 4 s through a 0.5 s RIR (8000 taps) against torch's own FFT convolution
    B, L, R = 64, 64000, 8000
    gen = torch.Generator(device="cuda").manual_seed(0)
    x = torch.randn(B, L, device="cuda", generator=gen)
    h = torch.randn(R, device="cuda", generator=gen) * torch.exp(-torch.arange(R, device="cuda") / 1500.0)
    y = fir_causal(x, h)
    nfft = 1 << 17
    want = torch.fft.irfft(torch.fft.rfft(x.double(), nfft) * torch.fft.rfft(h.double(), nfft), nfft)[:, :L]
    assert rel_l2(y, want.float()) < 1e-5


# ---------------------------------------------------------------------------------------------------------------- config 4
@pytest.mark.parametrize("grp,prec", [(1, "f32"), (4, "f32"), (1, "bf16")])
def test_deepfilter_training_step_vs_golden(golden, grp, prec):
    """BASELINE config 4: unet_2 -> DeepFilter(1,5) head -> WO-MALE as ONE engine step (loss, filtered spectrum, gradients)."""
    from cruse_amd.engine import TrainEngine
    from cruse_amd.model import cruse_net as M
    from oracle import cruse_oracle as O
    g = golden(f"g16_df_step_g{grp}.npz")
    o = O.unet_2(rnn_groups=grp); O.closed_form_init(o)
    m = M.unet_2(rnn_groups=grp, precision=prec)
    m.load_state_dict(o.state_dict(), strict=True)
    eng = TrainEngine(m.cuda(), use_graph=False, loss="wo_male_df")
    ls = eng._fwd_bwd(t(g["noisy"]), t(g["clean"]))
    ftol = 1e-4 if prec == "f32" else 1e-3
    assert abs(eng.loss_value(ls) - float(g["loss"])) <= ftol * abs(float(g["loss"]))
    est = eng._last_est.permute(1, 0, 2, 3)                                        # [2,B,T,F] -> [B,2,T,F]
    e = rel_l2(est, torch.from_numpy(g["est"]))
    print(f"[config 4 g={grp} {prec}] filtered-spectrum rel-L2 {e:.3e}")
    assert e <= ftol
    # the closed-form fixture has constant channels whose pre-activations sit exactly on the BatchNorm mean: a last-bit
    # difference in the 33-tap sums flips individual ReLU decisions, which moves a gradient norm by up to ~1 % in f32
    gtol = 2e-2 if prec == "f32" else 0.3
    for name in eng.flat.names:
        if "gn/" + name not in g.files or (name.endswith(".bias") and name.startswith("conv") and name != "conv1_t.bias"):
            continue
        gn, got = float(g["gn/" + name]), float(eng.flat.G[name].norm())
        assert abs(got - gn) <= gtol * gn + 2e-6, (name, got, gn)
        if prec == "f32":
            g8 = torch.from_numpy(g["g8/" + name])
            # (the 33-tap neighbourhood sums of the head carry more f32 rounding than the mask-only step of fixture G6)
            assert max_abs(eng.flat.G[name].flatten()[:8], g8) <= 2e-2 * float(g8.abs().max()) + 1e-6 * max(gn, 1.0) + 2e-7, name
    # and as a graph-captured optimizer step at a larger shape
    eng2 = TrainEngine(M.unet_2(rnn_groups=grp, precision=prec).cuda(), use_graph=True, loss="wo_male_df")
    noisy, clean = O.synth_pair(4, 16000, seed=3)
    l0 = eng2.loss_value(eng2.step(noisy.cuda(), clean.cuda()))
    for _ in range(4):
        l1 = eng2.loss_value(eng2.step(noisy.cuda(), clean.cuda()))
    assert np.isfinite(l0) and l1 < l0


def test_synthetic_batch_matches_its_cpu_definition():
    """cruse_amd.data.synth_batch on the device (HIP one-pole FIR + mix) vs the same filter through torch on the CPU."""
    from cruse_amd.data import synth_batch
    noisy, clean = synth_batch(3, 5000, "cuda", 7)
    g = torch.Generator(device="cuda").manual_seed(7)
    white = 0.05 * torch.randn(3, 5000, device="cuda", generator=g)
    noise = 0.1 * torch.randn(3, 5000, device="cuda", generator=g)
    w = (1 - 0.95) * 0.95 ** torch.arange(63, -1, -1, dtype=torch.float32)
    ref = torch.nn.functional.conv1d(white.cpu().unsqueeze(1), w.view(1, 1, -1), padding=63)[..., :5000].squeeze(1) * 4.0
    assert rel_l2(clean, ref) < 1e-5 and rel_l2(noisy, ref + noise.cpu()) < 1e-5


# ---------------------------------------------------------------------------------------------------------------- a2-a4
@pytest.mark.parametrize("mode", ["mag_mapping", "complex_mapping", "mapping"])
def test_preprocess_as_written_vs_reference(golden, mode):
    """utils.utils.PreProcess (constant-padded STFT, the three masking modes, reconstruction) vs the reference's class."""
    from utils.utils import PreProcess
    g = golden("g17_preprocess.npz")
    pp = PreProcess(320, 160, 320, "hanning", mode, "freq")
    stft_in, real, imag, mags, phase = pp.pre_stft(t(g["wav"]))
    assert stft_in.shape == (2, 2, 21, 161) and real.shape == (2, 1, 21, 161)
    assert rel_l2(stft_in, torch.from_numpy(g["stft"])) < 1e-5 and rel_l2(mags, torch.from_numpy(g["mags"])) < 1e-5
    big = torch.from_numpy(g["mags"]) > 1e-2                                   # phase is ill-conditioned where |X| ~ 0
    assert max_abs(phase.cpu()[big], torch.from_numpy(g["phase"])[big]) < 1e-3
    ms = pp.masking(t(g["mask_real"]), t(g["mask_imag"]))
    assert ms.shape == (2, 21, 161, 2) and rel_l2(ms, torch.from_numpy(g[f"{mode}/masked"])) < 1e-5
    spec = torch.complex(real[:, 0], imag[:, 0]).transpose(1, 2)                # [B,F,T], what torch.istft takes
    rec = pp.reconstruction(spec, sig_len=3200)
    assert rel_l2(rec, torch.from_numpy(g["reconstruction"])) < 1e-5
    rec2 = pp.reconstruction(torch.stack([real[:, 0], imag[:, 0]], dim=-1), sig_len=3200)   # the [B,T,F,2] masking() returns
    assert rel_l2(rec2, rec) < 1e-7
    # differs from the hot path's reflect-padded STFT in frames 0 and T-1 only
    from cruse_amd.acoustics.feature import pre_stft
    refl = pre_stft(t(g["wav"]), 320, 160, 320)["real"]
    assert max_abs(refl[:, :, 1:-1], real[:, :, 1:-1]) < 1e-5 and max_abs(refl[:, :, 0], real[:, :, 0]) > 1e-3
    pp.log_transform()
    assert rel_l2(pp.spec_mags, torch.log(torch.from_numpy(g["mags"]))) < 1e-5


def test_sdnr_kernel_vs_reference_fixture(golden):
    """cruse_mask_sdnr_fwd against the values the reference's own sdnr (vad == 1 repair) produced (fixture G18)."""
    from cruse_amd import ops
    g = golden("g18_sdnr.npz")
    clean, noise, gain = t(g["clean"]), t(g["noise"]), t(g["gain"])
    B, _, T, Fs = clean.shape
    noisy = clean + noise
    for k, snr in enumerate(g["snr"]):
        ls, _, _ = ops.mask_sdnr(gain[:, 0, :, :160].contiguous().view(B * T, 160), clean[:, 0].contiguous().view(B * T, Fs),
                                 clean[:, 1].contiguous().view(B * T, Fs), noisy[:, 0].contiguous().view(B * T, Fs),
                                 noisy[:, 1].contiguous().view(B * T, Fs), B * T, 160, Fs, B, float(snr), 20.0)
        want = float(g["value"][k])
        assert abs(float(ls) / (B * Fs) - want) <= 2e-5 * abs(want), (snr, float(ls) / (B * Fs), want)


def test_bf16_mode_with_channels_beyond_the_mfma_kernels_and_an_input_gradient():
    """the bf16 storage of backward-only tensors (EngineConfig.bf16_dy / bf16_de) is taken only where every consumer is an MFMA
    kernel.  A bf16-mode unet_2 with 12- / 24-channel levels (not powers of two: VALU convs and weight gradients, f32 tensors) and the upsample model with
    x.requires_grad (level 1's data gradient into the one-channel input is a VALU conv) run their backward passes and agree with the
    f32 mode of the same weights."""
    from model.cruse import CRUSE4MagAddSkipUpsample
    from model.cruse_net import unet_2
    torch.manual_seed(7)
    wide = unet_2(ch=(1, 8, 12, 24, 64), rnn_groups=1, precision="bf16").cuda()
    ref = unet_2(ch=(1, 8, 12, 24, 64), rnn_groups=1, precision="f32").cuda()
    ref.load_state_dict(wide.state_dict())
    x = (torch.rand(2, 1, 21, 160) + 0.05).cuda(); w = torch.randn(2, 1, 21, 160).cuda()
    for m in (wide, ref):
        m.train()
        (m(x) * w).sum().backward()
    gb = torch.cat([p.grad.flatten() for p in wide.parameters() if p.grad is not None])
    gf = torch.cat([p.grad.flatten() for p in ref.parameters() if p.grad is not None])
    assert torch.isfinite(gb).all() and rel_l2(gb, gf) < 5e-2
    up = CRUSE4MagAddSkipUpsample(rnn_groups=1, precision="bf16").cuda()
    upf = CRUSE4MagAddSkipUpsample(rnn_groups=1, precision="f32").cuda()
    upf.load_state_dict(up.state_dict())
    grads = []
    for m in (up, upf):
        m.train()
        xi = x.clone().requires_grad_(True)
        (m(xi) * w).sum().backward()
        grads.append(xi.grad.clone())
    assert torch.isfinite(grads[0]).all() and rel_l2(grads[0], grads[1]) < 5e-2


@pytest.mark.gpu
def test_nchw_handoff_tables_hold_nothing_once_a_steps_tensors_are_gone():
    """the address-keyed hand-off tables of nn_generic (BatchNorm sums out of a convolution's epilogue and back) used to keep whole
    activations / gradients alive across steps whenever no HIP consumer popped an entry (the last block's BatchNorm output, a forward under
    no_grad in train mode).  Entries now die with their key tensor: after a forward + backward of the TFCM stack, a train-mode forward under
    no_grad and a bare Conv2d -> BatchNorm2d whose output nothing consumes, the tables are empty -- and the fused paths still deliver
    (the f16 result equals a second run bit for bit and stays inside the f16 tolerance of the oracle test above)."""
    import gc
    from cruse_amd import nn_generic as NG
    from cruse_amd.nn_generic import to_f16, to_f32
    from model import mtfaa as M
    torch.manual_seed(1)
    tf = M.TFCM(24, (3, 3), 2).cuda().train()
    x = torch.randn(2, 24, 33, 41, device="cuda")

    def run():
        for q in tf.parameters():
            q.grad = None
        y = to_f32(tf(to_f16(x)))
        y.square().mean().backward()
        return y.detach().clone(), [q.grad.clone() for q in tf.parameters()]
    y1, g1 = run()
    y2, g2 = run()
    assert torch.equal(y1, y2)
    with torch.no_grad():
        tf(to_f16(x))                                   # train-mode forward whose BatchNorm outputs no backward ever collects
    from model.based_model.cust_conv import Conv2dNormAct
    blk = Conv2dNormAct(24, 8, (1, 3), fstride=1).cuda().train()
    out = blk(x)                                        # the block's last BatchNorm output: no HIP conv consumes it
    del out, y1, y2, g1, g2
    gc.collect()
    torch.cuda.synchronize()
    assert NG.handoff_entries() == 0, (len(NG._BN_OUT), len(NG._BN_R), len(NG._BN_SUMS), len(NG._DX_SUMS))


@pytest.mark.gpu
def test_f16_conv_batchnorm_pair_with_many_planes():
    """a Conv2d -> BatchNorm2d pair in f16 training with B * C >= 65536 planes -- the fused BatchNorm kernels that fold the
    convolution's 8 replicated batch sums take N * C < 65536 only, and the call used to fail with CRUSE_E_SHAPE; the replicas are now folded
    before the generic kernels.  Forward and parameter gradients against torch f32 on the same weights."""
    import torch.nn as nn
    from cruse_amd.nn_generic import HipSequential, to_f16, to_f32
    torch.manual_seed(0)
    B, C, H, W = 300, 256, 2, 16                       # 76 800 planes
    seq = HipSequential(nn.Conv2d(C, C, (1, 1), bias=True), nn.BatchNorm2d(C), nn.ReLU()).cuda().train()
    ref = nn.Sequential(nn.Conv2d(C, C, (1, 1), bias=True), nn.BatchNorm2d(C), nn.ReLU()).train()
    ref.load_state_dict({k: v.detach().cpu() for k, v in seq.state_dict().items()})
    x = torch.randn(B, C, H, W)
    wgt = torch.randn(B, C, H, W)
    y = to_f32(seq(to_f16(x.cuda())))
    (y * wgt.cuda()).sum().backward()
    yr = ref(x)
    (yr * wgt).sum().backward()
    assert rel_l2(y, yr) < 3e-3
    for (n, p), (_, q) in zip(seq.named_parameters(), ref.named_parameters()):
        if n == "0.bias":
            continue                                    # feeds a BatchNorm: the true gradient is 0
        assert rel_l2(p.grad, q.grad) < 2e-2, n
