"""CPU: the oracle against the golden vectors produced by the reference's own code
(tests/golden/make_golden.py).  Bit-exact where the op sequence is identical."""
import numpy as np
import torch

from oracle import cruse_oracle as O


def _c(a):
    return torch.view_as_complex(torch.from_numpy(a).contiguous())


def test_stft_golden(golden):
    g = golden("g1_stft.npz")
    for L, T in ((3200, 21), (3199, 20), (3201, 21)):
        X = O.stft(torch.from_numpy(g[f"x_{L}"]), 320, 160, 320)
        assert X.shape == (2, 161, T)
        assert torch.equal(torch.view_as_real(X), torch.from_numpy(g[f"X_{L}"]))


def test_istft_golden(golden):
    g = golden("g7_istft.npz")
    X = _c(g["X"])
    y = O.istft(X, 320, 160, 320, length=3200)
    assert torch.equal(y, torch.from_numpy(g["y_rt"]))
    assert (y - torch.from_numpy(g["x"])).abs().max() < 1e-5
    ym = O.istft(X * torch.from_numpy(g["m"]), 320, 160, 320, length=3200)
    assert torch.equal(ym, torch.from_numpy(g["y_masked"]))


def test_ggru_golden(golden):
    g = golden("g3_ggru.npz")
    x = torch.from_numpy(g["x"])
    for grp in (1, 2, 4):
        m = O.GGRU(hidden_size=640, groups=grp)
        O.closed_form_init(m)
        assert torch.allclose(m(x), torch.from_numpy(g[f"y_g{grp}"]), atol=1e-6)


def test_unet2_golden(golden):
    g = golden("g4_unet2.npz")
    x = torch.from_numpy(g["x"])
    for grp in (1, 4):
        m = O.unet_2(rnn_groups=grp)
        O.closed_form_init(m)
        m.train()
        y = m(x)
        assert y.shape == (2, 1, 21, 160)
        assert torch.allclose(y, torch.from_numpy(g[f"mask_train_g{grp}"]), atol=1e-6)
        assert torch.allclose(m.bn1.running_mean, torch.from_numpy(g[f"bn1_running_mean_g{grp}"]), atol=1e-7)
        m.eval()
        assert torch.allclose(m(x), torch.from_numpy(g[f"mask_eval_g{grp}"]), atol=1e-6)


def test_state_dict_keys():
    m = O.unet_2(rnn_groups=2)
    keys = set(m.state_dict().keys())
    for k in ("conv1.weight", "conv4_t.bias", "bn3.running_var", "bn2_t.num_batches_tracked",
              "skip_connect_4.weight", "gru.gru_list1.1.weight_ih_l0", "gru.gru_list2.0.bias_hh_l0",
              "gru.ln1.weight", "gru.ln2.bias", "fc.weight"):
        assert k in keys, k
    assert m.conv4_t.weight.shape == (64, 32, 1, 3) and m.conv1.weight.shape == (8, 1, 2, 3)
    n = sum(p.numel() for n_, p in m.named_parameters() if not n_.startswith("fc."))
    assert sum(p.numel() for n_, p in O.unet_2(rnn_groups=1).named_parameters() if not n_.startswith("fc.")) == 4966555
    assert sum(p.numel() for n_, p in O.unet_2(rnn_groups=4).named_parameters() if not n_.startswith("fc.")) == 1280155
    assert n > 0


def test_losses_golden(golden):
    g = golden("g5_loss.npz")
    w = O.wo_male(torch.from_numpy(g["ref"]), torch.from_numpy(g["est"]), torch.from_numpy(g["unproc"]))
    assert torch.equal(w, torch.from_numpy(g["wo_male"]))
    s1, s2 = torch.from_numpy(g["s1"]), torch.from_numpy(g["s2"])
    assert torch.equal(O.sisnr(s1, s2), torch.from_numpy(g["sisnr"]))
    assert torch.equal(O.si_snr_loss(s1, s2), torch.from_numpy(g["si_snr_loss"]))
    try:
        O.wo_male(torch.zeros(1, 2, 3, 4), torch.zeros(1, 2, 3, 5), torch.zeros(1, 2, 3, 4))
        assert False
    except RuntimeError:
        pass


def test_train_step_golden(golden):
    for grp in (1, 4):
        g = golden(f"g6_step_g{grp}.npz")
        m = O.unet_2(rnn_groups=grp)
        O.closed_form_init(m)
        m.train()
        noisy, clean = O.synth_pair(2, 3200, seed=1234)
        assert torch.equal(noisy, torch.from_numpy(g["noisy"]))
        loss, aux = O.train_step_loss(m, noisy, clean)
        loss.backward()
        assert abs(float(loss.detach()) - float(g["loss"])) < 1e-6
        assert aux["est"].shape == (2, 21, 161, 2) and float(aux["est"][..., 160, :].abs().max()) == 0.0
        for name, p in m.named_parameters():
            if p.grad is None:
                assert name.startswith("fc.") or name.startswith("bn1_t."), name
                continue
            gn = float(g["gn/" + name])
            # conv biases feeding a BatchNorm have a mathematically zero gradient: pure rounding noise (~1e-8)
            assert abs(float(p.grad.norm()) - gn) <= 1e-4 * gn + 1e-6, name


def test_deepfilter_and_mask_golden(golden):
    g = golden("g8_deepfilter.npz")
    y = O.DeepFilter(1, 5)([torch.from_numpy(g["xr"]), torch.from_numpy(g["xi"])],
                           [torch.from_numpy(g["hr"]), torch.from_numpy(g["hi"])])
    assert y.shape == (2, 32, 21) and torch.allclose(y, torch.from_numpy(g["y"]), atol=1e-6)
    m = golden("g9_mask.npz")       # train_base/acoustics/mask.py complex_mul, run by the reference
    a, b, c, d = (torch.from_numpy(m[k]) for k in "abcd")
    assert torch.equal(a * c - b * d, torch.from_numpy(m["cm_r"])) and torch.equal(a * d + b * c, torch.from_numpy(m["cm_i"]))


def test_interleave_is_not_groupgru_shuffle():
    """GGRU's stack+flatten is new[j*g+i] = cat[i*h+j]."""
    g, h = 4, 5
    cat = torch.arange(g * h)
    new = torch.stack(torch.chunk(cat, g), dim=-1).flatten()
    for i in range(g):
        for j in range(h):
            assert new[j * g + i] == cat[i * h + j]


# ---------------------------------------------------------------------------------------------------------------------
# round 2: oracle part 2 against the fixtures generated from the reference's own code (tests/golden/make_golden_r2.py)
# ---------------------------------------------------------------------------------------------------------------------
def _t(a):
    return torch.from_numpy(np.asarray(a))


def test_oracle_ext_losses_and_snr_mix(golden):
    from oracle import cruse_oracle_ext as X
    g = golden("g10_losses.npz")
    ref, est = _t(g["ref"]), _t(g["est"])
    assert torch.equal(X.rmse(ref, est), _t(g["rmse"])) and torch.equal(X.c_rmse(ref, est), _t(g["c_rmse"]))
    s = golden("g15_snr_mix.npz")
    for b in range(3):
        noisy, c, n = X.snr_mix(s["clean"][b].copy(), s["noise"][b].copy(), float(s["snr"][b]))
        assert np.array_equal(noisy, s["noisy"][b]) and np.array_equal(c, s["clean_n"][b]) and np.array_equal(n, s["noise_s"][b])


def test_oracle_ext_conv_blocks_and_group_gru(golden):
    from oracle import cruse_oracle as O
    from oracle import cruse_oracle_ext as X
    g = golden("g11_convblocks.npz")
    cases = {"cna": lambda: X.Conv2dNormAct(1, 16, (2, 3), fstride=2),
             "ctna_sep": lambda: X.ConvTranspose2dNormAct(16, 8, (1, 3), fstride=2, separable=True),
             "kxf_upsample": lambda: X.convkxf(16, 8, k=2, f=3, fstride=2, mode="upsample", batch_norm=True)}
    for name, build in cases.items():
        m = build(); O.closed_form_init(m); m.train()
        assert torch.equal(m(_t(g[f"{name}/x"])), _t(g[f"{name}/y_train"])), name
        m.eval()
        assert torch.equal(m(_t(g[f"{name}/x"])), _t(g[f"{name}/y_eval"])), name
    gg = golden("g12_groupgru.npz")
    m = X.GroupGRU(128, 128, num_layers=3, groups=4, add_outputs=True)
    O.closed_form_init(m, scale=2.0)
    y, s = m(_t(gg["x"]))
    assert torch.equal(y, _t(gg["g4_l3_add/y"])) and torch.equal(s, _t(gg["g4_l3_add/state"]))


def test_oracle_ext_stft_variants_and_mtfaa(golden):
    from oracle import cruse_oracle as O
    from oracle import cruse_oracle_ext as X
    g = golden("g13_stft_variants.npz")
    wav = _t(g["wav"])
    K = X.init_stft_kernel(320, 160)
    m, p, r, i = X.custom_stft(wav, K, 160)
    assert torch.equal(r, _t(g["custom512/r"])) and torch.equal(m, _t(g["custom512/m"]))
    assert torch.equal(X.custom_istft(_t(g["custom512/m"]), _t(g["custom512/p"]), K, 160), _t(g["custom512/y"]))
    cs = X.ConvSTFT(320, 160)
    sr, si, mag, _ = cs.stft(wav)
    assert torch.equal(sr, _t(g["conv/spec_r"])) and torch.equal(si, _t(g["conv/spec_i"]))
    assert float((cs.istft(torch.stack([sr, si], 1)) - wav).abs().max()) < 2e-6           # the (decided) inverse inverts
    a = golden("g14_mtfaa.npz")
    assert torch.equal(X.mtfaa_stft_transform(_t(a["sig"]), 320, 160, 320, "hamm"), _t(a["stft_hamm"]))
    blk = X.TFCM_Block(24, (3, 3), 4); O.closed_form_init(blk, 2.0); blk.train()
    assert torch.equal(blk(_t(a["tfcm/x"])), _t(a["tfcm_d4/y_train"]))


def test_oracle_ext_deepfilter_step(golden):
    from oracle import cruse_oracle as O
    from oracle import cruse_oracle_ext as X
    g = golden("g16_df_step_g1.npz")
    m = O.unet_2(rnn_groups=1); O.closed_form_init(m); m.train()
    loss, aux = X.train_step_loss_df(m, _t(g["noisy"]), _t(g["clean"]))
    assert torch.equal(loss, _t(g["loss"])) and torch.equal(aux["est"], _t(g["est"]))
    assert float(aux["est"][:, :, :, 160].abs().max()) > 0.0        # the 11-bin neighbourhood reaches the Nyquist bin


def test_oracle_ext_preprocess(golden):
    from oracle import cruse_oracle_ext as X
    g = golden("g17_preprocess.npz")
    pp = X.PreProcess(320, 160, 320, "hanning", "complex_mapping", "freq")
    out = pp.pre_stft(_t(g["wav"]))
    assert torch.equal(out[0], _t(g["stft"])) and torch.equal(out[3], _t(g["mags"]))
    assert torch.equal(pp.masking(_t(g["mask_real"]), _t(g["mask_imag"])), _t(g["complex_mapping/masked"]))


def test_oracle_sdnr_vs_reference_fixture(golden):
    """sdnr (loss_func/loss.py:151-175) was parity-unpinned in round 1: G18 holds the reference function's own values (vad == 1)."""
    g = golden("g18_sdnr.npz")
    for k, snr in enumerate(g["snr"]):
        v = O.sdnr(_t(g["clean"]), _t(g["gain"]), _t(g["noise"]), float(snr), beta=20.0)
        assert abs(float(v) - float(g["value"][k])) <= 1e-6 * abs(float(g["value"][k])), snr


def test_oracle_round3_fixtures(golden):
    """G19 (GroupGRU with a non-zero state) and G20 (snr_mix with RIRs): the oracle restatements against the reference's
    own outputs (tests/golden/make_golden_r3.py)."""
    from oracle import cruse_oracle as O
    from oracle import cruse_oracle_ext as X
    g = golden("g19_groupgru_state.npz")
    x = torch.from_numpy(g["x"])
    for name, kw in {"g2_l2": dict(num_layers=2, groups=2), "g4_l3_add": dict(num_layers=3, groups=4, add_outputs=True),
                     "g1_l1": dict(num_layers=1, groups=1)}.items():
        m = X.GroupGRU(128, 128, **kw)
        O.closed_form_init(m, scale=2.0)
        y, s = m(x, torch.from_numpy(g[f"{name}/state_in"]))
        assert torch.equal(y, torch.from_numpy(g[f"{name}/y"])) and torch.equal(s, torch.from_numpy(g[f"{name}/state"])), name
    # G22: GroupedGRULayer(bidirectional=True, dropout=0.3) as shipped
    g = golden("g22_grouped_gru_bidirectional.npz")
    x = torch.from_numpy(g["x"])
    for name, grp in {"g2": 2, "g1": 1}.items():
        m = X.GroupedGRULayer(128, 128, grp, dropout=0.3, bidirectional=True).train()
        O.closed_form_init(m, scale=2.0)
        y, st = m(x, torch.from_numpy(g[f"{name}/state_in"]))
        assert torch.equal(y, torch.from_numpy(g[f"{name}/y"])) and torch.equal(st, torch.from_numpy(g[f"{name}/state"])), name
        y0, st0 = m(x)
        assert torch.equal(y0, torch.from_numpy(g[f"{name}/y_zero_state"])) and torch.equal(st0, torch.from_numpy(g[f"{name}/state_zero_state"]))
    s = golden("g20_snr_mix_rir.npz")
    for b in range(3):
        noisy, c, n = X.snr_mix(s["clean"][b].copy(), s["noise"][b].copy(), float(s["snr"][b]), rir=s["rir"][b].copy(),
                                rir_noise=s["rir_noise"][b].copy())
        assert np.allclose(noisy, s["noisy"][b], rtol=0, atol=1e-6) and np.allclose(c, s["clean_n"][b], rtol=0, atol=1e-6)
    # G21: the CRUSE4MagAddSkipUpsample composition on the oracle's own restated blocks == on the reference's blocks
    g = golden("g21_cruse_upsample.npz")
    for grp in (1, 4):
        m = X.CRUSE4MagAddSkipUpsample(rnn_groups=grp)
        O.closed_form_init(m)
        m.train()
        assert torch.equal(m(torch.from_numpy(g["x"])), torch.from_numpy(g[f"g{grp}/mask_train"]))


def test_oracle_round4_fixtures(golden):
    """G23: GroupedGRULayer(batch_first=False) / (bias=False) as shipped (tests/golden/make_golden_r4.py); the waveform L1 / MSE
    step of the oracle is torch's own criterion on iSTFT(mask * N) (a decision, stated in oracle.train_step_loss)."""
    from oracle import cruse_oracle as O
    from oracle import cruse_oracle_ext as X
    g = golden("g23_grouped_gru_layouts.npz")
    x_tb = torch.from_numpy(g["x_tb"])
    for name, kw in {"tb_g2": dict(batch_first=False), "tb_g2_nobias": dict(batch_first=False, bias=False),
                     "bt_g2_nobias": dict(bias=False)}.items():
        m = X.GroupedGRULayer(128, 128, 2, **kw)
        O.closed_form_init(m, scale=2.0)
        x = x_tb if not kw.get("batch_first", True) else x_tb.transpose(0, 1).contiguous()
        y, s = m(x, torch.from_numpy(g[f"{name}/state_in"]))
        assert torch.equal(y, torch.from_numpy(g[f"{name}/y"])) and torch.equal(s, torch.from_numpy(g[f"{name}/state"])), name
        assert torch.equal(m(x)[0], torch.from_numpy(g[f"{name}/y0"]))
    o = O.unet_2(rnn_groups=1)
    O.closed_form_init(o)
    o.train()
    noisy, clean = O.synth_pair(2, 3200, seed=92)
    for mode, crit in (("L1", torch.nn.L1Loss()), ("MSE", torch.nn.MSELoss())):
        loss, aux = O.train_step_loss(o, noisy, clean, loss_mode=mode)
        assert aux["wave"].shape == clean.shape and torch.equal(loss, crit(aux["wave"], clean))
