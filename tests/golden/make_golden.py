#!/usr/bin/env python3
"""Generate tests/golden/*.npz by RUNNING THE REFERENCE'S OWN CODE in this container.

Usage (build container only; /root/reference does not exist on the GPU box):
    python tests/golden/make_golden.py [--ref /root/reference]

The reference does not import as shipped, so each piece is
lifted from its source file at run time, with the documented one-token repairs
applied as textual substitutions (listed in REPAIRS below, every one asserted to
hit exactly once), exec'd, and run on seeded inputs.  Only inputs and outputs are
stored -- no reference source text is written anywhere.

Weights are closed-form (oracle.cruse_oracle.closed_form_init) and are loaded
into the reference modules through load_state_dict, which also pins the
state-dict key names.
"""
from __future__ import annotations

import argparse
import os
import re
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
sys.path.insert(0, ROOT)

from oracle import cruse_oracle as O  # noqa: E402


def read_src(ref, rel):
    with open(os.path.join(ref, rel), "rb") as f:
        return f.read().decode("latin-1")


def top_level_block(src: str, header: str) -> str:
    """Text of the top-level `def`/`class` starting with `header` up to the next top-level statement."""
    lines = src.splitlines()
    start = next(i for i, l in enumerate(lines) if l.startswith(header))
    end = len(lines)
    for i in range(start + 1, len(lines)):
        l = lines[i]
        if l and not l[0].isspace() and not l.startswith("#"):
            end = i
            break
    return "\n".join(lines[start:end]) + "\n"


def sub_once(text: str, old: str, new: str) -> str:
    assert text.count(old) == 1, f"repair target not unique/present: {old!r} x{text.count(old)}"
    return text.replace(old, new)


# (old, new, tag) -- SURVEY.md 8(a) rows a7/a8/a11
REPAIRS_GGRU = [
    ("out = self.view(", "out = out.view(", "R1 cruse_net.py:53"),
]
REPAIRS_UNET2 = [
    ('setattr(self,"conv{}".format(tmp), nn.Conv2d(ch[tmp-1], ch[i+1],(1,3), self.stride))',
     'setattr(self,"conv{}_t".format(i+1), nn.ConvTranspose2d(ch[i+1], ch[i],(1,3), self.stride))', "R2 :140"),
    ('setattr(self,"bn{}_t".format(i+1), nn.BatchNorm2d(ch[tmp-1]))',
     'setattr(self,"bn{}_t".format(i+1), nn.BatchNorm2d(ch[i]))', "R3 :142"),
    ("nn.Conv2d(ch[i+1], ch[i+1],(1,3),bias=False)",
     "nn.Conv2d(ch[i+1], ch[i+1],(1,3),padding=(0,1),bias=False)", "R5 :143"),
    ("self.gru = GGRU(groups=rnn_groups)",
     "self.gru = GGRU(hidden_size=hidden_size, groups=rnn_groups)", "R7 :144"),
    ("e3 = self.elu(self.bn2(self.conv2(e2)", "e3 = self.elu(self.bn3(self.conv3(e2)", "R4 :151"),
    ("e4 = self.elu(self.bn2(self.conv2(e3)", "e4 = self.elu(self.bn4(self.conv4(e3)", "R4 :152"),
    ("skip2 = self.skip_connect_1(e2)", "skip2 = self.skip_connect_2(e2)", "R5 :154"),
    ("ski3 = self.skip_connect_1(e3)", "skip3 = self.skip_connect_3(e3)", "R5 :155"),
    ("skip4 = self.skip_connect_1(e4)", "skip4 = self.skip_connect_4(e4)", "R5 :156"),
    ("self.conv3_t(out)", "self.conv3_t(d4_1)", "R6 :162"),
    ("self.conv2_t(out)", "self.conv2_t(d3_1)", "R6 :163"),
    ("self.conv1_t(out)", "self.conv1_t(d2_1)", "R6 :164"),
]
REPAIRS_WOMALE = [
    ("B, C, T, F = torch.size(ref)", "B, C, T, F = ref.size()", "a11 loss.py:129"),
    ("unproc[:, 1, :, 1]**2", "unproc[:, 1, :, :]**2", "a11 loss.py:139"),
]


def load_reference(ref):
    ns = {}
    exec("import torch\nimport torch.nn as nn\nimport torch.nn.functional as F\nimport numpy as np\n", ns)
    # --- feature.stft / istft (run as shipped) -------------------------------
    fsrc = read_src(ref, "train_base/acoustics/feature.py")
    exec(top_level_block(fsrc, "def stft("), ns)
    exec(top_level_block(fsrc, "def istft("), ns)
    # --- cruse_net.GGRU (R1) and unet_2 (R2-R8) ------------------------------
    msrc = read_src(ref, "model/cruse_net.py")
    g = top_level_block(msrc, "class GGRU(")
    for old, new, _ in REPAIRS_GGRU:
        g = sub_once(g, old, new)
    exec(g, ns)
    u = top_level_block(msrc, "class unet_2(")
    n_r4 = u.count("[...,-self.padding[0],:]")
    assert n_r4 == 4, n_r4
    u = u.replace("[...,-self.padding[0],:]", "[...,:-self.padding[0],:]")  # R4 :149-152
    for old, new, _ in REPAIRS_UNET2:
        u = sub_once(u, old, new)
    exec(u, ns)
    # --- loss_func/loss.py: wo_male (a11 repairs), sisnr (as shipped) ---------
    lsrc = read_src(ref, "loss_func/loss.py")
    w = top_level_block(lsrc, "def wo_male(")
    for old, new, _ in REPAIRS_WOMALE:
        w = sub_once(w, old, new)
    exec(top_level_block(lsrc, "def l2_norm("), ns)
    exec(top_level_block(lsrc, "def sisnr("), ns)
    exec(w, ns)
    # --- model/deep_filter.py: ctor repair + the imaginary-part decision ----------------
    dsrc = read_src(ref, "model/deep_filter.py")
    d = top_level_block(dsrc, "class DeepFilter(")
    d = sub_once(d, "torch.reshape(kernel, [t_width(f_width, 1, f_width, t_width)])",
                 "torch.reshape(kernel, [t_width * f_width, 1, f_width, t_width])")
    d = sub_once(d, "output_i = inputs_r * filters_i + inputs_r * filters_i",
                 "output_i = inputs_r * filters_i + inputs_i * filters_r")
    exec(d, ns)
    # --- importable modules ---------------------------------------------------
    # (this repo ships drop-in packages with the SAME top-level names `model` / `train_base`: make sure the
    #  reference's own packages are the ones imported here, then restore the module table)
    import importlib
    tops = ("model", "train_base")
    saved = {k: sys.modules.pop(k) for k in list(sys.modules) if k.split(".")[0] in tops}
    old_path = list(sys.path)
    # the reference's `model/` has no __init__.py (namespace package) and would lose against this repo's regular
    # package anywhere on the path: take this repo (and the cwd) off the path for these imports
    sys.path[:] = [ref] + [q for q in old_path if os.path.abspath(q or os.getcwd()) != ROOT]
    try:
        ns["cust_conv"] = importlib.import_module("model.based_model.cust_conv")
        ns["mask_mod"] = importlib.import_module("train_base.acoustics.mask")
        ns["tb_loss"] = importlib.import_module("train_base.loss")
        assert ns["cust_conv"].__file__.startswith(ref)
    finally:
        sys.path[:] = old_path
        for k in [k for k in sys.modules if k.split(".")[0] in tops]:
            del sys.modules[k]
        sys.modules.update(saved)
    return ns


def npz(name, **arrs):
    out = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach()
            if v.is_complex():
                v = torch.view_as_real(v)
            v = v.cpu().numpy()
        out[k] = np.asarray(v)
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **out)
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} KiB  {len(out)} arrays")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    args = ap.parse_args()
    torch.manual_seed(0)
    torch.set_num_threads(4)
    R = load_reference(args.ref)
    n_fft, hop, win = 320, 160, 320

    # ---- G1: stft (feature.py:10-30), frame-count edges ----------------------
    g = torch.Generator().manual_seed(0)
    arrs = {}
    for L in (3200, 3199, 3201):
        x = 0.1 * torch.randn(2, L, generator=g)
        X = R["stft"](x, n_fft, hop, win)
        arrs[f"x_{L}"] = x
        arrs[f"X_{L}"] = X            # [B,F,T,2]
        assert torch.allclose(torch.view_as_real(O.stft(x, n_fft, hop, win)), torch.view_as_real(X), atol=0, rtol=0)
    npz("g1_stft.npz", **arrs)

    # ---- G7: istft round trip and istft(mask * X) (feature.py:33-61) ---------
    x = arrs["x_3200"]
    X = R["stft"](x, n_fft, hop, win)
    m = torch.rand(X.shape, generator=g)
    npz("g7_istft.npz", x=x, X=X, m=m,
        y_rt=R["istft"](X, n_fft, hop, win, length=x.shape[1]),
        y_masked=R["istft"](X * m, n_fft, hop, win, length=x.shape[1]),
        y_nolen=R["istft"](X, n_fft, hop, win))

    # ---- G3: GGRU g in {1,2,4} (cruse_net.py:14-55, R1) -----------------------
    arrs = {}
    xg = torch.randn(2, 64, 21, 10, generator=g)
    arrs["x"] = xg
    for grp in (1, 2, 4):
        mine = O.GGRU(hidden_size=640, groups=grp)
        O.closed_form_init(mine)
        theirs = R["GGRU"](hidden_size=640, groups=grp)
        theirs.load_state_dict(mine.state_dict(), strict=True)
        y = theirs(xg)
        arrs[f"y_g{grp}"] = y
        assert torch.equal(mine(xg), y), f"oracle GGRU != patched reference GGRU (g={grp})"
        # cust_conv.GroupedGRULayer (cust_conv.py:250-325) = first layer before the interleave
        lay = R["cust_conv"].GroupedGRULayer(640, 640, grp)
        for i in range(grp):
            lay.layers[i].load_state_dict(mine.gru_list1[i].state_dict())
        seq = xg.transpose(1, 2).reshape(2, 21, 640)
        o_cat, _ = lay(seq, lay.get_h0(2))
        arrs[f"l1cat_g{grp}"] = o_cat
    npz("g3_ggru.npz", **arrs)

    # ---- G4: unet_2 mask, train-mode and eval-mode BN (cruse_net.py:129-165) --
    arrs = {}
    xin = torch.rand(2, 1, 21, 160, generator=g) * 0.5
    arrs["x"] = xin
    for grp in (1, 4):
        mine = O.unet_2(rnn_groups=grp)
        O.closed_form_init(mine)
        theirs = R["unet_2"](rnn_groups=grp)
        theirs.load_state_dict(mine.state_dict(), strict=True)
        theirs.train(); mine.train()
        y_tr = theirs(xin)
        assert torch.equal(mine(xin), y_tr), "oracle unet_2 != patched reference (train)"
        arrs[f"mask_train_g{grp}"] = y_tr
        arrs[f"bn1_running_mean_g{grp}"] = theirs.bn1.running_mean.clone()
        arrs[f"bn4_running_var_g{grp}"] = theirs.bn4.running_var.clone()
        theirs.eval(); mine.eval()
        y_ev = theirs(xin)
        assert torch.equal(mine(xin), y_ev), "oracle unet_2 != patched reference (eval)"
        arrs[f"mask_eval_g{grp}"] = y_ev
    npz("g4_unet2.npz", **arrs)

    # ---- G2: causal top-pad == pad-both-then-crop-last (cust_conv.py:38-55 vs R4)
    blk = R["cust_conv"].Conv2dNormAct(1, 8, (2, 3), fstride=2, norm_layer=None, activation_layer=None)
    mine = O.unet_2(rnn_groups=1)
    O.closed_form_init(mine)
    blk[1].load_state_dict(mine.conv1.state_dict())
    xc = torch.randn(2, 1, 9, 160, generator=g)
    y_blk = blk(xc)
    y_mine = mine.conv1(xc)[..., :-1, :]
    assert torch.allclose(y_blk, y_mine, atol=1e-6), "R4 causal-crop equivalence failed"
    npz("g2_conv.npz", x=xc, y=y_blk, w=mine.conv1.weight, b=mine.conv1.bias)

    # ---- G5: losses ------------------------------------------------------------
    ref = torch.randn(2, 2, 21, 161, generator=g) * 0.3
    est = torch.randn(2, 2, 21, 161, generator=g) * 0.3
    unp = torch.randn(2, 2, 21, 161, generator=g) * 0.4
    wm = R["wo_male"](ref, est, unp)
    assert torch.equal(O.wo_male(ref, est, unp), wm)
    s1 = torch.randn(3, 3200, generator=g)
    s2 = s1 * 0.7 + 0.2 * torch.randn(3, 3200, generator=g)
    npz("g5_loss.npz", ref=ref, est=est, unproc=unp, wo_male=wm, s1=s1, s2=s2,
        sisnr=R["sisnr"](s1, s2), si_snr_loss=R["tb_loss"].si_snr_loss()(s1, s2))
    assert torch.equal(O.sisnr(s1, s2), R["sisnr"](s1, s2))
    assert torch.equal(O.si_snr_loss(s1, s2), R["tb_loss"].si_snr_loss()(s1, s2))

    # ---- G9: mask.py ------------------------------------------------------------
    mm = R["mask_mod"]
    a = torch.randn(2, 5, 7, generator=g); b = torch.randn(2, 5, 7, generator=g)
    c = torch.randn(2, 5, 7, generator=g); d = torch.randn(2, 5, 7, generator=g)
    cm_r, cm_i = mm.complex_mul(a, b, c, d)
    irm = mm.build_ideal_ratio_mask(a.abs(), c.abs())
    cirm = mm.build_complex_ideal_ratio_mask(torch.complex(a, b), torch.complex(c, d))
    npz("g9_mask.npz", a=a, b=b, c=c, d=d, cm_r=cm_r, cm_i=cm_i, irm=irm, cirm=cirm,
        decomp=mm.decompress_cIRM(cirm))

    # ---- G8: DeepFilter(1, 5) (deep_filter.py:15-41) -------------------------------------------------------
    dfx = [torch.randn(2, 16, 21, generator=g) for _ in range(2)]
    dfh = [torch.randn(2, 16, 21, generator=g) for _ in range(2)]
    y_df = R["DeepFilter"](1, 5)(dfx, dfh)
    assert torch.allclose(O.DeepFilter(1, 5)(dfx, dfh), y_df, atol=1e-6)
    npz("g8_deepfilter.npz", xr=dfx[0], xi=dfx[1], hr=dfh[0], hi=dfh[1], y=y_df)

    # ---- G6: one full training step, composed from the pinned pieces ------------
    for grp in (1, 4):
        model = O.unet_2(rnn_groups=grp)
        O.closed_form_init(model)
        model.train()
        noisy, clean = O.synth_pair(2, 3200, seed=1234)
        loss, aux = O.train_step_loss(model, noisy, clean)
        loss.backward()
        arrs = dict(noisy=noisy, clean=clean, loss=loss, mask=aux["mask"], est=aux["est"])
        for name, p in model.named_parameters():
            if p.grad is None:
                continue
            arrs["gn/" + name] = p.grad.norm()
            arrs["g8/" + name] = p.grad.flatten()[:8]
        # the reference's wo_male on the same tensors
        est_b2tf = aux["est"].permute(0, 3, 1, 2)
        unproc = torch.cat([aux["feats"]["real"], aux["feats"]["imag"]], dim=1)
        assert torch.equal(R["wo_male"](aux["ref"], est_b2tf, unproc), loss)
        npz(f"g6_step_g{grp}.npz", **arrs)
    print("all golden fixtures written and cross-checked against the reference's code")


if __name__ == "__main__":
    main()
