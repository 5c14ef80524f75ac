"""CPU oracle for the CRUSE hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this file.  The product path (cruse_amd/) never imports it and fails loudly when
the HIP library is missing.

What this is: a torch-CPU (f32) restatement of the reference's training hot path
  waveform -> STFT -> unet_2 (conv encoder, grouped GRU, convT decoder) -> mask
  -> mask*spectrum -> weighted spectral loss (WO-MALE)
built from stock torch.nn modules only.  Each function cites the reference
file:line it follows (paths relative to the reference checkout).

Pinning status: the reference ships NO golden vectors or known-answer tests on
this path, and model/cruse_net.py does not import as
shipped.  The oracle is therefore pinned against *outputs of the reference's own
code run in the build container*: tests/golden/make_golden.py executes the
reference's stft/istft source, its GGRU class (one-token repair R1), its unet_2
class (repairs R2-R8 applied as documented textual substitutions), its wo_male
/ sisnr / si_snr_loss source, mask.py and cust_conv.py, and stores inputs and
outputs as fixtures.  tests/test_oracle.py checks this file against those
fixtures.  Where a repair is a decision rather than a typo fix it is marked
DECISION below.
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Sequence, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

# ----------------------------------------------------------------------------
# Acoustic front end
# ----------------------------------------------------------------------------


def stft(y: torch.Tensor, n_fft: int, hop_length: int, win_length: int) -> torch.Tensor:
    """train_base/acoustics/feature.py:10-30.  [B,L] -> complex64 [B,F,T].

    periodic Hann of length n_fft (:27), center=True with torch's default
    reflect padding, one-sided, no normalisation.
    """
    assert y.dim() == 2  # feature.py:21
    return torch.stft(y, n_fft, hop_length, win_length,
                      window=torch.hann_window(n_fft).to(y.device),
                      return_complex=True, center=True)


def istft(features: torch.Tensor, n_fft: int, hop_length: int, win_length: int,
          length: Optional[int] = None, use_mag_phase: bool = False) -> torch.Tensor:
    """train_base/acoustics/feature.py:33-61.  complex [B,F,T] -> [B,L]."""
    if use_mag_phase:  # feature.py:47-51
        mag, phase = features
        features = torch.complex(mag * torch.cos(phase), mag * torch.sin(phase))
    return torch.istft(features, n_fft, hop_length, win_length,
                       window=torch.hann_window(n_fft).to(features.device),
                       length=length, center=True)


def pre_stft(y: torch.Tensor, n_fft: int, hop_length: int, win_length: int,
             f_net: Optional[int] = None) -> Dict[str, torch.Tensor]:
    """utils/utils.py:389-412 (PreProcess.pre_stft) layout + magnitude formula on
    top of feature.py:10-30 padding.

    DECISION: reflect padding of feature.stft, the [B,1,T,F]
    layout and mag = sqrt(re^2 + im^2 + 1e-8) of utils/utils.py:397-405.
    Returns real, imag, mag, phase as [B,1,T,F_stft] and, when f_net is given,
    `mag_net` = mag[..., :f_net] (R8: the network runs on 160 bins).
    """
    spec = stft(y, n_fft, hop_length, win_length)          # [B,F,T]
    ri = torch.view_as_real(spec).permute(0, 3, 2, 1)      # [B,2,T,F] (utils.py:397 transpose(1,3))
    real = ri[:, 0:1].contiguous()
    imag = ri[:, 1:2].contiguous()
    mag = torch.sqrt(real ** 2 + imag ** 2 + 1e-8)          # utils.py:400
    phase = torch.atan2(imag, real)                         # utils.py:401
    out = {"real": real, "imag": imag, "mag": mag, "phase": phase}
    if f_net is not None:
        out["mag_net"] = mag[..., :f_net].contiguous()
    return out


def masking(mask: torch.Tensor, real: torch.Tensor, imag: torch.Tensor) -> torch.Tensor:
    """utils/utils.py:417-433, post_process_mode == "mag_mapping" (:418-420).

    mask [B,1,T,Fn], real/imag [B,1,T,Fs] with Fs >= Fn.  DECISION (R8): bins
    Fn..Fs-1 of the enhanced spectrum are zero.  Returns [B,T,Fs,2] as :430-432.
    """
    fn, fs = mask.shape[-1], real.shape[-1]
    m = F.pad(mask, (0, fs - fn)) if fs > fn else mask
    out_real = (m * real).squeeze(1)
    out_imag = (m * imag).squeeze(1)
    return torch.stack([out_real, out_imag], dim=-1).contiguous()


# ----------------------------------------------------------------------------
# Model: model/cruse_net.py
# ----------------------------------------------------------------------------


class GGRU(nn.Module):
    """model/cruse_net.py:14-55 with repair R1 (`self.view` -> `out.view`, :53)."""

    def __init__(self, in_features=None, out_features=None, mid_features=None,
                 hidden_size=1024, groups=2):
        super().__init__()
        hidden_size_t = hidden_size // groups                       # :22
        self.gru_list1 = nn.ModuleList([nn.GRU(hidden_size_t, hidden_size_t, 1, batch_first=True)
                                        for _ in range(groups)])  # :23-26
        self.gru_list2 = nn.ModuleList([nn.GRU(hidden_size_t, hidden_size_t, 1, batch_first=True)
                                        for _ in range(groups)])  # :28-31
        self.ln1 = nn.LayerNorm(hidden_size)                        # :32
        self.ln2 = nn.LayerNorm(hidden_size)                        # :33
        self.groups = groups
        self.mid_features = mid_features

    def forward(self, x):
        out = x.transpose(1, 2).contiguous()                        # :39  [B,T,C,F]
        out = out.view(out.size(0), out.size(1), -1).contiguous()   # :40  [B,T,C*F]
        out = torch.chunk(out, self.groups, dim=-1)                 # :42
        out = torch.stack([self.gru_list1[i](out[i])[0] for i in range(self.groups)], dim=-1)  # :43-44
        out = torch.flatten(out, start_dim=-2, end_dim=-1)          # :45 interleave j*g+i
        out = self.ln1(out)                                         # :46
        out = torch.chunk(out, self.groups, dim=-1)                 # :48
        out = torch.cat([self.gru_list2[i](out[i])[0] for i in range(self.groups)], dim=-1)    # :49-50
        out = self.ln2(out)                                         # :51
        out = out.view(out.size(0), out.size(1), x.size(1), -1).contiguous()  # :53 (R1)
        out = out.transpose(1, 2).contiguous()                      # :54
        return out


class unet_2(nn.Module):
    """model/cruse_net.py:129-165 with repairs R2-R8.

    R2 :140  decoder layers are ConvTranspose2d(ch[k], ch[k-1], (1,3), stride) named conv{k}_t
    R3 :142  bn{k}_t = BatchNorm2d(ch[k-1])
    R4 :149-152  causal crop [..., :-padding[0], :] and per-level conv{k}/bn{k}
    R5 :143,:153-156  skip_connect_k(e_k) with padding (0,1)
    R6 :161-164  decoder chains d4 -> d3 -> d2 -> d1
    R7 :144  GGRU(hidden_size=hidden_size, groups=rnn_groups)
    R8 geometry: input has in_feat//2*2 = 160 bins; `fc` (:146) is unused but kept.
    """

    def __init__(self, in_feat=161, ch=(1, 8, 16, 32, 64), stride=(1, 2), rnn_groups=4):
        super().__init__()
        self.laynum = len(ch) - 1                                   # :132
        hidden_size = in_feat // 2 ** self.laynum * ch[-1]          # :133
        self.ker_x = 2                                              # :134
        self.stride = stride
        self.padding = [self.ker_x - stride[0], 3 - stride[1]]      # :136
        for i in range(len(ch) - 1):
            k = i + 1
            setattr(self, f"conv{k}", nn.Conv2d(ch[k - 1], ch[k], (self.ker_x, 3), self.stride, self.padding))  # :138
            setattr(self, f"conv{k}_t", nn.ConvTranspose2d(ch[k], ch[k - 1], (1, 3), self.stride))              # :140 R2
            setattr(self, f"bn{k}", nn.BatchNorm2d(ch[k]))                                                     # :141
            setattr(self, f"bn{k}_t", nn.BatchNorm2d(ch[k - 1]))                                               # :142 R3
            setattr(self, f"skip_connect_{k}", nn.Conv2d(ch[k], ch[k], (1, 3), padding=(0, 1), bias=False))     # :143 R5
        self.gru = GGRU(hidden_size=hidden_size, groups=rnn_groups)  # :144 R7
        self.elu = nn.ReLU()                                        # :145 (named elu, is ReLU)
        self.fc = nn.Linear(in_feat, in_feat)                       # :146 unused

    def forward(self, x, return_intermediates: bool = False):
        p0 = self.padding[0]
        e1 = self.elu(self.bn1(self.conv1(x)[..., :-p0, :]))        # :149 R4
        e2 = self.elu(self.bn2(self.conv2(e1)[..., :-p0, :]))       # :150 R4
        e3 = self.elu(self.bn3(self.conv3(e2)[..., :-p0, :]))       # :151 R4
        e4 = self.elu(self.bn4(self.conv4(e3)[..., :-p0, :]))       # :152 R4
        skip1 = self.skip_connect_1(e1)                             # :153 R5
        skip2 = self.skip_connect_2(e2)
        skip3 = self.skip_connect_3(e3)
        skip4 = self.skip_connect_4(e4)
        out_gru = self.gru(e4)                                      # :158
        out = out_gru + skip4                                       # :160
        d4 = self.elu(self.bn4_t(self.conv4_t(out)[..., :-1])) + skip3   # :161 R6
        d3 = self.elu(self.bn3_t(self.conv3_t(d4)[..., :-1])) + skip2    # :162 R6
        d2 = self.elu(self.bn2_t(self.conv2_t(d3)[..., :-1])) + skip1    # :163 R6
        d1 = torch.sigmoid(self.conv1_t(d2)[..., :-1])              # :164
        if return_intermediates:
            return d1, dict(e1=e1, e2=e2, e3=e3, e4=e4, skip1=skip1, skip2=skip2, skip3=skip3,
                            skip4=skip4, gru=out_gru, d4=d4, d3=d3, d2=d2)
        return d1


class DeepFilter(nn.Module):
    """model/deep_filter.py:15-41.  Repairs: ctor reshape `t_width(f_width, ...)` -> `[t_width*f_width, 1,
    f_width, t_width]` (:26).  DECISION: imaginary part r*fi + i*fr (the reference's :38 writes
    r*fi twice)."""

    def __init__(self, t_dim, f_dim):
        super().__init__()
        self.t_dim, self.f_dim = t_dim, f_dim
        t_width, f_width = t_dim * 2 + 1, f_dim * 2 + 1
        kernel = torch.eye(t_width * f_width)
        self.register_buffer("kernel", torch.reshape(kernel, [t_width * f_width, 1, f_width, t_width]))

    def forward(self, inputs, filters):
        ci = F.conv2d(torch.cat(inputs, 0)[:, None], self.kernel, padding=[self.f_dim, self.t_dim])   # :29-31
        ir, ii = torch.chunk(ci, 2, 0)
        cf = F.conv2d(torch.cat(filters, 0)[:, None], self.kernel, padding=[self.f_dim, self.t_dim])  # :33-35
        fr, fi = torch.chunk(cf, 2, 0)
        out_r = torch.sum(ir * fr - ii * fi, 1)                                                        # :37,39
        out_i = torch.sum(ir * fi + ii * fr, 1)                                                        # :38,40 (decision)
        return torch.cat([out_r, out_i], dim=1)


# ----------------------------------------------------------------------------
# Losses: loss_func/loss.py, train_base/loss.py
# ----------------------------------------------------------------------------


def wo_male(ref, est, unproc, alpha=2.0, beta=1.0, gamma=1.0):
    """loss_func/loss.py:121-148 with the two repairs of SURVEY row a11:
    `torch.size(ref)` -> `ref.size()` (:129) and `unproc[:, 1, :, 1]` -> `unproc[:, 1, :, :]` (:139).
    Inputs [B,2,T,F] (dim 1 = re/im)."""
    if ref.shape != est.shape:
        raise RuntimeError(f"Dimension mismatch when calculate wo-male, {ref.shape} vs {est.shape}")
    B, C, T, Fq = ref.size()
    mag_ref = torch.sqrt(ref[:, 0] ** 2 + ref[:, 1] ** 2)
    mag_est = torch.sqrt(est[:, 0] ** 2 + est[:, 1] ** 2)
    mag_unproc = torch.sqrt(unproc[:, 0] ** 2 + unproc[:, 1] ** 2)
    iam = (mag_ref / mag_unproc) ** gamma                            # :141
    w_iam = torch.exp(alpha / (beta + iam))                          # :142
    loss = w_iam * torch.abs(torch.log10(mag_est + 1) - torch.log10(mag_ref + 1))  # :144-145
    return torch.sum(loss) / (B * T * Fq * 1.0)                      # :146


def sdnr(ref_clean, est_g, ref_noise, snr, beta=20.0):
    """loss_func/loss.py:151-175; repairs: `torch.size` (:164) and vad == 1 because
    activity_detector_tf_frame is `pass` (utils/utils.py:217-219).  snr in dB (scalar or [B])."""
    l_noise = torch.mean(torch.norm(ref_noise * est_g, p=2, dim=(1, 2)) ** 2)       # :166
    s_sa = ref_clean                                                                 # :168-169, vad=1
    l_speech = torch.mean(torch.norm(s_sa - est_g * s_sa, p=2, dim=(1, 2)) ** 2)     # :170
    snr_t = torch.as_tensor(snr, dtype=torch.float32)
    snr_tmp = 10 ** (snr_t / 10)
    beta_tmp = 10 ** (beta / 10)
    alpha = snr_tmp / (snr_tmp + beta_tmp)                                           # :173
    return (alpha * l_speech + (1 - alpha) * l_noise).mean()                         # :174


def sisnr(s1, s2, eps=1e-8):
    """loss_func/loss.py:48-56 (runs as shipped)."""
    def l2(a, b):
        return torch.sum(a * b, -1, keepdim=True)
    s_target = l2(s1, s2) / (l2(s2, s2) + eps) * s2
    e_noise = s1 - s_target
    snr = 10 * torch.log10(l2(s_target, s_target) / (l2(e_noise, e_noise) + eps) + eps)
    return torch.mean(snr)


def si_snr_loss(x, s, eps=1e-8):
    """train_base/loss.py:7-25 (inner function of si_snr_loss())."""
    def l2norm(mat, keep_dim=False):
        return torch.norm(mat, dim=-1, keepdim=keep_dim)
    if x.shape != s.shape:
        raise RuntimeError(f"Dimension mismatch when calculate si_snr, {x.shape} vs {s.shape}")
    x_zm = x - torch.mean(x, dim=-1, keepdim=True)
    s_zm = s - torch.mean(s, dim=-1, keepdim=True)
    t = torch.sum(x_zm * s_zm, dim=-1, keepdim=True) * s_zm / (l2norm(s_zm, keep_dim=True) ** 2 + eps)
    return -torch.mean(20 * torch.log10(eps + l2norm(t) / (l2norm(x_zm - t) + eps)))


# ----------------------------------------------------------------------------
# The training step (the loop train/trainer_casual.py never wrote)
# ----------------------------------------------------------------------------


def closed_form_init(model: nn.Module, scale: float = 1.0) -> None:
    """Deterministic, RNG-free weights so fixtures only need input seeds.

    Every parameter p (named, in state-dict order, index i) gets
    p.flat[j] = a_i * sin(0.37*j + 1.3*i + 0.1) with a_i = scale/sqrt(fan),
    norm weights are 1 + 0.1*sin(.), norm biases 0.05*sin(.)."""
    with torch.no_grad():
        for i, (name, p) in enumerate(model.named_parameters()):
            j = torch.arange(p.numel(), dtype=torch.float64)
            s = torch.sin(0.37 * j + 1.3 * i + 0.1)
            is_norm = (".ln" in name or name.startswith("ln") or "bn" in name)
            if is_norm and name.endswith("weight"):
                v = 1.0 + 0.1 * s
            elif is_norm and name.endswith("bias"):
                v = 0.05 * s
            else:
                fan = p[0].numel() if p.dim() > 1 else p.numel()
                v = scale * s / math.sqrt(max(fan, 1))
            p.copy_(v.reshape(p.shape).to(p.dtype))


def synth_pair(batch: int, length: int, seed: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """Parity-fixture inputs: plain Gaussian, CPU generator.
    clean = 0.05*N(0,1), noise = 0.1*N(0,1), noisy = clean + noise."""
    g = torch.Generator().manual_seed(seed)
    clean = 0.05 * torch.randn(batch, length, generator=g)
    noise = 0.1 * torch.randn(batch, length, generator=g)
    return clean + noise, clean


def enhanced_spectrum(model: nn.Module, noisy: torch.Tensor, n_fft=320, hop=160, win=320):
    """noisy [B,L] -> (mask [B,1,T,Fn], est [B,T,Fs,2], feats dict)."""
    f_net = (n_fft // 2 + 1) // 2 * 2
    feats = pre_stft(noisy, n_fft, hop, win, f_net=f_net)
    mask = model(feats["mag_net"])
    est = masking(mask, feats["real"], feats["imag"])
    return mask, est, feats


def train_step_loss(model: nn.Module, noisy: torch.Tensor, clean: torch.Tensor,
                    n_fft=320, hop=160, win=320, loss_mode: str = "WO_MALE") -> Tuple[torch.Tensor, Dict]:
    """One forward of the hot loop: STFT(noisy), STFT(clean) ->
    unet_2 -> mask*spectrum -> loss.  loss_func.loss(inputs, labels, noisy) calls
    wo_male(labels, inputs, noisy) (loss_func/loss.py:24,30)."""
    mask, est, feats = enhanced_spectrum(model, noisy, n_fft, hop, win)
    cfe = pre_stft(clean, n_fft, hop, win)
    est_b2tf = est.permute(0, 3, 1, 2)                                 # [B,2,T,F]
    ref = torch.cat([cfe["real"], cfe["imag"]], dim=1)
    unproc = torch.cat([feats["real"], feats["imag"]], dim=1)
    wave = None
    if loss_mode == "WO_MALE":
        loss = wo_male(ref, est_b2tf, unproc)
    elif loss_mode == "SI_SNR":
        # PreProcess.reconstruction (utils/utils.py:443-455) + si_snr_loss
        # (train_base/loss.py:7-25), the loss reachable through tools/train_stand.py:73-75
        spec = torch.view_as_complex(est.contiguous()).transpose(1, 2)          # [B,F,T]
        wave = istft(spec, n_fft, hop, win, length=noisy.shape[1])
        loss = si_snr_loss(wave, clean)
    elif loss_mode in ("L1", "MSE"):
        # train_base/loss.py:3-4: l1_loss / mse_loss are torch.nn.L1Loss / MSELoss, instantiated by tools/train_stand.py:73-75.
        # DECISION (the reference never wrote the trainer that calls them): like si_snr_loss they compare WAVEFORMS -- the
        # enhanced waveform of PreProcess.reconstruction (utils/utils.py:443-455) against the clean clip, default reduction "mean"
        spec = torch.view_as_complex(est.contiguous()).transpose(1, 2)          # [B,F,T]
        wave = istft(spec, n_fft, hop, win, length=noisy.shape[1])
        loss = (torch.nn.L1Loss() if loss_mode == "L1" else torch.nn.MSELoss())(wave, clean)
    else:
        raise ValueError(loss_mode)
    return loss, dict(mask=mask, est=est, feats=feats, ref=ref, wave=wave)
