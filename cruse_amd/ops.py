"""Torch-tensor wrappers over the C ABI (include/cruse_hip.h).

torch is used here for device memory and the current HIP stream only; every
arithmetic op is a kernel in libcruse_hip.so.  All tensors are f32 CUDA(HIP)
tensors in frame-major [B,T,C,F] layout unless stated.
"""
from __future__ import annotations

import ctypes
from typing import Dict, List, Optional, Sequence

import torch

from ._lib import PREC_BF16, PREC_BF16X3, PREC_BY_NAME, PREC_F32, check, lib  # noqa: F401


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _p(t: Optional[torch.Tensor]):
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("cruse_amd ops need tensors on the HIP device (no CPU fallback)")
    if not t.is_contiguous():
        raise RuntimeError(f"cruse_amd ops need contiguous tensors, got strides {t.stride()} for shape {tuple(t.shape)}")
    return t.data_ptr()


def _xdt(t: torch.Tensor, name: str) -> int:
    """CRUSE_DT_* of a conv / weight-gradient INPUT: f32, or bf16 for the backward-only tensors of the bf16 mode."""
    if t.dtype == torch.float32:
        return 0
    if t.dtype == torch.bfloat16:
        return 2
    raise RuntimeError(f"{name}: expected float32 or bfloat16, got {t.dtype}")


def _f32(t: torch.Tensor, name: str) -> torch.Tensor:
    if t.dtype != torch.float32:
        raise RuntimeError(f"{name}: expected float32, got {t.dtype}")
    return t


def set_option(name: str, value=None) -> None:
    """Library option (cruse_set_option: kernel-selection A/B switches and profiling aids -- the library reads no
    environment variables); value None restores the default."""
    check(lib.cruse_set_option(name.encode(), 0 if value is None else int(value), 1 if value is None else 0))


def get_option(name: str):
    v, isset = ctypes.c_int(0), ctypes.c_int(0)
    check(lib.cruse_get_option(name.encode(), ctypes.byref(v), ctypes.byref(isset)))
    return v.value if isset.value else None


class options:
    """with ops.options(gru_bwd_rs=0, gru_wlo=0): ...  -- library options for the duration of the block."""

    def __init__(self, **kw):
        self.kw = kw

    def __enter__(self):
        self.saved = {k: get_option(k) for k in self.kw}
        for k, v in self.kw.items():
            set_option(k, v)
        return self

    def __exit__(self, *exc):
        for k, v in self.saved.items():
            set_option(k, v)


def prec_code(prec) -> int:
    if isinstance(prec, str):
        return PREC_BY_NAME[prec]
    return int(prec)


# ---------------------------------------------------------------- STFT / iSTFT
def stft_frames(L: int, hop: int) -> int:
    return 1 + L // hop


def stft(wave: torch.Tensor, n_fft: int, hop: int, want_ri: bool = True, mag_bins: int = 0,
         mag_eps: float = 0.0, out=None):
    """wave [B,L] -> (re [B,T,F], im [B,T,F], mag [B,T,mag_bins]); absent outputs are None.
    out = (re, im, mag): write into these (None entries are not produced) instead of allocating."""
    _f32(wave, "stft")
    if wave.dim() != 2:
        raise RuntimeError(f"stft expects [B,L], got {tuple(wave.shape)}")
    B, L = wave.shape
    T = stft_frames(L, hop)
    F = n_fft // 2 + 1
    if out is not None:
        re, im, mag = out
        mag_bins = mag.shape[-1] if mag is not None else 0
    else:
        re = torch.empty(B, T, F, device=wave.device, dtype=torch.float32) if want_ri else None
        im = torch.empty(B, T, F, device=wave.device, dtype=torch.float32) if want_ri else None
        mag = torch.empty(B, T, mag_bins, device=wave.device, dtype=torch.float32) if mag_bins > 0 else None
    check(lib.cruse_stft_fwd(_p(wave), B, L, n_fft, hop, _p(re), _p(im), _p(mag), mag_bins, mag_eps, _stream()))
    return re, im, mag


def istft(re: torch.Tensor, im: torch.Tensor, n_fft: int, hop: int, length: int) -> torch.Tensor:
    B, T, F = re.shape
    if F != n_fft // 2 + 1 or im.shape != re.shape:
        raise RuntimeError(f"istft: spectrum shape {tuple(re.shape)} does not match n_fft={n_fft}")
    wave = torch.empty(B, length, device=re.device, dtype=torch.float32)
    check(lib.cruse_istft_fwd(_p(re), _p(im), B, T, n_fft, hop, length, _p(wave), _stream()))
    return wave


def istft_bwd(dwave: torch.Tensor, T: int, n_fft: int, hop: int):
    B, L = dwave.shape
    F = n_fft // 2 + 1
    dre = torch.empty(B, T, F, device=dwave.device, dtype=torch.float32)
    dim = torch.empty(B, T, F, device=dwave.device, dtype=torch.float32)
    check(lib.cruse_istft_bwd(_p(dwave), B, T, n_fft, hop, L, _p(dre), _p(dim), _stream()))
    return dre, dim


# ---------------------------------------------------------------- convolutions
CONV_PREC = {"f32": PREC_F32, "bf16x3": PREC_BF16X3, "bf16": PREC_BF16X3, "valu": -1}


def conv_prec(prec) -> int:
    """MFMA precision used by the convolutions for a model precision mode: the convs are HBM-bound, so
    the bf16 mode runs them as split-bf16 x3 (~f32 accuracy) at no measurable cost."""
    if prec is None:
        return -1
    return CONV_PREC[prec] if isinstance(prec, str) else int(prec)


def conv_gather(x, w, bias, B, T, Cin, Fin, Cout, Fout, KT, S, pad, w_layout=0, act=0, out=None, accum=False,
                prec=None, bn_bwd=None, out_bf16=False):
    """bn_bwd = (bn_y, mean, rstd, gamma, beta, relu): the output is the gradient wrt the output of that BatchNorm(+ReLU); returns
    (out, sums) with the backward sums of bn_act_bwd accumulated by the conv's epilogue (cruse_conv_gather_bnbwd)."""
    if out is None:
        out = torch.empty(B, T, Cout, Fout, device=x.device, dtype=torch.bfloat16 if out_bf16 else torch.float32)
    if bn_bwd is not None:
        by, mean, rstd, gamma, beta, relu = bn_bwd
        sums, z = ARENA.take(2 * Cout * BN_STAT_REPLICAS, x.device)
        check(lib.cruse_conv_gather_bnbwd(_p(x), _p(w), _p(out), B, T, Cin, Fin, Cout, Fout, KT, S, pad, w_layout,
                                          1 if accum else 0, conv_prec(prec), _p(by), _p(mean), _p(rstd), _p(gamma), _p(beta),
                                          1 if relu else 0, _p(sums), z, _xdt(x, "conv_gather"), _xdt(out, "conv_gather out"), _stream()))
        return out, sums
    check(lib.cruse_conv_gather(_p(x), _p(w), _p(bias), _p(out), B, T, Cin, Fin, Cout, Fout, KT, S, pad,
                                w_layout, act, 1 if accum else 0, conv_prec(prec), _xdt(x, "conv_gather"), _xdt(out, "conv_gather out"),
                                _stream()))
    return out


def conv_scatter2(g, w, bias, B, T, Cs, Fg, Cout, KT, pad, act=0, out=None, accum=False, prec=None, bn_bwd=None, out_bf16=False):
    """bn_bwd: as for conv_gather (cruse_conv_scatter2_bnbwd)."""
    Fout = 2 * Fg
    if out is None:
        out = torch.empty(B, T, Cout, Fout, device=g.device, dtype=torch.bfloat16 if out_bf16 else torch.float32)
    if bn_bwd is not None:
        by, mean, rstd, gamma, beta, relu = bn_bwd
        sums, z = ARENA.take(2 * Cout * BN_STAT_REPLICAS, g.device)
        check(lib.cruse_conv_scatter2_bnbwd(_p(g), _p(w), _p(out), B, T, Cs, Fg, Cout, Fout, KT, pad, 1 if accum else 0,
                                            conv_prec(prec), _p(by), _p(mean), _p(rstd), _p(gamma), _p(beta), 1 if relu else 0,
                                            _p(sums), z, _xdt(g, "conv_scatter2"), _xdt(out, "conv_scatter2 out"), _stream()))
        return out, sums
    check(lib.cruse_conv_scatter2(_p(g), _p(w), _p(bias), _p(out), B, T, Cs, Fg, Cout, Fout, KT, pad, act,
                                  1 if accum else 0, conv_prec(prec), _xdt(g, "conv_scatter2"), _xdt(out, "conv_scatter2 out"), _stream()))
    return out


def _bwd_in_args(dout, bn_in):
    y, mean, rstd, gamma, beta, sums, relu, training, dgamma, dbeta, dbias = bn_in
    dv = torch.empty(y.shape, device=y.device, dtype=torch.bfloat16)
    return dv, (_p(dout), _xdt(dout, "conv bnbwd_in dout"), _p(y), _p(mean), _p(rstd), _p(gamma), _p(beta), _p(sums), BN_STAT_REPLICAS,
                1 if relu else 0, 1 if training else 0, _p(dv), _p(dgamma), _p(dbeta), _p(dbias))


def conv_gather_bwd_in(dout, bn_in, w, B, T, Cin, Fin, Cout, Fout, KT, S, pad, w_layout=0, out=None, accum=False, prec=None, bn_bwd=None,
                       out_bf16=False):
    """Data gradient with the BatchNorm(+ReLU) backward of its INPUT fused into the staging (cruse_conv_gather_bnbwd_in).
    bn_in = (y_pre, mean, rstd, gamma, beta, sums [BN_STAT_REPLICAS][2*Cin], relu, training, dgamma, dbeta, dbias or None): `dout` is the gradient
    wrt that BatchNorm's output.  Returns (out, output-side sums or None, dy [B,T,Cin,Fin] bf16 -- the weight gradient's operand)."""
    if out is None:
        out = torch.empty(B, T, Cout, Fout, device=dout.device, dtype=torch.bfloat16 if out_bf16 else torch.float32)
    dv, head = _bwd_in_args(dout, bn_in)
    sums, z = (ARENA.take(2 * Cout * BN_STAT_REPLICAS, dout.device) if bn_bwd is not None else (None, 0))
    by, mean, rstd, gamma, beta, relu = bn_bwd if bn_bwd is not None else (None, None, None, None, None, False)
    check(lib.cruse_conv_gather_bnbwd_in(*head, _p(w), _p(out), B, T, Cin, Fin, Cout, Fout, KT, S, pad, w_layout, 1 if accum else 0, conv_prec(prec),
                                         _p(by), _p(mean), _p(rstd), _p(gamma), _p(beta), 1 if relu else 0, _p(sums), z,
                                         _xdt(out, "conv_gather_bwd_in out"), _stream()))
    return out, sums, dv


def conv_scatter2_bwd_in(dout, bn_in, w, B, T, Cs, Fg, Cout, KT, pad, out=None, accum=False, prec=None, bn_bwd=None, out_bf16=False):
    """As conv_gather_bwd_in for the stride-2 transposed form (cruse_conv_scatter2_bnbwd_in)."""
    Fout = 2 * Fg
    if out is None:
        out = torch.empty(B, T, Cout, Fout, device=dout.device, dtype=torch.bfloat16 if out_bf16 else torch.float32)
    dv, head = _bwd_in_args(dout, bn_in)
    sums, z = (ARENA.take(2 * Cout * BN_STAT_REPLICAS, dout.device) if bn_bwd is not None else (None, 0))
    by, mean, rstd, gamma, beta, relu = bn_bwd if bn_bwd is not None else (None, None, None, None, None, False)
    check(lib.cruse_conv_scatter2_bnbwd_in(*head, _p(w), _p(out), B, T, Cs, Fg, Cout, Fout, KT, pad, 1 if accum else 0, conv_prec(prec),
                                           _p(by), _p(mean), _p(rstd), _p(gamma), _p(beta), 1 if relu else 0, _p(sums), z,
                                           _xdt(out, "conv_scatter2_bwd_in out"), _stream()))
    return out, sums, dv


class BnIn:
    """The BatchNorm2d(train) + ReLU a forward conv applies to its INPUT while staging it (cruse_conv_*_bnin): `y_pre` the layer's
    pre-BN tensor, `sums` its batch sums [BN_STAT_REPLICAS][2C], gamma / beta, count = rows * F; mean / rstd (+ running statistics)
    are published by the ONE consumer that is handed them (publish=True); add: tensor added after the ReLU (the decoder's skip)."""

    def __init__(self, y_pre, sums, nrep, count, eps, momentum, gamma, beta, mean, rstd, running_mean=None, running_var=None, add=None):
        self.y_pre, self.sums, self.nrep, self.count, self.eps, self.momentum = y_pre, sums, nrep, count, eps, momentum
        self.gamma, self.beta, self.mean, self.rstd, self.rm, self.rv, self.add = gamma, beta, mean, rstd, running_mean, running_var, add

    def args(self, publish, copy_bf16):
        return (_p(self.y_pre), _p(self.sums), self.nrep, self.count, self.eps, self.momentum, _p(self.gamma), _p(self.beta),
                _p(self.mean) if publish else None, _p(self.rstd) if publish else None, _p(self.rm) if publish else None,
                _p(self.rv) if publish else None, _p(self.add), _p(copy_bf16))


def bnin_eligible(prec, Cin, Cout) -> bool:
    """shapes / modes cruse_conv_*_bnin takes: the MFMA forward convs of the bf16 mode"""
    # (the bf16 MODE: its forward convs run split-bf16 x3 and its weight gradients take the bf16 copies as they are)
    return (prec is not None and prec_code(prec) == PREC_BF16 and conv_prec(prec) == PREC_BF16X3 and 8 <= Cin <= 64
            and (Cin & (Cin - 1)) == 0 and 8 <= Cout <= 64)


def conv_gather_bnin(bn: BnIn, w, bias, B, T, Cin, Fin, Cout, Fout, KT, S, pad, prec, publish=False, copy_bf16=None, want_sums=False,
                     out=None):
    """conv_gather on relu(bn(y_pre)) [+ add] without materialising it -> y, or (y, sums) with want_sums."""
    if out is None:
        out = torch.empty(B, T, Cout, Fout, device=w.device, dtype=torch.float32)
    sums, z = ARENA.take(2 * Cout * BN_STAT_REPLICAS, w.device) if want_sums else (None, 0)
    check(lib.cruse_conv_gather_bnin(*bn.args(publish, copy_bf16), _p(w), _p(bias), _p(out), B, T, Cin, Fin, Cout, Fout, KT, S, pad,
                                     conv_prec(prec), _p(sums), z, _stream()))
    return (out, sums) if want_sums else out


def conv_scatter2_bnin(bn: BnIn, w, bias, B, T, Cs, Fg, Cout, KT, pad, prec, publish=False, copy_bf16=None, want_sums=False):
    out = torch.empty(B, T, Cout, 2 * Fg, device=w.device, dtype=torch.float32)
    sums, z = ARENA.take(2 * Cout * BN_STAT_REPLICAS, w.device) if want_sums else (None, 0)
    check(lib.cruse_conv_scatter2_bnin(*bn.args(publish, copy_bf16), _p(w), _p(bias), _p(out), B, T, Cs, Fg, Cout, 2 * Fg, KT, pad,
                                       conv_prec(prec), _p(sums), z, _stream()))
    return (out, sums) if want_sums else out


def conv_gather_bnstats(x, w, bias, B, T, Cin, Fin, Cout, Fout, KT, S, pad, prec=None):
    """conv_gather + the BatchNorm batch sums of its output, accumulated by the conv's epilogue -> (y, sums)."""
    out = torch.empty(B, T, Cout, Fout, device=x.device, dtype=torch.float32)
    sums, z = ARENA.take(2 * Cout * BN_STAT_REPLICAS, x.device)
    check(lib.cruse_conv_gather_bnstats(_p(x), _p(w), _p(bias), _p(out), B, T, Cin, Fin, Cout, Fout, KT, S, pad,
                                        conv_prec(prec), _p(sums), z, _stream()))
    return out, sums


def conv_scatter2_bnstats(g, w, bias, B, T, Cs, Fg, Cout, KT, pad, prec=None):
    Fout = 2 * Fg
    out = torch.empty(B, T, Cout, Fout, device=g.device, dtype=torch.float32)
    sums, z = ARENA.take(2 * Cout * BN_STAT_REPLICAS, g.device)
    check(lib.cruse_conv_scatter2_bnstats(_p(g), _p(w), _p(bias), _p(out), B, T, Cs, Fg, Cout, Fout, KT, pad,
                                          conv_prec(prec), _p(sums), z, _stream()))
    return out, sums


_wgrad_ws = {}
_ws_retired = []          # outgrown workspaces: captured HIP graphs may still point at them, so they are never freed


def _ws(key, nbytes: int, device, zero: bool = True) -> torch.Tensor:
    """zero = False: a scratch whose user writes every byte it reads (weight-gradient / split-K slabs).  Those are keyed by STREAM, so a
    HIP-graph capture meets new keys -- and a torch.zeros made during capture is a fill NODE that replays with every step (round 6: eight
    5-15 us fills in the leaf chain of the captured step)."""
    device = torch.device(device)
    if device.type == "cuda" and device.index is None:       # "cuda" and "cuda:0" must name the same workspace
        device = torch.device("cuda", torch.cuda.current_device())
    buf = _wgrad_ws.get((key, device))
    if buf is None or buf.numel() < nbytes:
        old = buf
        # zero-filled: the GRU workspace starts with a STICKY 256-byte header (hand-off status word) that the library
        # never clears (include/cruse_hip.h, cruse_gru_seq_fwd); a grown buffer inherits the old header
        buf = (torch.zeros if zero else torch.empty)(nbytes, dtype=torch.uint8, device=device)
        if old is not None:
            _ws_retired.append((key, old))           # (graphs captured on the old buffer keep it alive)
        _wgrad_ws[(key, device)] = buf
    return buf


_gru_hdr: Dict[torch.device, torch.Tensor] = {}


def gru_header(device) -> torch.Tensor:
    """The STICKY 256-byte status header of the recurrence kernels (word 0: a hand-off timed out; bytes 64.. / 128..: the
    s_memtime stamps of the gru_dbg = 32 instances) -- ONE tensor per device for the life of the process, never regrown:
    HIP graphs captured before the panel scratch grew for a larger batch keep writing their time-outs where step_health and
    the guarded Adam look (the header used to be the first 256 bytes of the regrowable workspace)."""
    device = torch.device(device)
    if device.type == "cuda" and device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())
    h = _gru_hdr.get(device)
    if h is None:
        h = _gru_hdr[device] = torch.zeros(256, dtype=torch.uint8, device=device)
    return h


WGRAD_PREC = {"f32": PREC_F32, "bf16x3": PREC_BF16X3, "bf16": PREC_BF16, "valu": -1}


def conv_wgrad(a, bt, dw, B, T, Ca, Fa, Cb, Fb, KT, S, pad, prec=None):
    """dw[ca][cb][kt][kf] += ... (dw is a contiguous f32 tensor of Ca*Cb*KT*3 elements)."""
    nbytes = lib.cruse_conv_wgrad_ws_bytes(Ca, Cb, KT)
    # one partial-slab workspace PER STREAM: weight-gradient leaves may run concurrently on several side streams
    ws = _ws(("wgrad", _stream()), nbytes, a.device, zero=False)
    pc = -1 if prec is None else (WGRAD_PREC[prec] if isinstance(prec, str) else int(prec))
    check(lib.cruse_conv_wgrad(_p(a), _p(bt), _p(dw), B, T, Ca, Fa, Cb, Fb, KT, S, pad, pc, _xdt(a, "conv_wgrad a"),
                               _xdt(bt, "conv_wgrad bt"), _p(ws), _stream()))


def upsample_w(x, rows, W, up=2):
    """nearest FreqUpsample along the last axis (cust_conv.py:177-184): x [rows, W] f32 -> [rows, W * up]"""
    _f32(x, "upsample_w")
    out = torch.empty(rows, W * up, device=x.device, dtype=torch.float32)
    check(lib.cruse_upsample_w(_p(x), rows, W, up, _p(out), 0, _stream()))
    return out


def downsum_w(dxu, rows, W, up=2):
    """its gradient: [rows, W * up] f32 -> [rows, W] (sums of `up` neighbours)"""
    _f32(dxu, "downsum_w")
    out = torch.empty(rows, W, device=dxu.device, dtype=torch.float32)
    check(lib.cruse_downsum_w(_p(dxu), rows, W, up, _p(out), 0, _stream()))
    return out


def channel_sum(g, rows, C, F, out):
    check(lib.cruse_channel_sum(_p(g), rows, C, F, _p(out), _stream()))


def col_sum(g, g_off, rows, ncol, ld, out):
    check(lib.cruse_col_sum(g.data_ptr() + 4 * g_off, rows, ncol, ld, _p(out), _stream()))


# ---------------------------------------------------------------- zeroed scratch for the step's reductions
class _ZeroArena:
    """f64 accumulators for the reductions of ONE training step (BatchNorm batch sums, backward sums, loss sums), carved
    from one buffer that a single cruse_zero clears at the top of the step -- instead of one memset launch in front of
    every reduction kernel (15 launches per step).  Inactive outside `with ARENA.step(device):` -- callers then get a
    fresh tensor and the kernel clears it itself.  Under HIP-graph capture the slices are handed out in the same order on
    every capture, so the captured pointers stay valid."""

    CAP = 1 << 14           # doubles (128 KiB)

    def __init__(self):
        self.bufs = {}
        self.active = None
        self.off = 0

    def step(self, device):
        arena = self

        class _Ctx:
            def __enter__(self_):
                dev = torch.device(device)
                buf = arena.bufs.get(dev)
                if buf is None:
                    buf = arena.bufs[dev] = torch.zeros(arena.CAP, device=dev, dtype=torch.float64)
                arena.active, arena.off = buf, 0
                zero_(buf)
                return arena

            def __exit__(self_, *exc):
                arena.active = None
        return _Ctx()

    def take(self, n: int, device):
        """-> (f64 tensor of n elements, zeroed flag)."""
        if self.active is None or self.active.device != device or self.off + n > self.CAP:
            return torch.empty(n, device=device, dtype=torch.float64), 0
        t = self.active[self.off:self.off + n]
        self.off += (n + 1) // 2 * 2                    # keep 16-byte alignment
        return t, 1


ARENA = _ZeroArena()


# ---------------------------------------------------------------- BatchNorm
def bn_stats(y, rows, C, F):
    sums, z = ARENA.take(2 * C, y.device)
    check(lib.cruse_bn_stats(_p(y), rows, C, F, _p(sums), z, _stream()))
    return sums


BN_STAT_REPLICAS = 16      # CRUSE_BN_STAT_REPLICAS (include/cruse_hip.h): layout of the sums of conv_*_bnstats


def bn_finalize_act_fwd(y, sums, count, eps, momentum, gamma, beta, skip, rows, C, F, relu=True, running_mean=None,
                        running_var=None, out_bf16=None):
    """bn_finalize + bn_act_fwd in one launch -> (out, mean, rstd).  sums: [2*C] (bn_stats) or [replicas, 2*C]
    (conv_*_bnstats); the statistic is the sum over the replicas."""
    if sums.numel() % (2 * C) != 0:
        raise RuntimeError(f"bn_finalize_act_fwd: {sums.numel()} sums for {C} channels")
    out = torch.empty_like(y)
    mean = torch.empty(C, device=y.device, dtype=torch.float32)
    rstd = torch.empty(C, device=y.device, dtype=torch.float32)
    if out_bf16 is not None and (out_bf16.dtype not in (torch.bfloat16, torch.float16) or out_bf16.numel() < y.numel()):
        raise RuntimeError("bn_finalize_act_fwd: out_bf16 must be a bf16 (or f16) tensor of at least y.numel() elements")
    # (the operand copy takes the element type of the tensor handed in: bf16, or f16 for the single-pass f16 gate projection)
    check(lib.cruse_bn_finalize_act_fwd_c(_p(y), _p(sums), sums.numel() // (2 * C), count, eps, momentum, _p(gamma), _p(beta), _p(skip), _p(out),
                                          _p(out_bf16), _copy_dt(out_bf16), _p(mean), _p(rstd), _p(running_mean), _p(running_var), rows, C, F,
                                          1 if relu else 0, _stream()))
    return out, mean, rstd


def bn_finalize(sums, count, C, eps, momentum, running_mean=None, running_var=None):
    mean = torch.empty(C, device=sums.device, dtype=torch.float32)
    rstd = torch.empty(C, device=sums.device, dtype=torch.float32)
    check(lib.cruse_bn_finalize(_p(sums), count, C, eps, momentum, _p(mean), _p(rstd), _p(running_mean),
                                _p(running_var), _stream()))
    return mean, rstd


def bn_eval_stats(running_mean, running_var, eps):
    C = running_mean.numel()
    mean = torch.empty(C, device=running_mean.device, dtype=torch.float32)
    rstd = torch.empty(C, device=running_mean.device, dtype=torch.float32)
    check(lib.cruse_bn_eval_stats(_p(running_mean), _p(running_var), C, eps, _p(mean), _p(rstd), _stream()))
    return mean, rstd


def bn_act_fwd(y, mean, rstd, gamma, beta, skip, rows, C, F, relu=True):
    out = torch.empty_like(y)
    check(lib.cruse_bn_act_fwd(_p(y), _p(mean), _p(rstd), _p(gamma), _p(beta), _p(skip), _p(out), rows, C, F,
                               1 if relu else 0, _stream()))
    return out


def bn_act_bwd(dout, y, mean, rstd, gamma, beta, rows, C, F, relu, training, dgamma, dbeta, dbias=None, sums=None,
               out_bf16=False):
    """sums: the backward sums the convolution that produced dout has already accumulated (conv_gather / conv_scatter2 with
    bn_bwd: [BN_STAT_REPLICAS][2*C]); None: the reduce pass runs here.  out_bf16: dy as a bf16 tensor (its consumers -- the
    data-gradient conv and the weight gradient of the bf16 mode -- round it to bf16 operands anyway: the same bits, half the
    bytes written once and read twice)."""
    nrep = BN_STAT_REPLICAS
    if sums is None:
        nrep = 1
        sums, z = ARENA.take(2 * C, y.device)
        _f32(dout, "bn_act_bwd (reduce pass) dout")       # (a bf16 gradient arrives with its sums from the conv that stored it)
        check(lib.cruse_bn_act_bwd_reduce(_p(dout), _p(y), _p(mean), _p(rstd), _p(gamma), _p(beta), rows, C, F,
                                          1 if relu else 0, _p(sums), z, _stream()))
    dy = torch.empty_like(y, dtype=torch.bfloat16 if out_bf16 else torch.float32)
    check(lib.cruse_bn_act_bwd_apply(_p(dout), _p(y), _p(mean), _p(rstd), _p(gamma), _p(beta), _p(sums), nrep, rows, C, F,
                                     1 if relu else 0, 1 if training else 0, _xdt(dout, "bn_act_bwd dout"), _p(dy), 2 if out_bf16 else 0,
                                     _p(dgamma), _p(dbeta),
                                     _p(dbias), _stream()))
    return dy


def _copy_dt(t) -> int:
    """CRUSE_DT_* of a 2-byte operand copy (CRUSE_DT_F16 = 1, CRUSE_DT_BF16 = 2; None: bf16)"""
    return 1 if (t is not None and t.dtype == torch.float16) else 2


# ---------------------------------------------------------------- LayerNorm
def ln_fwd(x, gamma, beta, res, rows, H, interleave_g=1, eps=1e-5, save=True, out=None, out_bf16=None, seg=None, stats=None):
    """seg = (seg_len, seg_stride, seg_off): only the rows of one time chunk (cruse_ln_fwd row segments; rows = B * seg_len),
    into the full-size out / out_bf16 / stats = (mean, rstd) of the call that owns them."""
    y = torch.empty_like(x) if out is None else out
    if stats is not None:
        mean, rstd = stats
    else:
        nstat = x.numel() // H
        mean = torch.empty(nstat, device=x.device, dtype=torch.float32) if save else None
        rstd = torch.empty(nstat, device=x.device, dtype=torch.float32) if save else None
    sl, ss, so = seg if seg is not None else (0, 0, 0)
    if out_bf16 is not None and (out_bf16.dtype not in (torch.bfloat16, torch.float16) or out_bf16.numel() < x.numel()):
        raise RuntimeError("ln_fwd: out_bf16 must be a bf16 (or f16) tensor of at least x.numel() elements")
    check(lib.cruse_ln_fwd_c(_p(x), _p(gamma), _p(beta), _p(res), _p(y), _p(out_bf16), _copy_dt(out_bf16), _p(mean), _p(rstd), rows, H,
                             interleave_g, eps, sl, ss, so, _stream()))
    return y, mean, rstd


def ln_bwd(dy, x, mean, rstd, gamma, rows, H, interleave_g, dgamma, dbeta):
    dx = torch.empty_like(x)
    check(lib.cruse_ln_bwd(_p(dy), _p(x), _p(mean), _p(rstd), _p(gamma), rows, H, interleave_g, _p(dx), _p(dgamma),
                           _p(dbeta), _stream()))
    return dx


# ---------------------------------------------------------------- GEMM
def gemm(transA, transB, M, N, K, A, a_off, lda, Bm, b_off, ldb, C, c_off, ldc, bias=None, accumulate=False,
         splitk=1, b_shift_T=0, prec=PREC_F32):
    """Raw-pointer GEMM; *_off are element offsets into the tensors' storage views."""
    pa = A.data_ptr() + 4 * a_off
    pb = Bm.data_ptr() + 4 * b_off
    pc = C.data_ptr() + 4 * c_off
    check(lib.cruse_gemm(1 if transA else 0, 1 if transB else 0, M, N, K, pa, lda, pb, ldb, pc, ldc, _p(bias),
                         1 if accumulate else 0, splitk, b_shift_T, prec_code(prec), _stream()))


def gemm_bf16_nt(M, N, K, A, a_off, lda, Bm, b_off, ldb, C, c_off, ldc, bias=None, accumulate=False, splitk=1,
                 a_kstride=64, b_kstride=64, slabs=False):
    """C[M,N] (+)= A[M,K] . B[N,K]^T on bf16 operands (element offsets into the tensors' storage).
    *_kstride == 64: row-major operand; larger: the K-tiled time-major layout (lda == 64), see cruse_hip.h.
    slabs (split-K, accumulate): the k-slices store partial sums to a per-stream scratch that one kernel adds to C in slice order
    -- no atomics (cruse_gemm_bf16_nt_slabs)."""
    if slabs and accumulate and bias is None and abs(splitk) > 1 and N % 4 == 0 and C.dtype == torch.float32:
        if A.dtype != torch.bfloat16 or Bm.dtype != torch.bfloat16:
            raise RuntimeError("gemm_bf16_nt needs bf16 operands")
        nbytes = lib.cruse_gemm_bf16_slab_bytes(M, N, splitk)
        ws = _ws(("gemm_slabs", _stream()), nbytes, C.device, zero=False)
        check(lib.cruse_gemm_bf16_nt_slabs(M, N, K, A.data_ptr() + 2 * a_off, lda, a_kstride, Bm.data_ptr() + 2 * b_off, ldb, b_kstride,
                                           C.data_ptr() + 4 * c_off, ldc, splitk, _p(ws), ws.numel(), _stream()))
        return
    if C.dtype in _OUT16 and A.dtype == torch.bfloat16 and Bm.dtype == torch.bfloat16 and not accumulate and abs(splitk) <= 1:
        return _gemm_nt_out16(M, N, K, A, None, a_off, lda, a_kstride, Bm, None, b_off, ldb, b_kstride, C, c_off, ldc, bias, False)
    if A.dtype != torch.bfloat16 or Bm.dtype != torch.bfloat16 or C.dtype != torch.float32:
        raise RuntimeError("gemm_bf16_nt needs bf16 operands and an f32 result")
    check(lib.cruse_gemm_bf16_nt(M, N, K, A.data_ptr() + 2 * a_off, lda, a_kstride, Bm.data_ptr() + 2 * b_off, ldb, b_kstride,
                                 C.data_ptr() + 4 * c_off, ldc, _p(bias), 1 if accumulate else 0, splitk, _stream()))


def gemm_bf16_nt_cat(Ms, N, K, A, a_rows, lda, Bs, b_off, ldb, C, c_off, ldc, splitk, a_kstride=64, b_kstride=64):
    """C[sum Ms, N] += cat_i(A[a_rows[i] : a_rows[i] + Ms[i]] . Bs[i]^T): up to three split-K slab products in ONE launch
    (cruse_gemm_bf16_nt_slabs_cat; a_rows in rows of A, b_off / c_off in elements)."""
    n = len(Ms)
    if A.dtype != torch.bfloat16 or any(b.dtype != torch.bfloat16 for b in Bs) or C.dtype != torch.float32:
        raise RuntimeError("gemm_bf16_nt_cat needs bf16 operands and an f32 result")
    nbytes = lib.cruse_gemm_bf16_slab_bytes(int(sum(Ms)), N, splitk)
    ws = _ws(("gemm_slabs", _stream()), nbytes, C.device, zero=False)
    ms = (ctypes.c_int * n)(*[int(m) for m in Ms])
    ar = (ctypes.c_longlong * n)(*[int(r) for r in a_rows])
    bs = (ctypes.c_void_p * n)(*[b.data_ptr() + 2 * b_off for b in Bs])
    check(lib.cruse_gemm_bf16_nt_slabs_cat(n, ctypes.cast(ms, ctypes.c_void_p), N, K, A.data_ptr(), ctypes.cast(ar, ctypes.c_void_p), lda, a_kstride,
                                           ctypes.cast(bs, ctypes.c_void_p), ldb, b_kstride, C.data_ptr() + 4 * c_off, ldc, splitk, _p(ws),
                                           ws.numel(), _stream()))


def gemm_bf16_nt_atr(M, N, K, A_T, a_off, a_mb_stride, n_mb, Bm, b_off, ldb, C, c_off, ldc, accumulate=False, b_kstride=64):
    """C[M,N] (+)= A . B^T with A read from its time-major K-tiled image (element (m, k) at A_T[a_off + (m // 64) * a_mb_stride + k * 64 + m % 64]):
    cruse_gemm_bf16_nt_atr -- dX straight from the gate-gradient tensor dgT, no row-major dgi."""
    if A_T.dtype != torch.bfloat16 or Bm.dtype != torch.bfloat16 or C.dtype != torch.float32:
        raise RuntimeError("gemm_bf16_nt_atr needs bf16 operands and an f32 result")
    check(lib.cruse_gemm_bf16_nt_atr(M, N, K, A_T.data_ptr() + 2 * a_off, a_mb_stride, n_mb, Bm.data_ptr() + 2 * b_off, ldb, b_kstride,
                                     C.data_ptr() + 4 * c_off, ldc, 1 if accumulate else 0, _stream()))
    return C


def gemm_bf16_nt_groups(M, N, K, G, A_hi, A_lo, lda, a_gstep, B_hi, B_lo, ldb, b_gstep, C, ldc, c_gstep, bias=None, bias_gstep=0,
                        accumulate=False, b_kstride=64):
    """G equal-shape products side by side along N in one launch (cruse_gemm_bf16_nt_groups): the GRU groups of a layer.
    bias: the first group's bias tensor (the others bias_gstep floats apart)."""
    for t_ in (A_hi, A_lo, B_hi, B_lo):
        if t_ is not None and t_.dtype != torch.bfloat16:
            raise RuntimeError("gemm_bf16_nt_groups needs bf16 planes")
    if C.dtype != torch.float32:
        raise RuntimeError("gemm_bf16_nt_groups: f32 result")
    check(lib.cruse_gemm_bf16_nt_groups(M, N, K, G, _p(A_hi), _p(A_lo), lda, a_gstep, _p(B_hi), _p(B_lo), ldb, b_kstride, b_gstep,
                                        _p(C), ldc, c_gstep, _p(bias), bias_gstep, 1 if accumulate else 0, _stream()))
    return C


def uniform_stride(tensors) -> Optional[int]:
    """element stride between equally shaped f32 tensors if it is the same for all consecutive pairs (the per-group parameters in the flat
    buffer), else None"""
    if len(tensors) < 2:
        return 0
    d = [tensors[i + 1].data_ptr() - tensors[i].data_ptr() for i in range(len(tensors) - 1)]
    if any(x != d[0] for x in d) or d[0] <= 0 or d[0] % 4:
        return None
    return d[0] // 4


def gemm_bf16_nt_seg(M, N, K, A_hi, A_lo, a_off, lda, B_hi, B_lo, b_off, ldb, C, c_off, ldc, seg, bias=None, accumulate=False,
                     b_kstride=64):
    """gemm_bf16_nt / gemm_bf16x3_nt (A_lo / B_lo None: plain bf16) on the rows of ONE TIME CHUNK: seg = (seg_len, seg_stride,
    seg_off), M = B * seg_len (cruse_gemm_bf16_nt_seg)."""
    lo = lambda t_, off: None if t_ is None else t_.data_ptr() + 2 * off
    check(lib.cruse_gemm_bf16_nt_seg(M, N, K, A_hi.data_ptr() + 2 * a_off, lo(A_lo, a_off), lda, B_hi.data_ptr() + 2 * b_off,
                                     lo(B_lo, b_off), ldb, b_kstride, C.data_ptr() + 4 * c_off, ldc, _p(bias),
                                     1 if accumulate else 0, seg[0], seg[1], seg[2], _stream()))


def cast_bf16(x, out=None):
    y = torch.empty(x.shape, device=x.device, dtype=torch.bfloat16) if out is None else out
    check(lib.cruse_cast_bf16(_p(x), _p(y), x.numel(), _stream()))
    return y


def ktile_bf16(x, rows, cols, split=False, out=None):
    """x [rows, cols] f32 -> K-tiled bf16 [ceil(cols/64), rows, 64] (k = column index, zero padded): no transposition.
    split: returns (hi, lo) planes with x ~= hi + lo.  out = (y, lo or None): write into these."""
    if out is not None:
        y, lo = out
    else:
        y = torch.empty((cols + 63) // 64, rows, 64, device=x.device, dtype=torch.bfloat16)
        lo = torch.empty_like(y) if split else None
    check(lib.cruse_ktile_bf16(_p(x), rows, cols, cols, _p(y), _p(lo), _stream()))
    return (y, lo) if split else y


def ktile_f16(x, rows, cols, split=False):
    """x [rows, cols] f32 -> K-tiled IEEE f16 [ceil(cols/64), rows, 64] (the layout of ktile_bf16): the B operand of gemm_f16_nt.
    split: returns (hi, lo) with lo = f16(x - hi) (cruse_ktile_f16_split): the planes of gemm_f16_nt(B_lo=...)"""
    y = torch.empty((cols + 63) // 64, rows, 64, device=x.device, dtype=torch.float16)
    if split:
        lo = torch.empty_like(y)
        check(lib.cruse_ktile_f16_split(_p(x), rows, cols, cols, _p(y), _p(lo), _stream()))
        return y, lo
    check(lib.cruse_ktile_f16(_p(x), rows, cols, cols, _p(y), _stream()))
    return y


_OUT16 = {torch.float16: 1, torch.bfloat16: 2}        # CRUSE_DT_F16 / CRUSE_DT_BF16


def _gemm_nt_out16(M, N, K, A_hi, A_lo, a_off, lda, a_kstride, B_hi, B_lo, b_off, ldb, b_kstride, C, c_off, ldc, bias, operands_f16):
    """2-byte result rows (cruse_gemm_nt_out16): C is an f16 / bf16 tensor -- the gi rows gru_seq_fwd reads in that dtype"""
    check(lib.cruse_gemm_nt_out16(M, N, K, A_hi.data_ptr() + 2 * a_off, None if A_lo is None else A_lo.data_ptr() + 2 * a_off, lda, a_kstride,
                                  B_hi.data_ptr() + 2 * b_off, None if B_lo is None else B_lo.data_ptr() + 2 * b_off, ldb, b_kstride,
                                  C.data_ptr() + 2 * c_off, ldc, _p(bias), 1 if operands_f16 else 0, _OUT16[C.dtype], _stream()))
    return C


def gemm_f16_nt(M, N, K, A, a_off, lda, B, b_off, ldb, C, c_off, ldc, bias=None, a_kstride=64, b_kstride=64, B_lo=None):
    """C[M,N] = A[M,K] . B[N,K]^T + bias on IEEE-f16 operands (cruse_gemm_f16_nt; offsets in elements): the forward gate projection
    in one pass.  B_lo: the low plane of B (ktile_f16(split=True)) -- two passes on the same accumulators (cruse_gemm_f16x2_nt)."""
    if A.dtype != torch.float16 or B.dtype != torch.float16 or (B_lo is not None and B_lo.dtype != torch.float16):
        raise RuntimeError("gemm_f16_nt needs f16 operands")
    if C.dtype in _OUT16:
        return _gemm_nt_out16(M, N, K, A, None, a_off, lda, a_kstride, B, B_lo, b_off, ldb, b_kstride, C, c_off, ldc, bias, True)
    if C.dtype != torch.float32:
        raise RuntimeError("gemm_f16_nt: the result is f32, or f16 / bf16 rows (cruse_gemm_nt_out16)")
    if B_lo is not None:
        check(lib.cruse_gemm_f16x2_nt(M, N, K, A.data_ptr() + 2 * a_off, lda, a_kstride, B.data_ptr() + 2 * b_off, B_lo.data_ptr() + 2 * b_off, ldb,
                                      b_kstride, C.data_ptr() + 4 * c_off, ldc, _p(bias), _stream()))
        return C
    check(lib.cruse_gemm_f16_nt(M, N, K, A.data_ptr() + 2 * a_off, lda, a_kstride, B.data_ptr() + 2 * b_off, ldb, b_kstride,
                                C.data_ptr() + 4 * c_off, ldc, _p(bias), _stream()))
    return C


def cast_bf16_padded(x, pad=64, split=False):
    """bf16 copy of x followed by `pad` zero elements, so a GEMM whose K is rounded up to 64 may read past the last row.
    split: returns (hi, lo) planes with x ~= hi + lo (both padded)."""
    n = x.numel()
    buf = torch.empty(n + pad, device=x.device, dtype=torch.bfloat16)
    lo = torch.empty(n + pad, device=x.device, dtype=torch.bfloat16) if split else None
    if pad:
        buf[n:].zero_()
        if split:
            lo[n:].zero_()
    check(lib.cruse_cast_bf16_split(_p(x), _p(buf), _p(lo), n, _stream()))
    return (buf, lo) if split else buf


def gemm_bf16x3_nt(M, N, K, A_hi, A_lo, a_off, lda, B_hi, B_lo, b_off, ldb, C, c_off, ldc, bias=None, accumulate=False,
                   a_kstride=64, b_kstride=64):
    """Split-bf16 form of gemm_bf16_nt: (A_hi + A_lo) . (B_hi + B_lo)^T without the lo.lo term; A_lo = None drops
    the A_lo.B_hi term too (two passes: only B's rounding is corrected)."""
    for t_ in (A_hi, A_lo, B_hi, B_lo):
        if t_ is not None and t_.dtype != torch.bfloat16:
            raise RuntimeError("gemm_bf16x3_nt needs bf16 planes")
    if C.dtype in _OUT16 and not accumulate:
        return _gemm_nt_out16(M, N, K, A_hi, A_lo, a_off, lda, a_kstride, B_hi, B_lo, b_off, ldb, b_kstride, C, c_off, ldc, bias, False)
    if C.dtype != torch.float32:
        raise RuntimeError("gemm_bf16x3_nt needs an f32 result (f16 / bf16 rows without accumulation: cruse_gemm_nt_out16)")
    check(lib.cruse_gemm_bf16x3_nt(M, N, K, A_hi.data_ptr() + 2 * a_off,
                                   None if A_lo is None else A_lo.data_ptr() + 2 * a_off, lda, a_kstride,
                                   B_hi.data_ptr() + 2 * b_off, B_lo.data_ptr() + 2 * b_off, ldb, b_kstride,
                                   C.data_ptr() + 4 * c_off, ldc, _p(bias), 1 if accumulate else 0, _stream()))


def transpose_bf16(x, rows, cols, shift_T=0, out=None):
    """x [rows, cols] f32 -> K-tiled time-major bf16 [ceil(rows/64), cols, 64] (zero-padded frames);
    shift_T > 0 reads row r-1 within each clip."""
    ldT = (rows + 63) // 64 * 64
    y = torch.empty(ldT // 64, cols, 64, device=x.device, dtype=torch.bfloat16) if out is None else out
    check(lib.cruse_transpose_bf16(_p(x), rows, cols, cols, _p(y), ldT, shift_T, _stream()))
    return y


# ---------------------------------------------------------------- GRU recurrence
def _ptr_array(ts: Sequence[torch.Tensor]):
    arr = (ctypes.c_void_p * len(ts))()
    for i, t in enumerate(ts):
        arr[i] = _p(t)
    return arr


STEP_SLOT0 = 100            # slots STEP_SLOT0 .. +STEP_SLOTS-1: the scratches gru_step_ws_clear() clears with ONE launch
STEP_SLOTS = 4


def _gru_ws(B, G, Hg, dev, slot: int):
    """-> (panel scratch pointer, status word pointer).  slot 0 is the library's classic workspace (sticky header +
    panels); slots > 0 -- concurrent recurrences on other streams -- get their own panel scratch and share slot 0's
    status word, so one word still guards the optimizer step.  Slots STEP_SLOT0.. are carved from one buffer
    (gru_step_ws_clear)."""
    nbytes = lib.cruse_gru_ws_bytes(B, G, Hg)
    status = gru_header(dev).data_ptr()
    if slot == 0:
        return _ws("gru", nbytes, dev).data_ptr() + 256, status
    if STEP_SLOT0 <= slot < STEP_SLOT0 + STEP_SLOTS:
        per = (nbytes - 256 + 255) // 256 * 256
        buf = _ws(("gru_step", per), per * STEP_SLOTS, dev)
        return buf.data_ptr() + (slot - STEP_SLOT0) * per, status
    panels = _ws(("gru_panels", slot), nbytes - 256, dev)
    return panels.data_ptr(), status


def gru_step_ws_clear(B, G, Hg, dev) -> None:
    """One launch clears the panel scratches of slots STEP_SLOT0 .. STEP_SLOT0 + STEP_SLOTS - 1: the (up to) four recurrences of
    a training step then run with zeroed=True -- no memset launch in front of each (4 x ~14 us on the main stream)."""
    nbytes = lib.cruse_gru_ws_bytes(B, G, Hg)
    per = (nbytes - 256 + 255) // 256 * 256
    zero_(_ws(("gru_step", per), per * STEP_SLOTS, torch.device(dev)))


def _off(t: torch.Tensor, elems: int) -> int:
    return t.data_ptr() + elems * t.element_size()


def gru_seq_fwd(gi, w_hh: List[torch.Tensor], b_hh: List[torch.Tensor], B, T, G, Hg, prec, save=True, slot=0, xcd_rot=0,
                h0=None, out=None, chunk=None, wide=False, zeroed=False):
    """-> (h, coef, an, z); the last three are None when save is False (inference).  slot / xcd_rot: see
    cruse_gru_seq_fwd_on (concurrent recurrences).
    h0 [B, G*Hg] ("cat" layout: feature = group*Hg + unit): initial state (cust_conv.py:305-325); None = 0.
    chunk = (t0, n): run only frames [t0, t0+n) of the [B,T] tensors into `out` = (h, coef, an, z) from the call that ran
    the frames before them -- the initial state is then h[:, t0-1] (h0 for t0 == 0).  Consecutive chunks reproduce the
    single launch (cruse_gru_seq_fwd_ex).  wide: chains of 16 clips (half the workgroups; same results).
    zeroed: the slot's scratch is already clear
    (gru_step_ws_clear)."""
    if gi.dtype not in (torch.float32, torch.float16):
        raise RuntimeError(f"gru_seq_fwd: gi must be f32 (or f16 rows: cruse_gru_seq_fwd_gi16), got {gi.dtype}")
    if gi.dtype == torch.float16 and (h0 is not None or chunk is not None or wide):
        raise RuntimeError("gru_seq_fwd: f16 gi rows are served for whole sequences from h0 = 0 on chains of 8 clips")
    dev = gi.device
    H = G * Hg
    if out is not None:
        h, coef, an, z = out
    else:
        h = torch.empty(B, T, H, device=dev, dtype=torch.float32)
        if save:
            cdt = torch.bfloat16 if prec_code(prec) == PREC_BF16 else torch.float32
            coef = torch.empty(B, T, 3 * H, device=dev, dtype=cdt)
            an = torch.empty_like(h); z = torch.empty_like(h)
        else:
            coef = an = z = None
    t0, n = chunk if chunk is not None else (0, T)
    if not (0 <= t0 and n > 0 and t0 + n <= T):
        raise RuntimeError(f"gru_seq_fwd: chunk ({t0}, {n}) outside [0, {T})")
    h0p, h0s = None, 0
    if t0 > 0:
        h0p, h0s = _off(h, (t0 - 1) * H), T * H
    elif h0 is not None:
        if wide:
            raise RuntimeError("gru_seq_fwd: wide chains take no caller-supplied h0 (their hand-off needs |h| < 1); chunk "
                               "continuations are fine")
        if tuple(h0.shape) != (B, H) or h0.dtype != torch.float32:
            raise RuntimeError(f"gru_seq_fwd: h0 must be f32 [{B}, {H}], got {tuple(h0.shape)}")
        _p(h0)
        h0p, h0s = h0.data_ptr(), H
    panels, status = _gru_ws(B, G, Hg, dev, slot)
    wa, ba = _ptr_array(w_hh), _ptr_array(b_hh)
    _p(gi); _p(h); _p(coef); _p(an); _p(z)
    opt = lambda t_, k: None if t_ is None else _off(t_, t0 * k)
    if gi.dtype == torch.float16:
        check(lib.cruse_gru_seq_fwd_gi16(gi.data_ptr(), 1, ctypes.cast(wa, ctypes.c_void_p), ctypes.cast(ba, ctypes.c_void_p), h.data_ptr(),
                                         None if coef is None else coef.data_ptr(), None if an is None else an.data_ptr(),
                                         None if z is None else z.data_ptr(), B, T, T, G, Hg, prec_code(prec), panels, 1 if zeroed else 0,
                                         status, xcd_rot, _stream()))
        return h, coef, an, z
    check(lib.cruse_gru_seq_fwd_ex(_off(gi, t0 * 3 * H), ctypes.cast(wa, ctypes.c_void_p), ctypes.cast(ba, ctypes.c_void_p),
                                   _off(h, t0 * H), opt(coef, 3 * H), opt(an, H), opt(z, H), h0p, h0s, B, n, T, G, Hg,
                                   prec_code(prec), 16 if wide else 0, panels, 1 if zeroed else 0,
                                   status, xcd_rot, _stream()))
    return h, coef, an, z


def dgi_buffer(rows, G, Hg, device, slabs=3):
    """bf16 [rows, G, slabs, Hg] gate-gradient rows (slabs 3: r, z, n_i; 4: + n_h, see cruse_gru_seq_bwd_ex), followed by the
    zero K-padding the dX GEMM may read (K = 3*Hg rounded to 64; with 4 slabs the run-over lands in the n_h slab)."""
    n = rows * G * slabs * Hg
    pad = 64 if ((3 * Hg) % 64 and slabs == 3) else 0
    buf = torch.empty(n + pad, device=device, dtype=torch.bfloat16)
    if pad:
        buf[n:].zero_()
    return buf[:n].view(rows, G, slabs, Hg)


def gru_seq_bwd(dout, w_hh: List[torch.Tensor], coef, z, B, T, G, Hg, prec, slot=0, xcd_rot=0, an=None, want_dgi=False,
                out=None, chunk=None, dg_slabs=3, wide=False, zeroed=False):
    """dout [B,T,H] -> dh [B,T,H] (total gradient reaching every h_t).  want_dgi (CRUSE_PREC_BF16, with the a_n rows):
    -> (dh, dgi) with dgi = dh * (c_r, c_z, a_n) in bf16 written by the recurrence itself (cruse_gru_seq_bwd_on).
    chunk = (t0, n): only frames [t0, t0+n), into out = dh (or (dh, dgi)); chunks are run from the LAST to the first, and
    every chunk but the last picks the gradient carried across its end up from dh[:, t0+n] (cruse_gru_seq_bwd_ex)."""
    H = G * Hg
    dgi = None
    if out is not None:
        dh, dgi = out if want_dgi else (out, None)
    else:
        dh = torch.empty_like(dout)
        if want_dgi:
            dgi = dgi_buffer(B * T, G, Hg, dout.device, dg_slabs)
    if want_dgi and (an is None or prec_code(prec) != PREC_BF16):
        raise RuntimeError("gru_seq_bwd: want_dgi needs the a_n rows and the bf16 mode")
    t0, n = chunk if chunk is not None else (0, T)
    if not (0 <= t0 and n > 0 and t0 + n <= T):
        raise RuntimeError(f"gru_seq_bwd: chunk ({t0}, {n}) outside [0, {T})")
    carry = 1 if t0 + n < T else 0
    steps = n + carry                                   # a carried chunk re-visits frame t0+n to pick up its dh
    panels, status = _gru_ws(B, G, Hg, dout.device, slot)
    wa = _ptr_array(w_hh)
    _p(dout); _p(coef); _p(z); _p(dh); _p(an); _p(dgi)
    check(lib.cruse_gru_seq_bwd_ex(_off(dout, t0 * H), ctypes.cast(wa, ctypes.c_void_p), _off(coef, t0 * 3 * H), _off(z, t0 * H),
                                   _off(dh, t0 * H), _off(an, t0 * H) if want_dgi else None,
                                   None if dgi is None else _off(dgi, t0 * dg_slabs * H), dg_slabs, carry, B, steps, T, G, Hg, prec_code(prec),
                                   16 if wide else 0, panels, 1 if zeroed else 0, status, xcd_rot, _stream()))
    return (dh, dgi) if want_dgi else dh


def gru_gate_grads(dh, coef, an, rows, G, Hg, prec):
    if (coef.dtype == torch.bfloat16) != (prec_code(prec) == PREC_BF16):
        raise RuntimeError("gru_gate_grads: coef dtype does not match the precision mode")
    dgi = torch.empty(coef.shape, device=coef.device, dtype=torch.float32)
    dgh = torch.empty_like(dgi)
    check(lib.cruse_gru_gate_grads(_p(dh), _p(coef), _p(an), _p(dgi), _p(dgh), rows, G, Hg, prec_code(prec), _stream()))
    return dgi, dgh


def gru_gate_grads_bf16(dh, coef, an, rows, G, Hg, db_ih, db_hh, want_dgi=True, want_dgT=True):
    """bf16 gate gradients for gemm_bf16_nt: returns (dgi [rows,G,3,Hg] bf16, dgT [ldT/64,G,4,Hg,64] bf16, ldT) and
    accumulates the bias gradients into the per-group tensors db_ih[g], db_hh[g].  want_dgi / want_dgT False: that output
    is not made (None) -- the recurrence may have written dgi itself (gru_seq_bwd(want_dgi=True))."""
    if coef.dtype != torch.bfloat16:
        raise RuntimeError("gru_gate_grads_bf16 needs the bf16 coefficients of CRUSE_PREC_BF16")
    ldT = (rows + 63) // 64 * 64
    dgi = dgi_buffer(rows, G, Hg, dh.device) if want_dgi else None
    dgT = torch.empty(ldT // 64, G, 4, Hg, 64, device=dh.device, dtype=torch.bfloat16) if want_dgT else None
    check(lib.cruse_gru_gate_grads_bf16(_p(dh), _p(coef), _p(an), _p(dgi), _p(dgT), ldT, _ptr_array(db_ih),
                                        _ptr_array(db_hh), rows, G, Hg, _stream()))
    return dgi, dgT, ldT


def gru_status() -> int:
    """0 if no recurrence hand-off EVER timed out on any device of this process (synchronises).  The status word is
    sticky: neither the library nor later launches clear it (gru_status_reset and the per-step latch of step_health do)."""
    bad = 0
    for h in _gru_hdr.values():
        bad |= int(h[:4].view(torch.int32).item())
    return bad


def gru_status_word(device, B: int = 0, G: int = 0, Hg: int = 0) -> torch.Tensor:
    """The device status word itself (uint8[4] view of gru_header) -- handed to step_health / adam_step as skip_flag so that
    a step whose recurrence timed out never reaches the parameters.  (B, G, Hg are accepted for the round-3 callers.)"""
    return gru_header(device)[:4]


def gru_status_reset() -> None:
    for h in _gru_hdr.values():
        h.zero_()


def check_gru_status() -> None:
    """Raise (CRUSE_E_TIMEOUT) if a recurrence hand-off timed out since the last reset: the outputs of that launch
    were garbage.  Call once per epoch / before a checkpoint (synchronises)."""
    if gru_status() != 0:
        raise RuntimeError("cruse_hip error -5 (CRUSE_E_TIMEOUT): a GRU hand-off timed out -- the persistent recurrence "
                           "kernel's workgroups were not co-resident (another process or kernel holding CUs?); the "
                           "affected optimizer steps were skipped")


# ---------------------------------------------------------------- mask + loss, misc
def mask_loss(mask, nre, nim, cmag, rows, Fn, Fs, alpha=2.0, beta=1.0, want_dmask=False, want_dlogit=False,
              want_est=False):
    dev = mask.device
    loss_sum = torch.empty(1, device=dev, dtype=torch.float64)
    dmask = torch.empty(rows, Fn, device=dev, dtype=torch.float32) if want_dmask else None
    dlogit = torch.empty(rows, Fn, device=dev, dtype=torch.float32) if want_dlogit else None
    er = torch.empty(rows, Fs, device=dev, dtype=torch.float32) if want_est else None
    ei = torch.empty(rows, Fs, device=dev, dtype=torch.float32) if want_est else None
    check(lib.cruse_mask_loss_fwd(_p(mask), _p(nre), _p(nim), _p(cmag), rows, Fn, Fs, alpha, beta, _p(loss_sum),
                                  _p(dmask), _p(dlogit), _p(er), _p(ei), _stream()))
    return loss_sum, dmask, dlogit, er, ei


def mask_apply(mask, nre, nim, rows, Fn, Fs):
    er = torch.empty(rows, Fs, device=mask.device, dtype=torch.float32)
    ei = torch.empty_like(er)
    check(lib.cruse_mask_apply(_p(mask), _p(nre), _p(nim), rows, Fn, Fs, _p(er), _p(ei), _stream()))
    return er, ei


def mask_apply_bwd(dre, dim, nre, nim, mask, rows, Fn, Fs, through_sigmoid=True):
    out = torch.empty(rows, Fn, device=mask.device, dtype=torch.float32)
    check(lib.cruse_mask_apply_bwd(_p(dre), _p(dim), _p(nre), _p(nim), _p(mask), rows, Fn, Fs,
                                   1 if through_sigmoid else 0, _p(out), _stream()))
    return out


def mask_sdnr(mask, cre, cim, nre, nim, rows, Fn, Fs, B, snr_db, beta_db=20.0, want_dmask=False, want_dlogit=False):
    """sdnr (loss_func/loss.py:151-175) with the mask as gain -> (loss_sum f64[1] (divide by B*Fs), dmask, dlogit)."""
    dev = mask.device
    loss_sum = torch.empty(1, device=dev, dtype=torch.float64)
    dmask = torch.empty(rows, Fn, device=dev, dtype=torch.float32) if want_dmask else None
    dlogit = torch.empty(rows, Fn, device=dev, dtype=torch.float32) if want_dlogit else None
    check(lib.cruse_mask_sdnr_fwd(_p(mask), _p(cre), _p(cim), _p(nre), _p(nim), rows, Fn, Fs, B, snr_db, beta_db,
                                  _p(loss_sum), _p(dmask), _p(dlogit), _stream()))
    return loss_sum, dmask, dlogit


def sisnr_fwd(x, s, eps=1e-8):
    """-> (loss f64[1], coef [B,4]) for si_snr_loss(x, s) of train_base/loss.py:7-25."""
    B, L = x.shape
    if s.shape != x.shape:
        raise RuntimeError(f"Dimension mismatch when calculate si_snr, {tuple(x.shape)} vs {tuple(s.shape)}")
    mom = torch.empty(B, 5, device=x.device, dtype=torch.float64)
    loss = torch.empty(1, device=x.device, dtype=torch.float64)
    coef = torch.empty(B, 4, device=x.device, dtype=torch.float32)
    check(lib.cruse_sisnr_fwd(_p(x), _p(s), B, L, eps, _p(mom), _p(loss), _p(coef), _stream()))
    return loss, coef


def sisnr_bwd(x, s, coef, grad_scale=1.0):
    dx = torch.empty_like(x)
    B, L = x.shape
    check(lib.cruse_sisnr_bwd(_p(x), _p(s), _p(coef), B, L, grad_scale, _p(dx), _stream()))
    return dx


def wave_l1_mse(est, ref, mse: bool, want_grad=True):
    """-> (loss_sum f64[1], dest or None): torch.nn.L1Loss / MSELoss(reduction="mean") terms on waveforms -- the loss is
    loss_sum / est.numel(), dest its gradient wrt est (train_base/loss.py:3-4)."""
    if est.shape != ref.shape:
        raise RuntimeError(f"Dimension mismatch when calculate {'mse' if mse else 'l1'} loss, {tuple(est.shape)} vs {tuple(ref.shape)}")
    n = est.numel()
    loss = torch.empty(1, device=est.device, dtype=torch.float64)
    dest = torch.empty_like(est) if want_grad else None
    check(lib.cruse_wave_l1_mse(_p(est), _p(ref), n, 1 if mse else 0, 1.0 / n, _p(loss), _p(dest), _stream()))
    return loss, dest


def deepfilter_fwd(xr, xi, hr, hi, f_dim, t_dim, out=None):
    B, F, T = xr.shape
    o_r, o_i = (torch.empty_like(xr), torch.empty_like(xr)) if out is None else out
    check(lib.cruse_deepfilter_fwd(_p(xr), _p(xi), _p(hr), _p(hi), B, F, T, f_dim, t_dim, _p(o_r), _p(o_i), _stream()))
    return o_r, o_i


def deepfilter_bwd(dor, doi, xr, xi, hr, hi, f_dim, t_dim):
    B, F, T = xr.shape
    outs = [torch.empty_like(xr) for _ in range(4)]
    check(lib.cruse_deepfilter_bwd(_p(dor), _p(doi), _p(xr), _p(xi), _p(hr), _p(hi), B, F, T, f_dim, t_dim,
                                   *[_p(o) for o in outs], _stream()))
    return outs


def sigmoid_bwd(dmask, mask):
    out = torch.empty_like(mask)
    check(lib.cruse_sigmoid_bwd(_p(dmask), _p(mask), _p(out), mask.numel(), _stream()))
    return out


def axpby(out, x, y, a, b):
    check(lib.cruse_axpby(_p(out), _p(x), _p(y), a, b, out.numel(), _stream()))
    return out


def adam_step(p, g, m, v, lr, beta1, beta2, eps, weight_decay, step, grad_scale=1.0, max_norm=0.0, gsumsq=None,
              skip_flag=None, loss_check=None, skipped=None, loss_sum=None, loss_scale=1.0, loss_acc=None):
    """Fused Adam over flat buffers; the optional device-side guards are described at cruse_adam_step_guarded.
    skip_flag: a uint8[4k] / int32[k] tensor of k status words (any non-zero word skips the step).  `step` counts the CALLS;
    with `skipped` the bias corrections use step - skipped[0].  loss_acc f64[2] += (loss_sum * loss_scale, 1) when applied."""
    nw = 0 if skip_flag is None else skip_flag.numel() * skip_flag.element_size() // 4
    if loss_acc is not None and (loss_acc.dtype != torch.float64 or loss_acc.numel() < 2):
        raise RuntimeError("adam_step: loss_acc must be an f64[2] tensor (sum, applied steps)")
    check(lib.cruse_adam_step_guarded(_p(p), _p(g), _p(m), _p(v), p.numel(), lr, beta1, beta2, eps, weight_decay, step,
                                      grad_scale, max_norm, _p(gsumsq), _p(skip_flag), nw, _p(loss_check), _p(skipped),
                                      _p(loss_sum), float(loss_scale), _p(loss_acc), _stream()))


def step_health(gru_status, loss_sum, health, loss_acc=None, loss_scale=1.0) -> None:
    """health[0] = latched-and-cleared GRU status word, health[1] = non-finite loss; loss_acc += loss_sum * loss_scale
    (cruse_step_health).  health: int32[2]."""
    if health.dtype != torch.int32 or health.numel() < 2:
        raise RuntimeError("step_health needs an int32[2] health tensor")
    check(lib.cruse_step_health(_p(gru_status), _p(loss_sum), _p(health), _p(loss_acc), float(loss_scale), _stream()))


def gru_plan(B: int, groups: int, Hg: int, prec="bf16", fwd: bool = True) -> dict:
    """the launch plan of the recurrence for this shape (cruse_gru_plan): which kernels a batch size runs on"""
    out = (ctypes.c_int * 6)()
    check(lib.cruse_gru_plan(int(B), int(groups), int(Hg), prec_code(prec), 1 if fwd else 0, ctypes.cast(out, ctypes.c_void_p)))
    return {"clips_per_chain": out[0], "chains_per_group": out[1], "chains_per_launch": out[2], "launches": out[3],
            "workgroups_per_chain": out[4], "wide": bool(out[5])}


def cu_hog(nblocks: int, microseconds: float, clock_ghz: float = 2.1) -> None:
    """Test / probe rig: `nblocks` workgroups hold one CU each (128 KB of LDS) for about `microseconds` on the current stream
    (cruse_cu_hog) -- what a collective's channels or another tenant do to the persistent recurrences."""
    check(lib.cruse_cu_hog(int(nblocks), int(microseconds * 1e3 * clock_ghz), _stream()))


def zero_(t: torch.Tensor) -> torch.Tensor:
    """stream-ordered zero fill (kernel node; see cruse_zero)."""
    check(lib.cruse_zero(_p(t), t.numel() * t.element_size(), _stream()))
    return t


def accum_f64(acc: torch.Tensor, x: torch.Tensor) -> None:
    if acc.dtype != torch.float64 or x.dtype != torch.float64 or acc.numel() != x.numel():
        raise RuntimeError("accum_f64 needs two f64 tensors of the same size")
    check(lib.cruse_accum_f64(_p(acc), _p(x), acc.numel(), _stream()))


def counters_add(counters: Sequence[torch.Tensor], v: int = 1) -> None:
    """every int64 scalar tensor in `counters` += v with ONE launch (BatchNorm num_batches_tracked)."""
    for c in counters:
        if c.dtype != torch.int64:
            raise RuntimeError("counters_add needs int64 tensors")
    for i in range(0, len(counters), 32):
        part = counters[i:i + 32]
        check(lib.cruse_counters_add(ctypes.cast(_ptr_array(part), ctypes.c_void_p), len(part), v, _stream()))


def sumsq(x, out=None, accumulate=False):
    """sum of squares (f64[1]) of a flat f32 tensor: the squared total gradient norm of clip_grad_norm_."""
    if out is None:
        out = torch.empty(1, device=x.device, dtype=torch.float64)
    check(lib.cruse_sumsq(_p(x), x.numel(), _p(out), 1 if accumulate else 0, _stream()))
    return out
