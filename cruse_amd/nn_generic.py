"""HIP execution of the reference's general [B,C,H,W] conv blocks (cust_conv.py, mtfaa.py).

`HipSequential` is an nn.Sequential whose children are the SAME stock torch.nn modules the reference composes
(ConstantPad2d, Conv2d, ConvTranspose2d, BatchNorm2d, ReLU, PReLU, FreqUpsample) -- kept as parameter containers so
constructor arguments, state-dict keys and default initialisation match -- but whose forward walks the children and
runs fused HIP kernels (cruse_amd/csrc/generic.hip): zero pads and the nearest frequency upsampling fold into the
following convolution's gather index, an activation folds into the preceding BatchNorm (or convolution).
There is no CPU path.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import ctypes

import torch
import torch.nn as nn

from . import ops
from ._lib import DT_F16, DT_F32, check, lib

_p = ops._p
_stream = ops._stream


def _dt(x: torch.Tensor) -> int:
    """storage type code of an activation tensor (CRUSE_DT_*): f32, or f16 for BASELINE config 5."""
    if x.dtype == torch.float32:
        return DT_F32
    if x.dtype == torch.float16:
        return DT_F16
    raise RuntimeError(f"cruse_amd blocks take float32 or float16 activations, got {x.dtype}")


class _CastFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, to_f16):
        x = x.contiguous()
        want = torch.float16 if to_f16 else torch.float32
        # the gradient goes back in the INPUT's storage type: a cast that was a no-op forward (an f32 model calling
        # to_f32) must be a no-op backward too -- rounding an f32 gradient through f16 flushes 1e-9 to 0
        ctx.src_dtype = x.dtype
        if x.dtype == want:
            return x
        out = torch.empty_like(x, dtype=want)
        check(lib.cruse_cast_f16(_p(x), _p(out), x.numel(), 1 if to_f16 else 0, _stream()))
        return out

    @staticmethod
    def backward(ctx, g):
        if g.dtype == ctx.src_dtype:
            return g, None
        return _CastFn.apply(g, ctx.src_dtype == torch.float16), None


def to_f16(x):
    """f32 -> f16 activations (HIP kernel; the gradient comes back as f32): where a model's fp16 part begins."""
    return _CastFn.apply(x, True)


def to_f32(x):
    return _CastFn.apply(x, False)


def _pair(v):
    return (v, v) if isinstance(v, int) else (int(v[0]), int(v[1]))


# ------------------------------------------------------------------------------------------------------------------
# zeroed f32 accumulators for the parameter gradients (the kernels add into them with atomics)
# ------------------------------------------------------------------------------------------------------------------
_ZCHUNK = 1 << 16                  # floats per pool chunk (256 KB)
_ZPOOL = {}                        # device -> [chunk, next free offset]
import threading
_ZLOCK = threading.Lock()          # (autograd may run backward() on worker threads)


def _zeros_f32(shape, device) -> torch.Tensor:
    """A zero tensor carved from a pre-zeroed chunk: one fill launch per 64 K floats instead of one per tensor (the backward pass
    of config 5 made 77 of them per step, 4.7 us each, for tensors of 24-576 floats).  Chunks are never reused -- a slice stays
    valid for as long as anything (a .grad) refers to it -- and big requests get their own torch.zeros.
    Memory note: autograd's AccumulateGrad may keep a slice as the parameter's .grad, which keeps its 256 KB chunk alive;
    a step's ~80 small gradients share one or two chunks, so what stays pinned is bounded by a chunk or two per set of live .grad
    tensors -- but torch.save of such a .grad writes the whole chunk: clone() it first."""
    return _zeros_pool(shape, device, torch.float32)


def _zeros_f64(shape, device) -> torch.Tensor:
    """the same for the f64 accumulators of the BatchNorm passes (batch sums, backward sums): their fill launches were 24 of config 5's step"""
    return _zeros_pool(shape, device, torch.float64)


def _zeros_pool(shape, device, dtype) -> torch.Tensor:
    n = 1
    for d in shape:
        n *= int(d)
    if n > _ZCHUNK // 8 or torch.cuda.is_current_stream_capturing():
        return torch.zeros(shape, device=device, dtype=dtype)
    dev = torch.device(device)
    need = (n + 63) // 64 * 64                 # 256- / 512-byte slots
    with _ZLOCK:
        ent = _ZPOOL.get((dev, dtype))
        if ent is None or ent[1] + need > _ZCHUNK or ent[2] != torch.cuda.current_stream(dev):
            # (a chunk is zeroed on the stream that allocates it: slices are handed out on that stream only)
            ent = [torch.zeros(_ZCHUNK, device=dev, dtype=dtype), 0, torch.cuda.current_stream(dev)]
            _ZPOOL[(dev, dtype)] = ent
        out = ent[0][ent[1]:ent[1] + n].view(shape)
        ent[1] += need
    return out



# ------------------------------------------------------------------------------------------------------------------
# raw kernel wrappers
# ------------------------------------------------------------------------------------------------------------------
BN_STAT_REPLICAS = 8               # replicas of the batch sums a convolution's epilogue adds into (spreads its f64 atomics)


def _conv_raw(x, w, bias, out_hw, KH, KW, stride, dil, pt, pl, groups, up_w, transposed, Cout, act=0, slope=None,
              out=None, accumulate=False, bn_sums=None, residual=None):
    """bn_sums: a zeroed [BN_STAT_REPLICAS, 2 * Cout] f64 tensor -- the convolution also delivers the BatchNorm batch sums of its output;
    residual: y = conv(x) + residual (cruse_conv2d_nchw_ex)"""
    B, Cin, Hin, Win = x.shape
    Hout, Wout = out_hw
    y = torch.empty(B, Cout, Hout, Wout, device=x.device, dtype=x.dtype) if out is None else out
    if bn_sums is not None or residual is not None:
        check(lib.cruse_conv2d_nchw_ex(_p(x), _p(w), _p(bias), _p(residual), _p(y), B, Cin, Hin, Win, Cout, Hout, Wout, KH, KW, stride[0],
                                       stride[1], dil[0], dil[1], pt, pl, groups, up_w, 1 if transposed else 0, act, _p(slope), _p(bn_sums),
                                       bn_sums.shape[0] if bn_sums is not None else 1, _dt(x), _stream()))
        return y
    check(lib.cruse_conv2d_nchw(_p(x), _p(w), _p(bias), _p(y), B, Cin, Hin, Win, Cout, Hout, Wout, KH, KW, stride[0], stride[1],
                                dil[0], dil[1], pt, pl, groups, up_w, 1 if transposed else 0, act, _p(slope),
                                1 if accumulate else 0, _dt(x), _stream()))
    return y


def _wgrad_raw(S, Bg, dw, KH, KW, stride, dil, pt, pl, groups, up_w, db=None):
    """db (Conv2d form only: S = dy): the bias gradient accumulated in the same call (cruse_conv2d_nchw_wgrad_ex)"""
    N, CA, HS, WS = S.shape
    _, CB, HB, WB = Bg.shape
    if db is not None:
        check(lib.cruse_conv2d_nchw_wgrad_ex(_p(S), _p(Bg), _p(dw), _p(db), N, CA, HS, WS, CB, HB, WB, KH, KW, stride[0], stride[1], dil[0],
                                             dil[1], pt, pl, groups, up_w, _dt(S), _stream()))
        return
    check(lib.cruse_conv2d_nchw_wgrad(_p(S), _p(Bg), _p(dw), N, CA, HS, WS, CB, HB, WB, KH, KW, stride[0], stride[1], dil[0],
                                      dil[1], pt, pl, groups, up_w, _dt(S), _stream()))


# Hand-off tables between ADJACENT autograd nodes, keyed by a tensor's storage address (they used to hold strong references to
# whole activations and gradients until a consumer popped the entry -- an entry no HIP consumer takes, e.g. the last block's BatchNorm
# output, stayed alive across steps -- and were mutated without a lock although autograd runs backward on worker threads).  An entry
# now holds its key tensor WEAKLY and disappears with it (a finalizer pops the entry when the tensor dies, so the address cannot be
# matched by a later tensor that reuses it); what travels with the entry (batch sums, the BatchNorm's constants) lives exactly as long
# as the tensor it describes.  The version is the one at stash time: a gradient autograd accumulated into in place is the same tensor,
# but no longer what the sums describe.  One re-entrant lock (a finalizer may run while it is held).
import weakref
_HLOCK = threading.RLock()


def _stash(table, key_tensor, payload):
    k = key_tensor.data_ptr()

    def drop(ref, k=k, table=table):
        with _HLOCK:
            ent = table.get(k)
            if ent is not None and ent[0] is ref:
                del table[k]
    with _HLOCK:
        table[k] = (weakref.ref(key_tensor, drop), key_tensor._version, payload)


def _take(table, t):
    with _HLOCK:
        ent = table.pop(t.data_ptr(), None)
    if ent is None:
        return None
    src = ent[0]()
    if src is not None and src.shape == t.shape and src.dtype == t.dtype and ent[1] == t._version:
        return ent[2]
    return None


def handoff_entries() -> int:
    """entries waiting in the hand-off tables (tests: 0 once a step's tensors are gone)"""
    with _HLOCK:
        return len(_DX_SUMS) + len(_BN_SUMS) + len(_BN_OUT) + len(_BN_R)


# Bias gradient of a convolution that feeds a BatchNorm: the BatchNorm's backward apply pass (cruse_bn_nchw_bwd_ex) sums the dx it stores per
# channel -- the convolution's backward finds that sum here instead of re-reading dx (one 17 us pass per Conv2d -> BatchNorm2d pair at the
# config-5 shape).
_DX_SUMS = {}


def _stash_dx_sum(dx, sums):
    _stash(_DX_SUMS, dx, sums)


def _take_dx_sum(dy):
    return _take(_DX_SUMS, dy)


# ... and the other way round: a convolution that feeds a training-mode BatchNorm delivers that BatchNorm's batch sums from its epilogue
_BN_SUMS = {}


def _stash_bn_sums(y, sums):
    _stash(_BN_SUMS, y, sums)


def _take_bn_sums(x):
    return _take(_BN_SUMS, x)


# ... and backwards again: a convolution whose input came out of a training-mode BatchNorm (+ act) computes, in its data-gradient kernel's
# epilogue, that BatchNorm's backward sums (cruse_conv2d_nchw_bnbwd) -- the BatchNorm's forward leaves what that takes under its output's
# address (_BN_OUT), the convolution's backward leaves the sums under its input gradient's address (_BN_R)
_BN_OUT = {}
_BN_R = {}
BN_BWD_REPLICAS = 8


def _channel_sum(dy, out):
    N, C = dy.shape[:2]
    check(lib.cruse_nchw_channel_sum(_p(dy), N, C, dy[0, 0].numel(), _p(out), _dt(dy), _stream()))


class _ConvFn(torch.autograd.Function):
    """Conv2d (transposed=False) or ConvTranspose2d (True) on NCHW with folded zero pads / upsampling."""

    @staticmethod
    def forward(ctx, x, w, bias, cfg, res=None):
        """res: a tensor of the output's shape added to it (the residual of a block); its gradient is the output's"""
        x = x.contiguous(); w = w.contiguous().float()
        if res is not None:
            res = res.contiguous()
            if res.dtype != x.dtype:
                raise RuntimeError("conv + residual: storage types differ")
        (stride, dil, pt, pl, groups, up_w, transposed, out_hw) = cfg[:8]
        want_bn = len(cfg) > 8 and cfg[8]
        KH, KW = w.shape[2], w.shape[3]
        Cout = w.shape[1] * groups if transposed else w.shape[0]
        sums = _zeros_f64((BN_STAT_REPLICAS, 2 * Cout), x.device) if want_bn else None
        if res is not None and tuple(res.shape) != (x.shape[0], Cout, out_hw[0], out_hw[1]):
            raise RuntimeError(f"conv + residual: residual shape {tuple(res.shape)} is not the output's")
        y = _conv_raw(x, w, bias, out_hw, KH, KW, stride, dil, pt, pl, groups, up_w, transposed, Cout, bn_sums=sums, residual=res)
        if want_bn:
            _stash_bn_sums(y, sums)
        ctx.save_for_backward(x, w)
        ctx.cfg, ctx.has_bias, ctx.has_res = cfg, bias is not None, res is not None
        ctx.bn_src = _take(_BN_OUT, x) if (x.dtype == torch.float16 and up_w == 1) else None
        ctx.tap = len(cfg) > 9 and cfg[9]
        if ctx.tap:
            # the input handed on beside the output: a block that ALSO adds its input to its result (TFCM_Block) takes the residual from here,
            # so that the residual path's gradient arrives in THIS backward and is added inside the data-gradient kernel (no accumulation pass)
            return y, x.view(x.shape)
        return y

    @staticmethod
    def backward(ctx, dy, dtap=None):
        x, w = ctx.saved_tensors
        (stride, dil, pt, pl, groups, up_w, transposed, out_hw) = ctx.cfg[:8]
        dy = dy.contiguous()
        if dtap is not None:
            dtap = dtap.contiguous()
            if dtap.dtype != x.dtype:
                dtap = _CastFn.apply(dtap, x.dtype == torch.float16)
        if dy.dtype != x.dtype:
            dy = _CastFn.apply(dy, x.dtype == torch.float16)
        KH, KW = w.shape[2], w.shape[3]
        B, Cin, Hin, Win = x.shape
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            fold = dtap if (dtap is not None and up_w == 1) else None           # the tapped input's gradient rides the data-gradient kernel
            if ctx.bn_src is not None and fold is None and up_w == 1:
                # the input was a BatchNorm (+ act) output: the data-gradient kernel also delivers that BatchNorm's backward sums
                bx, bmean, brstd, bgamma, bbeta, bslope, bact = ctx.bn_src
                dx = torch.empty_like(x)
                r = _zeros_f64((BN_BWD_REPLICAS, 4, Cin), x.device)
                got = ctypes.c_int(0)
                check(lib.cruse_conv2d_nchw_bnbwd(_p(dy), _p(w), _p(dx), B, dy.shape[1], dy.shape[2], dy.shape[3], Cin, Hin, Win, KH, KW, stride[0],
                                                  stride[1], dil[0], dil[1], pt, pl, groups, 0 if transposed else 1, _p(bx), _p(bmean), _p(brstd),
                                                  _p(bgamma), _p(bbeta), _p(bslope), bact, _p(r), BN_BWD_REPLICAS, ctypes.addressof(got), _dt(x),
                                                  _stream()))
                if got.value:
                    _stash(_BN_R, dx, r)
            elif not transposed:
                dxu = _conv_raw(dy, w, None, (Hin, Win * up_w), KH, KW, stride, dil, pt, pl, groups, 1, True, Cin, residual=fold)
                if up_w > 1:
                    dx = torch.empty_like(x)
                    check(lib.cruse_downsum_w(_p(dxu), B * Cin * Hin, Win, up_w, _p(dx), _dt(dx), _stream()))
                else:
                    dx = dxu
            else:
                dx = _conv_raw(dy, w, None, (Hin, Win), KH, KW, stride, dil, pt, pl, groups, 1, False, Cin, residual=fold)
            if dtap is not None and fold is None:
                dx = add(dx, dtap)
        pre = _take_dx_sum(dy)
        want_db = ctx.has_bias and ctx.needs_input_grad[2]
        if want_db:
            db = pre if pre is not None else _zeros_f32((dy.shape[1],), dy.device)
        db_in_wgrad = want_db and pre is None and ctx.needs_input_grad[1] and not transposed
        if ctx.needs_input_grad[1]:
            dw = _zeros_f32(tuple(w.shape), w.device)
            if not transposed:
                _wgrad_raw(dy, x, dw, KH, KW, stride, dil, pt, pl, groups, up_w, db=db if db_in_wgrad else None)
            else:
                _wgrad_raw(x, dy, dw, KH, KW, stride, dil, pt, pl, groups, 1)
        if want_db and pre is None and not db_in_wgrad:
            _channel_sum(dy, db)
        return dx, dw, db, None, (dy if ctx.has_res and ctx.needs_input_grad[4] else None)


def conv2d(x, w, bias=None, stride=(1, 1), dilation=(1, 1), pad=(0, 0, 0, 0), groups=1, up_w=1, bn_stats=False, residual=None, tap_input=False):
    """F.conv2d on zero-padded x; pad = (top, bottom, left, right); up_w: nearest upsampling of W folded in front.
    bn_stats: a training-mode BatchNorm2d consumes the output next -- its batch sums come out of this call (see _BN_SUMS)."""
    stride, dilation = _pair(stride), _pair(dilation)
    pt, pb, pl, pr = pad
    _, _, Hin, Win = x.shape
    KH, KW = w.shape[2], w.shape[3]
    Hout = (Hin + pt + pb - dilation[0] * (KH - 1) - 1) // stride[0] + 1
    Wout = (Win * up_w + pl + pr - dilation[1] * (KW - 1) - 1) // stride[1] + 1
    if Hout <= 0 or Wout <= 0:
        raise RuntimeError(f"conv2d: kernel {KH}x{KW} does not fit the padded input {Hin}x{Win}")
    # tap_input: returns (y, x'): x' is x handed through the convolution's autograd node (see _ConvFn.forward)
    return _ConvFn.apply(x, w, bias, (stride, dilation, pt, pl, groups, up_w, False, (Hout, Wout), bool(bn_stats), bool(tap_input)), residual)


def conv_transpose2d(x, w, bias=None, stride=(1, 1), padding=(0, 0), output_padding=(0, 0), dilation=(1, 1), groups=1,
                     pre_pad_top=0):
    """F.conv_transpose2d; pre_pad_top: zero rows in front of x along H (ConstantPad2d before the layer), stride_h == 1."""
    stride, dilation, padding, output_padding = _pair(stride), _pair(dilation), _pair(padding), _pair(output_padding)
    if pre_pad_top and stride[0] != 1:
        raise RuntimeError("conv_transpose2d: a folded top pad needs stride 1 along H")
    _, _, Hin, Win = x.shape
    KH, KW = w.shape[2], w.shape[3]
    Hout = (Hin + pre_pad_top - 1) * stride[0] - 2 * padding[0] + dilation[0] * (KH - 1) + output_padding[0] + 1
    Wout = (Win - 1) * stride[1] - 2 * padding[1] + dilation[1] * (KW - 1) + output_padding[1] + 1
    return _ConvFn.apply(x, w, bias, (stride, dilation, padding[0] - pre_pad_top, padding[1], groups, 1, True, (Hout, Wout)))


# ------------------------------------------------------------------------------------------------------------------
# BatchNorm2d (+ ReLU / PReLU) and bare activations
# ------------------------------------------------------------------------------------------------------------------
class _BnActFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, slope, mean, rstd, act, training):
        x = x.contiguous()
        N, C = x.shape[:2]
        HW = x[0, 0].numel()
        y = torch.empty_like(x)
        check(lib.cruse_bn_nchw_fwd(_p(x), _p(mean), _p(rstd), _p(gamma), _p(beta), _p(slope), act, N, C, HW, _p(y), _dt(x), _stream()))
        ctx.save_for_backward(x, gamma, beta, slope, mean, rstd)
        ctx.act, ctx.training = act, training
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, beta, slope, mean, rstd = ctx.saved_tensors
        dy = dy.contiguous()
        if dy.dtype != x.dtype:
            dy = _CastFn.apply(dy, x.dtype == torch.float16)
        N, C = x.shape[:2]
        HW = x[0, 0].numel()
        dx = torch.empty_like(x)
        scratch = torch.empty(3 * C, device=x.device, dtype=torch.float64)
        dg = _zeros_f32((C,), x.device) if gamma is not None else None
        db = _zeros_f32((C,), x.device) if beta is not None else None
        ds = _zeros_f32((C,), x.device) if slope is not None else None
        check(lib.cruse_bn_nchw_bwd(_p(dy), _p(x), _p(mean), _p(rstd), _p(gamma), _p(beta), _p(slope), ctx.act,
                                    1 if ctx.training else 0, N, C, HW, _p(scratch), _p(dx), _p(dg), _p(db), _p(ds), _dt(x), _stream()))
        return dx, dg, db, ds, None, None, None, None


def _fused_bn_takes_replicas(x) -> bool:
    """cruse_bn_nchw_fwd_train / cruse_bn_nchw_bwd_ex fold replicated f64 sums only in their f16 kernels, whose grid is (blocks, N * C): N * C < 65536"""
    return x.dtype == torch.float16 and x.shape[0] * x.shape[1] < 65536


class _BnTrainActFn(torch.autograd.Function):
    """Training-mode BatchNorm2d (+ act) from the batch sums: statistics finalised inside the forward kernel (running statistics and batch
    counter updated there), parameter gradients and -- want_dx_sum -- the preceding convolution's bias gradient inside the backward apply
    pass (cruse_bn_nchw_fwd_train / cruse_bn_nchw_bwd_ex)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, slope, sums, eps, momentum, rmean, rvar, nbt, act, want_dx_sum):
        N, C = x.shape[:2]
        HW = x[0, 0].numel()
        y = torch.empty_like(x)
        mean = torch.empty(C, device=x.device, dtype=torch.float32)
        rstd = torch.empty(C, device=x.device, dtype=torch.float32)
        if sums.dim() == 2 and not _fused_bn_takes_replicas(x):
            sums = sums.sum(0)          # (the kernels that fold replicated sums are the f16 ones with N * C < 65536)
        check(lib.cruse_bn_nchw_fwd_train(_p(x), _p(sums), sums.shape[0] if sums.dim() == 2 else 1, float(eps), float(momentum), _p(gamma), _p(beta),
                                          _p(slope), act, N, C, HW, _p(y),
                                          _p(mean), _p(rstd), _p(rmean), _p(rvar), _p(nbt), _dt(x), _stream()))
        ctx.save_for_backward(x, gamma, beta, slope, mean, rstd)
        ctx.act, ctx.want_dx_sum = act, bool(want_dx_sum)
        if _fused_bn_takes_replicas(x):
            _stash(_BN_OUT, y, (x, mean, rstd, gamma, beta, slope, act))
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, beta, slope, mean, rstd = ctx.saved_tensors
        dy = dy.contiguous()
        if dy.dtype != x.dtype:
            dy = _CastFn.apply(dy, x.dtype == torch.float16)
        N, C = x.shape[:2]
        HW = x[0, 0].numel()
        dx = torch.empty_like(x)
        delivered = _take(_BN_R, dy)                       # the producer of dy has already formed the sums (see _BN_OUT / _BN_R)
        if delivered is not None and not _fused_bn_takes_replicas(x):
            delivered = None                               # (not produced for such shapes; recomputed by the kernel's own reduce pass)
        scratch = delivered if delivered is not None else _zeros_f64((4 * C,), x.device)
        dg = _zeros_f32((C,), x.device)
        db = _zeros_f32((C,), x.device)
        ds = _zeros_f32((C,), x.device) if slope is not None else None
        dxs = _zeros_f32((C,), x.device) if ctx.want_dx_sum else None
        check(lib.cruse_bn_nchw_bwd_ex(_p(dy), _p(x), _p(mean), _p(rstd), _p(gamma), _p(beta), _p(slope), ctx.act, 1, N, C, HW, _p(scratch), 1,
                                       delivered.shape[0] if delivered is not None else 0, _p(dx), _p(dg), _p(db), _p(ds), _p(dxs), _dt(x), _stream()))
        if dxs is not None:
            _stash_dx_sum(dx, dxs)
        return dx, dg, db, ds, None, None, None, None, None, None, None, None


def _act_code(m) -> Tuple[int, Optional[torch.Tensor]]:
    if m is None:
        return 0, None
    if isinstance(m, nn.ReLU):
        return 1, None
    if isinstance(m, nn.PReLU):
        return 2, m.weight
    if isinstance(m, nn.Sigmoid):
        return 3, None
    raise RuntimeError(f"HipSequential: activation {type(m).__name__} has no HIP kernel (ReLU, PReLU, Sigmoid)")


def batchnorm_act(x, bn: Optional[nn.BatchNorm2d], act_module=None, conv_bias_in_front: bool = False):
    """bn(x) then act, fused; bn None: activation only.  Training mode updates the running statistics like torch.
    conv_bias_in_front: x is the output of a convolution with a bias -- its bias gradient (the channel sums of this BatchNorm's input
    gradient) is then delivered by the backward pass of this call (see _DX_SUMS)."""
    act, slope = _act_code(act_module)
    C = x.shape[1]
    if act == 2 and slope.numel() != C:
        if slope.numel() != 1:
            raise RuntimeError("PReLU: num_parameters must be 1 or the channel count")
        slope = slope.expand(C).contiguous()
    if bn is None:
        return _BnActFn.apply(x, None, None, slope, None, None, act, False)
    training = bn.training or bn.running_mean is None
    xc = x.contiguous()
    N = xc.shape[0]
    HW = xc[0, 0].numel()
    if training:
        fusable = bn.weight is not None and bn.bias is not None and (bn.momentum is not None or not (bn.training and bn.running_mean is not None))
        sums = _take_bn_sums(xc)
        if sums is not None and not fusable:
            sums = sums.sum(0)
        if sums is None:
            sums = _zeros_f64((2 * C,), x.device)
            check(lib.cruse_bn_nchw_stats_ex(_p(xc), N, C, HW, _p(sums), 1, _dt(xc), _stream()))
        upd = bn.training and bn.running_mean is not None
        mom = bn.momentum if bn.momentum is not None else 0.1
        if fusable:
            # (momentum None = cumulative average: the separate finalize below)
            return _BnTrainActFn.apply(xc, bn.weight, bn.bias, slope, sums, bn.eps, mom, bn.running_mean if upd else None,
                                       bn.running_var if upd else None, bn.num_batches_tracked if upd else None, act,
                                       conv_bias_in_front)
        mean, rstd = ops.bn_finalize(sums, N * HW, C, bn.eps, mom, bn.running_mean if upd else None,
                                     bn.running_var if upd else None)
        if upd and bn.num_batches_tracked is not None:
            ops.counters_add([bn.num_batches_tracked], 1)
    else:
        mean, rstd = ops.bn_eval_stats(bn.running_mean, bn.running_var, bn.eps)
    return _BnActFn.apply(xc, bn.weight, bn.bias, slope, mean, rstd, act, training)


class _AddFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        a = a.contiguous(); b = b.contiguous()
        out = torch.empty_like(a)
        check(lib.cruse_add_nchw(_p(a), _p(b), _p(out), a.numel(), _dt(a), _stream()))
        return out

    @staticmethod
    def backward(ctx, g):
        return g, g


def add(a, b):
    if a.shape != b.shape or a.dtype != b.dtype:
        raise RuntimeError(f"add: shape / dtype mismatch {tuple(a.shape)} {a.dtype} vs {tuple(b.shape)} {b.dtype}")
    return _AddFn.apply(a, b)


class FreqUpsample(nn.Module):
    """cust_conv.py:177-184: nearest interpolation along the last axis; inside a HipSequential it is folded into the
    following Conv2d, stand-alone it is a 1x1 identity-free gather (conv kernel with a delta weight is avoided: a
    dedicated call with KH = KW = 1 and a ones depthwise weight)."""

    def __init__(self, factor, mode="nearest"):
        super().__init__()
        self.f = float(factor)
        self.mode = mode
        if mode != "nearest" or int(self.f) != self.f:
            raise RuntimeError("cruse_amd FreqUpsample: integer nearest-neighbour factors only")

    def forward(self, x):
        C = x.shape[1]
        ones = torch.ones(C, 1, 1, 1, device=x.device, dtype=torch.float32)          # (weights stay f32 in either storage mode)
        return conv2d(x, ones, None, groups=C, up_w=int(self.f))


# ------------------------------------------------------------------------------------------------------------------
# the executor
# ------------------------------------------------------------------------------------------------------------------
def _pad4(m: nn.ConstantPad2d):
    if float(getattr(m, "value", 0.0)) != 0.0:
        raise RuntimeError("HipSequential: only zero ConstantPad2d folds into the convolution")
    l, r, t, b = m.padding
    return int(t), int(b), int(l), int(r)


def run_sequential(mods: Sequence[nn.Module], x: torch.Tensor, tap_first_conv: bool = False):
    """tap_first_conv: returns (out, x') with x' the sequence's input handed through its first Conv2d's autograd node (conv2d(tap_input=True))"""
    if not x.is_cuda:
        raise RuntimeError("cruse_amd blocks need tensors on the HIP device (no CPU fallback)")
    mods = list(mods)
    tapped = None
    i = 0
    pad = (0, 0, 0, 0)
    up = 1
    biased_conv = False                           # the previous module was a Conv2d with a trainable bias (and nothing came in between)
    while i < len(mods):
        m = mods[i]
        was_conv, biased_conv = biased_conv, False
        if isinstance(m, nn.BatchNorm2d):
            biased_conv = was_conv
        if isinstance(m, nn.ConstantPad2d):
            t, b, l, r = _pad4(m)
            pad = (pad[0] + t, pad[1] + b, pad[2] + l, pad[3] + r)
            i += 1
        elif isinstance(m, FreqUpsample):
            if pad[2] or pad[3]:                   # (a time pad commutes with the frequency upsampling)
                raise RuntimeError("HipSequential: a frequency pad before FreqUpsample is not supported")
            up *= int(m.f)
            i += 1
        elif isinstance(m, nn.Conv2d) and not isinstance(m, nn.ConvTranspose2d):
            if m.padding_mode != "zeros" or isinstance(m.padding, str):
                raise RuntimeError("HipSequential: Conv2d needs numeric zero padding")
            ph, pw = _pair(m.padding)
            nxt = mods[i + 1] if i + 1 < len(mods) else None
            bn_next = isinstance(nxt, nn.BatchNorm2d) and (nxt.training or nxt.running_mean is None) and x.dtype == torch.float16
            tap = tap_first_conv and tapped is None and i == 0
            x = conv2d(x, m.weight, m.bias, m.stride, m.dilation, (pad[0] + ph, pad[1] + ph, pad[2] + pw, pad[3] + pw),
                       m.groups, up, bn_stats=bn_next, tap_input=tap)
            if tap:
                x, tapped = x
            pad, up = (0, 0, 0, 0), 1
            i += 1
            biased_conv = m.bias is not None and m.bias.requires_grad
            continue
        elif isinstance(m, nn.ConvTranspose2d):
            if up != 1 or pad[1] or pad[2] or pad[3]:
                raise RuntimeError("HipSequential: only a top zero pad folds into ConvTranspose2d")
            x = conv_transpose2d(x, m.weight, m.bias, m.stride, m.padding, m.output_padding, m.dilation, m.groups,
                                 pre_pad_top=pad[0])
            pad = (0, 0, 0, 0)
            i += 1
        elif isinstance(m, nn.BatchNorm2d):
            nxt = mods[i + 1] if i + 1 < len(mods) else None
            if isinstance(nxt, (nn.ReLU, nn.PReLU, nn.Sigmoid)):
                x = batchnorm_act(x, m, nxt, conv_bias_in_front=biased_conv)
                i += 2
            else:
                x = batchnorm_act(x, m, None, conv_bias_in_front=biased_conv)
                i += 1
        elif isinstance(m, (nn.ReLU, nn.PReLU, nn.Sigmoid)):
            x = batchnorm_act(x, None, m)
            i += 1
        elif isinstance(m, nn.Identity):
            i += 1
        else:
            x = m(x)                              # a nested HIP module (HipSequential, ComplexConv2d, ...)
            i += 1
    if pad != (0, 0, 0, 0) or up != 1:
        raise RuntimeError("HipSequential: trailing pad / upsample without a convolution")
    if tap_first_conv:
        if tapped is None:
            raise RuntimeError("HipSequential: tap_first_conv needs a Conv2d as the first module")
        return x, tapped
    return x


class HipSequential(nn.Sequential):
    def forward(self, x):
        return run_sequential(list(self), x)

    def forward_tap(self, x):
        """(self(x), x'): x' carries x through the first convolution's autograd node (for a residual taken from the block's input)"""
        return run_sequential(list(self), x, tap_first_conv=True)


class HipConv2d(nn.Conv2d):
    """nn.Conv2d parameters, HIP forward/backward."""

    def forward(self, x):
        return run_sequential([self], x)

    def forward_add(self, x, res):
        """self(x) + res in one launch where the kernel offers it (the residual of TFCM_Block, mtfaa.py:191)"""
        if not x.is_cuda:
            raise RuntimeError("cruse_amd blocks need tensors on the HIP device (no CPU fallback)")
        if self.padding_mode != "zeros" or isinstance(self.padding, str):
            raise RuntimeError("HipConv2d: numeric zero padding")
        ph, pw = _pair(self.padding)
        return conv2d(x, self.weight, self.bias, self.stride, self.dilation, (ph, ph, pw, pw), self.groups, 1, residual=res)
