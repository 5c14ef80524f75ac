"""ctypes binding of libcruse_hip.so (include/cruse_hip.h).

The product path has NO CPU fallback: if the shared library is missing or a symbol
is absent, importing this module raises.  Build it with
`python -c "import __graft_entry__ as g; g.build()"` or cruse_amd/csrc/build.sh.
"""
from __future__ import annotations

import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libcruse_hip.so")

ABI_VERSION = 13
PREC_F32, PREC_BF16X3, PREC_BF16, PREC_F16 = 0, 1, 2, 3
PREC_BY_NAME = {"f32": PREC_F32, "bf16x3": PREC_BF16X3, "bf16": PREC_BF16, "f16": PREC_F16}
DT_F32, DT_F16, DT_BF16 = 0, 1, 2

_T = {"p": ctypes.c_void_p, "i": ctypes.c_int, "f": ctypes.c_float, "q": ctypes.c_longlong, "d": ctypes.c_double, "c": ctypes.c_char_p,
      "z": ctypes.c_size_t, "Q": ctypes.c_ulonglong}

# name -> (argument type string, restype); mirrors include/cruse_hip.h one to one
SIGNATURES = {
    "cruse_abi_version": ("", "i"),
    "cruse_last_error": ("", "s"),
    "cruse_set_option": ("cii", "i"),
    "cruse_get_option": ("cpp", "i"),
    "cruse_stft_fwd": ("piiiipppifp", "i"),
    "cruse_istft_fwd": ("ppiiiiipp", "i"),
    "cruse_istft_bwd": ("piiiiippp", "i"),
    "cruse_conv_gather": ("ppppiiiiiiiiiiiiiiip", "i"),
    "cruse_conv_scatter2": ("ppppiiiiiiiiiiiiip", "i"),
    "cruse_conv_gather_bnstats": ("ppppiiiiiiiiiipip", "i"),
    "cruse_conv_scatter2_bnstats": ("ppppiiiiiiiiipip", "i"),
    "cruse_conv_mfma_stamps": ("p", "i"),
    "cruse_conv_gather_bnbwd": ("pppiiiiiiiiiiiipppppipiiip", "i"),
    "cruse_conv_scatter2_bnbwd": ("pppiiiiiiiiiipppppipiiip", "i"),
    "cruse_conv_wgrad_ws_bytes": ("iii", "z"),
    "cruse_conv_gather_bnbwd_in": ("pipppppp" + "iii" + "pppp" + "pp" + "iiiiiiiiiiii" + "pppppi" + "piip", "i"),
    "cruse_conv_scatter2_bnbwd_in": ("pipppppp" + "iii" + "pppp" + "pp" + "iiiiiiiiii" + "pppppi" + "piip", "i"),
    "cruse_conv_gather_bnin": ("ppiqffpppppppppppiiiiiiiiiipip", "i"),
    "cruse_conv_scatter2_bnin": ("ppiqffpppppppppppiiiiiiiiipip", "i"),
    "cruse_conv_wgrad": ("pppiiiiiiiiiiiipp", "i"),
    "cruse_channel_sum": ("pqiipp", "i"),
    "cruse_col_sum": ("pqiipp", "i"),
    "cruse_bn_stats": ("pqiipip", "i"),
    "cruse_bn_finalize_act_fwd": ("ppiqffpppppppppqiiip", "i"),
    "cruse_bn_finalize_act_fwd_c": ("ppiqffpppppippppqiiip", "i"),
    "cruse_bn_finalize": ("pqiffppppp", "i"),
    "cruse_bn_eval_stats": ("ppifppp", "i"),
    "cruse_bn_act_fwd": ("pppppppqiiip", "i"),
    "cruse_bn_act_bwd_reduce": ("ppppppqiiipip", "i"),
    "cruse_bn_act_bwd_apply": ("pppppppiqiiiiipipppp", "i"),
    "cruse_ln_fwd": ("ppppppppqiifiqqp", "i"),
    "cruse_ln_fwd_c": ("ppppppippqiifiqqp", "i"),
    "cruse_ln_bwd": ("pppppqiipppp", "i"),
    "cruse_gemm": ("iiiiipipipipiiiip", "i"),
    "cruse_gemm_bf16_nt": ("iiipqqpqqpqpiip", "i"),
    "cruse_gemm_bf16_slab_bytes": ("iii", "z"),
    "cruse_gemm_bf16_nt_slabs": ("iiipqqpqqpqipzp", "i"),
    "cruse_gemm_bf16_nt_slabs_cat": ("ipiippqqpqqpqipzp", "i"),
    "cruse_gemm_bf16_nt_groups": ("iiiippqqppqqqpqqpqip", "i"),
    "cruse_gemm_bf16_nt_seg": ("iiippqppqqpqpiiqqp", "i"),
    "cruse_cast_bf16": ("ppqp", "i"),
    "cruse_transpose_bf16": ("pqiqpqip", "i"),
    "cruse_ktile_bf16": ("piiqppp", "i"),
    "cruse_ktile_f16": ("piiqpp", "i"),
    "cruse_gemm_f16_nt": ("iiipqqpqqpqpp", "i"),
    "cruse_gemm_nt_out16": ("iiippqqppqqpqpiip", "i"),
    "cruse_gemm_bf16_nt_atr": ("iiipqipqqpqip", "i"),
    "cruse_gemm_f16x2_nt": ("iiipqqppqqpqpp", "i"),
    "cruse_ktile_f16_split": ("piiqppp", "i"),
    "cruse_cast_bf16_split": ("pppqp", "i"),
    "cruse_gemm_bf16x3_nt": ("iiippqqppqqpqpip", "i"),
    "cruse_gru_ws_bytes": ("iii", "z"),
    "cruse_gru_plan": ("iiiiip", "i"),
    "cruse_gru_seq_fwd": ("pppppppiiiiipp", "i"),
    "cruse_gru_seq_bwd": ("pppppiiiiipp", "i"),
    "cruse_gru_seq_fwd_on": ("pppppppiiiiippip", "i"),
    "cruse_gru_seq_bwd_on": ("pppppppiiiiippip", "i"),
    "cruse_gru_seq_fwd_ex": ("ppppppppqiiiiiiipipip", "i"),
    "cruse_gru_seq_fwd_gi16": ("pippppppiiiiiipipip", "i"),
    "cruse_gru_seq_bwd_ex": ("pppppppiiiiiiiiipipip", "i"),
    "cruse_gru_gate_grads": ("pppppqiiip", "i"),
    "cruse_gru_gate_grads_bf16": ("pppppqppqiip", "i"),
    "cruse_mask_loss_fwd": ("ppppqiiffpppppp", "i"),
    "cruse_mask_apply": ("pppqiippp", "i"),
    "cruse_mask_apply_bwd": ("pppppqiiipp", "i"),
    "cruse_mask_sdnr_fwd": ("pppppqiiiffpppp", "i"),
    "cruse_sisnr_fwd": ("ppiifpppp", "i"),
    "cruse_sisnr_bwd": ("pppiifpp", "i"),
    "cruse_wave_l1_mse": ("ppqifppp", "i"),
    "cruse_deepfilter_fwd": ("ppppiiiiippp", "i"),
    "cruse_deepfilter_bwd": ("ppppppiiiiippppp", "i"),
    "cruse_sigmoid_bwd": ("pppqp", "i"),
    "cruse_axpby": ("pppffqp", "i"),
    "cruse_adam_step": ("ppppqfffffifp", "i"),
    "cruse_adam_step_guarded": ("ppppqfffffiffppipppdpp", "i"),
    "cruse_step_health": ("ppppdp", "i"),
    "cruse_sumsq": ("pqpip", "i"),
    "cruse_conv2d_nchw": ("ppppiiiiiiiiiiiiiiiiiiipiip", "i"),
    "cruse_conv2d_nchw_wgrad": ("pppiiiiiiiiiiiiiiiiiip", "i"),
    "cruse_conv2d_nchw_wgrad_ex": ("ppppiiiiiiiiiiiiiiiiiip", "i"),
    "cruse_nchw_channel_sum": ("piiipip", "i"),
    "cruse_downsum_w": ("pqiipip", "i"),
    "cruse_upsample_w": ("pqiipip", "i"),
    "cruse_bn_nchw_stats": ("piiipip", "i"),
    "cruse_bn_nchw_fwd": ("ppppppiiiipip", "i"),
    "cruse_bn_nchw_bwd": ("pppppppiiiiipppppip", "i"),
    "cruse_bn_nchw_bwd_ex": ("pppppppiiiiipiipppppip", "i"),
    "cruse_conv2d_nchw_bnbwd": ("ppp" + "iiiiiii" + "iiiiiiii" + "ii" + "pppppp" + "i" + "pip" + "ip", "i"),
    "cruse_bn_nchw_stats_ex": ("piiipiip", "i"),
    "cruse_bn_nchw_fwd_train": ("ppiffpppiiiippppppip", "i"),
    "cruse_conv2d_nchw_ex": ("ppppp" + "iiiiiii" + "iiiiiiii" + "iiii" + "p" + "pi" + "ip", "i"),
    "cruse_add_nchw": ("pppqip", "i"),
    "cruse_cast_f16": ("ppqip", "i"),
    "cruse_stft_framed": ("ppiiiiiiiiifppp", "i"),
    "cruse_istft_framed": ("ppppiiiiiiiifipp", "i"),
    "cruse_mask_ops": ("ippppqfffppp", "i"),
    "cruse_polar": ("ippppqffppp", "i"),
    "cruse_rmse": ("ppqfppp", "i"),
    "cruse_c_rmse": ("ppiqffppp", "i"),
    "cruse_sisnr_plain_finalize": ("pifppp", "i"),
    "cruse_wo_male_spec": ("pppiqqqfffppp", "i"),
    "cruse_onepole_fir": ("piififpp", "i"),
    "cruse_snr_mix": ("pppiifppppp", "i"),
    "cruse_fir_causal": ("ppqiiipp", "i"),
    "cruse_stream_create_masked": ("ppi", "i"),
    "cruse_cu_census": ("piip", "i"),
    "cruse_cu_hog": ("iQp", "i"),
    "cruse_zero": ("pzp", "i"),
    "cruse_accum_f64": ("ppip", "i"),
    "cruse_counters_add": ("piqp", "i"),
}


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: the CRUSE HIP library must be built for gfx950 "
            "(run `python -c 'import __graft_entry__ as g; g.build()'`). There is no CPU fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (args, res) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.argtypes = [_T[c] for c in args]
        fn.restype = ctypes.c_char_p if res == "s" else _T[res]
    if lib.cruse_abi_version() != ABI_VERSION:
        raise ImportError(f"libcruse_hip.so ABI version {lib.cruse_abi_version()} != {ABI_VERSION}")
    return lib


lib = _load()


def check(rc: int) -> None:
    """Mirror the reference's error behaviour: Python RuntimeError on shape/dim mismatch."""
    if rc != 0:
        msg = lib.cruse_last_error()
        raise RuntimeError(f"cruse_hip error {rc}: {msg.decode() if msg else ''}")
