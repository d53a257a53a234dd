"""ctypes binding of libavhip.so (include/av_hip.h).  Fails loudly: there is NO CPU / PyTorch fallback.

The library is built in-tree by ``build.py`` (hipcc, gfx950) and travels to the GPU box with the repo snapshot.
"""
from __future__ import annotations

import ctypes as C
import os

from . import precision as _precision

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("AVAMD_LIB") or os.path.join(_HERE, "libavhip.so")      # AVAMD_LIB: another build of the same library (A/B tooling)

AV_F32, AV_BF16 = 0, 1
A_ROWMAJOR, A_TRANS, A_CONV2D, A_CONV3D1 = 0, 1, 2, 3
B_NK, B_KN = 0, 1
ACT_NONE, ACT_GELU, ACT_MUL_GELU_GRAD, ACT_GELU_GF, ACT_MUL_AUX = 0, 1, 2, 3, 4
LSTM_COUNTER_INTS = 1024         # AV_LSTM_COUNTER_INTS: device workspace of av_lstm_*_layer

vp, ll, i32, f32 = C.c_void_p, C.c_longlong, C.c_int, C.c_float


class GemmArgs(C.Structure):
    _fields_ = [
        ("A", vp), ("B", vp), ("C", vp), ("C2", vp), ("bias", vp), ("R", vp), ("aux", vp), ("stats", vp),
        ("M", i32), ("N", i32), ("K", i32), ("batch", i32),
        ("lda", ll), ("ldb", ll), ("ldc", ll), ("ldr", ll),
        ("sA", ll), ("sB", ll), ("sC", ll), ("sR", ll), ("sBias", ll),
        ("a_mode", i32), ("b_mode", i32), ("in_dtype", i32), ("out_dtype", i32), ("aux_dtype", i32), ("act", i32),
        ("alpha", f32),
        ("cT", i32), ("cH", i32), ("cW", i32), ("cCtot", i32), ("cCin", i32), ("cCoff", i32),
        ("cKt", i32), ("cKh", i32), ("cKw", i32), ("cSh", i32), ("cSw", i32), ("cPt", i32), ("cPh", i32), ("cPw", i32),
        ("cOh", i32), ("cOw", i32),
        ("batch_inner", i32), ("oA", ll), ("oB", ll), ("oC", ll),
        ("drop_p", f32), ("drop_stream", C.c_uint), ("drop_seed", C.c_ulonglong),
        ("k_total", i32), ("cNF", i32), ("cPM", i32),
    ]


class W2v2LayerArgs(C.Structure):
    _fields_ = [
        ("B", i32), ("T", i32), ("hidden", i32), ("heads", i32), ("inter", i32), ("lp", i32), ("gf", i32), ("stream_base", i32),
        ("eps", f32), ("scale", f32), ("hd_p", f32), ("at_p", f32), ("ac_p", f32),
        ("seed", C.c_ulonglong),
        ("ln1_g", vp), ("ln1_b", vp), ("ln2_g", vp), ("ln2_b", vp), ("b_qkv", vp), ("b_o", vp), ("b_1", vp), ("b_2", vp),
        ("w_qkv", vp), ("w_o", vp), ("w_1", vp), ("w_2", vp),
        ("h", vp), ("klen", vp), ("amask", vp),
        ("x1", vp), ("qkv", vp), ("ao", vp), ("x2", vp), ("u", vp), ("g", vp),
        ("mu1", vp), ("rs1", vp), ("lse", vp), ("h2", vp), ("mu2", vp), ("rs2", vp), ("h3", vp),
    ]


class W2v2LayerBwdArgs(C.Structure):
    _fields_ = [
        ("B", i32), ("T", i32), ("hidden", i32), ("heads", i32), ("inter", i32), ("lp", i32), ("gf", i32), ("stream_base", i32), ("lower_stream", i32),
        ("scale", f32), ("hd_p", f32), ("at_p", f32), ("ac_p", f32),
        ("seed", C.c_ulonglong),
        ("ln1_g", vp), ("ln2_g", vp),
        ("w_2t", vp), ("w_1t", vp), ("w_ot", vp), ("w_qkvt", vp),
        ("dh", vp), ("h", vp), ("mu1", vp), ("rs1", vp), ("lse", vp), ("h2", vp), ("mu2", vp), ("rs2", vp),
        ("dh_lp", vp), ("qkv", vp), ("ao", vp), ("amask", vp), ("u", vp),
        ("klen", vp),
        ("dh3_t", vp), ("du", vp), ("dx2", vp), ("dh2_lp", vp), ("dao", vp), ("dqkv", vp), ("dx1", vp), ("dh_out_lp", vp),
        ("dh2", vp), ("delta", vp), ("dh_out", vp),
    ]


# name -> argtypes (restype is int status unless listed in _RESTYPES); must list every symbol of include/av_hip.h
SIGNATURES = {
    "av_last_error": [],
    "av_version": [],
    "av_gemm": [C.POINTER(GemmArgs), vp],
    "av_w2v2_layer_fwd": [C.POINTER(W2v2LayerArgs), vp],
    "av_w2v2_layer_bwd_dx": [C.POINTER(W2v2LayerBwdArgs), vp],
    "av_transpose": [vp, i32, vp, i32, i32, i32, ll, i32, vp],
    "av_sum_slices": [vp, i32, ll, ll, f32, vp, i32, vp],
    "av_layernorm_fwd": [vp, i32, vp, vp, vp, i32, vp, vp, ll, i32, f32, i32, vp],
    "av_layernorm_bwd_drop": [vp, i32, vp, i32, vp, vp, vp, vp, vp, vp, i32, ll, i32, vp, f32, C.c_ulonglong, C.c_uint, vp],
    "av_layernorm_bwd": [vp, i32, vp, i32, vp, vp, vp, vp, vp, vp, i32, ll, i32, vp, vp],
    "av_log_softmax_fwd": [vp, i32, vp, ll, i32, vp],
    "av_log_softmax_bwd": [vp, vp, vp, i32, ll, i32, vp],
    "av_log_softmax_bwd_ld": [vp, vp, vp, i32, ll, i32, i32, vp],
    "av_colsum": [vp, i32, vp, ll, i32, ll, i32, vp],
    "av_cast": [vp, i32, vp, i32, ll, vp],
    "av_cast_dropout": [vp, i32, vp, i32, ll, f32, C.c_ulonglong, C.c_uint, vp],
    "av_dropout_uniform": [vp, ll, C.c_ulonglong, C.c_uint, vp],
    "av_fusion_xattn_fwd": [vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, f32, vp],
    "av_fusion_xattn_bwd": [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, f32, vp],
    "av_zero_feature_cols": [vp, i32, vp, i32, i32, i32, vp],
    "av_overwrite_rows": [vp, i32, vp, vp, ll, i32, vp],
    "av_axpby": [f32, vp, i32, f32, vp, ll, vp],
    "av_mask_rows": [vp, i32, vp, ll, i32, vp],
    "av_mul_scalar_dev": [vp, vp, vp, ll, vp],
    "av_attention_fwd": [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, ll, ll, ll, ll, ll, ll, ll, ll, vp, f32, f32, C.c_ulonglong, C.c_uint, vp],
    "av_attention_dropmask": [vp, i32, i32, i32, i32, f32, C.c_ulonglong, C.c_uint, vp],
    "av_attention_fwd_mask": [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, ll, ll, ll, ll, ll, ll, ll, ll, vp, f32, f32, C.c_ulonglong, C.c_uint, vp, vp],
    "av_attention_bwd_mask": [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp, vp, f32, f32, C.c_ulonglong, C.c_uint, vp, vp],
    "av_attention_bwd": [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp, vp, f32, f32, C.c_ulonglong, C.c_uint, vp],
    "av_softmax_rows": [vp, vp, i32, ll, i32, f32, vp, i32, i32, vp],
    "av_conv0_ln_gelu": [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, f32, vp],
    "av_lstm_fwd_step": [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp],
    "av_lstm_bwd_step": [vp, i32, ll, ll, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp],
    "av_lstm_fwd_layer": [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, vp],
    "av_lstm_bwd_layer": [vp, i32, ll, ll, vp, vp, vp, vp, vp, vp, i32, i32, i32, vp],
    "av_mask_downsample": [vp, vp, i32, i32, i32, vp],
    "av_fusion_gather_lerp_fwd": [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp],
    "av_fusion_gather_lerp_bwd": [vp, vp, vp, i32, i32, i32, i32, i32, vp],
    "av_permute_bt": [vp, i32, vp, i32, i32, i32, i32, vp],
    "av_gather_rows": [vp, i32, vp, vp, i32, ll, i32, vp],
    "av_scatter_rows": [vp, vp, vp, ll, i32, f32, i32, vp],
    "av_class_order": [vp, ll, vp, vp],
    "av_conv3d_front": [vp, vp, vp, vp, i32, i32, i32, i32, vp],
    "av_conv3d_front_pool": [vp, vp, vp, vp, vp, i32, i32, i32, i32, vp],
    "av_bn_prelu_minmax": [vp, vp, vp, vp, vp, vp, ll, vp],
    "av_conv3x3_c64": [vp, vp, vp, vp, i32, i32, i32, vp, vp, vp, vp],
    "av_bn_finalize": [vp, i32, ll, vp, vp, vp, vp, f32, f32, i32, vp, vp, i32, vp, i32, vp],
    "av_bn_act": [vp, vp, vp, vp, vp, vp, vp, vp, i32, ll, i32, vp],
    "av_bn_prelu_maxpool": [vp, vp, vp, vp, vp, i32, ll, i32, i32, i32, vp],
    "av_avgpool": [vp, i32, vp, ll, i32, i32, i32, vp],
    "av_l2norm_fwd": [vp, vp, vp, ll, i32, f32, vp],
    "av_l2norm_bwd": [vp, vp, vp, vp, ll, i32, f32, vp],
    "av_lse_rows": [vp, vp, vp, ll, i32, i32, vp],
    "av_lse_rows_chunk": [vp, vp, vp, ll, i32, i32, i32, vp],
    "av_contrastive_dsim_chunk": [vp, vp, vp, i32, ll, i32, i32, f32, i32, vp],
    "av_contrastive_dsim": [vp, vp, vp, i32, ll, i32, i32, f32, vp],
    "av_reduce_sum": [vp, ll, vp, f32, i32, vp],
    "av_loss_combine": [vp, vp, vp, vp, f32, i32, vp, vp],
    "av_ctc_greedy": [vp, vp, vp, vp, i32, i32, i32, i32, vp],
    "av_lip_gray_resize": [vp, i32, vp, i32, i32, i32, i32, i32, i32, f32, vp],
    "av_mix_pair": [vp, ll, vp, ll, vp, vp, vp, vp, vp],
    "av_adam_multi": [vp, vp, vp, vp, vp, i32, i32, vp, vp, i32, f32, vp, f32, f32, i32, vp],
    "av_adam_step": [vp, vp, vp, vp, ll, f32, f32, f32, f32, i32, f32, vp],
    "av_softmax_bwd_rows": [vp, i32, vp, vp, i32, ll, i32, f32, i32, vp],
    "av_nchw_to_nhwc": [vp, vp, i32, ll, i32, i32, i32, i32, vp],
    "av_relu_maxpool2_fwd": [vp, vp, i32, ll, i32, i32, i32, i32, vp],
    "av_relu_maxpool2_bwd": [vp, vp, vp, i32, ll, i32, i32, i32, i32, vp],
    "av_im2col3": [vp, vp, i32, ll, i32, i32, i32, vp],
    "av_gru_fwd_step": [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp],
    "av_gru_bwd_step": [vp, i32, ll, ll, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp],
    "av_wav_info": [C.c_char_p, C.POINTER(i32), C.POINTER(i32), C.POINTER(ll), C.POINTER(i32), C.POINTER(i32)],
    "av_wav_read_mono_f32": [C.c_char_p, ll, ll, vp],
    "av_resample_sinc": [vp, ll, vp, ll, vp, vp, i32, i32, i32, i32, vp],
}
_RESTYPES = {"av_last_error": C.c_char_p}

_libs = {}
_by_mode = {}                      # precision mode -> loaded library: lib() runs once per kernel launch (a dict hit, not a path computation)


def _path() -> str:
    if _precision.get_precision() == "fp16":
        return os.environ.get("AVAMD_LIB_F16") or os.path.join(_HERE, "libavhip_f16.so")
    return LIB_PATH


def lib() -> C.CDLL:
    """Load (once per library) and return the HIP library of the current precision mode (fp32 / bf16: libavhip.so, fp16: libavhip_f16.so);
    raise if it is missing - no fallback exists."""
    l = _by_mode.get(_precision._mode)
    if l is not None:
        return l
    path = _path()
    l = _libs.get(path)
    if l is None:
        if not os.path.exists(path):
            raise RuntimeError(
                f"{path} is missing: build it with `python {os.path.join(_HERE, 'build.py')}` "
                "(hipcc --offload-arch=gfx950). The MI355X path has no CPU/PyTorch fallback.")
        # torch first: libavhip.so links against libamdhip64, and the process must end up with ONE HIP runtime - the one PyTorch-ROCm
        # ships.  Loading this library before torch binds it to the system runtime instead; torch then shares that copy and kernels
        # registered here fail later in odd ways (hipFuncSetAttribute: "cannot raise dynamic LDS" in build() + smoke() of one process).
        import torch  # noqa: F401
        l = C.CDLL(path)
        for name, argtypes in SIGNATURES.items():
            fn = getattr(l, name)           # AttributeError if a declared symbol is not exported
            fn.argtypes = argtypes
            fn.restype = _RESTYPES.get(name, C.c_int)
        _libs[path] = l
    _by_mode[_precision._mode] = l
    return l


def check(status: int, what: str = "") -> None:
    if status != 0:
        msg = lib().av_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"libavhip {what} failed (status {status}): {msg}")
