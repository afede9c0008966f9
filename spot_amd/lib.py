"""ctypes binding of libstes_hip.so (the C-ABI declared in include/stes.h).

The library is the product: if it is missing or a kernel launch fails this module
raises, it never falls back to another implementation.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libstes_hip.so")

_lib = None


class StesError(RuntimeError):
    pass


_P, _I32, _I64, _F32, _F64, _U32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_double, C.c_uint32


def _struct(name, fields):
    return type(name, (C.Structure,), {"_fields_": fields})


GemmArgs = _struct("GemmArgs", [
    ("A", _P), ("lda", _I64), ("A0", _P), ("lda0", _I64), ("k0", _I32),
    ("a_scale", _P), ("a_shift", _P), ("a_relu", _I32),
    ("a_gate", _P), ("gate_group", _I32),
    ("map_stride", _I32), ("Ho", _I32), ("Wo", _I32), ("Hi", _I32), ("Wi", _I32),
    ("W", _P), ("ldw", _I64), ("bias", _P), ("act", _I32),
    ("C", _P), ("ldc", _I64), ("c_f32", _I32),
    ("col_sum", _P), ("col_sumsq", _P),
    ("relu_mask", _P), ("ldmask", _I64),
    ("C0", _P), ("ldc0", _I64), ("n_split", _I32),
    ("M", _I64), ("N", _I32), ("K", _I32), ("out_scale", _F32),
    ("a_raw", _P), ("bn_alpha", _P), ("bn_beta", _P), ("bn_delta", _P), ("a_out", _P), ("lda_out", _I64),
    ("wg_q", _P), ("wg_ldq", _I64), ("wg_q0", _P), ("wg_ldq0", _I64), ("wg_k0", _I32), ("wg_perm_fold", _I32),
    ("wg_dw", _P), ("wg_lddw", _I64), ("wg_map_stride", _I32),
    ("e_scale", _P), ("e_shift", _P), ("e_res", _P), ("e_ldres", _I64), ("e_rscale", _P), ("e_rshift", _P),
    ("e_slice", _P), ("e_slice_n", _I32)])

Gemm2Args = _struct("Gemm2Args", [
    ("A", _P), ("lda", _I64), ("M", _I64), ("K1", _I32),
    ("W1", _P), ("ldw1", _I32), ("N1", _I32), ("bias1", _P), ("act1", _I32),
    ("relu_mask1", _P), ("ldmask1", _I64),
    ("H", _P), ("ldh", _I64),
    ("W2", _P), ("ldw2", _I32), ("N2", _I32), ("bias2", _P), ("act2", _I32), ("out_scale", _F32),
    ("C", _P), ("ldc", _I64), ("c_f32", _I32)])

BlockEvalArgs = _struct("BlockEvalArgs", [
    ("X", _P), ("G", _P),
    ("W1", _P), ("s1", _P), ("b1", _P),
    ("wp2", _P), ("s2", _P), ("b2", _P),
    ("Wse1", _P), ("bse1", _P), ("Wse2", _P), ("bse2", _P),
    ("W3", _P), ("s3", _P), ("b3", _P),
    ("out", _P), ("slice", _P), ("slice_n", _I32),
    ("N", _I32), ("H", _I32), ("W", _I32), ("C", _I32), ("rd", _I32), ("fold", _I32)])

BlockFrontArgs = _struct("BlockFrontArgs", [
    ("X", _P), ("G", _P), ("fold", _I32),
    ("W1", _P), ("s1", _P), ("b1", _P),
    ("wp2", _P), ("s2", _P), ("b2", _P),
    ("c2", _P), ("pooled", _P),
    ("N", _I32), ("H", _I32), ("W", _I32), ("Cin", _I32), ("Cmid", _I32), ("stride", _I32)])

WgradArgs = _struct("WgradArgs", [
    ("P", _P), ("ldp", _I64), ("Q", _P), ("ldq", _I64), ("Q0", _P), ("ldq0", _I64), ("k0", _I32),
    ("q_scale", _P), ("q_shift", _P), ("q_relu", _I32), ("q_gate", _P), ("gate_group", _I32),
    ("map_stride", _I32), ("Ho", _I32), ("Wo", _I32), ("Hi", _I32), ("Wi", _I32),
    ("dW", _P), ("lddw", _I64), ("dbias", _P), ("perm_fold", _I32),
    ("M", _I64), ("N1", _I32), ("N2", _I32),
    ("p_raw", _P), ("bn_alpha", _P), ("bn_beta", _P), ("bn_delta", _P), ("ws", _P), ("ws_floats", _I64)])

BnBwdArgs = _struct("BnBwdArgs", [
    ("D", _P), ("mul", _P), ("add", _P), ("group", _I32),
    ("act_raw", _P), ("act_scale", _P), ("act_shift", _P),
    ("raw1", _P), ("raw2", _P),
    ("s_dy", _P), ("s_dyr1", _P), ("s_dyr2", _P),
    ("alpha", _P), ("beta", _P), ("delta", _P), ("out", _P),
    ("M", _I64), ("C", _I32)])

CastDesc = _struct("CastDesc", [("src", _P), ("dst", _P), ("rows", _I32), ("cols", _I32), ("ld", _I32), ("dst_rows", _I32),
                                ("transpose", _I32), ("perm_fold", _I32), ("block0", _I32), ("reserved", _I32)])

CAST_CHUNK = 2048        # STES_CAST_CHUNK in include/stes.h: destination elements per workgroup of stes_cast_bf16_multi

GsmGeom = _struct("GsmGeom", [("B", _I32), ("T", _I32), ("H", _I32), ("W", _I32), ("fold", _I32)])

GruArgs = _struct("GruArgs", [
    ("gi", _P), ("whh", _P), ("whhT", _P), ("bhh", _P),
    ("y", _P), ("hstate", _P), ("saved", _P), ("hprev_bf", _P),
    ("dy", _P), ("ld_dy", _I64), ("mask", _P),
    ("dgi", _P), ("dgh", _P), ("dhstate", _P),
    ("B", _I32), ("T", _I32), ("H", _I32), ("ws_floats", _I64)])

_SIGS = {
    "stes_gemm": [_P, _P],
    "stes_wgrad": [_P, _P],
    "stes_gemm_small2": [_P, _P],
    "stes_block_eval_s3": [_P, _P],
    "stes_block_front_eval": [_P, _P],
    "stes_bn_finalize": [_P, _P, _F64, _P, _P, _P, _P, _F32, _F32, _I32, _P, _P, _P, _P, _I32, _P],
    "stes_bn_act_residual": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I32, _P, _I64, _I32, _P, _P],
    "stes_bn_act_residual_tsm": [_P, _P, _P, _P, _P, _P, _P, _I64, _I32, _I32, _I32, _I32, _P],
    "stes_chan_pool": [_P, _P, _P, _P, _P, _P, _I32, _I32, _I32, _F32, _P],
    "stes_bn_act_gate": [_P, _P, _P, _P, _P, _I32, _I32, _I32, _P],
    "stes_sigmoid_bwd": [_P, _P, _P, _I64, _P],
    "stes_clip_gate_fwd": [_P, _P, _P, _I32, _I32, _I32, _P],
    "stes_clip_gate_bwd": [_P, _P, _P, _P, _P, _P, _P, _I32, _I32, _I32, _P],
    "stes_se_bwd_pool": [_P, _P, _P, _P, _P, _P, _P, _P, _I32, _I32, _I32, _P],
    "stes_se_bwd_bnsums": [_P, _P, _P, _P, _P, _I32, _I32, _P],
    "stes_bn_bwd_reduce": [_P, _P],
    "stes_bn_bwd_finalize": [_P, _P, _P, _P, _P, _F64, _P, _P, _P, _P, _P, _I32, _P],
    "stes_bn_bwd_apply": [_P, _P],
    "stes_block_in_grad": [_P, _I64, _P, _I32, _P, _I32, _I32, _I32, _I32, _I32, _P, _P, _I64, _I32, _P, _P, _P, _P, _P, _P, _P],
    "stes_block_in_grad_tsm": [_P, _I64, _P, _I32, _P, _I32, _I32, _I32, _I32, _I32, _P, _P, _I64, _I32, _P, _P, _P, _P, _P, _P, _I32, _I32, _P],
    "stes_pool_bwd": [_P, _P, _P, _P, _I32, _I32, _I32, _P],
    "stes_pack_conv3x3g": [_P, _P, _P, _I32, _I32, _P],
    "stes_conv3x3g_fwd": [_P, _P, _P, _P, _P, _P, _P, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _P],
    "stes_conv3x3g_fwd_pool": [_P, _P, _P, _P, _P, _P, _P, _P, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _P],
    "stes_conv3x3g_dgrad": [_P, _P, _P, _P, _P, _P, _P, _P, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _P],
    "stes_conv3x3g_wgrad": [_P, _P, _P, _P, _P, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _P],
    "stes_im2col_stem": [_P, _P, _I32, _I32, _I32, _I32, _P],
    "stes_im2col_stem_u8": [_P, _P, _P, _P, _I32, _I32, _I32, _I32, _I32, _I32, _P],
    "stes_stem_conv": [_P, _I32, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _P],
    "stes_augment_u8": [_P, _P, _P, _P, _P, _I64, _I32, _I32, _I32, _I32, _I32, _I32, _P],
    "stes_gsm_prep": [_P, _P, _I32, _P],
    "stes_gsm_gate": [_P, _P, _P, _P, _P, _P, _P, _P],
    "stes_gsm_shift": [_P, _P, _P, _P, _P],
    "stes_gsm_bwd_gate": [_P, _P, _P, _P, _P, _P],
    "stes_gsm_bwd_wgrad": [_P, _P, _P, _P, _P, _P, _P, _P],
    "stes_gsm_dxbn": [_P, _P, _P, _P, _P, _P, _P, _P],
    "stes_gsm_bwd_dx": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P],
    "stes_gsm_forward": [_P, _P, _P, _P, _P, _P, _P, _P, _P],
    "stes_gsm_bwd_conv": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P],
    "stes_col_stats": [_P, _I64, _I32, _I64, _P, _P, _P, _P],
    "stes_tsm_shift": [_P, _P, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _P],
    "stes_tsm_shift_inplace": [_P, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _P],
    "stes_gru_forward": [_P, _P],
    "stes_gru_backward": [_P, _P],
    "stes_gru_forward_noalloc": [_P, _P],
    "stes_mingru_scan_fwd": [_P, _P, _P, _I32, _I32, _I32, _P],
    "stes_mingru_scan_bwd": [_P, _P, _P, _P, _I64, _P, _I32, _I32, _I32, _P],
    "stes_dropout": [_P, _P, _P, _I64, _F32, _U32, _P],
    "stes_loss": [_P, _I64, _P, _I64, _P, _P, _I32, _I32, _F32, _F32, _P, _P, _P, _P, _P],
    "stes_adamw": [_P, _P, _P, _P, _I64, _F32, _F32, _F32, _F32, _F32, _I32, _F32, _P],
    "stes_cast_bf16": [_P, _P, _I32, _I32, _I32, _I32, _I32, _I32, _P],
    "stes_cast_bf16_multi": [_P, _I32, _I32, _P],
}
EXPORTS = ["stes_version", "stes_last_error", "stes_gsm_bwd_conv_ws_floats", "stes_gru_ws_floats", "stes_wgrad_ws_floats",
           "stes_augment_ws_floats", "stes_gru_set_spin_limit", "stes_gru_noalloc_prepare", "stes_set_deterministic"] + list(_SIGS)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise StesError(f"{LIB_PATH} not found: build it with `python -m spot_amd.build` "
                            "(hipcc --offload-arch=gfx950); there is no fallback path")
        L = C.CDLL(LIB_PATH)
        L.stes_last_error.restype = C.c_char_p
        L.stes_version.restype = C.c_int
        L.stes_gsm_bwd_conv_ws_floats.argtypes = [C.c_int]
        L.stes_gsm_bwd_conv_ws_floats.restype = C.c_int
        L.stes_gru_ws_floats.argtypes = [C.c_int, C.c_int, C.c_int]
        L.stes_gru_ws_floats.restype = C.c_int64
        L.stes_wgrad_ws_floats.argtypes = []
        L.stes_wgrad_ws_floats.restype = C.c_int64
        L.stes_augment_ws_floats.argtypes = [C.c_int, C.c_int, C.c_int]
        L.stes_augment_ws_floats.restype = C.c_int64
        L.stes_gru_set_spin_limit.argtypes = [C.c_int]
        L.stes_gru_set_spin_limit.restype = C.c_int
        L.stes_gru_noalloc_prepare.argtypes = []
        L.stes_gru_noalloc_prepare.restype = C.c_int
        L.stes_set_deterministic.argtypes = [_P, _P]
        L.stes_set_deterministic.restype = C.c_int
        for name, sig in _SIGS.items():
            f = getattr(L, name)
            f.argtypes = sig
            f.restype = C.c_int
        _lib = L
    return _lib


ERR_UNSUPPORTED = -3        # STES_ERR_UNSUPPORTED in include/stes.h

_det_ticket = None


def set_deterministic(on: bool):
    """The library's deterministic mode (stes_set_deterministic in include/stes.h): every cross-workgroup float sum is
    committed in a fixed order, so two runs of the same work give the same bits.  A test mode: slower, one stream."""
    global _det_ticket
    torch.cuda.synchronize()
    if on:
        t = torch.zeros(64, dtype=torch.int32, device="cuda")
        check(lib().stes_set_deterministic(t.data_ptr(), stream()), "stes_set_deterministic")
        _det_ticket = t
    else:
        check(lib().stes_set_deterministic(None, stream()), "stes_set_deterministic")
        _det_ticket = None
    torch.cuda.synchronize()


def deterministic_ticket():
    """The ticket tensor while the deterministic mode is on (word 1 != 0: a workgroup's wait for its turn ran out), else None."""
    return _det_ticket


def check(rc: int, what: str):
    if rc != 0:
        err = StesError(f"{what} failed ({rc}): {lib().stes_last_error().decode()}")
        err.rc = rc
        raise err


def stream():
    return torch.cuda.current_stream().cuda_stream


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()


# ---------------------------------------------------------------- optional per-launch timing (bench.py roofline leg)
# When TIMING is a list, every launch is bracketed by HIP events recorded on the launch stream and appended as
# (kernel, start, end, algorithmic_bytes or None, flops).  `kernel` is the name rocprofv3 shows for the launch.
TIMING = None


def _conv_bytes(a):
    n, hi, wi, ho, wo, _stride, c = a[-7:]
    return 2 * c * n * (hi * wi + ho * wo)


_RING1_ROWS = {7: 8, 14: 4, 28: 2}         # stride-1 row-ring kernel: map width -> rows per step (stes_conv_ring1_launch),
_RING1_ROWS_FWD = {7: 8, 14: 8, 28: 2}     # ... of the forward and of the weight gradient


def _ring2(a):
    n, hi, wi, ho, wo, stride, c = a[-7:]
    return stride == 2 and hi % 2 == 0 and wi in (112, 56, 28, 14)


def _stem_bytes(a):
    e_slice, e_slice_n, b, t, tv, cin, h, w = a[11:19]
    pix = b * t * ((h + 1) // 2) * ((w + 1) // 2)
    out = 64 * pix * (1 if a[9] is not None else 2) + (2 * e_slice_n * pix if e_slice is not None else 0)
    return b * tv * cin * h * w * (1 if a[1] else 4) + out


def _geom_of(a):
    """(rows, fold) of the gated-shift geometry passed by reference as the last argument."""
    g = a[-1]._obj
    return g.B * g.T * g.H * g.W, g.fold


def _gsm_bytes(slices, gates):
    """Algorithmic bytes of a gated-shift kernel: `slices` passes over the bf16 slice [M][fold], `gates` over an fp32 [M][2]."""
    def f(a):
        m, fold = _geom_of(a)
        return slices * m * fold * 2 + gates * m * 8
    return f


def _bn_act_bytes(a):
    # (c, scale, shift, res, rscale, rshift, out, sum, sumsq, stat_c, slice_out, M, C, relu_bits)
    m, c = a[11], a[12]
    return 2 * m * c * (2 + (a[3] is not None)) + (2 * m * a[9] if a[10] is not None else 0) + (m * c // 8 if a[13] is not None else 0)


# entry point -> (kernel name, algorithmic bytes of one call): every operand read once, every result written once
_KERNEL_OF = {
    "stes_clip_gate_bwd": (lambda a: "clip_gate_bwd_reduce_kernel" if a[5] is not None else "clip_gate_bwd_apply_kernel", None),
    # (names as a kernel trace prints them, for the shapes of the named configuration: all of them fit the LDS-window kernels)
    # (stride-1 maps of at most 14 columns go to the column-strip kernel: the rule of stes_conv_strip_takes)
    # (stride-2 maps 112, 56, 28 or 14 wide of even height and stride-1 maps 7 wide go to the row-ring kernels: the rules of
    # stes_conv_ring_takes, stes_dgrad_ring_takes and stes_conv_ring1_takes)
    "stes_conv3x3g_fwd": (lambda a: "conv_ring1_kernel<%d, %d, false, false>" % (a[-5], _RING1_ROWS_FWD[a[-5]]) if a[-2] == 1 and a[-5] in _RING1_ROWS else
                          "conv3x3g_strip_kernel<true, 1>" if a[-2] == 1 and a[-5] <= 14 else
                          "conv_ring2_kernel<%d, false>" % a[-5] if _ring2(a) else "conv3x3g_rows_kernel<%d, false>" % a[-2], _conv_bytes),
    "stes_conv3x3g_fwd_pool": (lambda a: "conv_ring1_kernel<%d, %d, false, true>" % (a[-5], _RING1_ROWS_FWD[a[-5]]) if a[-2] == 1 and a[-5] in _RING1_ROWS else
                               "conv3x3g_strip_kernel<true, 3>" if a[-2] == 1 and a[-5] <= 14 else
                               "conv_ring2_kernel<%d, true>" % a[-5] if _ring2(a) else "conv3x3g_rows_kernel<%d, false, true>" % a[-2], _conv_bytes),
    "stes_conv3x3g_dgrad": (lambda a: "conv_ring1_kernel<%d, %d, true, false>" % (a[-5], _RING1_ROWS[a[-5]]) if a[-2] == 1 and a[-5] in _RING1_ROWS else
                            "conv3x3g_rows_kernel<1, true>" if a[-2] == 1 else
                            "dgrad_ring2_kernel<%d>" % a[-5] if _ring2(a) else
                            "conv3x3g_dgrad2_kernel<%s>" % ("true" if a[3] is not None else "false"),
                            lambda a: _conv_bytes(a) + 2 * a[-1] * a[-7] * a[-6] * a[-5] * (a[3] is not None)),
    "stes_conv3x3g_wgrad": (lambda a: "wgrad_ring1_kernel<%d, %d>" % (a[-5], _RING1_ROWS_FWD[a[-5]]) if a[-2] == 1 and a[-5] in _RING1_ROWS else
                            "conv3x3g_wgrad_kernel", _conv_bytes),
    # (x, x_is_u8, mean, std, w, patches, out, sums..., e_scale, e_shift, e_slice, e_slice_n, B, T, Tv, Cin, H, W)
    "stes_stem_conv": (lambda a: "stem_conv_kernel<%s, %s>" % ("true" if a[1] else "false", "true" if a[9] is not None else "false"),
                       _stem_bytes),
    # (dY, ldy, dXf, fold, resid, ms, Ho, Wo, Hi, Wi, X, out, M, C, raw1, raw2, sums..., x_bits): every operand once (the
    # mask as X, 2 bytes per element, or as bits, an eighth of a byte)
    # gated shift: passes over the compact slice and over the gate-sized fp32 tensors (gate, gate gradient)
    "stes_gsm_forward": (lambda a: "gsm_fwd_small_kernel<%d>" % _geom_of(a)[1] if _geom_of(a)[1] < 32 else
                         "gsm_fwd_fused_kernel<%d>" % ((_geom_of(a)[1] // 2 + 31) // 32), _gsm_bytes(2, 1)),
    "stes_gsm_gate": ("gsm_gate_kernel", _gsm_bytes(1, 1)),
    "stes_gsm_shift": (lambda a: "gsm_shift_kernel<%d>" % (4 if _geom_of(a)[1] == 8 else 8), _gsm_bytes(2, 1)),
    "stes_gsm_bwd_gate": (lambda a: "gsm_bwd_gate_kernel<4>" if _geom_of(a)[1] == 8 else "gsm_bwd_gate_rows_kernel", _gsm_bytes(2, 2)),
    "stes_gsm_bwd_wgrad": ("gsm_bwd_wgrad_kernel", _gsm_bytes(1, 1)),
    "stes_gsm_dxbn": (lambda a: "gsm_dxbn_rows_kernel<%d>" % _geom_of(a)[1] if _geom_of(a)[1] <= 16 else "gsm_dxbn_kernel<8>", _gsm_bytes(2, 1)),
    "stes_gsm_bwd_conv": (lambda a: "gsm_bwd_conv_kernel<%d>" % ((_geom_of(a)[1] // 2 + 15) // 16), _gsm_bytes(2, 1)),
    "stes_gsm_bwd_dx": (lambda a: "gsm_bwd_dx_kernel<%d>" % (4 if _geom_of(a)[1] == 8 else 8), _gsm_bytes(4, 1)),
    "stes_bn_act_residual": ("bn_act_residual_kernel", _bn_act_bytes),
    # (c, scale, shift, mul, out, out_bf16, N, HW, C, norm): the map once (twice with a multiplier)
    "stes_bn_act_gate": ("bn_act_gate_kernel", lambda a: 4 * a[5] * a[6] * a[7]),
    "stes_chan_pool": ("chan_pool_kernel", lambda a: 2 * a[6] * a[7] * a[8] * (1 + (a[3] is not None))),
    # (c2, scale, shift, dA3, S, dgate, gate, dpre, N, HW, C): the raw map and the gradient once each
    "stes_se_bwd_pool": ("se_bwd_pool_kernel", lambda a: 4 * a[8] * a[9] * a[10]),
    "stes_block_in_grad": (lambda a: "block_in_grad_kernel<%s>" % ("true" if a[14] is not None else "false"),
                           lambda a: 2 * a[12] * a[13] * (2 + (a[14] is not None) + (a[15] is not None))
                           + (a[12] * a[13] // 8 if a[19] is not None else 2 * a[12] * a[13])
                           + (2 * a[12] * a[13] // (a[5] * a[5]) if a[4] is not None else 0)),
}
_KERNEL_OF["stes_block_in_grad_tsm"] = _KERNEL_OF["stes_block_in_grad"]          # (same kernel, same bytes)
_KERNEL_OF["stes_bn_act_residual_tsm"] = ("bn_act_residual_kernel", lambda a: 2 * a[7] * a[8] * (2 + (a[3] is not None)))


def _timed(kernel, nbytes, flops, fn):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    fn()
    e1.record()
    TIMING.append((kernel, e0, e1, nbytes, flops))


def call(name, *args):
    if TIMING is not None:
        kernel, fb = _KERNEL_OF.get(name, (name.replace("stes_", "") + "_kernel", None))
        if callable(kernel):
            kernel = kernel(args)
        _timed(kernel, fb(args) if fb else None, 0, lambda: check(getattr(lib(), name)(*args, stream()), name))
        return
    check(getattr(lib(), name)(*args, stream()), name)


def _gemm_work(a, out_bytes):
    m, n, k = a.M, a.N, a.K
    byt = m * k * 2 + n * k * 2 + m * n * out_bytes
    if a.a_gate:
        byt += (m // max(a.gate_group, 1)) * k * 4
    if a.relu_mask:
        byt += m * n * 2
    if a.a_raw:
        byt += m * k * 2 * (2 if a.a_out else 1)
    return byt, 2 * m * n * k


def gemm_streams(M, N, K):
    """True when stes_gemm runs a plain or BatchNorm-backward [M,K] x [N,K]^T bf16 product on its streaming kernel (the rule
    in stes_gemm_stream.hip), the only one with the BatchNorm-backward prologue."""
    return bool(_stream_slice(N)) and K in (64, 128, 320) and M >= 8192


def _stream_slice(N):
    """Columns per workgroup of the streaming GEMM for an N-column output, 0 if it does not take N (rule in stes_gemm_stream.hip)."""
    if N in (32, 64, 128, 160):
        return N
    if N == 320:
        return 160
    return 128 if N % 128 == 0 and N <= 1024 else 0


_RING = {320: (10, 1, 1), 128: (4, 1, 2), 64: (2, 2, 2), 32: (1, 4, 2)}       # K -> (k-steps, tiles in the ring, 16-row fragments)


def gemm_fuses_wgrad(M, N, K):
    """Shapes for which stes_gemm computes the layer's weight gradient in the launch of its input gradient."""
    return (K, N) == (64, 32) and M >= 8192


def gemm_has_tail(M, N, K, gate_group=None):
    """Shapes for which stes_gemm applies the block-tail epilogue (the last 1x1 conv of a stage-1, -2 or -3 bottleneck).
    gate_group: rows per squeeze-excite gate row when the operand is gated; the streaming kernel, the only one with the
    epilogue, takes gate groups of at least 32 rows (the rule in stes_gemm_stream.hip)."""
    return (N, K) in ((64, 64), (128, 128), (320, 320), (32, 32)) and M >= 8192 and (gate_group is None or gate_group >= 32)


def _gemm_kernel(a):
    """Which kernel stes_gemm dispatches to (the rules in stes_gemm_stream.hip and stes_gemm.hip)."""
    nb = _stream_slice(a.N)
    pro = 2 if a.a_raw else (1 if a.a_scale else 0)
    stream_ok = (nb and a.K in _RING and not (a.K == 32 and pro == 2) and not a.c_f32 and a.M >= 8192
                 and not a.relu_mask and not a.bias and a.act == 0 and a.out_scale in (0.0, 1.0)
                 and (not a.a_scale or 8 * a.K <= 16 * (nb * 2 + 16)) and (not a.a_gate or (a.a_scale and a.gate_group >= 32)))
    if stream_ok:
        return ("gemm_stream_wg_kernel" if a.wg_dw else "gemm_stream_tail_kernel" if a.e_scale else "gemm_stream_kernel") + "<%d, %d, %d, %d, %d, %d>" % (
            (nb // 16, 8 if nb >= 128 else 4, pro) + ((10, 2, 1) if a.K == 320 and pro == 0 else _RING[a.K]))    # (no prologue: two K = 320 tiles fit the ring)
    if (not a.a_scale and not a.a_gate and a.k0 == 0 and a.map_stride <= 1 and not a.col_sum and a.n_split == 0
            and a.M * a.N * a.K <= (1 << 28) and a.M <= 16 * 65535):
        return "gemm_small_kernel"
    wide = a.N > 64 and -(-a.M // 128) * -(-a.N // 128) >= 300 and not (a.N % 128 != 0 and a.N % 64 == 0)
    return "gemm_kernel<128>" if wide else "gemm_kernel<64>"


# ---------------------------------------------------------------- struct-argument entry points
def gemm(A, W, Cout, M, N, K, *, lda=None, ldw=None, ldc=None, A0=None, lda0=0, k0=0,
         scale=None, shift=None, relu=False, gate=None, gate_group=0, rowmap=None,
         bias=None, act=0, stats=None, relu_mask=None, ldmask=0, C0=None, ldc0=0, n_split=0, out_scale=0.0,
         bn_bwd=None, wgrad_fused=None, tail=None):
    """C = act(pro(A) W^T + bias); see stes_gemm in include/stes.h.  bn_bwd = (raw, (alpha, beta, delta), a_out | None);
    wgrad_fused = (Q, dW, Q0 | None, ldq0, k0, perm_fold[, rowmap]): the layer's weight gradient in the same launch
    (gemm_fuses_wgrad); rowmap = (stride, Ho, Wo, Hi, Wi) reads Q's rows through a strided 1x1 conv's pixel map."""
    a = GemmArgs()
    a.A, a.lda = ptr(A), lda if lda is not None else K
    a.A0, a.lda0, a.k0 = ptr(A0), lda0, k0
    a.a_scale, a.a_shift, a.a_relu = ptr(scale), ptr(shift), int(relu)
    a.a_gate, a.gate_group = ptr(gate), gate_group
    if rowmap is not None:
        a.map_stride, a.Ho, a.Wo, a.Hi, a.Wi = rowmap
    a.W, a.ldw = ptr(W), ldw if ldw is not None else K
    a.bias, a.act = ptr(bias), act
    a.C, a.ldc, a.c_f32 = ptr(Cout), ldc if ldc is not None else N, int(Cout.dtype == torch.float32)
    if stats is not None:
        a.col_sum, a.col_sumsq = ptr(stats[0]), ptr(stats[1])
    a.relu_mask, a.ldmask = ptr(relu_mask), ldmask
    a.C0, a.ldc0, a.n_split = ptr(C0), ldc0, n_split
    a.M, a.N, a.K, a.out_scale = M, N, K, out_scale
    if bn_bwd is not None:
        raw, coef, a_out = bn_bwd
        a.a_raw, a.bn_alpha, a.bn_beta, a.bn_delta = ptr(raw), ptr(coef[0]), ptr(coef[1]), ptr(coef[2])
        a.a_out, a.lda_out = ptr(a_out), K
    if wgrad_fused is not None:
        q, dw, q0, ldq0, wk0, perm = wgrad_fused[:6]
        a.wg_q, a.wg_ldq, a.wg_q0, a.wg_ldq0, a.wg_k0, a.wg_perm_fold = ptr(q), N, ptr(q0), ldq0, wk0, perm
        a.wg_dw, a.wg_lddw = ptr(dw), N
        if len(wgrad_fused) > 6 and wgrad_fused[6] is not None:
            a.wg_map_stride, a.Ho, a.Wo, a.Hi, a.Wi = wgrad_fused[6]
    if tail is not None:        # (scale, shift, res, res_scale | None, res_shift | None, slice | None, slice_n): block-tail epilogue
        es, eb, res, rs, rb, slc, slc_n = tail
        a.e_scale, a.e_shift, a.e_res, a.e_ldres, a.e_rscale, a.e_rshift = ptr(es), ptr(eb), ptr(res), N, ptr(rs), ptr(rb)
        a.e_slice, a.e_slice_n = ptr(slc), (slc_n if slc is not None else 0)
    if TIMING is not None:
        byt, fl = _gemm_work(a, 4 if a.c_f32 else 2)
        if a.e_scale:
            byt += 2 * a.M * ((a.N if a.e_res else 0) + a.e_slice_n)
        if a.wg_dw:
            byt, fl = byt + 2 * a.M * a.N, 2 * fl
        _timed(_gemm_kernel(a), byt, fl, lambda: check(lib().stes_gemm(C.byref(a), stream()), "stes_gemm"))
        return
    check(lib().stes_gemm(C.byref(a), stream()), "stes_gemm")


def gemm2(A, W1, H, W2, Cout, M, K1, N1, N2, *, bias1=None, act1=0, relu_mask1=None, ldmask1=0, bias2=None, act2=0, out_scale=0.0,
          ldw1=None, ldw2=None):
    """H = act1(A W1^T + bias1) [masked], Cout = act2(H W2^T + bias2) * out_scale in one launch; see stes_gemm_small2 in
    include/stes.h (the squeeze-excite FC pair and its input-gradient pair)."""
    a = Gemm2Args()
    a.A, a.lda, a.M, a.K1 = ptr(A), K1, M, K1
    a.W1, a.ldw1, a.N1, a.bias1, a.act1 = ptr(W1), ldw1 if ldw1 is not None else K1, N1, ptr(bias1), act1
    a.relu_mask1, a.ldmask1 = ptr(relu_mask1), ldmask1
    a.H, a.ldh = ptr(H), N1
    a.W2, a.ldw2, a.N2, a.bias2, a.act2, a.out_scale = ptr(W2), ldw2 if ldw2 is not None else N1, N2, ptr(bias2), act2, out_scale
    a.C, a.ldc, a.c_f32 = ptr(Cout), N2, int(Cout.dtype == torch.float32)
    fn = lambda: check(lib().stes_gemm_small2(C.byref(a), stream()), "stes_gemm_small2")
    if TIMING is not None:
        _timed("gemm_small2_kernel", 2 * M * (K1 + N1) + 2 * N1 * (K1 + N2) + M * N2 * (4 if a.c_f32 else 2), 2 * M * N1 * (K1 + N2), fn)
        return
    fn()


def block_front_takes(cin, cmid, H, W, stride, fold):
    """Shapes for which stes_block_front_eval runs conv1 -> bn/act -> conv2 (+ squeeze) of a bottleneck in one kernel (the rule
    in stes_block_front.hip: RegNet-Y 800MF's s1.b1, s2.b1, s3.b1 and the identity blocks of stage 2)."""
    return H == W and (stride, cin, cmid, W) in ((2, 32, 64, 112), (2, 64, 128, 56), (1, 128, 128, 28), (2, 128, 320, 28)) and fold % 8 == 0 and fold <= 32


def block_front(X, G, fold, W1, bn1, wp2, bn2, c2, pooled, N, H, W, cin, cmid, stride):
    """conv1 + bn1/ReLU + grouped conv2 + squeeze of a bottleneck in evaluation mode; see stes_block_front_eval."""
    a = BlockFrontArgs()
    a.X, a.G, a.fold = ptr(X), ptr(G), fold
    a.W1, a.s1, a.b1 = ptr(W1), ptr(bn1[0]), ptr(bn1[1])
    a.wp2, a.s2, a.b2 = ptr(wp2), ptr(bn2[0]), ptr(bn2[1])
    a.c2, a.pooled = ptr(c2), ptr(pooled)
    a.N, a.H, a.W, a.Cin, a.Cmid, a.stride = N, H, W, cin, cmid, stride
    fn = lambda: check(lib().stes_block_front_eval(C.byref(a), stream()), "stes_block_front_eval")
    if TIMING is not None:
        m = N * H * W
        mo = m // (stride * stride)
        _timed("block_front_kernel<%d, %d, %d, %d>" % (cin, cmid, W, stride), 2 * m * cin + 2 * mo * cmid, 2 * m * cin * cmid + 2 * mo * cmid * 144, fn)
        return
    fn()


def block_eval_takes(C_, H, W, rd, fold, stride, down):
    """Shapes for which stes_block_eval_s3 runs a whole bottleneck's evaluation forward in one kernel (the rule in
    stes_block_eval.hip: RegNet-Y 800MF's stage-3 identity blocks)."""
    return (C_, H, W, rd, stride) == (320, 14, 14, 80, 1) and fold in (0, 80) and not down


def block_eval(X, G, W1, bn1, wp2, bn2, Wse1, bse1, Wse2, bse2, W3, bn3, out, slc, slice_n, N):
    """One stage-3 bottleneck in evaluation mode; bn* = (scale, shift) of the evaluation-mode BatchNorms; see stes_block_eval_s3."""
    a = BlockEvalArgs()
    a.X, a.G = ptr(X), ptr(G)
    a.W1, a.s1, a.b1 = ptr(W1), ptr(bn1[0]), ptr(bn1[1])
    a.wp2, a.s2, a.b2 = ptr(wp2), ptr(bn2[0]), ptr(bn2[1])
    a.Wse1, a.bse1, a.Wse2, a.bse2 = ptr(Wse1), ptr(bse1), ptr(Wse2), ptr(bse2)
    a.W3, a.s3, a.b3 = ptr(W3), ptr(bn3[0]), ptr(bn3[1])
    a.out, a.slice, a.slice_n = ptr(out), ptr(slc), slice_n
    a.N, a.H, a.W, a.C, a.rd, a.fold = N, 14, 14, 320, 80, (80 if G is not None else 0)
    fn = lambda: check(lib().stes_block_eval_s3(C.byref(a), stream()), "stes_block_eval_s3")
    if TIMING is not None:
        m = N * 196
        _timed("block_eval_s3_kernel", 2 * m * (320 + a.fold + 320 + slice_n), 2 * m * (2 * 320 * 320 + 320 * 144), fn)     # (X counted once: its second read for the shortcut is not algorithmic)
        return
    fn()


def wgrad(P, Q, dW, M, N1, N2, *, ldp=None, ldq=None, lddw=None, Q0=None, ldq0=0, k0=0,
          scale=None, shift=None, relu=False, gate=None, gate_group=0, rowmap=None, dbias=None, perm_fold=0, bn_bwd=None, ws=None):
    """dW += P^T pro(Q); see stes_wgrad in include/stes.h.  bn_bwd = (raw, (alpha, beta, delta)): BatchNorm backward on P."""
    a = WgradArgs()
    a.P, a.ldp = ptr(P), ldp if ldp is not None else N1
    a.Q, a.ldq = ptr(Q), ldq if ldq is not None else N2
    a.Q0, a.ldq0, a.k0 = ptr(Q0), ldq0, k0
    a.q_scale, a.q_shift, a.q_relu = ptr(scale), ptr(shift), int(relu)
    a.q_gate, a.gate_group = ptr(gate), gate_group
    if rowmap is not None:
        a.map_stride, a.Ho, a.Wo, a.Hi, a.Wi = rowmap
    a.dW, a.lddw, a.dbias = ptr(dW), lddw if lddw is not None else N2, ptr(dbias)
    a.perm_fold = perm_fold
    a.M, a.N1, a.N2 = M, N1, N2
    if ws is not None:
        a.ws, a.ws_floats = ptr(ws), ws.numel()
    if bn_bwd is not None:
        a.p_raw, (a.bn_alpha, a.bn_beta, a.bn_delta) = ptr(bn_bwd[0]), [ptr(t) for t in bn_bwd[1]]
    if TIMING is not None:
        strip = N1 == 320 and N2 == 320 and M % 32 == 0 and M >= 8192 and dbias is None and bn_bwd is None    # the rule in stes_gemm.hip
        kernel = "wgrad_strip_kernel" if strip else ("wgrad_kernel<128, 8>" if N1 > 64 and N2 > 64 else "wgrad_kernel<64, 4>")
        _timed(kernel, 2 * M * (N1 + N2 + (N1 if bn_bwd else 0)), 2 * M * N1 * N2,
               lambda: check(lib().stes_wgrad(C.byref(a), stream()), "stes_wgrad"))
        return
    check(lib().stes_wgrad(C.byref(a), stream()), "stes_wgrad")


def _bn_bwd_args(D, raw1, M, Cc, *, mul=None, add=None, group=0, act=None, raw2=None,
                 sums=None, coef=None, out=None):
    a = BnBwdArgs()
    a.D, a.mul, a.add, a.group = ptr(D), ptr(mul), ptr(add), group
    if act is not None:
        a.act_raw, a.act_scale, a.act_shift = ptr(act[0]), ptr(act[1]), ptr(act[2])
    a.raw1, a.raw2 = ptr(raw1), ptr(raw2)
    if sums is not None:
        a.s_dy, a.s_dyr1 = ptr(sums[0]), ptr(sums[1])
        a.s_dyr2 = ptr(sums[2]) if len(sums) > 2 else None
    if coef is not None:
        a.alpha, a.beta, a.delta = ptr(coef[0]), ptr(coef[1]), ptr(coef[2])
    a.out = ptr(out)
    a.M, a.C = M, Cc
    return a


def bn_bwd_reduce(D, raw1, M, Cc, sums, **kw):
    a = _bn_bwd_args(D, raw1, M, Cc, sums=sums, **kw)
    if TIMING is not None:
        _timed("bn_bwd_reduce_kernel", 2 * M * Cc * (2 + (kw.get("raw2") is not None)), 0,
               lambda: check(lib().stes_bn_bwd_reduce(C.byref(a), stream()), "stes_bn_bwd_reduce"))
        return
    check(lib().stes_bn_bwd_reduce(C.byref(a), stream()), "stes_bn_bwd_reduce")


def bn_bwd_apply(D, raw1, M, Cc, coef, out, **kw):
    a = _bn_bwd_args(D, raw1, M, Cc, coef=coef, out=out, **kw)
    if TIMING is not None:
        # gradient and raw activation read once, result written once (the ReLU-mask source, when given, is raw1 itself)
        _timed("bn_bwd_apply_kernel", 6 * M * Cc, 0, lambda: check(lib().stes_bn_bwd_apply(C.byref(a), stream()), "stes_bn_bwd_apply"))
        return
    check(lib().stes_bn_bwd_apply(C.byref(a), stream()), "stes_bn_bwd_apply")


def gsm_geom(B, T, H, W, fold):
    g = GsmGeom()
    g.B, g.T, g.H, g.W, g.fold = B, T, H, W, fold
    return g


def gru_args(**kw):
    a = GruArgs()
    for k, v in kw.items():
        setattr(a, k, ptr(v) if isinstance(v, torch.Tensor) else v)
    return a
