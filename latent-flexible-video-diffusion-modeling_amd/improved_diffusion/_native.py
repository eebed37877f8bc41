"""ctypes binding of the gfx950 C ABI (include/lfvdm_hip.h).

PyTorch is used for device memory and streams only: every call passes raw device pointers,
integer shapes and the current HIP stream.  There is NO fallback: if ``liblfvdm_hip.so`` is
missing, importing the product on a GPU path raises immediately (build it with
``python latent-flexible-video-diffusion-modeling_amd/build.py``).
"""
import atexit
import ctypes as C
import json
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("LFVDM_LIB_PATH") or os.path.join(os.path.dirname(_HERE), "lib", "liblfvdm_hip.so")   # override: A/B builds

ACT_NONE, ACT_SILU = 0, 1
OUT_ROWS, OUT_NCHW = 0, 1

_lib = None

c_fp = C.c_void_p
c_i = C.c_int


class ConvArgs(C.Structure):
    _fields_ = [
        ("src0", c_fp), ("src1", c_fp), ("C0", C.c_int32), ("C1", C.c_int32), ("N", C.c_int32),
        ("Hs", C.c_int32), ("Ws", C.c_int32), ("up", C.c_int32), ("stride", C.c_int32), ("ksize", C.c_int32),
        ("Ho", C.c_int32), ("Wo", C.c_int32), ("coefA", c_fp), ("coefB", c_fp), ("act", C.c_int32),
        ("W", c_fp), ("bias", c_fp), ("Cout", C.c_int32),
        ("s2src0", c_fp), ("s2src1", c_fp), ("s2C0", C.c_int32), ("s2C1", C.c_int32), ("W2", c_fp), ("bias2", c_fp),
        ("res", c_fp), ("ldr", C.c_int32), ("resA", c_fp), ("resB", c_fp),
        ("out", c_fp), ("ldo", C.c_int32), ("out_mode", C.c_int32), ("tune", C.c_int32),
        ("splitk_ws", c_fp), ("splitk_cnt", c_fp), ("splitk_ws_floats", C.c_int64), ("splitk_cnt_ints", C.c_int64),
        ("gn_gamma", c_fp), ("gn_beta", c_fp), ("gn_film", c_fp), ("gn_out", c_fp), ("gn_film_ld", C.c_int32),
        ("gn_film_div", C.c_int32), ("gn_act", C.c_int32), ("gn_skip_raw", C.c_int32), ("gn_eps", C.c_float),
        ("gn_general", C.c_int32), ("gn_gw", C.c_int32), ("gn_ld", C.c_int32),
    ]


class GnArgs(C.Structure):
    _fields_ = [("src0", c_fp), ("src1", c_fp), ("C0", C.c_int32), ("C1", C.c_int32), ("N", C.c_int32), ("P", C.c_int32),
                ("gamma", c_fp), ("beta", c_fp), ("film", c_fp), ("film_div", C.c_int32), ("film_ld", C.c_int32),
                ("eps", C.c_float), ("act", C.c_int32), ("out", c_fp), ("cg", C.c_int32), ("ldo", C.c_int32), ("out_base", c_fp),
                ("out_col", C.c_int32), ("pad_", C.c_int32)]


class ChainStage(C.Structure):
    """lfvdm_chain_stage (include/lfvdm_hip.h): one stage of a persistent level chain."""
    _fields_ = [("kind", C.c_int32), ("n_items", C.c_int32), ("flag_base", C.c_int32), ("n_flags", C.c_int32),
                ("dep_base", C.c_int32), ("dep_stride", C.c_int32), ("cfg", C.c_int32), ("kz", C.c_int32), ("nt2", C.c_int32),
                ("wg_off", C.c_int32), ("side", C.c_int32), ("wg_lo", C.c_int32), ("wg_count", C.c_int32), ("ws_off", C.c_int64), ("cnt_off", C.c_int64), ("conv", ConvArgs), ("gn", GnArgs)]


CHAIN_CONV, CHAIN_GN, CHAIN_LOCAL = 0, 1, 2
CHAIN_CTL_EPOCH, CHAIN_CTL_EXIT, CHAIN_CTL_ABORT, CHAIN_CTL_INTS = 0, 32, 64, 96


class AdamWArgs(C.Structure):
    _fields_ = [("p", c_fp), ("g", c_fp), ("m", c_fp), ("v", c_fp), ("ema", c_fp * 4), ("ema_rate", C.c_float * 4),
                ("n_ema", C.c_int32), ("n", C.c_int64), ("lr", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float),
                ("eps", C.c_float), ("weight_decay", C.c_float), ("bias_corr1", C.c_float), ("bias_corr2_sqrt", C.c_float),
                ("grad_scale", C.c_float), ("grad_sqsum", c_fp), ("skip_flag", c_fp), ("skip_flag2", c_fp)]


class RowdotJob(C.Structure):
    _fields_ = [("W", c_fp), ("b", c_fp), ("inp", c_fp), ("out", c_fp), ("K", C.c_int32), ("O", C.c_int32),
                ("M", C.c_int32), ("ldin", C.c_int32), ("ldout", C.c_int32), ("in_mode", C.c_int32),
                ("row0", C.c_int32), ("pad_", C.c_int32)]


class RowdotBwdJob(C.Structure):
    _fields_ = [("W", c_fp), ("inp", c_fp), ("dout", c_fp), ("dW", c_fp), ("db", c_fp), ("din", c_fp), ("K", C.c_int32),
                ("O", C.c_int32), ("M", C.c_int32), ("ldin", C.c_int32), ("lddout", C.c_int32), ("lddin", C.c_int32),
                ("in_mode", C.c_int32), ("task0", C.c_int32)]


class PackJob(C.Structure):
    _fields_ = [("src", c_fp), ("dst", c_fp), ("Cout", C.c_int32), ("Cin", C.c_int32), ("taps", C.c_int32),
                ("transposed", C.c_int32), ("blk0", C.c_int32), ("ld", C.c_int32)]


# bumped by code that rewrites parameters through raw pointers (the fused AdamW): cached packed weights are stale
param_epoch = [0]


class UnpackJob(C.Structure):
    _fields_ = [("gp", c_fp), ("g", c_fp), ("Cout", C.c_int32), ("Cin", C.c_int32), ("taps", C.c_int32), ("row0", C.c_int32),
                ("ldp", C.c_int32), ("pad_", C.c_int32)]


class RpeJob(C.Structure):
    _fields_ = [("tproj", c_fp), ("Wd", c_fp), ("bd", c_fp), ("Wout", c_fp), ("bout", c_fp), ("R", c_fp),
                ("C", C.c_int32), ("tile0", C.c_int32), ("tproj_ld", C.c_int32), ("pad_", C.c_int32), ("act", c_fp)]


class RpeBwdJob(C.Structure):
    _fields_ = [("tproj", c_fp), ("Wd", c_fp), ("bd", c_fp), ("Wout_t", c_fp), ("dR", c_fp), ("dtproj", c_fp),
                ("dWd", c_fp), ("dbd", c_fp), ("C", C.c_int32), ("tile0", C.c_int32), ("tproj_ld", C.c_int32),
                ("dtproj_ld", C.c_int32)]


class WgradJob(C.Structure):
    _fields_ = [("a", ConvArgs), ("msplit", C.c_int32), ("task0", C.c_int32)]


_SIGS = {
    "lfvdm_abi_version": ([], c_i),
    "lfvdm_conv_igemm": ([C.POINTER(ConvArgs), c_fp], c_i),
    "lfvdm_conv_igemm_config": ([C.POINTER(ConvArgs), C.POINTER(c_i), C.POINTER(c_i)], c_i),
    "lfvdm_conv_igemm_candidates": ([C.POINTER(ConvArgs), C.POINTER(c_i), c_i], c_i),
    "lfvdm_pack_conv_weight": ([c_fp, c_fp, c_i, c_i, c_i, c_fp], c_i),
    "lfvdm_conv_wgrad": ([C.POINTER(ConvArgs), c_fp], c_i),
    "lfvdm_pack_conv_weight_t": ([c_fp, c_fp, c_i, c_i, c_i, c_fp], c_i),
    "lfvdm_unpack_conv_grad": ([c_fp, c_fp, c_i, c_i, c_i, c_i, c_fp], c_i),
    "lfvdm_unpack_conv_grads": ([c_fp, c_i, c_i, c_i, c_fp], c_i),
    "lfvdm_pack_conv_weights": ([c_fp, c_i, c_i, c_fp], c_i),
    "lfvdm_conv_in": ([c_fp] * 6 + [c_i] * 5 + [c_fp], c_i),
    "lfvdm_gn_coef": ([c_fp, c_fp, c_i, c_i, c_i, c_i, c_fp, c_fp, c_fp, c_i, c_i, C.c_float, c_fp, c_fp, c_fp], c_i),
    "lfvdm_gn_coef_stats": ([c_fp, c_fp, c_i, c_i, c_i, c_i, c_fp, c_fp, c_fp, c_i, c_i, C.c_float, c_fp, c_fp, c_fp, c_fp], c_i),
    "lfvdm_gn_apply": ([c_fp, c_fp, c_i, c_i, c_i, c_i, c_fp, c_fp, c_fp, c_i, c_i, C.c_float, c_i, c_fp, c_fp, c_fp, c_fp, c_fp], c_i),
    "lfvdm_gn_apply_part": ([c_fp, c_i, c_i, c_i, c_i, c_fp, c_fp, C.c_float, c_i, c_fp, c_i, c_fp], c_i),
    "lfvdm_gn_apply_ws_floats": ([c_i, c_i, c_i], C.c_long),
    "lfvdm_gn_apply_ws": ([c_fp, c_fp, c_i, c_i, c_i, c_i, c_fp, c_fp, c_fp, c_i, c_i, C.c_float, c_i, c_fp, c_fp, c_fp, c_fp, c_fp,
                           C.c_long, c_fp], c_i),
    "lfvdm_compose_rows": ([c_fp, c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_fp], c_i),
    "lfvdm_conv_in_tick": ([c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_fp, c_fp, c_fp, c_i, c_fp, c_i, c_fp, c_i,
                            c_fp], c_i),
    "lfvdm_p_sample_rng": ([c_fp] * 9 + [c_i, c_fp, c_fp, c_fp, c_i, c_i, c_fp, c_fp], c_i),
    "lfvdm_conv_out_psample_ok": ([c_i] * 5, c_i),
    "lfvdm_conv_out_psample": ([c_fp] * 13 + [c_i, c_fp, c_fp, c_fp] + [c_i] * 6 + [c_fp, c_fp], c_i),
    "lfvdm_masked_mse_bwd": ([c_fp, c_fp, c_fp, c_fp, c_fp, c_i, c_i, c_i, c_fp], c_i),
    "lfvdm_gn_bwd_stats": ([c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_fp, c_fp, c_fp, c_i, c_fp, c_fp], c_i),
    "lfvdm_gn_bwd_apply": ([c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_fp, c_fp, c_fp, c_fp, c_i, c_fp, c_fp, c_i, c_i, c_fp], c_i),
    "lfvdm_gn_bwd_apply_params": ([c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_fp, c_fp, c_fp, c_fp, c_i, c_fp, c_fp, c_i, c_i,
                                   c_fp, c_fp, c_fp, c_i, c_i, c_fp, c_fp, c_fp, c_i, c_fp, c_i, c_fp], c_i),
    "lfvdm_gn_bwd_fused_sums": ([c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_fp, c_fp, c_fp, c_i, c_fp, c_fp, c_fp, c_i, c_fp, c_i,
                                 c_fp, c_fp], c_i),
    "lfvdm_gn_bwd_ws_floats": ([c_i, c_i, c_i], C.c_long),
    "lfvdm_gn_bwd_ws": ([c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_fp, c_fp, c_fp, c_i, c_fp, c_fp, c_fp, c_fp, c_fp, c_i, c_i,
                         c_fp, c_fp, c_fp, c_i, c_fp, c_i, c_fp, c_i, c_fp, c_fp, C.c_long, c_fp], c_i),
    "lfvdm_gn_bwd_fused": ([c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_fp, c_fp, c_fp, c_i, c_fp, c_fp, c_fp, c_fp, c_fp, c_i, c_i,
                            c_fp, c_fp, c_fp, c_i, c_fp, c_i, c_fp, c_i, c_fp], c_i),
    "lfvdm_gn_param_grads": ([c_fp, c_fp, c_fp, c_fp, c_i, c_i, c_fp, c_fp, c_fp, c_i, c_i, c_i, c_fp], c_i),
    "lfvdm_rowdot_bwd": ([c_fp, c_i, c_i, c_fp], c_i),
    "lfvdm_gn_temporal_bwd": ([c_fp, c_fp, c_fp, C.c_float, c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_fp], c_i),
    "lfvdm_gn_temporal": ([c_fp, c_fp, c_fp, C.c_float, c_fp, c_i, c_i, c_i, c_i, c_fp], c_i),
    "lfvdm_gn_temporal_qkv_ok": ([c_i, c_i, c_i, c_i], c_i),
    "lfvdm_proj_gn_ok": ([c_i, c_i, c_i], c_i),
    "lfvdm_proj_gn": ([c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, C.c_float, c_i, c_fp, c_fp, c_i, c_i, c_i, c_fp], c_i),
    "lfvdm_gn_temporal_qkv": ([c_fp, c_fp, c_fp, C.c_float, c_fp, c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_fp], c_i),
    "lfvdm_rowdot": ([c_fp, c_i, c_i, c_fp], c_i),
    "lfvdm_silu": ([c_fp, c_fp, C.c_int64, c_fp], c_i),
    "lfvdm_rpe_nets": ([c_fp, c_i, c_i, c_fp, c_i, c_i, c_fp], c_i),
    "lfvdm_rpe_nets_maxc": ([c_fp, c_i, c_i, c_fp, c_i, c_i, c_i, c_fp], c_i),
    "lfvdm_rpe_nets_bwd": ([c_fp, c_i, c_i, c_fp, c_i, c_i, c_fp], c_i),
    "lfvdm_conv_wgrad_grouped": ([c_fp, c_i, c_i, c_fp], c_i),
    "lfvdm_attn_spatial": ([c_fp, c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_fp], c_i),
    "lfvdm_attn_spatial_fused_ok": ([c_i, c_i, c_i, c_i], c_i),
    "lfvdm_attn_spatial_fused": ([c_fp, c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_fp], c_i),
    "lfvdm_det_reduce": ([c_fp, c_fp, C.c_long, C.c_long, c_fp], c_i),
    "lfvdm_gn_temporal_bwd_det": ([c_fp, c_fp, c_fp, C.c_float, c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_fp, C.c_int64, c_fp], c_i),
    "lfvdm_rowdot_bwd_det": ([c_fp, c_i, c_i, c_fp, C.c_int64, c_fp, C.c_int64, c_fp], c_i),
    "lfvdm_rpe_nets_bwd_det": ([c_fp, c_i, c_i, c_fp, c_i, c_i, c_fp, C.c_int64, c_fp], c_i),
    "lfvdm_rpe_front": ([c_fp, c_i, c_fp, c_fp, c_fp, c_fp, c_i, c_i, c_i, c_fp], c_i),
    "lfvdm_rpe_front_bwd": ([c_fp, c_i, c_fp, c_fp, c_fp, c_fp, c_fp, c_i, c_fp, c_fp, c_i, c_i, c_i, c_fp], c_i),
    "lfvdm_sampler_tick": ([c_fp, c_fp, c_fp, c_i, c_fp], c_i),
    "lfvdm_sampler_tick_fetch": ([c_fp, c_fp, c_fp, c_i, c_fp, c_i, c_fp, c_i, c_fp], c_i),
    "lfvdm_attn_temporal_sel": ([c_fp] * 7 + [c_i] * 5 + [c_fp, c_fp], c_i),
    "lfvdm_attn_temporal_ring": ([c_fp] * 7 + [c_i] * 5 + [c_fp, c_i, c_fp], c_i),
    "lfvdm_attn_temporal_bwd": ([c_fp] * 12 + [c_i] * 5 + [c_fp], c_i),
    "lfvdm_attn_spatial_bwd": ([c_fp] * 6 + [c_i, c_i, c_i, c_i, c_fp], c_i),
    "lfvdm_attn_temporal": ([c_fp] * 7 + [c_i] * 5 + [c_fp], c_i),
    "lfvdm_adamw_ema": ([C.POINTER(AdamWArgs), c_fp], c_i),
    "lfvdm_q_sample": ([c_fp] * 6 + [c_i, c_i, c_fp], c_i),
    "lfvdm_p_sample": ([c_fp] * 9 + [c_i] + [c_fp] * 3 + [c_i, c_i, c_fp], c_i),
    "lfvdm_masked_mse": ([c_fp] * 4 + [c_i, c_i, c_i, c_fp], c_i),
    "lfvdm_prepare_batch": ([c_fp] * 6 + [c_i] * 4 + [c_fp], c_i),
    "lfvdm_chain_plan": ([C.POINTER(ChainStage), c_i, C.POINTER(C.c_int32), C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_int32),
                          C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int32), C.POINTER(C.c_int32)], c_i),
    "lfvdm_chain_conv_ok": ([C.POINTER(ConvArgs)], c_i),
    "lfvdm_chain_gn_ok": ([c_i, c_i, c_i, c_i], c_i),
    "lfvdm_chain_local_ok": ([C.POINTER(ConvArgs), c_i], c_i),
    "lfvdm_chain_capacity": ([c_i], c_i),
    "lfvdm_level_chain": ([c_fp, c_i, c_fp, c_fp, c_fp, c_i, c_i, C.c_double, c_fp], c_i),
    "lfvdm_flag_add": ([c_fp, c_fp], c_i),
    "lfvdm_flag_wait": ([c_fp, c_i, C.c_double, c_fp, c_fp], c_i),
    "lfvdm_flag_wait2": ([c_fp, c_i, C.c_double, c_fp, c_fp, c_fp], c_i),
}

EXPORTS = tuple(_SIGS)


def lib():
    """Load (once) and return the shared library; raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"native library {LIB_PATH} is missing - build it with "
                "`python latent-flexible-video-diffusion-modeling_amd/build.py` (there is no fallback path)")
        L = C.CDLL(LIB_PATH)
        for name, (argt, rest) in _SIGS.items():
            fn = getattr(L, name)
            fn.argtypes = argt
            fn.restype = rest
        _lib = L
    return _lib


_ERR = {1: "invalid shape", 2: "HIP launch error", 3: "unsupported configuration"}


def check(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what} failed: {_ERR.get(rc, rc)}")


def stream():
    return torch.cuda.current_stream().cuda_stream


def ptr(t, dtype=torch.float32):
    """Device pointer of a contiguous CUDA tensor (None -> NULL)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("native ops need device (HIP) tensors; the product has no CPU path")
    if t.dtype != dtype:
        raise RuntimeError(f"expected {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise RuntimeError("expected a contiguous tensor")
    return t.data_ptr()


# ------------------------------------------------------------------------------ thin wrappers
def conv_igemm(**kw):
    """Fill an lfvdm_conv_args from keyword tensors/ints (see fill_conv_args) and launch.  Required: src0, C0, N, Hs,
    Ws, Ho, Wo, W, Cout, out.  Everything else defaults to 'absent'."""
    a = fill_conv_args(**kw)
    ws, cnt = splitk_workspace(kw["out"].device)
    a.splitk_ws, a.splitk_cnt, a.splitk_ws_floats, a.splitk_cnt_ints = ws.data_ptr(), cnt.data_ptr(), ws.numel(), cnt.numel()
    a.tune = tuned_code(a)
    check(lib().lfvdm_conv_igemm(C.byref(a), stream()), "lfvdm_conv_igemm")


_splitk = {}


def splitk_workspace(device):
    """Per-device split-K workspace shared by all eager launches (they are stream-ordered): 8 MiB of slabs and
    zero-initialised, self-cleaning arrival tickets."""
    ent = _splitk.get(device)
    if ent is None:
        ent = _splitk[device] = (torch.empty(2 * 1024 * 1024, device=device, dtype=torch.float32),
                                 torch.zeros(4096, device=device, dtype=torch.int32))
    return ent


# ------------------------------------------------------------------------------------------- launch tuning
# Launch shape -> tile/K-chunk/split-K code (lfvdm_conv_args.tune), measured on the device the first time a
# shape is launched outside of stream capture.  LFVDM_TUNE_CACHE names a JSON table that is loaded READ-ONLY
# (entries are only valid for the library ABI version they were measured with; the committed
# profiles/tune_cache_mi355x.json is used this way by bench.py and the profiling tools, so ordinary runs never modify a
# tracked file).  Newly measured entries are written at exit only if LFVDM_TUNE_CACHE_OUT names a file (may equal
# LFVDM_TUNE_CACHE to refresh it in place - tools/refresh_profiles.sh does), and only by rank 0 of a multi-process job.
_tune = None
_tune_saved = 0


# markers appended to a launch-shape key (integers: the table is stored as JSON lists of ints): the fastest code among the
# variants the persistent level chain holds, measured per launch / measured inside its chain
TUNE_CHAIN, TUNE_IN_CHAIN, TUNE_LOCAL_RT = -101, -102, -103      # (-103: row tiles per item of a sample-local chain stage)


def tune_key(a):
    return (a.N, a.Hs, a.Ws, a.up, a.stride, a.ksize, a.Ho, a.Wo, a.C0, a.C1, a.Cout, a.s2C0, a.s2C1, bool(a.coefA), a.act,
            bool(a.res), bool(a.resA), a.out_mode, bool(a.splitk_ws),
            (1 + bool(a.gn_film) + 2 * bool(a.gn_skip_raw) + 4 * bool(a.gn_gw)) if a.gn_out else 0)


def tune_cache():
    global _tune, _tune_saved
    if _tune is None:
        _tune = {}
        for path in (os.environ.get("LFVDM_TUNE_CACHE", ""), os.environ.get("LFVDM_TUNE_CACHE_OUT", "")):
            if path and os.path.exists(path):
                try:
                    with open(path) as f:
                        blob = json.load(f)
                    if blob.get("abi") == int(lib().lfvdm_abi_version()):
                        _tune.update({tuple(json.loads(k)): int(v) for k, v in blob["entries"].items()})
                except (OSError, ValueError, KeyError):
                    pass
        _tune_saved = len(_tune)
        atexit.register(tune_cache_save)
    return _tune


def tune_cache_save():
    global _tune_saved
    path = os.environ.get("LFVDM_TUNE_CACHE_OUT", "")
    if not path or _tune is None or len(_tune) == _tune_saved or os.environ.get("RANK", "0") != "0":
        return
    try:
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        tmp = f"{path}.{os.getpid()}.tmp"
        with open(tmp, "w") as f:
            json.dump({"abi": int(lib().lfvdm_abi_version()),
                       "entries": {json.dumps([int(x) for x in k]): int(v) for k, v in sorted(_tune.items())}}, f, indent=0)
        os.replace(tmp, path)
        _tune_saved = len(_tune)
    except OSError:
        pass


def autotune_launch(a, rounds=3, reps=6, chain_only=False):
    """Time every legal variant of this implicit-GEMM launch (the launch is idempotent) and return the fastest
    code.  Must not be called while the stream is capturing.  LFVDM_TUNE_REPS / LFVDM_TUNE_ROUNDS: longer measurements
    (what the committed table was made with: 20 x 5).  chain_only: only the variants the persistent level chain holds
    (lfvdm_chain_conv_ok: 4-wave tiles, 32-channel chunks, plain split-K); 0 if there is none."""
    rounds = int(os.environ.get("LFVDM_TUNE_ROUNDS", rounds))
    reps = int(os.environ.get("LFVDM_TUNE_REPS", reps))
    L, s = lib(), stream()
    codes = (C.c_int * 256)()
    n = L.lfvdm_conv_igemm_candidates(C.byref(a), codes, 256)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best, best_t = 0, float("inf")
    for code in ([] if chain_only else [0]) + [codes[i] for i in range(n)]:
        a.tune = code
        if chain_only and L.lfvdm_chain_conv_ok(C.byref(a)) != 0:
            continue
        if L.lfvdm_conv_igemm(C.byref(a), s) != 0:
            continue
        t_min = float("inf")
        for _ in range(rounds):
            ev0.record()
            for _ in range(reps):
                L.lfvdm_conv_igemm(C.byref(a), s)
            ev1.record()
            ev1.synchronize()
            t_min = min(t_min, ev0.elapsed_time(ev1))
        if t_min < best_t * 0.98:      # prefer earlier (simpler) candidates on ties
            best, best_t = code, t_min
    a.tune = best
    return best


def tuned_code(a):
    """Cached code for this launch shape; measures it on first sight unless LFVDM_AUTOTUNE=0 or capturing."""
    cache = tune_cache()
    key = tune_key(a)
    code = cache.get(key)
    if code is None:
        if os.environ.get("LFVDM_AUTOTUNE", "1") == "0" or torch.cuda.is_current_stream_capturing():
            return 0
        code = cache[key] = autotune_launch(a)
    return code


def fill_conv_args(**kw):
    """ConvArgs from keyword tensors/ints (same keys as conv_igemm)."""
    a = ConvArgs()
    g = kw.get
    a.src0 = ptr(kw["src0"]); a.src1 = ptr(g("src1")); a.C0 = kw["C0"]; a.C1 = g("C1", 0)
    a.N = kw["N"]; a.Hs = kw["Hs"]; a.Ws = kw["Ws"]; a.up = g("up", 0); a.stride = g("stride", 1)
    a.ksize = g("ksize", 3); a.Ho = kw["Ho"]; a.Wo = kw["Wo"]
    a.coefA = ptr(g("coefA")); a.coefB = ptr(g("coefB")); a.act = g("act", ACT_NONE)
    a.W = ptr(g("W")); a.bias = ptr(g("bias")); a.Cout = kw["Cout"]
    a.s2src0 = ptr(g("s2src0")); a.s2src1 = ptr(g("s2src1")); a.s2C0 = g("s2C0", 0); a.s2C1 = g("s2C1", 0)
    a.W2 = ptr(g("W2")); a.bias2 = ptr(g("bias2"))
    a.res = ptr(g("res")); a.ldr = g("ldr", kw["Cout"]); a.resA = ptr(g("resA")); a.resB = ptr(g("resB"))
    a.out = ptr(kw["out"]); a.ldo = g("ldo", kw["Cout"]); a.out_mode = g("out_mode", OUT_ROWS)
    if g("gn_out") is not None:     # fused GroupNorm(+FiLM)(+activation) of the output (see lfvdm_conv_args)
        film = g("gn_film")
        a.gn_gamma = ptr(kw["gn_gamma"]); a.gn_beta = ptr(kw["gn_beta"]); a.gn_out = ptr(kw["gn_out"])
        a.gn_film = film.data_ptr() if film is not None else None
        a.gn_film_ld = film.stride(0) if film is not None else 0
        a.gn_film_div = g("gn_film_div", 1); a.gn_act = g("gn_act", ACT_NONE); a.gn_skip_raw = int(g("gn_skip_raw", 0))
        a.gn_eps = g("gn_eps", 1e-5)
    return a


# ------------------------------------------------------------------------------------------- deterministic gradients
def deterministic():
    """LFVDM_DETERMINISTIC=1: partial gradient sums go through ordered slabs instead of float atomics (include/lfvdm_hip.h,
    "Deterministic gradients"): bitwise reproducible training steps, at the price of the slab traffic."""
    return os.environ.get("LFVDM_DETERMINISTIC", "0") == "1"


_det_ws = {}


def det_workspace(device):
    """The per-device slab workspace of the deterministic mode (LFVDM_DET_WS_MB, default 1024 MiB; launches are stream-ordered,
    one workspace serves all of them).  Allocated on first use, outside of stream capture."""
    ws = _det_ws.get(device)
    if ws is None:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("deterministic workspace missing under stream capture: run the step eagerly once first")
        ws = _det_ws[device] = torch.empty(int(os.environ.get("LFVDM_DET_WS_MB", "1024")) * 2 ** 18, device=device, dtype=torch.float32)
    return ws


def _wgrad_codes(a):
    """Candidate launch codes of lfvdm_conv_wgrad for this layer shape (1 + tile + 4 * stages + 16 * M slices; tile 1 / 2 / 3 =
    64 x 64 / 128 x 64 / 128 x 128 filters x channels, stages 1 / 2 / 3 = two / three stages / two stages of 64-row chunks)."""
    Cin, M = a.C0 + a.C1, a.N * a.Ho * a.Wo
    if Cin % 64 or a.C0 % 64 or a.Cout < 64 or a.coefA:
        return []
    nchunks = (M + 31) // 32
    codes = []
    for tile, cot in ((1, 2), (2, 4), (3, 4)):
        if cot == 4 and a.Cout < 128:
            continue
        kt = 4 if tile == 3 else 2
        if tile == 3 and (Cin % 128 or a.C0 % 128):
            continue
        tiles = a.ksize * a.ksize * (Cin // (32 * kt)) * ((a.Cout + 32 * cot - 1) // (32 * cot))
        for stages in (2, 1, 3):          # 3 LDS-DMA stages, 2 stages, 2 stages of 64-row chunks
            for target in (256, 320, 384, 448, 512, 640, 768, 1024):
                ms = max(1, min((target + tiles - 1) // tiles, (nchunks // 2 if stages == 3 else nchunks) // 2))
                code = 1 + tile + 4 * stages + 16 * ms
                if code not in codes:
                    codes.append(code)
    # tap-fused kernels (stage field 0): tile 1 = one filter row (three taps) per workgroup, tile 2 = all nine taps
    HW = a.Ho * a.Wo
    if (a.ksize == 3 and a.stride == 1 and a.up == 0 and HW % 32 == 0 and (a.Wo % 32 == 0 or a.Wo in (8, 16))
            and a.Hs == a.Ho and a.Ws == a.Wo and a.out_mode == 0):        # (packed accumulator layout only: no OIHW store)
        for tile, rows in ((1, 3), (2, 1)):
            tiles = rows * (Cin // 64) * ((a.Cout + 63) // 64)
            for target in (128, 192, 256, 320, 384, 512):
                ms = max(1, min((target + tiles - 1) // tiles, (M // 32) // 2))
                code = 1 + tile + 16 * ms
                if code not in codes:
                    codes.append(code)
    return codes


def _capturing():
    """Is the current stream being captured?  (False on a host without a device: the rank-agreement logic also runs in the
    CPU tests of the exchange)"""
    return torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()


def _tuned_wgrad_code(a, out_floats):
    """Cached launch code of this weight-gradient shape; measured on first sight (outside stream capture) on SCRATCH
    outputs - the real launch accumulates.  The weight-gradient kernels are a quarter of a training step and how their
    (filter tile, channel tile, M slice) workgroups divide the 256 CUs differs per layer."""
    cache = tune_cache()
    key = _wgrad_key(a)
    store = _rank_store()
    skey = "lfvdm/wgrad/" + json.dumps([int(x) for x in key])

    def publish(code):
        # rank 0, every resolution once a store exists - also the ones made BEFORE the process group existed (the first
        # call found no store) and the heuristic 0 taken under stream capture: a peer blocked on a key that is never set
        # would sit in the backward pass until the store's timeout
        if store is not None and store[1] == 0 and key not in _wgrad_published:
            store[0].set(skey, str(int(code)))
            _wgrad_published.add(key)

    code = _wgrad_agreed.get(key)
    if code is not None:
        publish(code)
        return code
    if store is not None and store[1] != 0:
        # data-parallel job: ranks must run the SAME kernels (bitwise-equal replicas in deterministic mode, no skew from
        # one rank tuning while its peers sit in a bucket all-reduce): rank 0 decides, the others read its choice from
        # the rendezvous store (once per shape and process; no collective that could mismatch).  Bounded: if rank 0 does
        # not publish within LFVDM_WGRAD_AGREE_S seconds the cached / heuristic code is used and said on stderr
        if _capturing():
            return cache.get(key, 0)
        import datetime
        try:
            store[0].wait([skey], datetime.timedelta(seconds=float(os.environ.get("LFVDM_WGRAD_AGREE_S", "60"))))
            code = int(store[0].get(skey).decode())
        except Exception:
            code = cache.get(key, 0)
            import sys
            print(f"[lfvdm] rank {store[1]}: no weight-gradient launch code from rank 0 for {skey}; using {code}", file=sys.stderr,
                  flush=True)
        _wgrad_agreed[key] = code
        return code
    code = cache.get(key)
    if code is None:
        if _capturing() or os.environ.get("LFVDM_AUTOTUNE", "1") == "0" or deterministic():
            code = 0                    # the heuristic (nothing may be measured inside a capture): kept for the process and
                                        # published below, so that every rank runs it and nobody waits
        else:
            code = _measure_wgrad_code(a, out_floats)
            cache[key] = code
    _wgrad_agreed[key] = code
    publish(code)
    return code


_wgrad_agreed = {}
_wgrad_published = set()
_store_memo = []


def _wgrad_key(a):
    # the bias reduction and a strided dout (ldr != Cout: the head's padded rows) change the kernel's work: own entries
    return (-1,) + tune_key(a) + (int(bool(a.bias)), int(a.ldr != a.Cout and a.ldr != 0))


def _rank_store():
    """(c10d store, rank) of an initialised multi-rank job, else None."""
    if not _store_memo:
        import torch.distributed as dist
        st = None
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            try:
                st = (dist.distributed_c10d._get_default_store(), dist.get_rank())
            except Exception:
                st = None
        if not (dist.is_available() and dist.is_initialized()):
            return None                 # not memoised: the process group may still be created later
        _store_memo.append(st)
    return _store_memo[0]


def _measure_wgrad_code(a, out_floats):
    codes = _wgrad_codes(a)
    if not codes:
        return 0
    L, s = lib(), stream()
    dev = torch.device("cuda", torch.cuda.current_device())
    scratch, sb = torch.zeros(out_floats, device=dev), torch.zeros(max(a.Cout, 1), device=dev)
    real_out, real_bias = a.out, a.bias
    a.out = scratch.data_ptr()
    if real_bias:
        a.bias = sb.data_ptr()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best, best_t = 0, float("inf")
    for code in [0] + codes:
        a.tune = code
        if L.lfvdm_conv_wgrad(C.byref(a), s) != 0:
            continue
        t_min = float("inf")
        for _ in range(int(os.environ.get("LFVDM_TUNE_ROUNDS", 3))):
            ev0.record()
            for _ in range(int(os.environ.get("LFVDM_TUNE_REPS", 4))):
                L.lfvdm_conv_wgrad(C.byref(a), s)
            ev1.record()
            ev1.synchronize()
            t_min = min(t_min, ev0.elapsed_time(ev1))
        if t_min < best_t * 0.98:
            best, best_t = code, t_min
    a.out, a.bias, a.tune = real_out, real_bias, 0
    return best


def conv_wgrad(**kw):
    """Weight/bias gradient launch; kw as conv_igemm plus res=dout rows, out=packed dW, bias=db."""
    a = fill_conv_args(**kw)
    if deterministic():
        # tile shape / M slices: the table entry measured in the default (atomics) mode, if there is one - nothing is timed in
        # this mode (same table on every rank: same kernels, bitwise-equal replicas); key without the slab pointer
        key = _wgrad_key(a)
        ws = det_workspace(kw["out"].device)
        a.splitk_ws, a.splitk_ws_floats = ws.data_ptr(), ws.numel()
        a.tune = tune_cache().get(key, 0)
    else:
        a.tune = _tuned_wgrad_code(a, kw["out"].numel())
    check(lib().lfvdm_conv_wgrad(C.byref(a), stream()), "lfvdm_conv_wgrad")


def pack_conv_weight_t(w, out):
    Cout, Cin, k = w.shape[0], w.shape[1], (w.shape[2] if w.dim() == 4 else 1)
    check(lib().lfvdm_pack_conv_weight_t(ptr(w), ptr(out), Cout, Cin, k, stream()), "lfvdm_pack_conv_weight_t")


def unpack_conv_grad(gp, g, accumulate):
    Cout, Cin, k = g.shape[0], g.shape[1], (g.shape[2] if g.dim() == 4 else 1)
    check(lib().lfvdm_unpack_conv_grad(ptr(gp), ptr(g), Cout, Cin, k, int(bool(accumulate)), stream()), "lfvdm_unpack_conv_grad")


def conv_igemm_struct(a):
    check(lib().lfvdm_conv_igemm(C.byref(a), stream()), "lfvdm_conv_igemm")


def pack_conv_weight(w, out):
    Cout, Cin, k, _ = w.shape
    check(lib().lfvdm_pack_conv_weight(ptr(w), ptr(out), Cout, Cin, k, stream()), "lfvdm_pack_conv_weight")


def conv_in(x, x0, obs, w, bias, out, N, Cc, H, W, Cout):
    check(lib().lfvdm_conv_in(ptr(x), ptr(x0), ptr(obs), ptr(w), ptr(bias), ptr(out), N, Cc, H, W, Cout, stream()),
          "lfvdm_conv_in")


def gn_coef(src0, src1, C0, C1, N, P, gamma, beta, film, film_div, film_ld, eps, coefA, coefB):
    check(lib().lfvdm_gn_coef(ptr(src0), ptr(src1), C0, C1, N, P, ptr(gamma), ptr(beta), ptr(film), film_div, film_ld,
                              eps, ptr(coefA), ptr(coefB), stream()), "lfvdm_gn_coef")


def gn_temporal(x, gamma, beta, eps, y, B, T, P, Cc):
    check(lib().lfvdm_gn_temporal(ptr(x), ptr(gamma), ptr(beta), eps, ptr(y), B, T, P, Cc, stream()), "lfvdm_gn_temporal")


def rowdot(jobs_dev, njobs, total_rows):
    check(lib().lfvdm_rowdot(ptr(jobs_dev, torch.uint8), njobs, total_rows, stream()), "lfvdm_rowdot")


def rpe_nets(jobs_dev, njobs, total_tiles, frame_indices, B, T):
    check(lib().lfvdm_rpe_nets(ptr(jobs_dev, torch.uint8), njobs, total_tiles, ptr(frame_indices, torch.int64), B, T,
                               stream()), "lfvdm_rpe_nets")


def attn_spatial(qkv, o, attn_out, N, P, Cc, heads, lse=None):
    check(lib().lfvdm_attn_spatial(ptr(qkv), ptr(o), ptr(attn_out), ptr(lse), N, P, Cc, heads, stream()),
          "lfvdm_attn_spatial")


def attn_spatial_bwd(qkv, o, d_o, lse, delta_ws, dqkv, N, P, Cc, heads):
    check(lib().lfvdm_attn_spatial_bwd(ptr(qkv), ptr(o), ptr(d_o), ptr(lse), ptr(delta_ws), ptr(dqkv), N, P, Cc, heads,
                                       stream()), "lfvdm_attn_spatial_bwd")


def attn_temporal(qkv, Rq, Rk, Rv, mask, o, attn_out, B, T, P, Cc, heads):
    check(lib().lfvdm_attn_temporal(ptr(qkv), ptr(Rq), ptr(Rk), ptr(Rv), ptr(mask), ptr(o), ptr(attn_out), B, T, P, Cc,
                                    heads, stream()), "lfvdm_attn_temporal")


def attn_temporal_bwd(qkv, d_o, Rq, Rk, Rv, mask, ws_p, ws_ds, dqkv, dRq, dRk, dRv, B, T, P, Cc, heads):
    check(lib().lfvdm_attn_temporal_bwd(ptr(qkv), ptr(d_o), ptr(Rq), ptr(Rk), ptr(Rv), ptr(mask), ptr(ws_p), ptr(ws_ds),
                                        ptr(dqkv), ptr(dRq), ptr(dRk), ptr(dRv), B, T, P, Cc, heads, stream()),
          "lfvdm_attn_temporal_bwd")


def q_sample(x0, noise, t, sa, sb, out):
    B = x0.shape[0]
    check(lib().lfvdm_q_sample(ptr(x0), ptr(noise), ptr(t, torch.int64), ptr(sa), ptr(sb), ptr(out), B,
                               x0.numel() // B, stream()), "lfvdm_q_sample")


def p_sample(x, eps, noise, t, recip, recipm1, c1, c2, logvar, clip, sample, pred=None, mean=None):
    B = x.shape[0]
    check(lib().lfvdm_p_sample(ptr(x), ptr(eps), ptr(noise), ptr(t, torch.int64), ptr(recip), ptr(recipm1), ptr(c1),
                               ptr(c2), ptr(logvar), int(bool(clip)), ptr(sample), ptr(pred), ptr(mean), B,
                               x.numel() // B, stream()), "lfvdm_p_sample")


def p_sample_rng(x, eps, noise_out, t, recip, recipm1, c1, c2, logvar, clip, sample, seed, pred=None, mean=None):
    """p_sample with the noise drawn in the kernel (Philox keyed by the device int64 ``seed``)."""
    B = x.shape[0]
    check(lib().lfvdm_p_sample_rng(ptr(x), ptr(eps), ptr(noise_out), ptr(t, torch.int64), ptr(recip), ptr(recipm1), ptr(c1),
                                   ptr(c2), ptr(logvar), int(bool(clip)), ptr(sample), ptr(pred), ptr(mean), B,
                                   x.numel() // B, ptr(seed, torch.int64), stream()), "lfvdm_p_sample_rng")


def conv_out_psample(act, wp, bias, eps_out, x, noise_in, noise_out, t, recip, recipm1, c1, c2, logvar, clip, sample, seed,
                     pred=None, mean=None):
    """The U-Net's output conv and the x_{t-1} update in one launch (lfvdm_conv_out_psample).  act: channels-last rows
    [B*T*H*W][C]; wp: packed filters [Cout][9][C]; x / sample / eps_out: (B, T, Cout, H, W)."""
    B, T, Cout, H, W = x.shape
    check(lib().lfvdm_conv_out_psample(ptr(act), ptr(wp), ptr(bias), ptr(eps_out), ptr(x), ptr(noise_in), ptr(noise_out),
                                       ptr(t, torch.int64), ptr(recip), ptr(recipm1), ptr(c1), ptr(c2), ptr(logvar),
                                       int(bool(clip)), ptr(sample), ptr(pred), ptr(mean), B, T, H, W, act.shape[-1], Cout,
                                       ptr(seed, torch.int64) if seed is not None else None, stream()),
          "lfvdm_conv_out_psample")


def prepare_batch(pool, table, batch, frame_indices, obs_mask, latent_mask):
    """pool (B, Tp, ...) fp32, table (B, F, 4) int32 -> batch (B, F, ...), frame_indices (B, F) int64, masks (B, F, 1, 1, 1)."""
    B, Tp = pool.shape[:2]
    F = table.shape[1]
    check(lib().lfvdm_prepare_batch(ptr(pool), ptr(table, torch.int32), ptr(batch), ptr(frame_indices, torch.int64), ptr(obs_mask),
                                    ptr(latent_mask), B, F, Tp, pool[0, 0].numel(), stream()), "lfvdm_prepare_batch")


def masked_mse(a, b, mask, out, B, T, frame_inner):
    check(lib().lfvdm_masked_mse(ptr(a), ptr(b), ptr(mask), ptr(out), B, T, frame_inner, stream()), "lfvdm_masked_mse")


class StreamFlags:
    """Counters in device memory that order another stream behind points INSIDE a replayed graph (lfvdm_flag_add /
    lfvdm_flag_wait; one 128-byte line per counter)."""

    def __init__(self, n, device):
        self.buf = torch.zeros(n + 1, 32, device=device, dtype=torch.int32)     # row n: the timed-out word
        self.n = n

    def add(self, k):
        """Counter k += 1 behind everything enqueued so far on the current stream (capturable)."""
        check(lib().lfvdm_flag_add(self.buf[k].data_ptr(), stream()), "lfvdm_flag_add")

    def wait(self, k, target, torch_stream, timeout_s=20.0, also_f32=None):
        """``torch_stream`` proceeds once counter k has reached ``target``.  also_f32: device address of a float that a
        timeout raises to 1.0 as well (the skip word that rides in the last gradient bucket)."""
        wrapped = ((int(target) + 2 ** 31) % 2 ** 32) - 2 ** 31        # the device counter is a wrapping int32: so is the target
        check(lib().lfvdm_flag_wait2(self.buf[k].data_ptr(), wrapped, float(timeout_s), self.buf[self.n].data_ptr(), also_f32,
                                     torch_stream.cuda_stream), "lfvdm_flag_wait2")

    def timed_out_ptr(self):
        """Device address of the timed-out word (lfvdm_adamw_args.skip_flag)."""
        return self.buf[self.n].data_ptr()

    def timed_out(self):
        """Host check (synchronises): did any wait give up?"""
        return bool(self.buf[self.n, 0].item())


def jobs_to_device(jobs, device):
    """Pack a list of ctypes Structures into one uint8 device tensor (job table)."""
    if not jobs:
        return None
    arr = (type(jobs[0]) * len(jobs))(*jobs)
    raw = bytes(memoryview(arr))
    return torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(device)
