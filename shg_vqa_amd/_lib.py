"""ctypes binding of libshgvqa.so (include/shg_vqa.h).  No torch types cross this boundary:
tensors are handed over as raw device pointers + sizes, the stream as a void*.

The product never falls back to anything else: if the shared library is missing or a call
fails, this module raises."""
import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_int64, c_uint64, c_void_p

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libshgvqa.so")

F32, BF16 = 0, 1
ACT_NONE, ACT_GELU, ACT_RELU = 0, 1, 2
MASK_NONE, MASK_KEY, MASK_FULL = 0, 1, 2

P, I, L, F, U = c_void_p, c_int, c_int64, c_float, c_uint64

_SIGNATURES = {
    "shg_version": ([], c_int),
    "shg_last_error_string": ([], c_char_p),
    "shg_set_tuning": ([c_char_p, L], c_int),
    "shg_get_tuning": ([c_char_p], L),
    "shg_tuning_name": ([I], c_char_p),
    "shg_gemm_streamk_launches": ([], L),
    "shg_streamk_plan": ([I, I, I, I, P], c_int),
    "shg_streamk_plan_weighted": ([I, P, I, I, P], c_int),
    "shg_hungarian_per_frame": ([P, I, I, I, I, P, P, L, P, P, P, P], c_int),
    "shg_lsap_batched": ([P, I, I, I, P, P, P, P], c_int),
    "shg_weighted_ce_fwd": ([P, I, L, I, P, P, L, P, P, P], c_int),
    "shg_weighted_ce_bwd": ([P, I, L, I, P, P, P, P, P, P, L, P], c_int),
    "shg_bce_logits_fwd_bwd": ([P, I, L, I, P, P, P, P, L, P], c_int),
    "shg_loss_combine_fwd": ([P, P, P, F, P, P, P], c_int),
    "shg_loss_combine_bwd": ([P, P, P, F, P, P, P, P], c_int),
    "shg_bias_act_fwd": ([P, P, P, I, L, I, I, F, P, U, P], c_int),
    "shg_bias_act_bwd": ([P, P, P, P, P, I, I, L, I, I, F, P, U, P], c_int),
    "shg_bias_act_bwd_view": ([P, P, P, P, P, I, I, L, I, I, F, P, U, L, L, L, P, P, P], c_int),
    "shg_bias_act_bwd_rows": ([P, P, P, P, P, I, I, L, I, I, F, P, U, L, L, L, P, P, P, P], c_int),
    "shg_bias_act_drop_res_ln_fwd": ([P, P, P, P, P, P, P, P, P, I, L, I, I, F, F, P, U, P], c_int),
    "shg_bias_act_drop_res_ln_bwd": ([P, P, P, P, P, P, P, P, P, P, P, P, I, I, L, I, I, F, P, U, P], c_int),
    "shg_colsum_partial": ([P, I, L, I, L, P, I, P], c_int),
    "shg_colsum_accumulate": ([P, I, L, I, L, P, P], c_int),
    "shg_colsum_finish": ([P, I, I, P, I, P], c_int),
    "shg_colsum_finish_multi": ([P, P, I, I, I, P], c_int),
    "shg_colsum_partials": ([L], c_int),
    "shg_attention_keep_mask_bytes": ([I, I, I, I], c_int64),
    "shg_attention_fwd": ([P, P, P, P, P, I, I, I, I, I, L, L, L, L, L, L, I, P, F, F, P, U, P, P], c_int),
    "shg_attention_bwd": ([P, P, P, P, P, P, P, P, P, P, I, I, I, I, I, L, L, L, L, L, L, L, L, L, L, L, L,
                           I, P, F, F, P, U, P, P, P, P, P], c_int),
    "shg_gemm": ([P, P, P, P, I, I, L, L, L, L, L, L, I, I, I, P], c_int),
    "shg_gemm_kseg": ([P, P, P, I, L, L, L, I, L, L, L, L, L, I, P], c_int),
    "shg_gemm_dact": ([P, P, P, P, P, I, L, L, L, L, L, L, I, F, P, U, P], c_int),
    "shg_gemm_act": ([P, P, P, P, I, I, L, L, L, L, L, L, I, I, I, P, F, P, U, P], c_int),
    "shg_conv3d_k533_workspace_bytes": ([I, I, I, I], c_int64),
    "shg_conv3d_k533_prepare": ([P, I, I, I, I, P], c_int),
    "shg_conv3d_k533_fwd": ([P, P, P, P, I, I, I, I, I, I, I, I, I, P, P, P, P], c_int),
    "shg_conv3d_k533_fwd_rows": ([P, P, P, P, I, I, I, I, I, I, I, I, I, P, P, P, I, P, P, P], c_int),
    "shg_streamk_workspace_bytes": ([], c_int64),
    "shg_streamk_workspace_init": ([P, P], c_int),
    "shg_conv3d_k533_wgrad": ([P, P, P, I, I, I, I, I, I, I, I, P, P], c_int),
    "shg_conv3d_k533_wgrad_slice": ([P, P, P, I, I, I, I, I, I, I, I, I, I, P, P], c_int),
    "shg_conv3d_k533_wgrad_sumsq": ([P, P, P, I, I, I, I, I, I, I, I, I, P, P, P], c_int),
    "shg_conv3d_k533_wgrad_ex": ([P, P, P, I, I, I, I, I, I, I, I, I, I, P, I, P, P], c_int),
    "shg_conv3d_k533_dgrad_rows": ([P, P, P, I, I, I, I, I, I, I, P, I, P, P, P], c_int),
    "shg_conv3d_k533_workspace_bytes_ex": ([I, I, I, I, I], c_int64),
    "shg_conv3d_k533_prepare_ex": ([P, I, I, I, I, I, P], c_int),
    "shg_conv3d_k533_dgrad": ([P, P, P, I, I, I, I, I, I, I, P, P], c_int),
    "shg_ncdhw_to_padded_cl": ([P, P, I, I, I, I, I, I, P], c_int),
    "shg_sumsq": ([P, L, P, I, P, P], c_int),
    "shg_sumsq_partial": ([P, L, P, I, P], c_int),
    "shg_sumsq_final": ([P, I, P, P, P], c_int),
    "shg_bertadam_arena": ([P, P, P, P, P, L, P, F, F, F, L, F, F, F, F, P, I, P], c_int),
    "shg_add_i64": ([P, L, P], c_int),
    "shg_cast_f32": ([P, P, I, L, P], c_int),
    "shg_add": ([P, P, P, I, L, P], c_int),
    "shg_tokens_assemble": ([P, P, P, P, I, I, I, I, P], c_int),
    "shg_add2_accumulate": ([P, P, P, I, I, L, P], c_int),
    "shg_bias_act_drop_res_ln_fwd_pos": ([P, P, P, P, P, P, P, P, P, P, P, I, L, I, I, F, F, P, U, P], c_int),
    "shg_exec_create": ([I], c_void_p),
    "shg_exec_destroy": ([P], None),
    "shg_exec_pending_tiles": ([P], c_int64),
    "shg_exec_flush_wgrads": ([P, I, P], c_int),
    "shg_wgrad_group": ([P, I, I, P], c_int),
    "shg_abi_sizeof": ([I], c_int),
    "shg_attn_sublayer_saved_bytes": ([I, I, I, I, I, I], c_int64),
    "shg_attn_sublayer_scratch_bytes": ([I, I, I, I, I, I], c_int64),
    "shg_attn_sublayer_fwd": ([P, P, I, I, I, P, P, P, P, P, P, P, U], c_int),
    "shg_attn_sublayer_bwd": ([P, P, I, I, I, P, P, P, P, P, P, P, P, I, P, U], c_int),
    "shg_ffn_sublayer_saved_bytes": ([I, L, I, I], c_int64),
    "shg_ffn_sublayer_scratch_bytes": ([I, L, I, I], c_int64),
    "shg_ffn_sublayer_fwd": ([P, P, L, I, I, P, P, P, P, P, U], c_int),
    "shg_ffn_sublayer_bwd": ([P, P, L, I, I, P, P, P, P, P, U], c_int),
    "shg_decoder_saved_bytes": ([I, I, I, I, I, I, I], c_int64),
    "shg_decoder_scratch_bytes": ([I, I, I, I, I, I, I], c_int64),
    "shg_decoder_fwd": ([P, I, P, I, I, I, I, P, P, P, P, P, U], c_int),
    "shg_decoder_bwd": ([P, I, P, I, I, I, I, P, P, P, P, P, P, P, P, P, U], c_int),
}


# ---- structs of the sub-layer executor (include/shg_vqa.h); sizes are checked against the library at load time
class LinearT(ctypes.Structure):
    _fields_ = [("w", c_void_p), ("bias", c_void_p), ("gw", c_void_p), ("gb", c_void_p)]


class NormT(ctypes.Structure):
    _fields_ = [("gamma", c_void_p), ("beta", c_void_p), ("g_gamma", c_void_p), ("g_beta", c_void_p), ("eps", c_float),
                ("pad_", c_float)]


class AttnSublayerT(ctypes.Structure):
    _fields_ = [("mode", ctypes.c_int32), ("heads", ctypes.c_int32), ("mask_kind", ctypes.c_int32), ("pad_", ctypes.c_int32),
                ("scale", c_float), ("p_attn", c_float), ("p_out", c_float), ("pad2_", c_float), ("mask", c_void_p),
                ("a", LinearT), ("b", LinearT), ("o", LinearT), ("ln", NormT)]


class FfnSublayerT(ctypes.Structure):
    _fields_ = [("act", ctypes.c_int32), ("pad_", ctypes.c_int32), ("p_inner", c_float), ("p_out", c_float), ("l1", LinearT),
                ("l2", LinearT), ("ln", NormT)]


class DecoderLayerT(ctypes.Structure):
    _fields_ = [("self_attn", AttnSublayerT), ("cross_attn", AttnSublayerT), ("ffn", FfnSublayerT)]


class RunT(ctypes.Structure):
    _fields_ = [("dtype", ctypes.c_int32), ("training", ctypes.c_int32), ("stream", c_void_p), ("wgrad_stream", c_void_p),
                ("exec", c_void_p), ("seed_state", c_void_p), ("defer_wgrad", ctypes.c_int32), ("kv_ahead", ctypes.c_int32)]


class WgradProblemT(ctypes.Structure):
    _fields_ = [("dy", c_void_p), ("x", c_void_p), ("gw", c_void_p), ("rows", c_int64), ("n_out", c_int64), ("n_in", c_int64),
                ("ldy", c_int64), ("ldx", c_int64)]


_ABI_STRUCTS = [RunT, LinearT, NormT, AttnSublayerT, FfnSublayerT, DecoderLayerT]

_lib = None


class ShgError(RuntimeError):
    pass


def lib():
    """Loads the shared library (once).  Raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ShgError("libshgvqa.so is missing: run `python -m shg_vqa_amd.build` (or __graft_entry__.build())")
        # ONE HIP runtime per process: PyTorch bundles its own libamdhip64 (same SONAME as /opt/rocm's).  When it is
        # already loaded the dynamic loader binds our NEEDED entry to it; loaded the other way round the process would
        # end up with two runtimes (torch's allocations are then unknown to the one our kernels launch through:
        # "no ROCm-capable device").  So torch - and its runtime - always come first.
        import torch
        bundled = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")
        if os.path.exists(bundled):
            ctypes.CDLL(bundled, mode=ctypes.RTLD_GLOBAL)
        handle = ctypes.CDLL(LIB_PATH)
        for name, (args, ret) in _SIGNATURES.items():
            fn = getattr(handle, name)
            fn.argtypes = args
            fn.restype = ret
        for i, st in enumerate(_ABI_STRUCTS):
            if handle.shg_abi_sizeof(i) != ctypes.sizeof(st):
                raise ShgError("struct layout mismatch with libshgvqa.so for %s: rebuild (python -m shg_vqa_amd.build)" % st.__name__)
        _lib = handle
        _apply_env_tuning(handle)
    return _lib


def tuning_names():
    h, out, i = lib(), [], 0
    while True:
        nm = h.shg_tuning_name(i)
        if nm is None:
            return out
        out.append(nm.decode())
        i += 1


def _apply_env_tuning(handle):
    """The library itself reads no environment: SHG_<NAME> variables (e.g. SHG_GEMM8_MIN_TILES=130) are mapped onto its tuning
    table here, once, when the library is loaded."""
    i = 0
    while True:
        nm = handle.shg_tuning_name(i)
        if nm is None:
            break
        env = os.environ.get("SHG_" + nm.decode().upper())
        if env is not None:
            if handle.shg_set_tuning(nm, int(env)) != 0:
                raise ShgError("cannot set tuning switch %s" % nm.decode())
        i += 1


def set_tuning(name, value):
    """Sets one tuning switch of the library (include/shg_vqa.h shg_set_tuning); returns the previous value."""
    h = lib()
    old = h.shg_get_tuning(name.encode())
    if h.shg_set_tuning(name.encode(), int(value)) != 0:
        raise ShgError("unknown tuning switch %r (known: %s)" % (name, ", ".join(tuning_names())))
    return old


def get_tuning(name):
    v = lib().shg_get_tuning(name.encode())
    if v == -(1 << 63):
        raise ShgError("unknown tuning switch %r" % name)
    return v


def exported_names():
    return sorted(_SIGNATURES)


try:                                     # host-side trampoline (csrc_host/_fastcall.c, built by build.py); optional
    from . import _fastcall as _fast
except ImportError:                      # pragma: no cover
    _fast = None
_FAST = {}                               # name -> (function address, signature bytes)
_SIG_CHAR = {c_void_p: "p", c_int: "i", c_int64: "l", c_float: "f", c_uint64: "u"}


def _fast_entry(name):
    h = lib()
    args, ret = _SIGNATURES[name]
    if ret is not c_int or len(args) > 30 or sum(a is c_float for a in args) > 8:
        ent = None
    else:
        ent = (ctypes.cast(getattr(h, name), c_void_p).value, "".join(_SIG_CHAR[a] for a in args).encode())
    _FAST[name] = ent
    return ent


def call(name, *args):
    """Calls an int-returning entry point and raises ShgError on a non-zero status."""
    if _fast is not None:
        ent = _FAST.get(name) or _fast_entry(name)
        rc = _fast.call(ent[0], ent[1], *args) if ent is not None else getattr(lib(), name)(*args)
    else:
        rc = getattr(lib(), name)(*args)
    if rc != 0:
        msg = lib().shg_last_error_string()
        raise ShgError("%s failed (rc=%d): %s" % (name, rc, msg.decode() if msg else ""))
    return rc
