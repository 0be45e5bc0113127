"""ctypes binding of libinternnav_amd.so (C-ABI declared in include/internnav_amd.h).

The library is the product path: there is no Python/PyTorch fallback. `lib()` raises if the shared object has
not been built (`python -m internnav_amd.build`), and every op raises RuntimeError with ina_last_error() on failure.
"""
from __future__ import annotations

import ctypes as C
from pathlib import Path

_HERE = Path(__file__).resolve().parent
LIB_PATH = _HERE / "libinternnav_amd.so"

c_void_p, c_int32, c_int64, c_float = C.c_void_p, C.c_int32, C.c_int64, C.c_float


class GemmArgs(C.Structure):
    _fields_ = [
        ("A", c_void_p), ("W", c_void_p), ("C", c_void_p),
        ("bias", c_void_p), ("colscale", c_void_p), ("rowscale", c_void_p), ("R", c_void_p),
        ("M", c_int32), ("N", c_int32), ("K", c_int32),
        ("lda", c_int32), ("ldw", c_int32), ("ldc", c_int32), ("ldr", c_int32),
        ("act", c_int32), ("out_dtype", c_int32), ("res_dtype", c_int32), ("glu", c_int32),
        ("rowscale_div", c_int32), ("batch", c_int32),
        ("strideA", c_int64), ("strideW", c_int64), ("strideC", c_int64), ("strideR", c_int64),
        ("force_cfg", c_int32), ("group_m", c_int32),
        ("norm_gamma", c_void_p), ("norm_eps", c_float), ("a_dtype", c_int32),
        ("seg_stats", c_void_p), ("seg_eps", c_float), ("_pad_seg", c_int32),
        ("Wp", c_void_p),
    ]


class AttnArgs(C.Structure):
    _fields_ = [
        ("Q", c_void_p), ("K", c_void_p), ("V", c_void_p), ("O", c_void_p),
        ("q_bs", c_int64), ("q_rs", c_int64), ("q_hs", c_int64),
        ("k_bs", c_int64), ("k_rs", c_int64), ("k_hs", c_int64),
        ("v_bs", c_int64), ("v_rs", c_int64), ("v_hs", c_int64),
        ("o_bs", c_int64), ("o_rs", c_int64), ("o_hs", c_int64),
        ("B", c_int32), ("H", c_int32), ("Hkv", c_int32),
        ("Lq", c_int32), ("Lk", c_int32), ("D", c_int32),
        ("causal", c_int32), ("kv_start", c_int32), ("kv_bdiv", c_int32),
        ("scale", c_float),
        ("cu_q", c_void_p), ("cu_k", c_void_p), ("head_gate", c_void_p), ("k_len", c_void_p),
        ("accumulate", c_int32), ("drop_seed", C.c_uint32), ("drop_thresh", C.c_uint32), ("drop_scale", c_float),
        ("kernel", c_int32), ("_pad0", c_int32), ("drop_salt", c_void_p),
        ("rope_cos", c_void_p), ("rope_sin", c_void_p), ("k_new", c_void_p), ("v_new", c_void_p),
        ("kn_bs", c_int64), ("kn_rs", c_int64), ("kn_hs", c_int64),
    ]


class RowMap(C.Structure):
    _fields_ = [("seg_len", c_int32), ("seg_stride", c_int32), ("off", c_int32), ("_pad", c_int32)]


class NormArgs(C.Structure):
    _fields_ = [
        ("X", c_void_p), ("Y", c_void_p), ("Y32", c_void_p),
        ("gamma", c_void_p), ("beta", c_void_p), ("mod_scale", c_void_p), ("gate", c_void_p), ("G", c_void_p), ("P", c_void_p),
        ("in_map", RowMap), ("out_map", RowMap),
        ("rows", c_int32), ("C", c_int32),
        ("ldx", c_int32), ("ldy", c_int32), ("ldy32", c_int32), ("ldg", c_int32),
        ("x_dtype", c_int32), ("g_dtype", c_int32),
        ("mod_div", c_int32), ("mod_ld", c_int32),
        ("p_mod", c_int32), ("rms", c_int32),
        ("eps", c_float), ("ldy2", c_int32),
        ("Y2", c_void_p), ("gamma2", c_void_p), ("mod_scale2", c_void_p),
    ]


class PatchifyArgs(C.Structure):
    _fields_ = [
        ("img", c_void_p), ("out", c_void_p),
        ("mean", c_float * 3), ("inv_std", c_float * 3),
        ("n", c_int32), ("H", c_int32), ("W", c_int32), ("C", c_int32),
        ("ps", c_int32), ("ldo", c_int32), ("in_dtype", c_int32), ("_pad", c_int32),
    ]


class Embed3Args(C.Structure):
    _fields_ = [
        ("X", c_void_p), ("W", c_void_p), ("b", c_void_p), ("P", c_void_p), ("Y", c_void_p),
        ("out_map", RowMap),
        ("rows", c_int32), ("C", c_int32), ("ldy", c_int32), ("p_mod", c_int32),
        ("out_dtype", c_int32), ("x_div", c_int32),
    ]


class Head3Args(C.Structure):
    _fields_ = [
        ("X", c_void_p), ("gamma", c_void_p), ("beta", c_void_p), ("mod_scale", c_void_p), ("W", c_void_p), ("b", c_void_p),
        ("sample", c_void_p), ("noise", c_void_p), ("eps_out", c_void_p),
        ("coef", c_float * 5), ("clip", c_float), ("eps", c_float),
        ("rows", c_int32), ("C", c_int32), ("ldx", c_int32), ("x_dtype", c_int32),
        ("mode", c_int32), ("mod_div", c_int32), ("mod_ld", c_int32),
    ]


class SeqpoolArgs(C.Structure):
    _fields_ = [
        ("X", c_void_p), ("gamma", c_void_p), ("beta", c_void_p), ("w", c_void_p), ("b", c_void_p), ("out", c_void_p),
        ("eps", c_float),
        ("nseq", c_int32), ("T", c_int32), ("C", c_int32), ("ldx", c_int32), ("x_dtype", c_int32), ("_pad", c_int32),
    ]


class PoolActArgs(C.Structure):
    _fields_ = [
        ("X", c_void_p), ("P", c_void_p), ("Y", c_void_p),
        ("nseq", c_int32), ("T", c_int32), ("C", c_int32), ("ldx", c_int32), ("ldy", c_int32), ("p_mod", c_int32),
        ("x_dtype", c_int32), ("out_dtype", c_int32), ("act", c_int32), ("_pad", c_int32),
    ]


class GatherArgs(C.Structure):
    _fields_ = [
        ("X", c_void_p), ("Y", c_void_p), ("src", c_void_p), ("dst", c_void_p),
        ("ldx_bytes", c_int64), ("ldy_bytes", c_int64),
        ("rows", c_int32), ("row_bytes", c_int32),
    ]


class RopeArgs(C.Structure):
    _fields_ = [
        ("X", c_void_p), ("cos", c_void_p), ("sin", c_void_p), ("tab", c_void_p),
        ("map", RowMap),
        ("rows", c_int32), ("heads", c_int32), ("D", c_int32), ("ldx", c_int32), ("col0", c_int32), ("_pad", c_int32),
        ("KV", c_void_p), ("kv_dst", c_void_p), ("kv_head0", c_int32), ("v_heads", c_int32), ("ldkv", c_int32), ("_pad2", c_int32),
    ]


class MropeTableArgs(C.Structure):
    _fields_ = [
        ("pos", c_void_p), ("inv_freq", c_void_p), ("axis_of", c_void_p), ("cos", c_void_p), ("sin", c_void_p),
        ("n", c_int32), ("D", c_int32),
    ]


class ArgmaxArgs(C.Structure):
    _fields_ = [("X", c_void_p), ("out", c_void_p), ("rows", c_int32), ("n", c_int32), ("ldx", c_int32), ("_pad", c_int32)]


class DitAttnArgs(C.Structure):
    _fields_ = [
        ("X", c_void_p), ("O", c_void_p), ("g_q1", c_void_p), ("b_q1", c_void_p), ("g_k1", c_void_p), ("b_k1", c_void_p),
        ("g_q2", c_void_p), ("b_q2", c_void_p), ("K2", c_void_p), ("V2T", c_void_p), ("V2T_src", c_void_p), ("head_gate", c_void_p),
        ("k2_bs", c_int64), ("k2_rs", c_int64), ("v2_bs", c_int64), ("v2_rs", c_int64),
        ("nseq", c_int32), ("T", c_int32), ("heads", c_int32), ("seq_per_env", c_int32), ("Lz", c_int32), ("ldx", c_int32), ("ldo", c_int32),
        ("scale", c_float), ("eps", c_float), ("stats_ld", c_int32), ("stats", c_void_p),
    ]


class DitRowchainArgs(C.Structure):
    _fields_ = [
        ("A", c_void_p), ("W1", c_void_p), ("gamma1", c_void_p), ("gate", c_void_p), ("X", c_void_p), ("gamma2", c_void_p),
        ("mod_scale2", c_void_p), ("H", c_void_p), ("W2", c_void_p), ("C2", c_void_p),
        ("M", c_int32), ("K1", c_int32), ("N2", c_int32),
        ("lda", c_int32), ("ldw1", c_int32), ("ldx", c_int32), ("ldh", c_int32), ("ldw2", c_int32), ("ldc2", c_int32),
        ("glu2", c_int32), ("mod_div", c_int32), ("mod_ld", c_int32), ("eps", c_float), ("_reserved", c_int32),
        ("seg_stats", c_void_p), ("seg_eps", c_float), ("_pad", c_int32),
    ]


class GnMishArgs(C.Structure):
    _fields_ = [("X", c_void_p), ("Y", c_void_p), ("R", c_void_p), ("gamma", c_void_p), ("beta", c_void_p), ("film_env", c_void_p),
                ("film_step", c_void_p)] + [(n, c_int32) for n in ("seqs", "T", "C", "groups", "pad", "in_seq_stride", "ldx", "ldy", "ldr",
                                                                    "seq_per_env", "film_ld", "film_off")] + [("eps", c_float), ("x_f32", c_int32)]


class PadRowsArgs(C.Structure):
    _fields_ = [("X", c_void_p), ("bias", c_void_p)] + [(n, c_int32) for n in ("seqs", "T", "pad", "C", "ldx", "_pad")]


class DdimStepArgs(C.Structure):
    _fields_ = [("eps", c_void_p), ("sample", c_void_p), ("Xin", c_void_p)] + [(n, c_int32) for n in ("seqs", "T", "D", "pad", "lde", "ldx")] + \
               [(n, c_float) for n in ("inv_sqrt_a", "sqrt_b", "sqrt_ap", "sqrt_bp", "clip")] + [("use_clipped_model_output", c_int32)]


class ResizeU8Args(C.Structure):
    _fields_ = [("in_", c_void_p), ("out", c_void_p), ("bounds", c_void_p), ("coefs", c_void_p),
                ("outer", c_int32), ("n_in", c_int32), ("n_out", c_int32), ("inner", c_int32), ("ksize", c_int32), ("_pad", c_int32)]


class QwenPatchifyArgs(C.Structure):
    _fields_ = [("img", c_void_p), ("out", c_void_p), ("lut", c_void_p),
                ("n", c_int32), ("H", c_int32), ("W", c_int32), ("ps", c_int32), ("merge", c_int32), ("tdup", c_int32), ("ldo", c_int32), ("_pad", c_int32)]


class U8LutArgs(C.Structure):
    _fields_ = [("in_", c_void_p), ("out", c_void_p), ("lut", c_void_p), ("n", c_int64)]


class ResizeF32Args(C.Structure):
    _fields_ = [("in_", c_void_p), ("out", c_void_p), ("bounds", c_void_p), ("coefs", c_void_p),
                ("outer", c_int32), ("n_in", c_int32), ("n_out", c_int32), ("inner", c_int32), ("ksize", c_int32), ("_pad", c_int32)]


class SelectArgs(C.Structure):
    _fields_ = [
        ("critic", c_void_p), ("sample", c_void_p), ("neg", c_void_p), ("pos", c_void_p),
        ("scale", c_float),
        ("B", c_int32), ("S", c_int32), ("T", c_int32), ("k", c_int32), ("_pad", c_int32),
    ]


# every symbol include/internnav_amd.h declares: name -> (restype, argtypes)

# ---- SFT step (include/internnav_amd.h, second half) ------------------------------------------------------------------
class EwArgs(C.Structure):
    _fields_ = [("A", c_void_p), ("B", c_void_p), ("D", c_void_p), ("S", c_void_p), ("Y", c_void_p), ("Y2", c_void_p), ("tab", c_void_p),
                ("op", c_int32), ("rows", c_int32), ("C", c_int32),
                ("a_dt", c_int32), ("b_dt", c_int32), ("d_dt", c_int32), ("s_dt", c_int32), ("y_dt", c_int32), ("y2_dt", c_int32),
                ("lda", c_int32), ("ldb", c_int32), ("ldd", c_int32), ("lds", c_int32), ("ldy", c_int32), ("ldy2", c_int32),
                ("s_div", c_int32), ("s_f", c_int32), ("tab_mod", c_int32), ("act", c_int32), ("accumulate", c_int32),
                ("drop_seed", C.c_uint32), ("drop_thresh", C.c_uint32), ("drop_scale", c_float), ("drop_salt", c_void_p)]


class ColsumArgs(C.Structure):
    _fields_ = [("X", c_void_p), ("X2", c_void_p), ("out", c_void_p), ("partial", c_void_p), ("partial_elems", c_int64),
                ("rows", c_int32), ("C", c_int32), ("group_rows", c_int32),
                ("x_dt", c_int32), ("x2_dt", c_int32), ("ldx", c_int32), ("ldx2", c_int32), ("x_cs", c_int32), ("x2_cs", c_int32),
                ("ldo", c_int32), ("out_cs", c_int32), ("accumulate", c_int32), ("scale", c_float)]


class NormBwdArgs(C.Structure):
    _fields_ = [("X", c_void_p), ("DY", c_void_p), ("gamma", c_void_p), ("DX", c_void_p), ("XHAT", c_void_p),
                ("rows", c_int32), ("C", c_int32), ("x_dt", c_int32), ("dy_dt", c_int32), ("dx_dt", c_int32),
                ("ldx", c_int32), ("lddy", c_int32), ("lddx", c_int32), ("ldxh", c_int32), ("rms", c_int32), ("accumulate", c_int32),
                ("eps", c_float)]


class TransposeArgs(C.Structure):
    _fields_ = [("X", c_void_p), ("Y", c_void_p), ("rows", c_int32), ("cols", c_int32), ("x_dt", c_int32), ("ldx", c_int32),
                ("ldy", c_int32), ("_pad", c_int32)]


class SparseRowsArgs(C.Structure):
    _fields_ = [("inp", c_void_p), ("out", c_void_p), ("idx", c_void_p), ("coef", c_void_p),
                ("n_out", c_int32), ("C", c_int32), ("taps", c_int32), ("accumulate", c_int32)]


class SmallLinearArgs(C.Structure):
    _fields_ = [("X", c_void_p), ("W", c_void_p), ("bias", c_void_p), ("tab", c_void_p), ("Y", c_void_p),
                ("rows", c_int32), ("N", c_int32), ("K", c_int32), ("x_dt", c_int32), ("y_dt", c_int32), ("ldx", c_int32), ("ldy", c_int32),
                ("w_ns", c_int32), ("w_ks", c_int32), ("tab_mod", c_int32)]


class MseArgs(C.Structure):
    _fields_ = [("pred", c_void_p), ("target", c_void_p), ("mask", c_void_p), ("loss", c_void_p), ("dpred", c_void_p),
                ("nseq", c_int32), ("T", c_int32), ("D", c_int32), ("pred_dt", c_int32), ("dpred_dt", c_int32), ("ldp", c_int32),
                ("lddp", c_int32), ("loss_scale", c_float)]


class AdamwArgs(C.Structure):
    _fields_ = [("p", c_void_p), ("g", c_void_p), ("m", c_void_p), ("v", c_void_p), ("p_bf16", c_void_p), ("sumsq_parts", c_void_p),
                ("norm_out", c_void_p), ("n", c_int64), ("n_parts", c_int32), ("zero_grad", c_int32),
                ("lr", c_float), ("beta1", c_float), ("beta2", c_float), ("eps", c_float), ("wd", c_float), ("bc1", c_float), ("bc2", c_float),
                ("max_norm", c_float), ("grad_scale", c_float), ("_pad", c_int32)]


class GemmDwArgs(C.Structure):
    _fields_ = [("DY", c_void_p), ("X", c_void_p), ("dW", c_void_p), ("db", c_void_p), ("partial", c_void_p), ("partial_elems", c_int64),
                ("rows", c_int32), ("N", c_int32), ("K", c_int32), ("dy_dt", c_int32), ("lddy", c_int32), ("ldx", c_int32), ("ldw", c_int32), ("splits", c_int32)]


class GemmNnArgs(C.Structure):
    _fields_ = [("X", c_void_p), ("W", c_void_p), ("partial", c_void_p), ("partial_elems", c_int64),
                ("M", c_int32), ("N", c_int32), ("K", c_int32), ("ldx", c_int32), ("ldw", c_int32), ("splits", c_int32)]


class AttnBwdArgs(C.Structure):
    _fields_ = [("f", AttnArgs), ("dO", c_void_p), ("dQ", c_void_p), ("dK", c_void_p), ("dV", c_void_p), ("lse", c_void_p), ("delta", c_void_p),
                ("dq_bs", c_int64), ("dq_rs", c_int64), ("dq_hs", c_int64), ("dkv_bs", c_int64), ("dkv_rs", c_int64), ("dkv_hs", c_int64),
                ("kv_row0", c_int32), ("nsplit", c_int32), ("part", c_void_p), ("dq32", c_void_p), ("stage", c_int32), ("_pad", c_int32)]


SYMBOLS = {
    "ina_abi_version": (C.c_int, []),
    "ina_last_error": (C.c_char_p, []),
    "ina_device_check": (C.c_int, [C.c_char_p, C.c_int]),
    "ina_gemm_bf16": (C.c_int, [C.POINTER(GemmArgs), c_void_p]),
    "ina_gemm_select": (C.c_int, [C.POINTER(GemmArgs), C.POINTER(C.c_int)]),
    "ina_gemm_preshuffle": (C.c_int, [c_void_p, c_void_p, c_int32, c_int32, c_int64, c_void_p]),
    "ina_attention_bf16": (C.c_int, [C.POINTER(AttnArgs), c_void_p]),
    "ina_norm_bf16": (C.c_int, [C.POINTER(NormArgs), c_void_p]),
    "ina_patchify": (C.c_int, [C.POINTER(PatchifyArgs), c_void_p]),
    "ina_embed3": (C.c_int, [C.POINTER(Embed3Args), c_void_p]),
    "ina_head3": (C.c_int, [C.POINTER(Head3Args), c_void_p]),
    "ina_seqpool_head": (C.c_int, [C.POINTER(SeqpoolArgs), c_void_p]),
    "ina_select_traj": (C.c_int, [C.POINTER(SelectArgs), c_void_p]),
    "ina_pool_act": (C.c_int, [C.POINTER(PoolActArgs), c_void_p]),
    "ina_gather_rows": (C.c_int, [C.POINTER(GatherArgs), c_void_p]),
    "ina_rope_bf16": (C.c_int, [C.POINTER(RopeArgs), c_void_p]),
    "ina_mrope_table": (C.c_int, [C.POINTER(MropeTableArgs), c_void_p]),
    "ina_argmax_rows": (C.c_int, [C.POINTER(ArgmaxArgs), c_void_p]),
    "ina_dit_attention": (C.c_int, [C.POINTER(DitAttnArgs), c_void_p]),
    "ina_resize_u8": (C.c_int, [C.POINTER(ResizeU8Args), c_void_p]),
    "ina_qwen_patchify_u8": (C.c_int, [C.POINTER(QwenPatchifyArgs), c_void_p]),
    "ina_u8_lut": (C.c_int, [C.POINTER(U8LutArgs), c_void_p]),
    "ina_resize_f32": (C.c_int, [C.POINTER(ResizeF32Args), c_void_p]),
    "ina_dit_rowchain": (C.c_int, [C.POINTER(DitRowchainArgs), c_void_p]),
    "ina_gn_mish": (C.c_int, [C.POINTER(GnMishArgs), c_void_p]),
    "ina_pad_rows": (C.c_int, [C.POINTER(PadRowsArgs), c_void_p]),
    "ina_ddim_step": (C.c_int, [C.POINTER(DdimStepArgs), c_void_p]),
    "ina_ew": (C.c_int, [C.POINTER(EwArgs), c_void_p]),
    "ina_colsum": (C.c_int, [C.POINTER(ColsumArgs), c_void_p]),
    "ina_norm_bwd": (C.c_int, [C.POINTER(NormBwdArgs), c_void_p]),
    "ina_transpose": (C.c_int, [C.POINTER(TransposeArgs), c_void_p]),
    "ina_sparse_rows": (C.c_int, [C.POINTER(SparseRowsArgs), c_void_p]),
    "ina_small_linear": (C.c_int, [C.POINTER(SmallLinearArgs), c_void_p]),
    "ina_mse_masked": (C.c_int, [C.POINTER(MseArgs), c_void_p]),
    "ina_adamw": (C.c_int, [C.POINTER(AdamwArgs), c_void_p]),
    "ina_gemm_nn_bf16": (C.c_int, [C.POINTER(GemmNnArgs), c_void_p]),
    "ina_gemm_dw": (C.c_int, [C.POINTER(GemmDwArgs), c_void_p]),
    "ina_attention_bwd_bf16": (C.c_int, [C.POINTER(AttnBwdArgs), c_void_p]),
    "ina_struct_size": (C.c_int, [C.c_int]),
    "ina_set_workspace_slot": (C.c_int, [C.c_int]),
    "ina_workspace_retired": (C.c_int, []),
    "ina_prof_enable": (C.c_int, [C.c_int]),
    "ina_prof_read": (C.c_int, [C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "ina_prof_read_sub": (C.c_int, [C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
}

_lib = None
# the struct layouts above mirror include/internnav_amd.h at THIS version of the C-ABI (INA_ABI_VERSION there): lib() refuses a shared object
# built from another version - a stale .so would read pointers at the wrong offsets
ABI_VERSION = 8


class EngineError(RuntimeError):
    """the HIP library is missing / stale, or one of its entry points reported an error: never recoverable by retrying a policy call."""


def lib() -> C.CDLL:
    """Load the shared library (once). Fails loudly when it is missing: there is no fallback path."""
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            raise EngineError(
                f"{LIB_PATH} not found: the HIP extension is not built. Run `python -m internnav_amd.build` "
                "(or __graft_entry__.build()). internnav_amd has no CPU/PyTorch fallback."
            )
        h = C.CDLL(str(LIB_PATH))
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(h, name)  # AttributeError if the .so is stale
            fn.restype, fn.argtypes = res, args
        if h.ina_abi_version() != ABI_VERSION:
            raise EngineError(f"{LIB_PATH} was built for C-ABI version {h.ina_abi_version()}, these bindings are version {ABI_VERSION}: "
                              "rebuild it (`python -m internnav_amd.build`)")
        _lib = h
    return _lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = lib().ina_last_error().decode("utf-8", "replace")
        raise EngineError(f"internnav_amd.{what} failed (rc={rc}): {msg}")
