"""Torch-tensor front end of the op-level C-ABI (gemm / attention / norm).

PyTorch is plumbing here: it owns device memory and the stream; all arithmetic runs in the hand-written gfx950
kernels of libinternnav_amd.so. Tensors must live on the current HIP device.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import _lib
from ._lib import AttnArgs, GemmArgs, NormArgs

ACT = {None: 0, "none": 0, "gelu": 1, "gelu_erf": 1, "gelu_tanh": 2, "relu": 3, "silu": 4}
_DT = {torch.bfloat16: 0, torch.float32: 1}


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _f32(t: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
    if t is None:
        return None
    assert t.dtype == torch.float32 and t.is_contiguous(), "f32 parameter vectors must be contiguous float32"
    return t


def _as2d(x: torch.Tensor) -> torch.Tensor:
    assert x.stride(-1) == 1, "last dimension must be contiguous"
    return x.reshape(-1, x.shape[-1]) if x.is_contiguous() else x


def linear(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None, act=None,
           colscale: Optional[torch.Tensor] = None, residual: Optional[torch.Tensor] = None,
           out: Optional[torch.Tensor] = None, out_dtype=torch.bfloat16, glu: bool = False,
           rowscale: Optional[torch.Tensor] = None, rowscale_div: int = 1, force_cfg: int = 0) -> torch.Tensor:
    """out = epilogue(x @ w.T). x: bf16 [..., K] (or a 2-D row-strided view), w: bf16 [N, K]."""
    assert x.dtype == torch.bfloat16 and w.dtype == torch.bfloat16
    lead = x.shape[:-1]
    x2 = _as2d(x)
    assert x2.dim() == 2 and w.dim() == 2 and w.stride(1) == 1
    M, K = x2.shape
    N = w.shape[0]
    assert w.shape[1] == K, f"K mismatch {w.shape} vs {x2.shape}"
    n_out = N // 2 if glu else N
    if out is None:
        out = torch.empty(*lead, n_out, dtype=out_dtype, device=x.device)
    o2 = _as2d(out)
    assert o2.shape == (M, n_out)
    a = GemmArgs()
    a.A, a.W, a.C = x2.data_ptr(), w.data_ptr(), o2.data_ptr()
    a.bias, a.colscale, a.rowscale = _ptr(_f32(bias)), _ptr(_f32(colscale)), _ptr(_f32(rowscale))
    a.M, a.N, a.K = M, N, K
    a.lda, a.ldw, a.ldc = x2.stride(0), w.stride(0), o2.stride(0)
    if residual is not None:
        r2 = _as2d(residual)
        assert r2.shape == (M, n_out)
        a.R, a.ldr, a.res_dtype = r2.data_ptr(), r2.stride(0), _DT[r2.dtype]
    a.act = ACT[act] if not isinstance(act, int) else act
    a.out_dtype = _DT[out.dtype]
    a.glu = 1 if glu else 0
    a.rowscale_div = rowscale_div
    a.batch = 1
    a.force_cfg = force_cfg
    _lib.check(_lib.lib().ina_gemm_bf16(C.byref(a), _stream()), "gemm_bf16")
    return out


def attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, scale: Optional[float] = None, causal: bool = False,
              kv_start: int = 0, kv_bdiv: int = 1, cu_q: Optional[torch.Tensor] = None,
              cu_k: Optional[torch.Tensor] = None, max_q: int = 0, max_k: int = 0,
              head_gate: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None,
              accumulate: bool = False) -> torch.Tensor:
    """softmax(q k^T * scale [+ masks]) v.

    Dense: q [B, Lq, H, D], k/v [Bk, Lk, Hkv, D] (arbitrary strides, last dim contiguous; Bk = B / kv_bdiv).
    Varlen: q [T, H, D], k/v [Tk, Hkv, D] packed with int32 cu_q / cu_k offsets and max_q / max_k.
    """
    assert q.dtype == k.dtype == v.dtype == torch.bfloat16
    a = AttnArgs()
    if cu_q is not None:
        assert q.dim() == 3 and cu_k is not None
        T, H, D = q.shape
        Hkv = k.shape[1]
        B = cu_q.numel() - 1
        if out is None:
            out = torch.empty(T, H, D, dtype=torch.bfloat16, device=q.device)
        a.q_bs, a.q_rs, a.q_hs = 0, q.stride(0), q.stride(1)
        a.k_bs, a.k_rs, a.k_hs = 0, k.stride(0), k.stride(1)
        a.v_bs, a.v_rs, a.v_hs = 0, v.stride(0), v.stride(1)
        a.o_bs, a.o_rs, a.o_hs = 0, out.stride(0), out.stride(1)
        a.Lq, a.Lk = max_q, max_k
        a.cu_q, a.cu_k = cu_q.data_ptr(), cu_k.data_ptr()
        assert cu_q.dtype == torch.int32 and cu_k.dtype == torch.int32
    else:
        B, Lq, H, D = q.shape
        Bk, Lk, Hkv, _ = k.shape
        assert Bk * kv_bdiv == B, f"kv batch {Bk} x {kv_bdiv} != {B}"
        if out is None:
            out = torch.empty(B, Lq, H, D, dtype=torch.bfloat16, device=q.device)
        a.q_bs, a.q_rs, a.q_hs = q.stride(0), q.stride(1), q.stride(2)
        a.k_bs, a.k_rs, a.k_hs = k.stride(0), k.stride(1), k.stride(2)
        a.v_bs, a.v_rs, a.v_hs = v.stride(0), v.stride(1), v.stride(2)
        a.o_bs, a.o_rs, a.o_hs = out.stride(0), out.stride(1), out.stride(2)
        a.Lq, a.Lk = Lq, Lk
    assert q.stride(-1) == 1 and k.stride(-1) == 1 and v.stride(-1) == 1 and out.stride(-1) == 1
    a.Q, a.K, a.V, a.O = q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr()
    a.B, a.H, a.Hkv, a.D = B, H, Hkv, D
    a.causal = 1 if causal else 0
    a.kv_start, a.kv_bdiv = kv_start, kv_bdiv
    a.scale = float(scale) if scale is not None else float(D) ** -0.5
    a.head_gate = _ptr(_f32(head_gate))
    a.accumulate = 1 if accumulate else 0
    _lib.check(_lib.lib().ina_attention_bf16(C.byref(a), _stream()), "attention_bf16")
    return out


def norm(x: torch.Tensor, gamma: Optional[torch.Tensor] = None, beta: Optional[torch.Tensor] = None,
         eps: float = 1e-5, rms: bool = False, residual: Optional[torch.Tensor] = None,
         sum_out: Optional[torch.Tensor] = None, mod_scale: Optional[torch.Tensor] = None,
         gate: Optional[torch.Tensor] = None, gate_base: Optional[torch.Tensor] = None, mod_div: int = 1,
         out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """LayerNorm / RMSNorm over the last dim of bf16 x (+ optional residual-in, modulation, tanh-gated base)."""
    assert x.dtype == torch.bfloat16
    x2 = _as2d(x)
    rows, Cdim = x2.shape
    if out is None:
        out = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
    y2 = _as2d(out)
    a = NormArgs()
    a.X, a.Y = x2.data_ptr(), y2.data_ptr()
    a.rows, a.C, a.ldx, a.ldy = rows, Cdim, x2.stride(0), y2.stride(0)
    if residual is not None:
        r2 = _as2d(residual)
        a.R, a.ldr = r2.data_ptr(), r2.stride(0)
    if sum_out is not None:
        s2 = _as2d(sum_out)
        assert s2.stride(0) == y2.stride(0)
        a.S = s2.data_ptr()
    a.gamma, a.beta = _ptr(_f32(gamma)), _ptr(_f32(beta))
    if mod_scale is not None:
        assert mod_scale.dtype == torch.float32 and mod_scale.stride(-1) == 1
        a.mod_scale, a.mod_ld = mod_scale.data_ptr(), mod_scale.stride(0)
    if gate is not None:
        assert gate.dtype == torch.float32 and gate.stride(-1) == 1
        a.gate, a.mod_ld = gate.data_ptr(), gate.stride(0)
        if mod_scale is not None:
            assert mod_scale.stride(0) == gate.stride(0)
    if gate_base is not None:
        g2 = _as2d(gate_base)
        a.G, a.ldg = g2.data_ptr(), g2.stride(0)
    a.mod_div = mod_div
    a.rms = 1 if rms else 0
    a.eps = eps
    _lib.check(_lib.lib().ina_norm_bf16(C.byref(a), _stream()), "norm_bf16")
    return out
