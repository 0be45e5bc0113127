"""Torch-tensor front end of the op-level C-ABI (gemm / attention / norm).

PyTorch is plumbing here: it owns device memory and the stream; all arithmetic runs in the hand-written gfx950
kernels of libinternnav_amd.so. Tensors must live on the current HIP device.
"""
from __future__ import annotations

import contextlib
import ctypes as C
from typing import Optional

import torch

from . import _lib
from ._lib import AttnArgs, GemmArgs, NormArgs

ACT = {None: 0, "none": 0, "gelu": 1, "gelu_erf": 1, "gelu_tanh": 2, "relu": 3, "silu": 4, "mish": 5}
_DT = {torch.bfloat16: 0, torch.float32: 1}


# Tile selection of the GEMMs that leave it to the library (force_cfg = 0): inside `shared_tail()` they are issued with force_cfg = -1 -
# "this launch runs beside another stream's GEMMs" (include/internnav_amd.h) - so the tile cost model does not charge the under-filled last
# round that the other stream's workgroups fill. Every tile shape accumulates K in the same order, so the choice changes no bit of the
# result; that also makes the process-wide switch harmless if a launch of another host thread picks it up.
_AUTO_CFG = 0


@contextlib.contextmanager
def shared_tail():
    global _AUTO_CFG
    keep, _AUTO_CFG = _AUTO_CFG, -1
    try:
        yield
    finally:
        _AUTO_CFG = keep


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
    if x.dim() == 2:          # the common case: no new view object (every ops call runs on the host's launch-issue path)
        return x
    return x.reshape(-1, x.shape[-1]) if x.is_contiguous() else x


def linear(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None, act=None,
           colscale: Optional[torch.Tensor] = None, residual: Optional[torch.Tensor] = None,
           out: Optional[torch.Tensor] = None, out_dtype=torch.bfloat16, glu: bool = False,
           rowscale: Optional[torch.Tensor] = None, rowscale_div: int = 1, force_cfg: int = 0,
           batched: bool = False, group_m: int = 0, prenorm=None, w_frag: Optional[torch.Tensor] = None, seg_stats=None) -> torch.Tensor:
    """out = epilogue(x @ w.T). x: bf16 [..., K] (or a 2-D row-strided view), w: bf16 [N, K].

    seg_stats=(stats f32 [M, N // 384, 2], eps): the row-panel GEMM (K = 384, >= 16384 rows, plain epilogue) also writes (mean, rstd) of every 384-wide
    segment of its fp32 output rows - the LayerNorm statistics dit_attention(stats=) consumes. Refused where another kernel would run.
    w_frag: the same weight in MFMA fragment order (gemm_preshuffle(w)): the wide no-residual GEMMs that would run tile config 39 then take
    their B fragments from it straight into registers (config 40, bit-equal); ignored by every other tile.

    batched=True: x is [Bt, M, K] and out [Bt, M, N] views with arbitrary batch strides (rows contiguous-strided inside a
    batch); residual is [Bt, M, N] or a [M, N] table broadcast over the batch (e.g. positional embeddings).
    prenorm=(gamma f32 [K], eps): RMSNorm of x fused in front (x f32 or bf16, at most 16 rows): out = epilogue(bf16(rmsnorm(x) * gamma) @ w.T).
    """
    assert (x.dtype == torch.bfloat16 or prenorm is not None) and w.dtype == torch.bfloat16
    assert w.dim() == 2 and w.stride(1) == 1
    N, K = w.shape
    n_out = N // 2 if glu else N
    a = GemmArgs()
    if batched:
        assert x.dim() == 3 and x.stride(2) == 1 and out is not None and out.dim() == 3 and out.stride(2) == 1
        Bt, M, _ = x.shape
        assert out.shape == (Bt, M, n_out)
        a.A, a.C = x.data_ptr(), out.data_ptr()
        a.lda, a.ldc = x.stride(1), out.stride(1)
        a.batch, a.strideA, a.strideC = Bt, x.stride(0), out.stride(0)
        if residual is not None:
            assert residual.stride(-1) == 1 and residual.shape[-2:] == (M, n_out)
            a.R, a.ldr, a.res_dtype = residual.data_ptr(), residual.stride(-2), _DT[residual.dtype]
            a.strideR = residual.stride(0) if residual.dim() == 3 else 0
    else:
        lead = x.shape[:-1]
        x2 = _as2d(x)
        assert x2.dim() == 2
        M = x2.shape[0]
        if out is None:
            out = torch.empty(*lead, n_out, dtype=out_dtype, device=x.device)
        o2 = _as2d(out)
        assert o2.shape == (M, n_out), f"out {tuple(o2.shape)} vs {(M, n_out)}"
        a.A, a.C = x2.data_ptr(), o2.data_ptr()
        a.lda, a.ldc = x2.stride(0), o2.stride(0)
        a.batch = 1
        if residual is not None:
            r2 = _as2d(residual)
            assert r2.shape == (M, n_out)
            a.R, a.ldr, a.res_dtype = r2.data_ptr(), r2.stride(0), _DT[r2.dtype]
    assert x.shape[-1] == K, f"K mismatch {tuple(w.shape)} vs {tuple(x.shape)}"
    a.W, a.ldw = w.data_ptr(), w.stride(0)
    a.bias, a.colscale, a.rowscale = _ptr(_f32(bias)), _ptr(_f32(colscale)), _ptr(_f32(rowscale))
    a.M, a.N, a.K = M, N, K
    a.act = ACT[act] if not isinstance(act, int) else act
    a.out_dtype = _DT[out.dtype]
    a.glu = 1 if glu else 0
    a.rowscale_div = rowscale_div
    a.force_cfg = force_cfg if force_cfg else _AUTO_CFG
    a.group_m = group_m
    if seg_stats is not None:
        st, se = seg_stats
        assert not batched and st.dtype == torch.float32 and st.is_contiguous() and st.numel() >= M * (N // 384) * 2 and N % 384 == 0
        a.seg_stats, a.seg_eps = st.data_ptr(), float(se)
    if w_frag is not None:
        assert w_frag.dtype == torch.bfloat16 and w_frag.is_contiguous() and w_frag.numel() == N * K and N % 16 == 0 and K % 32 == 0
        a.Wp = w_frag.data_ptr()
    if prenorm is not None:
        gamma, eps = prenorm
        assert not batched and x.dtype in (torch.bfloat16, torch.float32) and gamma.dtype == torch.float32 and gamma.is_contiguous() and gamma.numel() == K
        a.norm_gamma, a.norm_eps, a.a_dtype = gamma.data_ptr(), float(eps), _DT[x.dtype]
    _lib.check(_lib.lib().ina_gemm_bf16(C.byref(a), _stream()), "gemm_bf16")
    return out


def gemm_preshuffle(w: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """w bf16 [N, K] -> the same elements in MFMA fragment order (flat [N * K]): fragment (n // 16, k // 32) is one contiguous KiB whose lane
    (k % 32 // 8) * 16 + n % 16 holds 8 consecutive k. Done once per weight at load time; consumed by linear(w_frag=)."""
    assert w.dtype == torch.bfloat16 and w.dim() == 2 and w.stride(1) == 1
    N, K = w.shape
    if out is None:
        out = torch.empty(N * K, dtype=torch.bfloat16, device=w.device)
    assert out.dtype == torch.bfloat16 and out.is_contiguous() and out.numel() == N * K
    _lib.check(_lib.lib().ina_gemm_preshuffle(w.data_ptr(), out.data_ptr(), N, K, w.stride(0), _stream()), "gemm_preshuffle")
    return out


def attention_rope_ok(Lq: int, Lk: int, H: int, Hkv: int, D: int) -> bool:
    """shapes whose attention launch can carry the rotary embedding + KV append of its new tokens (the one-launch decode kernel of the d = 128
    heads; csrc/attention.hip ina_launch_attention)"""
    return D == 128 and 256 <= Lk <= 1024 and Lq <= 8 and (H // Hkv) * Lq <= 48


def attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, scale: Optional[float] = None, causal: bool = False,
              kv_start: int = 0, kv_bdiv: int = 1, cu_q: Optional[torch.Tensor] = None,
              cu_k: Optional[torch.Tensor] = None, max_q: int = 0, max_k: int = 0,
              head_gate: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None,
              accumulate: bool = False, k_len: Optional[torch.Tensor] = None, drop_p: float = 0.0, drop_seed: int = 0,
              kernel: int = 0, drop_salt: Optional[torch.Tensor] = None, rope=None) -> torch.Tensor:
    """softmax(q k^T * scale [+ masks]) v.   kernel: 0 = automatic, 1 = the 16-rows-per-wave kernel, 2 = the 32-rows-per-wave kernel
    (ina_attn_args.kernel; the parity tests compare the two).
    rope=(cos f32 [B * Lq, 128], sin, k_new bf16 [B, Lq, Hkv, 128], v_new) - single-token decoder passes (see `attention_rope_ok`): q, k_new are
    UN-rotated; the launch rotates them, appends rotate(k_new) | v_new to the last Lq rows of k / v (the cache views) and attends.

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
    if k_len is not None:
        assert k_len.dtype == torch.int32 and k_len.is_contiguous() and cu_q is None
        a.k_len = k_len.data_ptr()
    a.accumulate = 1 if accumulate else 0
    if rope is not None:
        cos, sin, k_new, v_new = rope
        assert cu_q is None and cos.dtype == torch.float32 and cos.is_contiguous() and sin.is_contiguous() and cos.shape[-1] == 128 and cos.numel() >= B * Lq * 128
        assert k_new.dtype == torch.bfloat16 and k_new.shape == (B, Lq, Hkv, D) and v_new.shape == k_new.shape and k_new.stride() == v_new.stride() and k_new.stride(-1) == 1
        a.rope_cos, a.rope_sin, a.k_new, a.v_new = cos.data_ptr(), sin.data_ptr(), k_new.data_ptr(), v_new.data_ptr()
        a.kn_bs, a.kn_rs, a.kn_hs = k_new.stride(0), k_new.stride(1), k_new.stride(2)
    a.kernel = kernel
    if drop_p > 0.0:       # training only: attention-probability dropout, mask = counter hash of (seed, element index)
        assert cu_q is None, "dropout: dense layouts only"
        a.drop_seed, a.drop_thresh, a.drop_scale = drop_params(drop_p, drop_seed)
        if drop_salt is not None:          # one int32 device word added to the seed at kernel start (fresh masks per replay of a captured launch sequence)
            assert drop_salt.dtype == torch.int32 and drop_salt.numel() == 1 and drop_salt.is_cuda
            a.drop_salt = drop_salt.data_ptr()
    _lib.check(_lib.lib().ina_attention_bf16(C.byref(a), _stream()), "attention_bf16")
    return out


def drop_params(p: float, seed: int):
    """(seed, threshold, scale) of the counter-based dropout mask: keep iff hash(seed, index) >= p * 2^32, kept values scaled by 1 / (1 - p)."""
    assert 0.0 < p < 1.0
    return seed & 0xFFFFFFFF, max(1, min(0xFFFFFFFF, int(p * 4294967296.0))), 1.0 / (1.0 - p)


def _rowmap(m) -> _lib.RowMap:
    r = _lib.RowMap()
    if m is not None:
        r.seg_len, r.seg_stride, r.off = int(m[0]), int(m[1]), int(m[2])
    return r


def norm(x: torch.Tensor, gamma: Optional[torch.Tensor] = None, beta: Optional[torch.Tensor] = None,
         eps: float = 1e-5, rms: bool = False, mod_scale: Optional[torch.Tensor] = None,
         gate: Optional[torch.Tensor] = None, base: Optional[torch.Tensor] = None, mod_div: int = 1,
         pos: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None,
         out32: Optional[torch.Tensor] = None, rows: Optional[int] = None, in_map=None, out_map=None,
         out2: Optional[torch.Tensor] = None, gamma2: Optional[torch.Tensor] = None, mod_scale2: Optional[torch.Tensor] = None) -> torch.Tensor:
    """LayerNorm / RMSNorm over the last dim of x (bf16 or f32) -> out (bf16) and/or out32 (f32).

    t = norm(x) * gamma + beta; t *= 1 + mod_scale[r // mod_div]; t *= tanh(gate[r // mod_div]); t += base[r]; t += pos[r % len(pos)].
    in_map / out_map = (seg_len, seg_stride, off) gather / scatter logical rows; `rows` = number of logical rows (default: all of x).
    out2 (bf16): chained pre-norm of the produced row, out2 = norm(t) * gamma2 * (1 + mod_scale2[r // mod_div]) (same rms / eps).
    """
    assert x.dtype in _DT
    x2 = _as2d(x)
    Cdim = x2.shape[1]
    if rows is None:
        rows = x2.shape[0]
    if out is None and out32 is None:
        out = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
    a = NormArgs()
    a.X, a.x_dtype, a.ldx = x2.data_ptr(), _DT[x2.dtype], x2.stride(0)
    a.rows, a.C = rows, Cdim
    if out is not None:
        y2 = _as2d(out)
        assert y2.dtype == torch.bfloat16 and y2.shape[1] == Cdim
        a.Y, a.ldy = y2.data_ptr(), y2.stride(0)
    if out32 is not None:
        z2 = _as2d(out32)
        assert z2.dtype == torch.float32 and z2.shape[1] == Cdim
        a.Y32, a.ldy32 = z2.data_ptr(), z2.stride(0)
    a.gamma, a.beta = _ptr(_f32(gamma)), _ptr(_f32(beta))
    if mod_scale is not None:
        assert mod_scale.dtype == torch.float32 and mod_scale.stride(-1) == 1
        a.mod_scale, a.mod_ld = mod_scale.data_ptr(), mod_scale.stride(0)
    if gate is not None:
        assert gate.dtype == torch.float32 and gate.stride(-1) == 1
        a.gate, a.mod_ld = gate.data_ptr(), gate.stride(0)
        if mod_scale is not None:
            assert mod_scale.stride(0) == gate.stride(0)
    if base is not None:
        g2 = _as2d(base)
        a.G, a.ldg, a.g_dtype = g2.data_ptr(), g2.stride(0), _DT[g2.dtype]
    if pos is not None:
        assert pos.dtype == torch.float32 and pos.is_contiguous() and pos.shape[-1] == Cdim
        a.P, a.p_mod = pos.data_ptr(), pos.numel() // Cdim
    if out2 is not None:
        w2 = _as2d(out2)
        assert w2.dtype == torch.bfloat16 and w2.shape[1] == Cdim
        a.Y2, a.ldy2, a.gamma2 = w2.data_ptr(), w2.stride(0), _ptr(_f32(gamma2))
        if mod_scale2 is not None:
            assert mod_scale2.dtype == torch.float32 and mod_scale2.stride(-1) == 1 and (a.mod_ld in (0, mod_scale2.stride(0)))
            a.mod_scale2, a.mod_ld = mod_scale2.data_ptr(), mod_scale2.stride(0)
    a.in_map, a.out_map = _rowmap(in_map), _rowmap(out_map)
    a.mod_div = mod_div
    a.rms = 1 if rms else 0
    a.eps = eps
    _lib.check(_lib.lib().ina_norm_bf16(C.byref(a), _stream()), "norm_bf16")
    return out if out is not None else out32


def patchify(img: torch.Tensor, out: torch.Tensor, mean=(0.0, 0.0, 0.0), std=(1.0, 1.0, 1.0), ps: int = 14) -> torch.Tensor:
    """img [n, H, W, C] (f32|bf16, C = 3 or 1) -> out bf16 [n * (H/ps) * (W/ps), ldo] im2col rows (k = c*ps*ps + y*ps + x)."""
    assert img.is_contiguous() and img.dim() == 4 and out.dtype == torch.bfloat16 and out.stride(1) == 1
    a = _lib.PatchifyArgs()
    a.img, a.out = img.data_ptr(), out.data_ptr()
    for i in range(3):
        a.mean[i], a.inv_std[i] = float(mean[i]), 1.0 / float(std[i])
    a.n, a.H, a.W, a.C = img.shape
    assert out.shape[0] == a.n * (a.H // ps) * (a.W // ps)
    a.ps, a.ldo, a.in_dtype = ps, out.stride(0), _DT[img.dtype]
    _lib.check(_lib.lib().ina_patchify(C.byref(a), _stream()), "patchify")
    return out


def embed3(x: Optional[torch.Tensor], w: Optional[torch.Tensor], bias: Optional[torch.Tensor], out: torch.Tensor,
           pos: Optional[torch.Tensor] = None, rows: Optional[int] = None, out_map=None, x_div: int = 1) -> torch.Tensor:
    """out[out_map(r)] = w @ x[r // x_div] + bias + pos[r % len(pos)]  (nn.Linear(3, C); x None = table fill)."""
    o2 = _as2d(out)
    a = _lib.Embed3Args()
    if x is not None:
        assert x.dtype == torch.float32 and x.is_contiguous() and x.shape[-1] == 3 and w is not None
        a.X = x.data_ptr()
        if rows is None:
            rows = (x.numel() // 3) * x_div
    if w is not None:
        assert w.dtype == torch.float32 and w.is_contiguous() and w.shape == (o2.shape[1], 3)
        a.W = w.data_ptr()
    a.b = _ptr(_f32(bias))
    if pos is not None:
        assert pos.dtype == torch.float32 and pos.is_contiguous()
        a.P, a.p_mod = pos.data_ptr(), pos.numel() // o2.shape[1]
    assert rows is not None
    a.Y, a.ldy, a.out_dtype = o2.data_ptr(), o2.stride(0), _DT[o2.dtype]
    a.rows, a.C, a.x_div = rows, o2.shape[1], x_div
    a.out_map = _rowmap(out_map)
    _lib.check(_lib.lib().ina_embed3(C.byref(a), _stream()), "embed3")
    return out


def head3(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor, gamma=None, beta=None, eps: float = 1e-5, mode: int = 0,
          sample: Optional[torch.Tensor] = None, noise: Optional[torch.Tensor] = None, eps_out: Optional[torch.Tensor] = None,
          coef=(0.0, 0.0, 0.0, 0.0, 0.0), clip: float = 1.0, mod_scale: Optional[torch.Tensor] = None, mod_div: int = 1):
    """final norm + Linear(C,3) + sampler update (mode 0: write eps_out; 1: DDPM step; 2: Euler step) on f32 samples [rows,3]."""
    x2 = _as2d(x)
    a = _lib.Head3Args()
    a.X, a.ldx, a.x_dtype = x2.data_ptr(), x2.stride(0), _DT[x2.dtype]
    a.rows, a.C = x2.shape
    a.gamma, a.beta = _ptr(_f32(gamma)), _ptr(_f32(beta))
    assert w.dtype == torch.float32 and w.is_contiguous() and w.shape == (3, a.C) and b.dtype == torch.float32
    a.W, a.b = w.data_ptr(), b.data_ptr()
    for t in (sample, noise, eps_out):
        assert t is None or (t.dtype == torch.float32 and t.is_contiguous() and t.numel() == a.rows * 3)
    a.sample, a.noise, a.eps_out = _ptr(sample), _ptr(noise), _ptr(eps_out)
    for i in range(5):
        a.coef[i] = float(coef[i])
    a.clip, a.eps, a.mode = clip, eps, mode
    if mod_scale is not None:
        assert mod_scale.dtype == torch.float32 and mod_scale.stride(-1) == 1
        a.mod_scale, a.mod_ld = mod_scale.data_ptr(), mod_scale.stride(0)
    a.mod_div = mod_div
    _lib.check(_lib.lib().ina_head3(C.byref(a), _stream()), "head3")


def seqpool_head(x: torch.Tensor, T: int, gamma, beta, w: torch.Tensor, b: torch.Tensor, out: torch.Tensor, eps: float = 1e-5):
    """out[s] = w . mean_t(LayerNorm(x[s*T + t])) + b."""
    x2 = _as2d(x)
    a = _lib.SeqpoolArgs()
    a.X, a.ldx, a.x_dtype = x2.data_ptr(), x2.stride(0), _DT[x2.dtype]
    a.nseq, a.T, a.C = x2.shape[0] // T, T, x2.shape[1]
    a.gamma, a.beta, a.w, a.b = _ptr(_f32(gamma)), _ptr(_f32(beta)), _f32(w.reshape(-1)).data_ptr(), _f32(b).data_ptr()
    assert out.dtype == torch.float32 and out.numel() == a.nseq
    a.out, a.eps = out.data_ptr(), eps
    _lib.check(_lib.lib().ina_seqpool_head(C.byref(a), _stream()), "seqpool_head")
    return out


def select_traj(critic: torch.Tensor, sample: torch.Tensor, neg: torch.Tensor, pos: torch.Tensor, k: int = 8, scale: float = 0.25):
    """critic [B,S], sample [B,S,T,3] -> neg/pos [B,k,T,3] cumulative trajectories of the k lowest / highest critic samples."""
    B, S, T, _ = sample.shape
    for t in (critic, sample, neg, pos):
        assert t.dtype == torch.float32 and t.is_contiguous()
    a = _lib.SelectArgs()
    a.critic, a.sample, a.neg, a.pos = critic.data_ptr(), sample.data_ptr(), neg.data_ptr(), pos.data_ptr()
    a.scale, a.B, a.S, a.T, a.k = scale, B, S, T, k
    _lib.check(_lib.lib().ina_select_traj(C.byref(a), _stream()), "select_traj")


def pool_act(x: torch.Tensor, out: torch.Tensor, T: int = 1, pos: Optional[torch.Tensor] = None, act=None) -> torch.Tensor:
    """out[s] = act(mean_t x[s*T + t] + pos[s % len(pos)]); x bf16|f32 [nseq*T, C] -> out bf16|f32 [nseq, C]."""
    x2, o2 = _as2d(x), _as2d(out)
    a = _lib.PoolActArgs()
    a.X, a.ldx, a.x_dtype = x2.data_ptr(), x2.stride(0), _DT[x2.dtype]
    a.Y, a.ldy, a.out_dtype = o2.data_ptr(), o2.stride(0), _DT[o2.dtype]
    a.nseq, a.T, a.C = x2.shape[0] // T, T, x2.shape[1]
    assert o2.shape == (a.nseq, a.C)
    if pos is not None:
        assert pos.dtype == torch.float32 and pos.is_contiguous() and pos.shape[-1] == a.C
        a.P, a.p_mod = pos.data_ptr(), pos.numel() // a.C
    a.act = ACT[act]
    _lib.check(_lib.lib().ina_pool_act(C.byref(a), _stream()), "pool_act")
    return out


def gather_rows(x: torch.Tensor, out: torch.Tensor, src: Optional[torch.Tensor] = None, dst: Optional[torch.Tensor] = None,
                rows: Optional[int] = None) -> torch.Tensor:
    """out[dst[r] or r] = x[src[r] or r] (2-D tensors of equal dtype / width, int32 index vectors on the device)."""
    assert x.dim() == 2 and out.dim() == 2 and x.dtype == out.dtype and x.shape[1] == out.shape[1] and x.stride(1) == 1 and out.stride(1) == 1
    a = _lib.GatherArgs()
    es = x.element_size()
    a.X, a.Y = x.data_ptr(), out.data_ptr()
    a.ldx_bytes, a.ldy_bytes, a.row_bytes = x.stride(0) * es, out.stride(0) * es, x.shape[1] * es
    for t in (src, dst):
        assert t is None or (t.dtype == torch.int32 and t.is_contiguous())
    a.src, a.dst = _ptr(src), _ptr(dst)
    a.rows = rows if rows is not None else (src.numel() if src is not None else (dst.numel() if dst is not None else x.shape[0]))
    _lib.check(_lib.lib().ina_gather_rows(C.byref(a), _stream()), "gather_rows")
    return out


def rope(x: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor, heads: int, D: int, col0: int = 0, rows: Optional[int] = None,
         row_map=None, tab: Optional[torch.Tensor] = None, kv_out: Optional[torch.Tensor] = None, kv_dst: Optional[torch.Tensor] = None,
         kv_head0: int = 0, v_heads: int = 0) -> torch.Tensor:
    """in-place rotary embedding on `heads` heads of width D at columns [col0, col0 + heads*D) of bf16 rows x[row_map(r)].
    kv_out (fused KV-cache append of a decoder layer): heads [kv_head0, heads) are key heads - their rotated values go to cache row
    kv_dst[r] of kv_out (not back into x) together with the v_heads value heads that follow them in x: one launch instead of rope +
    gather."""
    assert x.dtype == torch.bfloat16 and x.dim() == 2 and x.stride(1) == 1
    assert cos.dtype == torch.float32 and cos.is_contiguous() and sin.is_contiguous() and cos.shape[-1] == D
    a = _lib.RopeArgs()
    a.X, a.cos, a.sin, a.tab = x.data_ptr(), cos.data_ptr(), sin.data_ptr(), _ptr(tab)
    a.map = _rowmap(row_map)
    a.rows = rows if rows is not None else x.shape[0]
    a.heads, a.D, a.ldx, a.col0 = heads, D, x.stride(0), col0
    if kv_out is not None:
        assert kv_out.dtype == torch.bfloat16 and kv_out.stride(1) == 1 and kv_dst.dtype == torch.int32 and kv_dst.is_contiguous()
        assert kv_out.shape[1] >= (heads - kv_head0 + v_heads) * D
        a.KV, a.kv_dst, a.kv_head0, a.v_heads, a.ldkv = kv_out.data_ptr(), kv_dst.data_ptr(), kv_head0, v_heads, kv_out.stride(0)
    _lib.check(_lib.lib().ina_rope_bf16(C.byref(a), _stream()), "rope_bf16")
    return x


def mrope_table(pos: torch.Tensor, inv_freq: torch.Tensor, axis_of: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor):
    """pos int32 [3, n] -> cos/sin f32 [n, D] of the multimodal rope (section axis per frequency in axis_of int32 [D/2])."""
    assert pos.dtype == torch.int32 and pos.is_contiguous() and pos.shape[0] == 3
    n, D = pos.shape[1], cos.shape[-1]
    assert cos.is_contiguous() and sin.is_contiguous() and cos.numel() >= n * D and inv_freq.numel() == D // 2 and axis_of.dtype == torch.int32
    a = _lib.MropeTableArgs()
    a.pos, a.inv_freq, a.axis_of, a.cos, a.sin = pos.data_ptr(), _f32(inv_freq).data_ptr(), axis_of.data_ptr(), cos.data_ptr(), sin.data_ptr()
    a.n, a.D = n, D
    _lib.check(_lib.lib().ina_mrope_table(C.byref(a), _stream()), "mrope_table")


def argmax_rows(x: torch.Tensor, out: torch.Tensor) -> torch.Tensor:
    """out[r] = argmax of f32 row r (first maximum), out int32 [rows]."""
    assert x.dtype == torch.float32 and x.dim() == 2 and x.stride(1) == 1 and out.dtype == torch.int32 and out.numel() == x.shape[0]
    a = _lib.ArgmaxArgs()
    a.X, a.out, a.rows, a.n, a.ldx = x.data_ptr(), out.data_ptr(), x.shape[0], x.shape[1], x.stride(0)
    _lib.check(_lib.lib().ina_argmax_rows(C.byref(a), _stream()), "argmax_rows")
    return out


def dit_v2t(kv2: torch.Tensor, heads: int, v2t: torch.Tensor) -> torch.Tensor:
    """condition V of a NextDiT block -> the transposed key-permuted image dit_attention consumes.
    kv2 bf16 [envs, Lz, 2, heads, 64] (K | V as produced by the fused kv projection); v2t bf16 [envs, heads, 64, 64]."""
    envs, Lz = kv2.shape[0], kv2.shape[1]
    assert kv2.dtype == torch.bfloat16 and kv2.stride(-1) == 1 and v2t.dtype == torch.bfloat16 and v2t.is_contiguous()
    assert v2t.numel() >= envs * heads * 64 * 64
    v = kv2[:, :, 1]
    a = _lib.DitAttnArgs()
    a.X, a.V2T, a.V2T_src = None, v2t.data_ptr(), v.data_ptr()
    a.v2_bs, a.v2_rs = v.stride(0), v.stride(1)
    a.nseq, a.seq_per_env, a.T, a.heads, a.Lz = envs, 1, 1, heads, Lz
    a.ldx, a.ldo, a.k2_bs, a.k2_rs = 8, 8, 8, 8
    _lib.check(_lib.lib().ina_dit_attention(C.byref(a), _stream()), "dit_attention(v2t)")
    return v2t


def dit_attention(x: torch.Tensor, out: torch.Tensor, norms, kv2: torch.Tensor, v2t: torch.Tensor, head_gate: Optional[torch.Tensor],
                  T: int, seq_per_env: int, heads: int, eps: float = 1e-5, scale: Optional[float] = None,
                  stats: Optional[torch.Tensor] = None) -> torch.Tensor:
    """NextDiT attention stage in one launch: out = SDPA(LN(q1), LN(k1), v1) + tanh(gate) * SDPA(LN(q2), K2, V2).

    x bf16 [rows, 4*heads*64] = [q1 | k1 | v1 | q2] per token, rows = nseq * T; norms = ((g_q1, b_q1), (g_k1, b_k1), (g_q2, b_q2)) f32;
    kv2 bf16 [envs, Lz, 2, heads, 64]; v2t from dit_v2t(kv2); out bf16 [rows, heads*64]. stats f32 [rows, 4, 2]: (mean, rstd) of the four
    segments of every x row from the producer (dit_rowchain(seg_stats=)); None = the kernel computes them itself."""
    D = heads * 64
    assert x.dtype == torch.bfloat16 and x.dim() == 2 and x.shape[1] == 4 * D and x.stride(1) == 1 and x.shape[0] % T == 0
    assert out.dtype == torch.bfloat16 and out.shape == (x.shape[0], D) and out.stride(1) == 1
    nseq = x.shape[0] // T
    assert kv2.shape[0] * seq_per_env >= nseq and kv2.stride(-1) == 1 and kv2.stride(-2) == 64
    a = _lib.DitAttnArgs()
    a.X, a.O = x.data_ptr(), out.data_ptr()
    (a.g_q1, a.b_q1), (a.g_k1, a.b_k1), (a.g_q2, a.b_q2) = [(_f32(g).data_ptr(), _f32(b).data_ptr()) for g, b in norms]
    k = kv2[:, :, 0]
    a.K2, a.V2T, a.V2T_src, a.head_gate = k.data_ptr(), v2t.data_ptr(), None, _ptr(head_gate)
    a.k2_bs, a.k2_rs = k.stride(0), k.stride(1)
    a.nseq, a.T, a.heads, a.seq_per_env, a.Lz, a.ldx, a.ldo = nseq, T, heads, seq_per_env, kv2.shape[1], x.stride(0), out.stride(0)
    a.scale, a.eps = (scale if scale is not None else 64 ** -0.5), eps
    if stats is not None:
        assert stats.dtype == torch.float32 and stats.shape == (x.shape[0], 4, 2) and stats.is_contiguous()
        a.stats, a.stats_ld = stats.data_ptr(), 8
    _lib.check(_lib.lib().ina_dit_attention(C.byref(a), _stream()), "dit_attention")
    return out


def dit_rowchain(a_in: torch.Tensor, w1: torch.Tensor, gamma1: torch.Tensor, x: torch.Tensor, gate: Optional[torch.Tensor] = None,
                 gamma2: Optional[torch.Tensor] = None, mod_scale2: Optional[torch.Tensor] = None, h: Optional[torch.Tensor] = None,
                 w2: Optional[torch.Tensor] = None, c2: Optional[torch.Tensor] = None, glu2: bool = False, mod_div: int = 0, eps: float = 1e-5,
                 seg_stats: Optional[torch.Tensor] = None, seg_eps: float = 1e-5) -> torch.Tensor:
    """the row-local chain of a NextDiT block in one launch (csrc/dit_rowchain.hip):
        x += tanh(gate[r // mod_div]) * rmsnorm(bf16(a_in @ w1.T)) * gamma1
        H  = rmsnorm(x) * gamma2 * (1 + mod_scale2[r // mod_div])
        c2 = H @ w2.T   (glu2: silu(H @ wg.T) * (H @ wu.T) with w2's rows interleaved [gate16 | up16])
    a_in bf16 [M, K1] (K1 = 384 | 1024; M a multiple of 128), w1 bf16 [384, K1], x f32 [M, 384] (in place); w2 bf16 [N2, 384] -> c2 bf16 [M, N2 (/ 2)], or w2 None
    (then h bf16 [M, 384] may be given to receive H). The projection and H never leave the chip.
    seg_stats f32 [M, N2 // 384, 2] (plain second GEMM): receives (mean, rstd) of every 384-wide segment of the c2 rows = the LayerNorm
    statistics dit_attention(stats=) consumes."""
    assert a_in.dtype == torch.bfloat16 and w1.dtype == torch.bfloat16 and a_in.dim() == 2 and a_in.stride(1) == 1 and w1.stride(1) == 1
    M, K1 = a_in.shape
    assert w1.shape == (384, K1) and x.dtype == torch.float32 and x.shape == (M, 384) and x.stride(1) == 1
    a = _lib.DitRowchainArgs()
    a.A, a.W1, a.gamma1, a.X = a_in.data_ptr(), w1.data_ptr(), _f32(gamma1).data_ptr(), x.data_ptr()
    a.M, a.K1, a.lda, a.ldw1, a.ldx = M, K1, a_in.stride(0), w1.stride(0), x.stride(0)
    if gate is not None:
        assert gate.dtype == torch.float32 and gate.stride(-1) == 1
        a.gate, a.mod_ld = gate.data_ptr(), gate.stride(0)
    a.gamma2 = _ptr(_f32(gamma2))
    if mod_scale2 is not None:
        assert mod_scale2.dtype == torch.float32 and mod_scale2.stride(-1) == 1 and a.mod_ld in (0, mod_scale2.stride(0))
        a.mod_scale2, a.mod_ld = mod_scale2.data_ptr(), mod_scale2.stride(0)
    if h is not None:
        assert w2 is None and h.dtype == torch.bfloat16 and h.shape == (M, 384) and h.stride(1) == 1
        a.H, a.ldh = h.data_ptr(), h.stride(0)
    if w2 is not None:
        N2 = w2.shape[0]
        assert w2.dtype == torch.bfloat16 and w2.shape == (N2, 384) and w2.stride(1) == 1 and c2 is not None
        assert c2.dtype == torch.bfloat16 and c2.shape == (M, N2 // 2 if glu2 else N2) and c2.stride(1) == 1
        a.W2, a.C2, a.N2, a.ldw2, a.ldc2, a.glu2 = w2.data_ptr(), c2.data_ptr(), N2, w2.stride(0), c2.stride(0), int(glu2)
    if seg_stats is not None:
        assert w2 is not None and not glu2 and N2 % 384 == 0
        assert seg_stats.dtype == torch.float32 and seg_stats.shape == (M, N2 // 384, 2) and seg_stats.is_contiguous()
        a.seg_stats, a.seg_eps = seg_stats.data_ptr(), seg_eps
    a.mod_div, a.eps = mod_div, eps
    _lib.check(_lib.lib().ina_dit_rowchain(C.byref(a), _stream()), "dit_rowchain")
    return x


def resize_u8(x: torch.Tensor, out: torch.Tensor, bounds: torch.Tensor, coefs: torch.Tensor, axis: int) -> torch.Tensor:
    """one axis of PIL's 8-bit bicubic resample: x u8 contiguous, out the same shape with `axis` resized to bounds.shape[0];
    bounds int32 [n_out, 2], coefs int32 [n_out, ksize] (internnav_amd.preprocess builds PIL's tables)."""
    assert x.dtype == torch.uint8 and out.dtype == torch.uint8 and x.is_contiguous() and out.is_contiguous()
    assert bounds.dtype == torch.int32 and coefs.dtype == torch.int32 and bounds.is_contiguous() and coefs.is_contiguous()
    n_out = bounds.shape[0]
    outer = 1
    for d in x.shape[:axis]:
        outer *= d
    inner = 1
    for d in x.shape[axis + 1:]:
        inner *= d
    assert out.shape[axis] == n_out and out.numel() == outer * n_out * inner and coefs.shape[0] == n_out
    a = _lib.ResizeU8Args()
    a.in_, a.out, a.bounds, a.coefs = x.data_ptr(), out.data_ptr(), bounds.data_ptr(), coefs.data_ptr()
    a.outer, a.n_in, a.n_out, a.inner, a.ksize = outer, x.shape[axis], n_out, inner, coefs.shape[1]
    _lib.check(_lib.lib().ina_resize_u8(C.byref(a), _stream()), "resize_u8")
    return out


def qwen_patchify_u8(img: torch.Tensor, out: torch.Tensor, lut: torch.Tensor, ps: int = 14, merge: int = 2, tdup: int = 2) -> torch.Tensor:
    """img u8 [n, H, W, 3] -> out bf16 [n * (H/ps) * (W/ps), 3*tdup*ps*ps] in the HF Qwen2-VL patch layout; lut f32 [3, 256]."""
    assert img.dtype == torch.uint8 and img.is_contiguous() and img.dim() == 4 and img.shape[3] == 3
    assert out.dtype == torch.bfloat16 and out.dim() == 2 and out.stride(1) == 1 and lut.dtype == torch.float32 and lut.is_contiguous() and lut.numel() == 768
    n, H, W = img.shape[:3]
    assert out.shape[0] == n * (H // ps) * (W // ps) and out.shape[1] == 3 * tdup * ps * ps
    a = _lib.QwenPatchifyArgs()
    a.img, a.out, a.lut = img.data_ptr(), out.data_ptr(), lut.data_ptr()
    a.n, a.H, a.W, a.ps, a.merge, a.tdup, a.ldo = n, H, W, ps, merge, tdup, out.stride(0)
    _lib.check(_lib.lib().ina_qwen_patchify_u8(C.byref(a), _stream()), "qwen_patchify_u8")
    return out


def u8_lut(x: torch.Tensor, out: torch.Tensor, lut: torch.Tensor) -> torch.Tensor:
    """out[i] = bf16(lut[x[i]]); x u8, lut f32 [256]."""
    assert x.dtype == torch.uint8 and x.is_contiguous() and out.dtype == torch.bfloat16 and out.is_contiguous() and out.numel() == x.numel()
    assert lut.dtype == torch.float32 and lut.is_contiguous() and lut.numel() == 256
    a = _lib.U8LutArgs()
    a.in_, a.out, a.lut, a.n = x.data_ptr(), out.data_ptr(), lut.data_ptr(), x.numel()
    _lib.check(_lib.lib().ina_u8_lut(C.byref(a), _stream()), "u8_lut")
    return out


def resize_f32(x: torch.Tensor, out: torch.Tensor, bounds: torch.Tensor, coefs: torch.Tensor, axis: int) -> torch.Tensor:
    """one axis of PIL's float ("F" mode) bicubic resample: x f32 contiguous, coefs f64 [n_out, ksize], bounds int32 [n_out, 2]."""
    assert x.dtype == torch.float32 and out.dtype == torch.float32 and x.is_contiguous() and out.is_contiguous()
    assert bounds.dtype == torch.int32 and coefs.dtype == torch.float64 and bounds.is_contiguous() and coefs.is_contiguous()
    n_out = bounds.shape[0]
    outer = 1
    for d in x.shape[:axis]:
        outer *= d
    inner = 1
    for d in x.shape[axis + 1:]:
        inner *= d
    assert out.shape[axis] == n_out and out.numel() == outer * n_out * inner and coefs.shape[0] == n_out
    a = _lib.ResizeF32Args()
    a.in_, a.out, a.bounds, a.coefs = x.data_ptr(), out.data_ptr(), bounds.data_ptr(), coefs.data_ptr()
    a.outer, a.n_in, a.n_out, a.inner, a.ksize = outer, x.shape[axis], n_out, inner, coefs.shape[1]
    _lib.check(_lib.lib().ina_resize_f32(C.byref(a), _stream()), "resize_f32")
    return out


def gn_mish(x: torch.Tensor, out: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, seqs: int, T: int, pad: int, in_seq_stride: int,
            groups: int = 8, residual: Optional[torch.Tensor] = None, film_env: Optional[torch.Tensor] = None, film_step: Optional[torch.Tensor] = None,
            film_off: int = 0, seq_per_env: int = 1, eps: float = 1e-5) -> torch.Tensor:
    """GroupNorm(groups) -> Mish [-> scale * y + bias (FiLM)] [-> + residual] of one ConditionalUnet1D Conv1dBlock.
    x bf16|f32 [rows, C]: conv output, row (b, t) at b * in_seq_stride + t; out / residual bf16 padded [seqs * (T + 2 pad), C]."""
    assert x.dtype in (torch.bfloat16, torch.float32) and out.dtype == torch.bfloat16 and x.stride(1) == 1 and out.stride(1) == 1
    C_ = out.shape[1]
    a = _lib.GnMishArgs()
    a.x_f32 = 1 if x.dtype == torch.float32 else 0
    a.X, a.Y, a.gamma, a.beta = x.data_ptr(), out.data_ptr(), _f32(gamma).data_ptr(), _f32(beta).data_ptr()
    a.seqs, a.T, a.C, a.groups, a.pad, a.in_seq_stride, a.ldx, a.ldy = seqs, T, C_, groups, pad, in_seq_stride, x.stride(0), out.stride(0)
    if residual is not None:
        assert residual.dtype == torch.bfloat16 and residual.stride(1) == 1
        a.R, a.ldr = residual.data_ptr(), residual.stride(0)
    if film_env is not None:
        assert film_env.dtype == torch.float32 and film_step.dtype == torch.float32 and film_env.stride(1) == 1 and film_step.is_contiguous()
        a.film_env, a.film_step, a.film_ld, a.film_off, a.seq_per_env = film_env.data_ptr(), film_step.data_ptr(), film_env.stride(0), film_off, seq_per_env
    a.eps = eps
    _lib.check(_lib.lib().ina_gn_mish(C.byref(a), _stream()), "gn_mish")
    return out


def pad_rows(x: torch.Tensor, seqs: int, T: int, pad: int, bias: Optional[torch.Tensor] = None) -> torch.Tensor:
    """zero the pad rows of a padded bf16 buffer [seqs * (T + 2 pad), C] (+ optional per-channel bias on the valid rows)."""
    assert x.dtype == torch.bfloat16 and x.stride(1) == 1
    a = _lib.PadRowsArgs()
    a.X, a.bias, a.seqs, a.T, a.pad, a.C, a.ldx = x.data_ptr(), _ptr(_f32(bias)), seqs, T, pad, x.shape[1], x.stride(0)
    _lib.check(_lib.lib().ina_pad_rows(C.byref(a), _stream()), "pad_rows")
    return x


def ddim_step(eps: torch.Tensor, sample: torch.Tensor, xin: torch.Tensor, seqs: int, T: int, D: int, pad: int, coefs, clip: float = 1.0,
              use_clipped_model_output: bool = False):
    """DDIMScheduler.step (eta 0) on the fp32 sample [seqs * T, D]; eps f32 in padded row indexing; xin bf16 padded network input.
    use_clipped_model_output as diffusers' step() argument (default False: the direction term keeps the network's eps)."""
    assert eps.dtype == torch.float32 and sample.dtype == torch.float32 and sample.is_contiguous() and xin.dtype == torch.bfloat16
    a = _lib.DdimStepArgs()
    a.eps, a.sample, a.Xin, a.seqs, a.T, a.D, a.pad, a.lde, a.ldx = eps.data_ptr(), sample.data_ptr(), xin.data_ptr(), seqs, T, D, pad, eps.stride(0), xin.stride(0)
    a.inv_sqrt_a, a.sqrt_b, a.sqrt_ap, a.sqrt_bp, a.clip = [float(c) for c in coefs] + [float(clip)]
    a.use_clipped_model_output = int(bool(use_clipped_model_output))
    _lib.check(_lib.lib().ina_ddim_step(C.byref(a), _stream()), "ddim_step")
    return sample
