"""Torch-tensor front end of the SFT-step kernels (backward / optimiser half of the C-ABI, include/internnav_amd.h).

Same rules as `ops.py`: PyTorch owns device memory and the stream, the arithmetic runs in libinternnav_amd.so; there is
no fallback. 2-D tensors are [rows, C] with a contiguous last dim (row stride free), dtype bf16 or f32.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import _lib
from .ops import ACT as _ACT, _DT, _stream

ACT = dict(_ACT, tanh=6)
EW_AFFINE, EW_ACT_FWD, EW_ACT_BWD, EW_GLU_FWD, EW_GLU_BWD, EW_DROPOUT = range(6)
SCALE_F = {None: 0, "id": 0, "one_plus": 1, "tanh": 2}


def _2d(t: torch.Tensor) -> torch.Tensor:
    assert t.dim() == 2 and t.stride(1) == 1 and t.dtype in _DT, f"need a 2-D bf16|f32 tensor with contiguous rows, got {tuple(t.shape)} {t.dtype} {t.stride()}"
    return t


def _ew(op, A, B=None, D=None, S=None, s_div=1, s_f=0, tab=None, act=0, out=None, out2=None, out_dtype=None, accumulate=False, drop=None, drop_salt=None):
    A = _2d(A)
    rows, Cd = A.shape
    if out is None:
        out = torch.empty(rows, Cd, dtype=out_dtype or A.dtype, device=A.device)
    a = _lib.EwArgs()
    a.op, a.rows, a.C = op, rows, Cd
    a.A, a.a_dt, a.lda = A.data_ptr(), _DT[A.dtype], A.stride(0)
    if B is not None:
        B = _2d(B)
        assert B.shape == A.shape
        a.B, a.b_dt, a.ldb = B.data_ptr(), _DT[B.dtype], B.stride(0)
    if D is not None:
        D = _2d(D)
        assert D.shape == A.shape
        a.D, a.d_dt, a.ldd = D.data_ptr(), _DT[D.dtype], D.stride(0)
    if S is not None:
        S = _2d(S)
        assert S.shape[1] == Cd and S.shape[0] * s_div >= rows, f"scale {tuple(S.shape)} x {s_div} vs rows {rows}"
        a.S, a.s_dt, a.lds, a.s_div, a.s_f = S.data_ptr(), _DT[S.dtype], S.stride(0), s_div, s_f
    if tab is not None:
        assert tab.dtype == torch.float32 and tab.is_contiguous() and tab.shape[-1] == Cd
        a.tab, a.tab_mod = tab.data_ptr(), tab.numel() // Cd
    out = _2d(out)
    assert out.shape == A.shape
    a.Y, a.y_dt, a.ldy = out.data_ptr(), _DT[out.dtype], out.stride(0)
    if out2 is not None:
        out2 = _2d(out2)
        a.Y2, a.y2_dt, a.ldy2 = out2.data_ptr(), _DT[out2.dtype], out2.stride(0)
    a.act = act
    a.accumulate = 1 if accumulate else 0
    if drop is not None:
        a.drop_seed, a.drop_thresh, a.drop_scale = drop
        if drop_salt is not None:
            a.drop_salt = _salt_ptr(drop_salt)
    _lib.check(_lib.lib().ina_ew(C.byref(a), _stream()), "ew")
    return out


def affine(x, scale=None, s_div=1, s_f=None, base=None, tab=None, out=None, out_dtype=None, accumulate=False):
    """out (+)= x * f(scale[r // s_div]) + base + tab[r % len(tab)]   (f: None|'id' s, 'one_plus' 1+s, 'tanh' tanh s)."""
    return _ew(EW_AFFINE, x, B=base, S=scale, s_div=s_div, s_f=SCALE_F[s_f], tab=tab, out=out, out_dtype=out_dtype, accumulate=accumulate)


def act_fwd(x, act, out=None, out_dtype=None):
    return _ew(EW_ACT_FWD, x, act=ACT[act], out=out, out_dtype=out_dtype)


def act_bwd(x, dy, act, out=None, out_dtype=None, accumulate=False):
    """out (+)= dy * act'(x)."""
    return _ew(EW_ACT_BWD, x, B=dy, act=ACT[act], out=out, out_dtype=out_dtype or dy.dtype, accumulate=accumulate)


def _salt_ptr(salt: torch.Tensor) -> int:
    """device word added to a dropout seed when the kernel starts (ina_*_args.drop_salt): int32 [1] on the launch device"""
    assert salt.dtype == torch.int32 and salt.numel() == 1 and salt.is_cuda, "drop_salt: one int32 device word"
    return salt.data_ptr()


def dropout(x, p: float, seed: int, out=None, out_dtype=None, salt=None):
    """nn.Dropout in train mode with the library's counter-based mask (element index = r * C + c of the [rows, C] tensor); applying
    it to dy with the same (p, seed) is the backward."""
    from .ops import drop_params

    return _ew(EW_DROPOUT, x, out=out, out_dtype=out_dtype, drop=drop_params(p, seed), drop_salt=salt)


def glu_fwd(a, b, out=None):
    return _ew(EW_GLU_FWD, a, B=b, out=out)


def glu_bwd(a, b, dy, da=None, db=None):
    """silu(a) * b backward -> (da, db)."""
    if da is None:
        da = torch.empty(a.shape, dtype=a.dtype, device=a.device)
    if db is None:
        db = torch.empty(b.shape, dtype=b.dtype, device=b.device)
    _ew(EW_GLU_BWD, a, B=b, D=dy, out=da, out2=db)
    return da, db


def colsum_chunks(group_rows: int) -> int:
    """row chunks (= partial sums per column) the library cuts a group of `group_rows` rows into: csrc/train.hip colsum_chunk_rows - 32-row chunks
    up to 2048 rows, 64 up to 4096, 128 up to 8192, 256 beyond (the library checks the partial buffer against its own count)."""
    if group_rows <= 32:
        return 1
    chunk = 32 if group_rows <= 2048 else 64 if group_rows <= 4096 else 128 if group_rows <= 8192 else 256
    return (group_rows + chunk - 1) // chunk


def colsum(x, x2=None, out=None, group_rows=0, accumulate=False, scale=1.0, x2_bcast=False, out_cs=1, x_bcast=False):
    """out[g, c] (+)= scale * sum_{rows of group g} x[r, c] * x2[r, c];  x2_bcast: x2 is [rows] (one value per row)."""
    rows = x.shape[0]
    a = _lib.ColsumArgs()
    if x_bcast:
        assert x.dim() == 1 or x.shape[1] == 1
        Cd = x2.shape[1]
        a.x_cs = 0
        a.ldx = x.stride(0)
    else:
        x = _2d(x)
        Cd = x.shape[1]
        a.x_cs, a.ldx = 1, x.stride(0)
    a.X, a.x_dt = x.data_ptr(), _DT[x.dtype]
    if x2 is not None:
        if x2_bcast:
            assert x2.shape[0] == rows
            a.x2_cs, a.ldx2 = 0, x2.stride(0)
        else:
            x2 = _2d(x2)
            assert x2.shape == (rows, Cd)
            a.x2_cs, a.ldx2 = 1, x2.stride(0)
        a.X2, a.x2_dt = x2.data_ptr(), _DT[x2.dtype]
    gr = group_rows or rows
    groups = rows // gr
    if out is None:
        out = torch.empty(groups, Cd, dtype=torch.float32, device=x.device)
        assert not accumulate
    assert out.dtype == torch.float32
    a.out = out.data_ptr()
    a.out_cs = out_cs
    a.ldo = out.stride(0) if out.dim() == 2 else Cd * out_cs
    a.rows, a.C, a.group_rows = rows, Cd, gr
    nchunk = colsum_chunks(gr)
    if nchunk > 1:
        part = torch.empty(groups * nchunk * Cd, dtype=torch.float32, device=x.device)
        a.partial, a.partial_elems = part.data_ptr(), part.numel()
    a.accumulate = 1 if accumulate else 0
    a.scale = scale
    _lib.check(_lib.lib().ina_colsum(C.byref(a), _stream()), "colsum")
    return out


def norm_bwd(x, dy, gamma=None, eps=1e-5, rms=False, dx=None, dx_dtype=None, accumulate=False, want_xhat=False):
    """backward of y = norm(x) * gamma (+ beta): returns (dx, xhat or None); dx (+)= when accumulate."""
    x, dy = _2d(x), _2d(dy)
    rows, Cd = x.shape
    assert dy.shape == x.shape
    if dx is None:
        assert not accumulate
        dx = torch.empty(rows, Cd, dtype=dx_dtype or x.dtype, device=x.device)
    dx = _2d(dx)
    a = _lib.NormBwdArgs()
    a.X, a.x_dt, a.ldx = x.data_ptr(), _DT[x.dtype], x.stride(0)
    a.DY, a.dy_dt, a.lddy = dy.data_ptr(), _DT[dy.dtype], dy.stride(0)
    a.DX, a.dx_dt, a.lddx = dx.data_ptr(), _DT[dx.dtype], dx.stride(0)
    if gamma is not None:
        assert gamma.dtype == torch.float32 and gamma.is_contiguous() and gamma.numel() == Cd
        a.gamma = gamma.data_ptr()
    xhat = None
    if want_xhat:
        xhat = torch.empty(rows, Cd, dtype=torch.bfloat16, device=x.device)
        a.XHAT, a.ldxh = xhat.data_ptr(), Cd
    a.rows, a.C, a.rms, a.eps = rows, Cd, 1 if rms else 0, eps
    a.accumulate = 1 if accumulate else 0
    _lib.check(_lib.lib().ina_norm_bwd(C.byref(a), _stream()), "norm_bwd")
    return dx, xhat


def transpose(x, pad: int = 8, out=None):
    """x [rows, cols] -> bf16 [cols, ceil(rows / pad) * pad] (zero-filled tail): the K-contiguous operand of a backward GEMM."""
    x = _2d(x)
    rows, cols = x.shape
    ldy = (rows + pad - 1) // pad * pad
    if out is None:
        out = torch.empty(cols, ldy, dtype=torch.bfloat16, device=x.device)
    assert out.shape == (cols, ldy) and out.is_contiguous() and out.dtype == torch.bfloat16
    a = _lib.TransposeArgs()
    a.X, a.Y, a.rows, a.cols, a.x_dt, a.ldx, a.ldy = x.data_ptr(), out.data_ptr(), rows, cols, _DT[x.dtype], x.stride(0), ldy
    _lib.check(_lib.lib().ina_transpose(C.byref(a), _stream()), "transpose")
    return out


def sparse_rows(inp, idx, coef, out=None, accumulate=False):
    """out[t] (+)= sum_j coef[t, j] * inp[idx[t, j]];  inp f32 [n_in, C], idx int32 [n_out, taps] (-1 = unused), coef f32."""
    assert inp.dtype == torch.float32 and inp.is_contiguous() and idx.dtype == torch.int32 and coef.dtype == torch.float32
    assert idx.is_contiguous() and coef.is_contiguous() and idx.shape == coef.shape
    n_out, taps = idx.shape
    Cd = inp.shape[1]
    if out is None:
        out = torch.empty(n_out, Cd, dtype=torch.float32, device=inp.device)
    assert out.dtype == torch.float32 and out.is_contiguous() and out.shape == (n_out, Cd)
    a = _lib.SparseRowsArgs()
    a.inp, a.out, a.idx, a.coef = inp.data_ptr(), out.data_ptr(), idx.data_ptr(), coef.data_ptr()
    a.n_out, a.C, a.taps, a.accumulate = n_out, Cd, taps, 1 if accumulate else 0
    _lib.check(_lib.lib().ina_sparse_rows(C.byref(a), _stream()), "sparse_rows")
    return out


def small_linear(x, w, bias=None, tab=None, out=None, out_dtype=torch.float32, w_transposed=False):
    """out[r, n] = sum_k x[r, k] * W[n, k] + bias[n] + tab[r % len(tab), n]; w f32 [N, K] (w_transposed: w is [K, N], used as W^T)."""
    x = _2d(x)
    assert w.dtype == torch.float32 and w.dim() == 2
    N, K = (w.shape[1], w.shape[0]) if w_transposed else w.shape
    assert x.shape[1] == K
    if out is None:
        out = torch.empty(x.shape[0], N, dtype=out_dtype, device=x.device)
    out = _2d(out)
    a = _lib.SmallLinearArgs()
    a.X, a.x_dt, a.ldx = x.data_ptr(), _DT[x.dtype], x.stride(0)
    a.W = w.data_ptr()
    a.w_ns, a.w_ks = (w.stride(1), w.stride(0)) if w_transposed else (w.stride(0), w.stride(1))
    if bias is not None:
        assert bias.dtype == torch.float32 and bias.is_contiguous()
        a.bias = bias.data_ptr()
    if tab is not None:
        assert tab.dtype == torch.float32 and tab.is_contiguous() and tab.shape[-1] == N
        a.tab, a.tab_mod = tab.data_ptr(), tab.numel() // N
    a.Y, a.y_dt, a.ldy = out.data_ptr(), _DT[out.dtype], out.stride(0)
    a.rows, a.N, a.K = x.shape[0], N, K
    _lib.check(_lib.lib().ina_small_linear(C.byref(a), _stream()), "small_linear")
    return out


def mse_masked(pred, target, mask, T: int, loss_scale: float = 1.0, want_grad: bool = True):
    """masked MSE of internvla_n1.py:283-286: pred [nseq*T, >=D] (row stride free), target f32 [nseq*T, D], mask f32 [nseq].
    Returns (loss f32 [1], dpred f32 [nseq*T, D] or None)."""
    pred = _2d(pred)
    assert target.dtype == torch.float32 and target.is_contiguous() and mask.dtype == torch.float32 and mask.is_contiguous()
    rows, D = target.shape
    nseq = rows // T
    assert nseq * T == rows and mask.numel() == nseq and pred.shape[0] == rows and pred.shape[1] >= D
    loss = torch.empty(1, dtype=torch.float32, device=pred.device)
    dpred = torch.empty(rows, D, dtype=torch.float32, device=pred.device) if want_grad else None
    a = _lib.MseArgs()
    a.pred, a.pred_dt, a.ldp = pred.data_ptr(), _DT[pred.dtype], pred.stride(0)
    a.target, a.mask, a.loss = target.data_ptr(), mask.data_ptr(), loss.data_ptr()
    if dpred is not None:
        a.dpred, a.dpred_dt, a.lddp = dpred.data_ptr(), 1, D
    a.nseq, a.T, a.D, a.loss_scale = nseq, T, D, loss_scale
    _lib.check(_lib.lib().ina_mse_masked(C.byref(a), _stream()), "mse_masked")
    return loss, dpred


def adamw(p, g, m, v, lr, beta1, beta2, eps, wd, step: int, p_bf16=None, sumsq_parts=None, max_norm: float = 0.0, grad_scale: float = 1.0,
          norm_out=None, zero_grad: bool = False):
    """one fused torch.optim.AdamW step (+ clip_grad_norm_(max_norm) from sumsq_parts, + grad averaging) on flat f32 buffers."""
    for t in (p, g, m, v):
        assert t.dtype == torch.float32 and t.is_contiguous() and t.numel() == p.numel()
    a = _lib.AdamwArgs()
    a.p, a.g, a.m, a.v, a.n = p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel()
    if p_bf16 is not None:
        assert p_bf16.dtype == torch.bfloat16 and p_bf16.is_contiguous() and p_bf16.numel() == p.numel()
        a.p_bf16 = p_bf16.data_ptr()
    if sumsq_parts is not None:
        assert sumsq_parts.dtype == torch.float32 and sumsq_parts.is_contiguous()
        a.sumsq_parts, a.n_parts = sumsq_parts.data_ptr(), sumsq_parts.numel()
    if norm_out is not None:
        a.norm_out = norm_out.data_ptr()
    a.lr, a.beta1, a.beta2, a.eps, a.wd = lr, beta1, beta2, eps, wd
    a.bc1, a.bc2 = 1.0 - beta1 ** step, 1.0 - beta2 ** step
    a.max_norm, a.grad_scale, a.zero_grad = max_norm, grad_scale, 1 if zero_grad else 0
    _lib.check(_lib.lib().ina_adamw(C.byref(a), _stream()), "adamw")


def sumsq_parts(flat: torch.Tensor, width: int = 1024) -> torch.Tensor:
    """partial sums of squares of a flat f32 buffer (numel a multiple of `width`): f32 [width]; their sum is ||flat||^2."""
    assert flat.dtype == torch.float32 and flat.is_contiguous() and flat.numel() % width == 0
    v = flat.view(-1, width)
    return colsum(v, v).view(-1)


GEMM_DW_MAX_TILES_LONG = 512     # 64 x 64 tiles of the gradient up to which gemm_dw also takes reductions over more than 2048 rows (2304 x 768 = 432)
GEMM_DW_SLABS_PER_RANGE = 24      # 64-row slabs a workgroup of gemm_dw walks before the rows are cut into ranges (grid.z) with a reduce launch behind


def gemm_dw_splits(rows: int) -> int:
    """row ranges of gemm_dw: 1 (one launch) up to 2048 rows, then ranges of ~1536 rows, at most 16"""
    if rows <= 2048:
        return 1
    return min(16, (rows + 64 * GEMM_DW_SLABS_PER_RANGE - 1) // (64 * GEMM_DW_SLABS_PER_RANGE))


def gemm_dw_ok(dy: torch.Tensor, x: torch.Tensor, gW: torch.Tensor) -> bool:
    """can gemm_dw take these operands (else: two transposes + ops.linear + colsum)? dy [rows, N] f32 | bf16, x [rows, K] bf16, gW f32 [N, K] view.
    (profiles/r06z_gemm_dw_vs_five_launches.txt: 14 vs 29 us at 768 rows, 8 vs 20 at 96, 35 vs 170 at 12 288; long reductions run as row ranges +
    a reduce launch)"""
    if dy.dim() != 2 or x.dim() != 2 or gW.dim() != 2 or x.dtype != torch.bfloat16 or gW.dtype != torch.float32 or dy.dtype not in _DT:
        return False
    N, K = gW.shape
    # measured ahead of the five launches on every System-1 layer shape (24 ... 12 288 rows, 384 x 384 ... 2304 x 768); bigger gradients over long
    # reductions are not its case (and would want splits x N x K floats of scratch): they keep the tiled GEMM
    if dy.shape[0] > 2048 and ((N + 63) // 64) * ((K + 63) // 64) > GEMM_DW_MAX_TILES_LONG:
        return False
    es = dy.element_size()
    return (dy.shape == (x.shape[0], N) and x.shape[1] == K and dy.stride(1) == 1 and x.stride(1) == 1 and gW.stride(1) == 1 and N % 8 == 0 and K % 8 == 0
            and dy.stride(0) % 8 == 0 and x.stride(0) % 8 == 0 and gW.stride(0) % 4 == 0 and dy.data_ptr() % (8 * es) == 0 and x.data_ptr() % 16 == 0
            and gW.data_ptr() % 16 == 0)


def gemm_dw(dy: torch.Tensor, x: torch.Tensor, gW: torch.Tensor, gb: Optional[torch.Tensor] = None, splits: Optional[int] = None):
    """gW[n, k] += sum_r dy[r, n] * x[r, k]; gb[n] += sum_r dy[r, n]: weight / bias gradient of nn.Linear in ONE launch from the tape's row-major
    tensors (csrc/gemm_dw.hip) - the tiled GEMM needed two transposed copies and the bias two column-sum launches."""
    assert gemm_dw_ok(dy, x, gW)
    a = _lib.GemmDwArgs()
    a.DY, a.X, a.dW = dy.data_ptr(), x.data_ptr(), gW.data_ptr()
    if gb is not None:
        assert gb.dtype == torch.float32 and gb.is_contiguous() and gb.numel() == gW.shape[0]
        a.db = gb.data_ptr()
    a.rows, a.N, a.K, a.dy_dt = dy.shape[0], gW.shape[0], gW.shape[1], _DT[dy.dtype]
    a.lddy, a.ldx, a.ldw = dy.stride(0), x.stride(0), gW.stride(0)
    a.splits = gemm_dw_splits(a.rows) if splits is None else splits
    if a.splits > 1:
        part = torch.empty(a.splits * a.N * (a.K + 1), dtype=torch.float32, device=dy.device)
        a.partial, a.partial_elems = part.data_ptr(), part.numel()
    _lib.check(_lib.lib().ina_gemm_dw(C.byref(a), _stream()), "gemm_dw")


def gemm_nn(x, w, out=None, out_dtype=torch.float32, splits: Optional[int] = None):
    """out[m, k] = sum_n x[m, n] * w[n, k] for a few rows (m <= 16): dX of a frozen nn.Linear with w [N, K] in its stored layout."""
    assert x.dtype == torch.bfloat16 and w.dtype == torch.bfloat16 and x.dim() == 2 and w.dim() == 2
    assert x.stride(1) == 1 and w.stride(1) == 1
    M, N = x.shape
    K = w.shape[1]
    assert w.shape[0] == N and M <= 16
    mr = 8 if M <= 8 else 16
    if splits is None:
        colblocks = (K + 511) // 512
        splits = max(1, min((N + 15) // 16, (512 + colblocks - 1) // colblocks))      # ~2 workgroups per CU
        while (N + splits - 1) // splits * mr * 4 > 48 * 1024:
            splits *= 2
    part = torch.empty(splits * 4, mr * K, dtype=torch.float32, device=x.device)
    a = _lib.GemmNnArgs()
    a.X, a.W, a.partial, a.partial_elems = x.data_ptr(), w.data_ptr(), part.data_ptr(), part.numel()
    a.M, a.N, a.K, a.ldx, a.ldw, a.splits = M, N, K, x.stride(0), w.stride(0), splits
    _lib.check(_lib.lib().ina_gemm_nn_bf16(C.byref(a), _stream()), "gemm_nn_bf16")
    red = colsum(part).view(mr, K)[:M]
    if out is None and out_dtype == torch.float32:
        return red
    if out is None:
        out = torch.empty(M, K, dtype=out_dtype, device=x.device)
    return affine(red, out=out)


def attention_bwd(q, k, v, o, do, scale=None, causal=False, k_len=None, kv_bdiv=1, dq=None, dk=None, dv=None, kv_row0=0, need_dkv=True,
                  nsplit: Optional[int] = None, drop_p: float = 0.0, drop_seed: int = 0, drop_salt=None):
    """backward of ops.attention (dense): q/o/do [B, Lq, H, D], k/v [Bk, Lk, Hkv, D] (last dim contiguous, other strides free).
    Returns (dq [B,Lq,H,D], dk, dv [B, Lk - kv_row0, H, D]) - dk / dv are per QUERY head (sum GQA groups outside)."""
    assert q.dtype == k.dtype == v.dtype == o.dtype == do.dtype == torch.bfloat16
    B, Lq, H, D = q.shape
    Bk, Lk, Hkv, _ = k.shape
    assert o.shape == q.shape and do.shape == q.shape and do.stride() == o.stride(), "dO must share O's layout"
    a = _lib.AttnBwdArgs()
    f = a.f
    f.Q, f.K, f.V, f.O = q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr()
    f.q_bs, f.q_rs, f.q_hs = q.stride(0), q.stride(1), q.stride(2)
    f.k_bs, f.k_rs, f.k_hs = k.stride(0), k.stride(1), k.stride(2)
    f.v_bs, f.v_rs, f.v_hs = v.stride(0), v.stride(1), v.stride(2)
    f.o_bs, f.o_rs, f.o_hs = o.stride(0), o.stride(1), o.stride(2)
    f.B, f.H, f.Hkv, f.Lq, f.Lk, f.D = B, H, Hkv, Lq, Lk, D
    f.causal, f.kv_bdiv = 1 if causal else 0, kv_bdiv
    f.scale = float(scale) if scale is not None else float(D) ** -0.5
    if k_len is not None:
        assert k_len.dtype == torch.int32 and k_len.is_contiguous()
        f.k_len = k_len.data_ptr()
    for t in (q, k, v, o, do):
        assert t.stride(-1) == 1
    if dq is None:
        dq = torch.empty(B, Lq, H, D, dtype=torch.bfloat16, device=q.device)
    assert dq.stride(-1) == 1 and dq.shape == q.shape
    a.dO, a.dQ = do.data_ptr(), dq.data_ptr()
    a.dq_bs, a.dq_rs, a.dq_hs = dq.stride(0), dq.stride(1), dq.stride(2)
    stats = torch.empty(2, B, H, Lq, dtype=torch.float32, device=q.device)
    a.lse, a.delta = stats[0].data_ptr(), stats[1].data_ptr()
    if drop_p > 0.0:
        from .ops import drop_params

        f.drop_seed, f.drop_thresh, f.drop_scale = drop_params(drop_p, drop_seed)
        if drop_salt is not None:
            f.drop_salt = _salt_ptr(drop_salt)
        nsplit = 1
    if nsplit is None:      # few query rows against a long key axis: spread the keys over the chip
        nsplit = min(32, (Lk + 127) // 128) if (Lq <= 32 and Lk >= 512) else 1
    dq32 = None
    if nsplit > 1:
        part = torch.empty(B, H, nsplit, 2, Lq, dtype=torch.float32, device=q.device)
        dq32 = torch.zeros(B, Lq, H, D, dtype=torch.float32, device=q.device)
        a.nsplit, a.part, a.dq32 = nsplit, part.data_ptr(), dq32.data_ptr()
    if need_dkv:
        rows = min(Lq, Lk) if kv_row0 < 0 else Lk - kv_row0
        if dk is None:
            dk = torch.empty(B, rows, H, D, dtype=torch.bfloat16, device=q.device)
        if dv is None:
            dv = torch.empty(B, rows, H, D, dtype=torch.bfloat16, device=q.device)
        assert dk.shape == (B, rows, H, D) and dv.shape == dk.shape and dk.stride() == dv.stride() and dk.stride(-1) == 1
        a.dK, a.dV = dk.data_ptr(), dv.data_ptr()
        a.dkv_bs, a.dkv_rs, a.dkv_hs = dk.stride(0), dk.stride(1), dk.stride(2)
        a.kv_row0 = kv_row0
    _lib.check(_lib.lib().ina_attention_bwd_bf16(C.byref(a), _stream()), "attention_bwd_bf16")
    if dq32 is not None:
        dq.copy_(dq32)
    return dq, dk, dv
