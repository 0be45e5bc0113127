"""TEST INFRASTRUCTURE: torch (CPU) stand-ins for the kernel wrappers of `internnav_amd.ops` / `internnav_amd.train_ops`, with the same
call contracts (dtypes, strides, in-place outputs, accumulate flags). They let the CPU test run exercise the WIRING of the SFT tape
(`internnav_amd/sft.py`: which op feeds which, what every backward closure accumulates where) against the oracle's autograd without a GPU.
They are never imported by the product package; the kernels themselves are tested on the GPU against the same formulas
(tests/test_train_ops_gpu.py) and the whole step against the oracle (tests/test_sft_gpu.py)."""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

BF, F32 = torch.bfloat16, torch.float32
_ACT = {"gelu_erf": F.gelu, "gelu": F.gelu, "gelu_tanh": lambda t: F.gelu(t, approximate="tanh"), "relu": F.relu, "silu": F.silu, "tanh": torch.tanh,
        None: lambda t: t, "none": lambda t: t}
_SCALE = {None: lambda s: s, "id": lambda s: s, "one_plus": lambda s: 1 + s, "tanh": torch.tanh}


def _store(out, val, out_dtype=None, accumulate=False):
    if out is None:
        return val.to(out_dtype or val.dtype)
    if accumulate:
        out.copy_((out.float() + val.float()).to(out.dtype))
    else:
        out.copy_(val.to(out.dtype))
    return out


# ---------------------------------------------------------------------------------------------------------------- ops.*
def linear(x, w, bias=None, act=None, colscale=None, residual=None, out=None, out_dtype=BF, glu=False, batched=False, **_):
    assert x.dtype == BF and w.dtype == BF and not glu
    y = x.float() @ w.float().t()
    if bias is not None:
        y = y + bias
    if colscale is not None:
        y = y * colscale
    y = _ACT[act](y)
    if residual is not None:
        y = y + residual.float()
    return _store(out, y, out_dtype if out is None else None)


def norm(x, gamma=None, beta=None, eps=1e-5, rms=False, out=None, out32=None, **_):
    xf = x.float()
    if rms:
        y = xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + eps)
    else:
        y = F.layer_norm(xf, (xf.shape[-1],), None, None, eps)
    if gamma is not None:
        y = y * gamma
    if beta is not None:
        y = y + beta
    if out32 is not None:
        out32.copy_(y)
        return out32
    return _store(out, y, BF)


def _attn(q, k, v, causal):
    B, Lq, H, D = q.shape
    Lk = k.shape[1]
    s = torch.einsum("bqhd,bkhd->bhqk", q, k) * D ** -0.5
    if causal:
        s = s.masked_fill(~(torch.arange(Lk)[None, :] <= torch.arange(Lq)[:, None] + (Lk - Lq)), float("-inf"))
    return torch.einsum("bhqk,bkhd->bqhd", torch.softmax(s, -1), v).contiguous()      # the kernel writes a dense [B, Lq, H, D] tensor


COUNT_ONLY = False      # test_dropout_sites_*: run the train-mode graph for its mask-site COUNT, masks themselves are not applied here


def attention(q, k, v, scale=None, causal=False, out=None, drop_p=0.0, drop_seed=0, **_):
    assert drop_p == 0.0 or COUNT_ONLY, "the CPU stand-in covers the eval-mode graph"
    o = _attn(q.float(), k.float(), v.float(), causal)
    return _store(out, o, BF)


def patchify(img, out, mean=(0.0, 0.0, 0.0), std=(1.0, 1.0, 1.0), ps=14):
    n, H, W, C = img.shape
    x = img.float().permute(0, 3, 1, 2)
    if C == 1:
        x = x.expand(n, 3, H, W)
    x = (x - torch.tensor(mean).view(1, 3, 1, 1)) / torch.tensor(std).view(1, 3, 1, 1)
    p = x.unfold(2, ps, ps).unfold(3, ps, ps)                      # [n, 3, gh, gw, ps, ps]
    p = p.permute(0, 2, 3, 1, 4, 5).reshape(n * (H // ps) * (W // ps), 3 * ps * ps)
    out.zero_()
    out[:, : p.shape[1]] = p.to(out.dtype)
    return out


# ---------------------------------------------------------------------------------------------------------------- train_ops.*
def affine(x, scale=None, s_div=1, s_f=None, base=None, tab=None, out=None, out_dtype=None, accumulate=False):
    y = x.float()
    if scale is not None:
        y = y * _SCALE[s_f](scale.float()).repeat_interleave(s_div, 0)[: y.shape[0]]
    if base is not None:
        y = y + base.float()
    if tab is not None:
        t = tab.reshape(-1, y.shape[1])
        y = y + t.repeat(y.shape[0] // t.shape[0], 1)
    return _store(out, y, out_dtype or x.dtype, accumulate)


def act_fwd(x, act, out=None, out_dtype=None):
    return _store(out, _ACT[act](x.float()), out_dtype or x.dtype)


def act_bwd(x, dy, act, out=None, out_dtype=None, accumulate=False):
    xr = x.float().detach().requires_grad_(True)
    (g,) = torch.autograd.grad(_ACT[act](xr), xr, dy.float())
    return _store(out, g, out_dtype or dy.dtype, accumulate)


def glu_fwd(a, b, out=None):
    return _store(out, F.silu(a.float()) * b.float(), a.dtype)


def glu_bwd(a, b, dy, da=None, db=None):
    ar, br = a.float().detach().requires_grad_(True), b.float().detach().requires_grad_(True)
    ga, gb = torch.autograd.grad(F.silu(ar) * br, (ar, br), dy.float())
    return _store(da, ga, a.dtype), _store(db, gb, b.dtype)


def colsum(x, x2=None, out=None, group_rows=0, accumulate=False, scale=1.0, x2_bcast=False, out_cs=1, **_):
    v = x.float()
    if x2 is not None:
        v = v * (x2.float().reshape(-1, 1) if x2_bcast else x2.float())
    gr = group_rows or v.shape[0]
    s = v.view(v.shape[0] // gr, gr, v.shape[1]).sum(1) * scale
    if out is None:
        return s
    s = s.reshape(out.shape)                      # 1-D bias / strided weight-column views, [G, C] tables, [1, n] flattened tables
    out.copy_(out + s if accumulate else s)
    return out


def norm_bwd(x, dy, gamma=None, eps=1e-5, rms=False, dx=None, dx_dtype=None, accumulate=False, want_xhat=False):
    xr = x.float().detach().requires_grad_(True)
    if rms:
        xhat = xr * torch.rsqrt(xr.pow(2).mean(-1, keepdim=True) + eps)
    else:
        xhat = F.layer_norm(xr, (xr.shape[-1],), None, None, eps)
    y = xhat * gamma if gamma is not None else xhat
    (g,) = torch.autograd.grad(y, xr, dy.float())
    return _store(dx, g, dx_dtype or x.dtype, accumulate), (xhat.detach().to(BF) if want_xhat else None)


def transpose(x, pad=8, out=None):
    rows, cols = x.shape
    ldy = (rows + pad - 1) // pad * pad
    y = torch.zeros(cols, ldy, dtype=BF)
    y[:, :rows] = x.t().to(BF)
    return y


def gemm_dw_ok(dy, x, gW):
    return dy.dim() == 2 and x.dim() == 2 and gW.dim() == 2 and gW.shape[0] % 8 == 0 and gW.shape[1] % 8 == 0


def gemm_dw(dy, x, gW, gb=None):
    gW.add_(dy.to(BF).float().t() @ x.float())
    if gb is not None:
        gb.add_(dy.float().sum(0))


def sparse_rows(inp, idx, coef, out=None, accumulate=False):
    w = torch.where(idx >= 0, coef, torch.zeros_like(coef))
    s = (inp[idx.clamp_min(0).long()] * w.unsqueeze(-1)).sum(1)
    if out is None:
        return s
    out.copy_(out + s if accumulate else s)
    return out


def small_linear(x, w, bias=None, tab=None, out=None, out_dtype=F32, w_transposed=False):
    W = w.t() if w_transposed else w
    y = x.float() @ W.float().t()
    if bias is not None:
        y = y + bias
    if tab is not None:
        t = tab.reshape(-1, y.shape[1])
        y = y + t.repeat(y.shape[0] // t.shape[0], 1)
    return _store(out, y, out_dtype)


def mse_masked(pred, target, mask, T, loss_scale=1.0, want_grad=True):
    D = target.shape[1]
    m = mask.repeat_interleave(T)[:, None]
    e = pred.float()[:, :D] - target
    denom = mask.sum() * T * D
    loss = ((m * e * e).sum() / denom).view(1)
    return loss, (2.0 * m * e / denom * loss_scale if want_grad else None)


def dropout(x, p, seed, out=None, out_dtype=None, salt=None):
    assert COUNT_ONLY, "the CPU stand-in covers the eval-mode graph (dropout = 0)"
    return _store(out, x.float(), out_dtype or x.dtype)


def attention_bwd(q, k, v, o, do, scale=None, causal=False, dq=None, dk=None, dv=None, drop_p=0.0, **_):
    assert drop_p == 0.0 or COUNT_ONLY
    qr, kr, vr = (t.float().detach().requires_grad_(True) for t in (q, k, v))
    gq, gk, gv = torch.autograd.grad(_attn(qr, kr, vr, causal), (qr, kr, vr), do.float())
    return _store(dq, gq, BF), _store(dk, gk, BF), _store(dv, gv, BF)


def sumsq_parts(flat, width=1024):
    return flat.view(-1, width).pow(2).sum(0)


def adamw(p, g, m, v, lr, beta1, beta2, eps, wd, step, p_bf16=None, sumsq_parts=None, max_norm=0.0, grad_scale=1.0, norm_out=None, zero_grad=False):
    total = float(sumsq_parts.sum().sqrt()) * grad_scale if sumsq_parts is not None else 0.0
    clip = min(1.0, max_norm / (total + 1e-6)) if max_norm > 0 else 1.0
    gg = g * (grad_scale * clip)
    p.mul_(1 - lr * wd)
    m.mul_(beta1).add_(gg, alpha=1 - beta1)
    v.mul_(beta2).addcmul_(gg, gg, value=1 - beta2)
    p.addcdiv_(m, v.sqrt() / math.sqrt(1 - beta2 ** step) + eps, value=-lr / (1 - beta1 ** step))
    if p_bf16 is not None:
        p_bf16.copy_(p)
    if norm_out is not None:
        norm_out.fill_(total)
    if zero_grad:
        g.zero_()


OPS = dict(linear=linear, norm=norm, attention=attention, patchify=patchify)
TRAIN_OPS = dict(affine=affine, act_fwd=act_fwd, act_bwd=act_bwd, glu_fwd=glu_fwd, glu_bwd=glu_bwd, colsum=colsum, norm_bwd=norm_bwd, transpose=transpose,
                 sparse_rows=sparse_rows, small_linear=small_linear, mse_masked=mse_masked, dropout=dropout, attention_bwd=attention_bwd,
                 sumsq_parts=sumsq_parts, adamw=adamw, gemm_dw=gemm_dw, gemm_dw_ok=gemm_dw_ok)


def install(monkeypatch):
    """replace the kernel wrappers for the duration of a test (pytest monkeypatch restores them)."""
    from internnav_amd import ops, train_ops

    for k, fn in OPS.items():
        monkeypatch.setattr(ops, k, fn)
    for k, fn in TRAIN_OPS.items():
        monkeypatch.setattr(train_ops, k, fn)
