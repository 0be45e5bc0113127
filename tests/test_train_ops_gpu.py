"""SFT-step kernels (train.hip, attention_bwd.hip) against torch fp32 autograd of the same formulas, through the C-ABI."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


def _rel(a, b):
    return ((a.float() - b.float()).abs().max() / b.float().abs().max().clamp_min(1e-12)).item()


def test_ew_affine_act_glu(dev):
    from internnav_amd import train_ops as T

    g = torch.Generator(device="cpu").manual_seed(0)
    rows, Cd, div = 96, 384, 32
    x = torch.randn(rows, Cd, generator=g).to(dev)
    s = torch.randn(rows // div, Cd, generator=g).to(dev)
    base = torch.randn(rows, Cd, generator=g).to(dev)
    tab = torch.randn(32, Cd, generator=g).to(dev)
    for f, fn in (("id", lambda t: t), ("one_plus", lambda t: 1 + t), ("tanh", torch.tanh)):
        ref = x * fn(s).repeat_interleave(div, 0) + base + tab.repeat(rows // 32, 1)
        out = T.affine(x, scale=s, s_div=div, s_f=f, base=base, tab=tab)
        assert _rel(out, ref) < 1e-6
        outb = T.affine(x.bfloat16(), scale=s, s_div=div, s_f=f, base=base, out_dtype=torch.float32)
        assert _rel(outb, x.bfloat16().float() * fn(s).repeat_interleave(div, 0) + base) < 1e-6
    # odd widths / unaligned column views take the scalar kernel
    xv, bv = x[:, 1:7], base[:, 3:9]
    assert _rel(T.affine(xv, scale=s[:, 2:8], s_div=div, s_f="tanh", base=bv), xv * torch.tanh(s[:, 2:8]).repeat_interleave(div, 0) + bv) < 1e-6
    acc = base.clone()
    T.affine(x, out=acc, accumulate=True)
    assert _rel(acc, base + x) < 1e-6
    for name, fn in (("gelu_erf", F.gelu), ("gelu_tanh", lambda t: F.gelu(t, approximate="tanh")), ("relu", F.relu), ("silu", F.silu),
                     ("tanh", torch.tanh)):
        xr = (x * 2).clone().requires_grad_(True)
        y = fn(xr)
        dy = torch.randn_like(y)
        y.backward(dy)
        assert _rel(T.act_fwd(xr.detach(), name), y.detach()) < 2e-6, name
        assert _rel(T.act_bwd(xr.detach(), dy, name), xr.grad) < 2e-5, name
        # bf16 storage of the same values
        xb = xr.detach().bfloat16()
        assert _rel(T.act_bwd(xb, dy.bfloat16(), name, out_dtype=torch.float32), _bf_grad(fn, xb, dy.bfloat16())) < 2e-5, name
    a = x.clone().requires_grad_(True)
    b = base.clone().requires_grad_(True)
    y = F.silu(a) * b
    dy = torch.randn_like(y)
    y.backward(dy)
    assert _rel(T.glu_fwd(a.detach(), b.detach()), y.detach()) < 2e-6
    da, db = T.glu_bwd(a.detach(), b.detach(), dy)
    assert _rel(da, a.grad) < 2e-5 and _rel(db, b.grad) < 2e-5


def _bf_grad(fn, xb, dyb):
    xr = xb.float().requires_grad_(True)
    fn(xr).backward(dyb.float())
    return xr.grad


def test_colsum(dev):
    from internnav_amd import train_ops as T

    g = torch.Generator(device="cpu").manual_seed(1)
    x = torch.randn(1000, 200, generator=g).to(dev)
    x2 = torch.randn(1000, 200, generator=g).to(dev)
    assert _rel(T.colsum(x).view(-1), x.double().sum(0)) < 1e-5                    # 4 chunks -> two-stage
    assert _rel(T.colsum(x, x2).view(-1), (x.double() * x2.double()).sum(0)) < 1e-5
    out = torch.ones(200, device=dev)
    T.colsum(x.bfloat16(), x2, out=out, accumulate=True, scale=0.5)
    assert _rel(out, 1 + 0.5 * (x.bfloat16().double() * x2.double()).sum(0)) < 1e-5
    grp = T.colsum(x[:960], x2[:960], group_rows=32)
    assert grp.shape == (30, 200) and _rel(grp, (x[:960] * x2[:960]).view(30, 32, 200).double().sum(1)) < 1e-5
    # one value per row (dW[:, k] of nn.Linear(3, C)) into a strided output column
    col = torch.randn(1000, 3, generator=g).to(dev)
    W = torch.zeros(200, 3, device=dev)
    for k in range(3):
        T.colsum(x, col[:, k], out=W[:, k], x2_bcast=True, out_cs=3)
    assert _rel(W, x.double().t() @ col.double()) < 1e-5
    flat = torch.randn(5 * 1024, generator=g).to(dev)
    assert abs(T.sumsq_parts(flat).double().sum().item() - flat.double().pow(2).sum().item()) < 1e-2


@pytest.mark.parametrize("rms", [False, True])
@pytest.mark.parametrize("Cd", [384, 768, 3584])
def test_norm_bwd(dev, rms, Cd):
    from internnav_amd import train_ops as T

    g = torch.Generator(device="cpu").manual_seed(2)
    rows, eps = 70, 1e-5
    x = (torch.randn(rows, Cd, generator=g) * 2 + 0.3).to(dev).requires_grad_(True)
    gamma = (1 + 0.2 * torch.randn(Cd, generator=g)).to(dev).requires_grad_(True)
    beta = torch.randn(Cd, generator=g).to(dev).requires_grad_(True)
    if rms:
        y = x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + eps) * gamma
    else:
        y = F.layer_norm(x, (Cd,), gamma, beta, eps)
    dy = torch.randn(rows, Cd, generator=g).to(dev)
    y.backward(dy)
    dx, xhat = T.norm_bwd(x.detach(), dy, gamma.detach(), eps, rms, want_xhat=True)
    assert _rel(dx, x.grad) < 2e-5
    assert _rel(T.colsum(dy, xhat).view(-1), gamma.grad) < 1e-2          # xhat is stored in bf16
    if not rms:
        assert _rel(T.colsum(dy).view(-1), beta.grad) < 1e-5
    acc = torch.ones(rows, Cd, device=dev)
    T.norm_bwd(x.detach().bfloat16(), dy.bfloat16(), None, eps, rms, dx=acc, accumulate=True)
    xr = x.detach().bfloat16().float().requires_grad_(True)
    yr = xr * torch.rsqrt(xr.pow(2).mean(-1, keepdim=True) + eps) if rms else F.layer_norm(xr, (Cd,), None, None, eps)
    yr.backward(dy.bfloat16().float())
    assert _rel(acc, 1 + xr.grad) < 2e-5


def test_transpose_sparse_small_linear_mse(dev):
    from internnav_amd import train_ops as T

    g = torch.Generator(device="cpu").manual_seed(3)
    x = torch.randn(257, 130, generator=g).to(dev)
    y = T.transpose(x)
    assert y.shape == (130, 264) and torch.equal(y[:, :257], x.t().bfloat16()) and (y[:, 257:] == 0).all()
    yv = T.transpose(x.bfloat16()[:, 2:66])                     # strided view
    assert torch.equal(yv[:, :257], x.bfloat16()[:, 2:66].t())

    inp = torch.randn(50, 96, generator=g).to(dev)
    idx = torch.randint(-1, 50, (20, 16), generator=g).to(dev).int()
    coef = torch.randn(20, 16, generator=g).to(dev)
    ref = torch.zeros(20, 96, device=dev)
    for t in range(20):
        for j in range(16):
            if idx[t, j] >= 0:
                ref[t] += coef[t, j] * inp[idx[t, j]]
    out = T.sparse_rows(inp, idx, coef)
    assert _rel(out, ref) < 1e-5
    T.sparse_rows(inp, idx, coef, out=out, accumulate=True)
    assert _rel(out, 2 * ref) < 1e-5

    xs = torch.randn(64, 3, generator=g).to(dev)
    W = torch.randn(384, 3, generator=g).to(dev)
    b = torch.randn(384, generator=g).to(dev)
    tab = torch.randn(32, 384, generator=g).to(dev)
    assert _rel(T.small_linear(xs, W, b, tab), xs @ W.t() + b + tab.repeat(2, 1)) < 1e-5
    h = torch.randn(64, 384, generator=g).to(dev)
    W2 = torch.randn(3, 384, generator=g).to(dev)
    assert _rel(T.small_linear(h.bfloat16(), W2, b[:3]), h.bfloat16().float() @ W2.t() + b[:3]) < 1e-5
    dyy = torch.randn(64, 3, generator=g).to(dev)
    assert _rel(T.small_linear(dyy, W2, w_transposed=True), dyy @ W2) < 1e-5

    nseq, Tn, D = 6, 32, 3
    pred = torch.randn(nseq * Tn, 8, generator=g).to(dev)[:, :3].requires_grad_(True)
    tgt = torch.randn(nseq * Tn, D, generator=g).to(dev)
    mask = torch.tensor([1, 1, 0, 1, 0, 1.0], device=dev)
    loss = F.mse_loss(pred, tgt, reduction="none").view(nseq, Tn, D) * mask[:, None, None]
    loss = loss.sum() / mask.sum() / (Tn * D)
    loss.backward()
    l2, dp = T.mse_masked(pred.detach(), tgt, mask, Tn)
    assert abs(l2.item() - loss.item()) < 1e-6 * max(1, abs(loss.item()))
    assert _rel(dp, pred.grad) < 1e-5


def test_adamw_matches_torch(dev):
    from internnav_amd import train_ops as T

    g = torch.Generator(device="cpu").manual_seed(4)
    n = 3 * 1024
    p0 = torch.randn(n, generator=g).to(dev)
    ref = torch.nn.Parameter(p0.clone())
    opt = torch.optim.AdamW([ref], lr=1e-2, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01)
    p, m, v = p0.clone(), torch.zeros(n, device=dev), torch.zeros(n, device=dev)
    pb = torch.empty(n, dtype=torch.bfloat16, device=dev)
    norm = torch.zeros(1, device=dev)
    for step in range(1, 4):
        grad = (torch.randn(n, generator=g) * (3.0 if step == 2 else 0.01)).to(dev)
        ref.grad = grad.clone()
        tn = torch.nn.utils.clip_grad_norm_([ref], 1.0)
        opt.step()
        gbuf = grad.clone()
        T.adamw(p, gbuf, m, v, 1e-2, 0.9, 0.999, 1e-8, 0.01, step, p_bf16=pb, sumsq_parts=T.sumsq_parts(gbuf), max_norm=1.0,
                norm_out=norm, zero_grad=True)
        assert abs(norm.item() - tn.item()) < 1e-4 * tn.item()
        assert _rel(p, ref.detach()) < 2e-6
        assert (gbuf == 0).all() and torch.equal(pb, p.bfloat16())


@pytest.mark.parametrize("M,N,K", [(8, 3584, 3584), (8, 3584, 18944), (8, 18944, 3584), (4, 512, 1024), (12, 1000, 520)])
def test_gemm_nn(dev, M, N, K):
    from internnav_amd import train_ops as T

    g = torch.Generator(device="cpu").manual_seed(5)
    x = torch.randn(M, N, generator=g).to(dev).bfloat16()
    w = (torch.randn(N, K, generator=g) * N ** -0.5).to(dev).bfloat16()
    out = T.gemm_nn(x, w)
    ref = x.double() @ w.double()
    assert _rel(out, ref) < 1e-5
    outb = T.gemm_nn(x, w, out_dtype=torch.bfloat16)
    assert _rel(outb, ref) < 1e-2


def _attn_ref(q, k, v, do, scale, causal, k_len=None):
    """fp32 autograd of softmax(q k^T scale) v with GQA; q [B,Lq,H,D], k/v [B,Lk,Hkv,D]."""
    q, k, v = (t.float().detach().requires_grad_(True) for t in (q, k, v))
    B, Lq, H, D = q.shape
    Lk, Hkv = k.shape[1], k.shape[2]
    G = H // Hkv
    kk = k.repeat_interleave(G, dim=2)
    vv = v.repeat_interleave(G, dim=2)
    s = torch.einsum("bqhd,bkhd->bhqk", q, kk) * scale
    mask = torch.ones(B, 1, Lq, Lk, dtype=torch.bool, device=q.device)
    if causal:
        mask &= (torch.arange(Lk, device=q.device)[None, :] <= torch.arange(Lq, device=q.device)[:, None] + (Lk - Lq))[None, None]
    if k_len is not None:
        mask &= (torch.arange(Lk, device=q.device)[None, :] < k_len[:, None].to(q.device))[:, None, None, :]
    s = s.masked_fill(~mask, float("-inf"))
    o = torch.einsum("bhqk,bkhd->bqhd", torch.softmax(s, -1), vv)
    o.backward(do.float())
    return o.detach(), q.grad, k.grad, v.grad


@pytest.mark.parametrize("B,Lq,Lk,H,Hkv,D,causal", [
    (3, 257, 257, 6, 6, 64, False),     # DINOv2
    (2, 512, 512, 6, 6, 64, False),     # MemoryEncoder
    (2, 32, 512, 12, 12, 64, False),    # QFormer cross
    (5, 32, 36, 6, 6, 64, False),       # NextDiT cross
    (5, 32, 32, 6, 6, 64, False),       # NextDiT self
    (2, 100, 100, 4, 4, 64, True),
    (2, 4, 333, 28, 4, 128, True),      # LLM latent-query rows, GQA
    (1, 70, 200, 8, 2, 128, True),
    (4, 32, 32, 8, 8, 48, True),        # NavDP decoder self-attention (head dim 48 padded to 64, causal)
    (4, 32, 34, 8, 8, 48, False),       # NavDP decoder cross-attention
    (4, 1, 4, 8, 8, 48, False),         # goal compressor: one query over the 4 latent tokens
    (2, 4, 2100, 28, 4, 128, True),     # long cached prefix: the dQ pass splits the keys over workgroups (f32 atomics)
])
def test_attention_bwd(dev, B, Lq, Lk, H, Hkv, D, causal):
    from internnav_amd import ops
    from internnav_amd import train_ops as T

    g = torch.Generator(device="cpu").manual_seed(6)
    q = torch.randn(B, Lq, H, D, generator=g).to(dev).bfloat16()
    k = torch.randn(B, Lk, Hkv, D, generator=g).to(dev).bfloat16()
    v = torch.randn(B, Lk, Hkv, D, generator=g).to(dev).bfloat16()
    do = torch.randn(B, Lq, H, D, generator=g).to(dev).bfloat16()
    scale = D ** -0.5
    o = ops.attention(q, k, v, scale=scale, causal=causal)
    o_ref, dq_ref, dk_ref, dv_ref = _attn_ref(q, k, v, do, scale, causal)
    assert _rel(o, o_ref) < 2e-2
    dq, dk, dv = T.attention_bwd(q, k, v, o, do, scale=scale, causal=causal)
    G = H // Hkv
    dk = dk.float().view(B, Lk, Hkv, G, D).sum(3)
    dv = dv.float().view(B, Lk, Hkv, G, D).sum(3)
    for name, a, b in (("dq", dq, dq_ref), ("dk", dk, dk_ref), ("dv", dv, dv_ref)):
        err = (a.float() - b).abs()
        assert err.max().item() < 2.5e-2 * b.abs().max().item() and err.mean().item() < 4e-3 * b.abs().mean().item() + 1e-6, \
            f"{name}: max {err.max().item():.3e} / {b.abs().max().item():.3e}, mean {err.mean().item():.3e} / {b.abs().mean().item():.3e}"
    if causal and Lq < Lk:
        # only the last Lq key rows (the query rows themselves)
        _, dk2, dv2 = T.attention_bwd(q, k, v, o, do, scale=scale, causal=True, kv_row0=Lk - Lq)
        assert torch.equal(dk2.float().view(B, Lq, Hkv, G, D).sum(3), dk[:, Lk - Lq:])
        assert torch.equal(dv2.float().view(B, Lq, Hkv, G, D).sum(3), dv[:, Lk - Lq:])


def test_attention_bwd_packed_views(dev):
    """q / k / v as column slices of one packed [rows, 3C] projection, gradients written into a packed buffer (the nn.MultiheadAttention /
    DINOv2 qkv layout)."""
    from internnav_amd import ops
    from internnav_amd import train_ops as T

    g = torch.Generator(device="cpu").manual_seed(7)
    B, L, H, D = 2, 257, 6, 64
    Cd = H * D
    qkv = torch.randn(B * L, 3 * Cd, generator=g).to(dev).bfloat16()
    q, k, v = (qkv[:, i * Cd:(i + 1) * Cd].view(B, L, H, D) for i in range(3))
    o = ops.attention(q, k, v)
    do = torch.randn(B, L, H, D, generator=g).to(dev).bfloat16()
    dqkv = torch.empty_like(qkv)
    dq, dk, dv = (dqkv[:, i * Cd:(i + 1) * Cd].view(B, L, H, D) for i in range(3))
    T.attention_bwd(q, k, v, o, do, dq=dq, dk=dk, dv=dv)
    _, dq_ref, dk_ref, dv_ref = _attn_ref(q, k, v, do, D ** -0.5, False)
    for a, b in ((dq, dq_ref), (dk, dk_ref), (dv, dv_ref)):
        assert (a.float() - b).abs().max().item() < 2.5e-2 * b.abs().max().item()


def _fmix32(h):
    M = 0xFFFFFFFF
    h = h ^ (h >> 16)
    h = (h * 0x85EBCA6B) & M
    h = h ^ (h >> 13)
    h = (h * 0xC2B2AE35) & M
    return h ^ (h >> 16)


def _keep_mask(seed, idx, p):
    """host replica of ina_hash (csrc/common.h) on int64 tensors: keep iff hash(seed, idx) >= p * 2^32."""
    M = 0xFFFFFFFF
    lo, hi = idx & M, idx >> 32
    h = _fmix32((_fmix32(torch.full_like(lo, seed)) + 0x9E3779B9 * lo) & M)
    h = _fmix32((h + 0x9E3779B9 * hi + 0x7F4A7C15) & M)
    return h >= max(1, int(p * 4294967296.0))


def test_dropout_mask_matches_host_replica(dev):
    from internnav_amd import train_ops as T

    g = torch.Generator(device="cpu").manual_seed(8)
    rows, Cd, p, seed = 300, 384, 0.1, 12345
    x = torch.randn(rows, Cd, generator=g).to(dev)
    keep = _keep_mask(seed, torch.arange(rows * Cd, dtype=torch.int64, device=dev).view(rows, Cd), p)
    y = T.dropout(x, p, seed)
    assert torch.equal(y, torch.where(keep, x * (1.0 / (1.0 - p)), torch.zeros_like(x)))
    assert abs(keep.float().mean().item() - 0.9) < 5e-3
    yb = T.dropout(x.bfloat16()[:, 1:7], p, seed, out_dtype=torch.float32)          # scalar kernel, same indexing rule on the view
    keep6 = _keep_mask(seed, torch.arange(rows * 6, dtype=torch.int64, device=dev).view(rows, 6), p)
    assert torch.equal(yb, torch.where(keep6, x.bfloat16()[:, 1:7].float() * (1.0 / (1.0 - p)), torch.zeros(rows, 6, device=dev)))
    assert not torch.equal(T.dropout(x, p, seed + 1), y)


@pytest.mark.parametrize("B,Lq,Lk,H,D,causal", [(3, 257, 257, 6, 64, False), (2, 32, 512, 12, 64, False), (4, 32, 32, 8, 48, True), (4, 32, 34, 8, 48, False)])
def test_attention_dropout_forward_backward(dev, B, Lq, Lk, H, D, causal):
    """nn.MultiheadAttention(dropout=p) in train mode: the mask is regenerated identically by the forward and both backward passes; checked
    against fp32 autograd with the host replica of the mask."""
    from internnav_amd import ops
    from internnav_amd import train_ops as T

    g = torch.Generator(device="cpu").manual_seed(9)
    p, seed = 0.1, 777
    q, k, v, do = (torch.randn(B, L, H, D, generator=g).to(dev).bfloat16() for L in (Lq, Lk, Lk, Lq))
    o = ops.attention(q, k, v, causal=causal, drop_p=p, drop_seed=seed)
    idx = ((torch.arange(B, device=dev).view(B, 1, 1, 1) * H + torch.arange(H, device=dev).view(1, H, 1, 1)) * Lq
           + torch.arange(Lq, device=dev).view(1, 1, Lq, 1)).to(torch.int64) * Lk + torch.arange(Lk, device=dev).view(1, 1, 1, Lk)
    m = _keep_mask(seed, idx, p).float() / (1.0 - p)
    qr, kr, vr = (t.float().detach().requires_grad_(True) for t in (q, k, v))
    s = torch.einsum("bqhd,bkhd->bhqk", qr, kr) * D ** -0.5
    if causal:
        s = s.masked_fill(~(torch.arange(Lk, device=dev)[None, :] <= torch.arange(Lq, device=dev)[:, None] + (Lk - Lq)), float("-inf"))
    oref = torch.einsum("bhqk,bkhd->bqhd", torch.softmax(s, -1) * m, vr)
    oref.backward(do.float())
    assert _rel(o, oref.detach()) < 2e-2
    assert not torch.allclose(o.float(), ops.attention(q, k, v, causal=causal).float(), atol=1e-2)
    dq, dk, dv = T.attention_bwd(q, k, v, o, do, causal=causal, drop_p=p, drop_seed=seed)
    for name, a, b in (("dq", dq, qr.grad), ("dk", dk, kr.grad), ("dv", dv, vr.grad)):
        err = (a.float() - b).abs()
        assert err.max().item() < 2.5e-2 * b.abs().max().item() and err.mean().item() < 4e-3 * b.abs().mean().item() + 1e-6, name


@pytest.mark.parametrize("rows,N,K,dy32", [(768, 384, 384, True), (864, 1536, 384, False), (100, 72, 136, True), (5, 8, 8, False), (33, 64, 200, True), (2048, 128, 64, True),
                                              (768, 3072, 384, False)])
def test_gemm_dw_weight_and_bias_gradient_in_one_launch(dev, rows, N, K, dy32):
    """ina_gemm_dw (csrc/gemm_dw.hip): gW += bf16(dy)^T x, gb += sum_r dy from the row-major operands - against the fp64 product of the same
    bf16-rounded operands (tolerance: fp32 accumulation over `rows` terms) and against the launches it replaces (two transposes + the tiled GEMM +
    the column sums). Ragged tile edges, rows that are not a multiple of the 32-row slab, row-strided views, accumulation into existing values."""
    from internnav_amd import ops
    from internnav_amd import train_ops as T

    g = torch.Generator(device="cpu").manual_seed(rows * 7 + N)
    big_dy = torch.randn(rows, N + 16, generator=g).to(dev)
    big_x = torch.randn(rows, K + 8, generator=g).to(dev).bfloat16()
    dy = (big_dy if dy32 else big_dy.bfloat16())[:, 8:8 + N]               # row-strided views (lddy = N + 16, ldx = K + 8)
    x = big_x[:, :K]
    store = torch.randn(N + 8, K + 4, generator=g).to(dev)                  # gradient rows inside a larger buffer (ldw = K + 4)
    store0 = store.clone()
    gW = store[8:, :K]
    gb = torch.randn(N, generator=g).to(dev)
    gW0, gb0 = gW.clone(), gb.clone()
    assert T.gemm_dw_ok(dy, x, gW)
    T.gemm_dw(dy, x, gW, gb)
    torch.cuda.synchronize()
    ref_w = gW0.double() + dy.bfloat16().double().t() @ x.double()
    ref_b = gb0.double() + dy.double().sum(0)
    scale = (dy.bfloat16().double().abs().t() @ x.double().abs()).max().item()
    assert (gW.double() - ref_w).abs().max().item() <= 2e-6 * scale + 1e-6, ((gW.double() - ref_w).abs().max().item(), scale)
    assert (gb.double() - ref_b).abs().max().item() <= 1e-5 * dy.double().abs().sum(0).max().item() + 1e-6
    assert torch.equal(store[:8], store0[:8]) and torch.equal(store[:, K:], store0[:, K:])      # the rest of the buffer is untouched
    # the launches it replaces give the same gradient to fp32 summation order
    gW2, gb2 = gW0.clone(), gb0.clone()
    ops.linear(T.transpose(dy.bfloat16().contiguous()), T.transpose(x.contiguous()), out=gW2, residual=gW2)
    T.colsum(dy.contiguous(), out=gb2, accumulate=True)
    assert (gW - gW2).abs().max().item() <= 4e-6 * scale + 1e-6
    assert (gb - gb2).abs().max().item() <= 2e-5 * dy.double().abs().sum(0).max().item() + 1e-6
    # without a bias pointer nothing but gW moves; a shape the kernel does not take is refused by the predicate, not by a crash
    gW3 = gW0.clone()
    T.gemm_dw(dy, x, gW3)
    assert torch.equal(gW3, gW)
    assert not T.gemm_dw_ok(dy[:, :N - 4], x, gW[:N - 4]) and not T.gemm_dw_ok(dy, x.float(), gW)


@pytest.mark.parametrize("rows,N,K,splits", [(12288, 384, 384, None), (2100, 128, 72, None), (3000, 72, 136, 5), (130, 64, 64, 4), (4096, 1536, 384, None), (12288, 2304, 768, None)])
def test_gemm_dw_row_ranges_and_ordered_reduce(dev, rows, N, K, splits):
    """long reductions: the rows in `splits` ranges (grid.z), per-range tiles in the partial buffer, a second launch adding them in ascending order -
    uneven ranges, an EMPTY last range (130 rows = 3 slabs in 4 ranges), the automatic range count; same bounds as the one-launch form, and the
    result does not depend on how often it is run (no atomics)."""
    from internnav_amd import train_ops as T

    g = torch.Generator(device="cpu").manual_seed(rows + N)
    dy = torch.randn(rows, N, generator=g).to(dev)
    x = torch.randn(rows, K, generator=g).to(dev).bfloat16()
    gW0, gb0 = torch.randn(N, K, generator=g).to(dev), torch.randn(N, generator=g).to(dev)
    if splits is None:
        assert T.gemm_dw_splits(rows) > 1
    outs = []
    for _ in range(2):
        gW, gb = gW0.clone(), gb0.clone()
        T.gemm_dw(dy, x, gW, gb, splits=splits)
        torch.cuda.synchronize()
        outs.append((gW, gb))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    gW, gb = outs[0]
    ref_w = gW0.double() + dy.bfloat16().double().t() @ x.double()
    ref_b = gb0.double() + dy.double().sum(0)
    scale = (dy.bfloat16().double().abs().t() @ x.double().abs()).max().item()
    assert (gW.double() - ref_w).abs().max().item() <= 2e-6 * scale + 1e-6
    assert (gb.double() - ref_b).abs().max().item() <= 1e-5 * dy.double().abs().sum(0).max().item() + 1e-6
    assert T.gemm_dw_splits(2048) == 1 and T.gemm_dw_splits(2049) == 2 and T.gemm_dw_splits(10 ** 6) == 16
    # offered for long reductions only while the gradient has few tiles (the tiled GEMM keeps the throughput-bound shapes)
    big = torch.zeros(4096, 4096, device=dev)
    assert T.gemm_dw_ok(dy, x, gW0) and not T.gemm_dw_ok(big, torch.zeros(4096, 1024, device=dev, dtype=torch.bfloat16), torch.zeros(4096, 1024, device=dev))
    assert T.gemm_dw_ok(big[:2048], torch.zeros(2048, 1024, device=dev, dtype=torch.bfloat16), torch.zeros(4096, 1024, device=dev))
