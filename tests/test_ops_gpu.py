"""GPU numerics of the op-level kernels against a plain PyTorch fp32 reference of the same op (same bf16 inputs).

Tolerances: outputs are bf16 (8 mantissa bits): |err| <= 2^-8 * |ref| + small abs slack from fp32 accumulation order.
"""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


def _dev():
    return torch.device("cuda:0")


def _rand(shape, gen, scale=1.0, dtype=torch.bfloat16):
    return (torch.randn(shape, generator=gen, dtype=torch.float32) * scale).to(dtype).to(_dev())


def _close(out, ref, rtol=1.0 / 128, atol=2e-2):
    out = out.float().cpu()
    ref = ref.float().cpu()
    err = (out - ref).abs()
    bound = atol + rtol * ref.abs()
    bad = (err > bound).sum().item()
    assert bad == 0, f"{bad}/{err.numel()} elements out of tolerance, max err {err.max().item():.4g}, ref max {ref.abs().max().item():.4g}"


@pytest.fixture(scope="module")
def ops(built_lib):
    from internnav_amd import _lib, ops

    name = (b" " * 64)
    import ctypes

    buf = ctypes.create_string_buffer(64)
    _lib.check(_lib.lib().ina_device_check(buf, 64), "device_check")
    return ops


GEMM_SHAPES = [
    # M, N, K  (hot-path shapes: ViT-S qkv/proj/mlp, NavDP decoder, DiT, Qwen ViT/LLM slices, ragged edges)
    (257 * 3, 1152, 384), (257 * 3, 384, 1536), (768, 1536, 384), (100, 384, 384), (1, 384, 384),
    (24 * 64, 1152, 384), (4096, 3840, 1280), (1000, 1280, 3424), (920, 4608, 3584), (64, 3584, 3584),
    (130, 132, 72), (513, 68, 200), (256, 384, 592), (33, 4, 8), (2048, 2048, 2048),
]


@pytest.mark.parametrize("M,N,K", GEMM_SHAPES)
def test_gemm_plain(ops, M, N, K):
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K)
    x, w = _rand((M, K), g), _rand((N, K), g, scale=K ** -0.5)
    out = ops.linear(x, w)
    ref = x.float() @ w.float().t()
    _close(out, ref)


@pytest.mark.parametrize("cfg", [1, 2, 3, 4, 5])
def test_gemm_every_tile_config(ops, cfg):
    g = torch.Generator().manual_seed(cfg)
    M, N, K = 300, 260, 328
    x, w = _rand((M, K), g), _rand((N, K), g, scale=K ** -0.5)
    bias = torch.randn(N, generator=g).to(_dev())
    out = ops.linear(x, w, bias=bias, force_cfg=cfg)
    _close(out, x.float() @ w.float().t() + bias)


@pytest.mark.parametrize("act", ["gelu", "gelu_tanh", "relu", "silu"])
def test_gemm_epilogue_full(ops, act):
    g = torch.Generator().manual_seed(11)
    M, N, K = 771, 384, 384
    x, w = _rand((M, K), g), _rand((N, K), g, scale=K ** -0.5)
    bias = torch.randn(N, generator=g).to(_dev())
    cs = torch.randn(N, generator=g).to(_dev())
    res = _rand((M, N), g)
    rs = torch.randn((M + 2) // 3, generator=g).to(_dev())
    out = ops.linear(x, w, bias=bias, act=act, colscale=cs, residual=res, rowscale=rs, rowscale_div=3)
    y = x.float() @ w.float().t() + bias
    y = {"gelu": torch.nn.functional.gelu, "gelu_tanh": lambda t: torch.nn.functional.gelu(t, approximate="tanh"),
         "relu": torch.relu, "silu": torch.nn.functional.silu}[act](y)
    y = y * cs * rs.repeat_interleave(3)[:M, None] + res.float()
    _close(out, y)


def test_gemm_f32_out_and_f32_residual(ops):
    g = torch.Generator().manual_seed(5)
    M, N, K = 130, 1536, 384
    x, w = _rand((M, K), g), _rand((N, K), g, scale=K ** -0.5)
    res = torch.randn((M, N), generator=g).to(_dev())
    out = ops.linear(x, w, residual=res, out_dtype=torch.float32)
    assert out.dtype == torch.float32
    _close(out, x.float() @ w.float().t() + res, rtol=1e-4, atol=1e-3)


def test_gemm_glu(ops):
    g = torch.Generator().manual_seed(6)
    M, K, I = 333, 384, 1024
    x = _rand((M, K), g)
    wg, wu = _rand((I, K), g, scale=K ** -0.5), _rand((I, K), g, scale=K ** -0.5)
    bg, bu = torch.randn(I, generator=g), torch.randn(I, generator=g)
    # interleave rows in 16-row blocks: [gate16 | up16]
    w = torch.stack([wg.view(I // 16, 16, K), wu.view(I // 16, 16, K)], dim=1).reshape(2 * I, K).contiguous()
    b = torch.stack([bg.view(I // 16, 16), bu.view(I // 16, 16)], dim=1).reshape(2 * I).contiguous().to(_dev())
    out = ops.linear(x, w, bias=b, act="silu", glu=True)
    ref = torch.nn.functional.silu(x.float() @ wg.float().t() + bg.to(_dev())) * (x.float() @ wu.float().t() + bu.to(_dev()))
    _close(out, ref)


def test_gemm_strided_views(ops):
    g = torch.Generator().manual_seed(8)
    big = _rand((200, 1152), g)
    x = big[:, 384:768]  # row-strided view (lda = 1152)
    w = _rand((384, 384), g, scale=384 ** -0.5)
    outbuf = torch.zeros(200, 768, dtype=torch.bfloat16, device=_dev())
    ops.linear(x, w, out=outbuf[:, 384:])
    _close(outbuf[:, 384:], x.float() @ w.float().t())
    assert outbuf[:, :384].abs().max().item() == 0


def _ref_attn(q, k, v, scale, causal=False, kv_start=0):
    # q [B,Lq,H,D], k/v [B,Lk,Hkv,D]
    B, Lq, H, D = q.shape
    Lk, Hkv = k.shape[1], k.shape[2]
    qf = q.float().permute(0, 2, 1, 3)
    kf = k.float().permute(0, 2, 1, 3).repeat_interleave(H // Hkv, dim=1)
    vf = v.float().permute(0, 2, 1, 3).repeat_interleave(H // Hkv, dim=1)
    s = qf @ kf.transpose(-1, -2) * scale
    mask = torch.zeros(Lq, Lk, device=q.device)
    if causal:
        i = torch.arange(Lq, device=q.device)[:, None]
        j = torch.arange(Lk, device=q.device)[None, :]
        mask = mask.masked_fill(j > i + (Lk - Lq), float("-inf"))
    if kv_start:
        mask[:, :kv_start] = float("-inf")
    p = torch.softmax(s + mask, dim=-1)
    return (p @ vf).permute(0, 2, 1, 3)


ATTN_CASES = [
    # B, Lq, Lk, H, Hkv, D, causal, kv_start
    (3, 257, 257, 6, 6, 64, False, 0),      # DINOv2 ViT-S
    (5, 24, 24, 8, 8, 48, True, 0),         # NavDP decoder self-attn (T=24, causal)
    (4, 32, 32, 8, 8, 48, True, 0),         # N1 NavDP head (T=32)
    (4, 24, 132, 8, 8, 48, False, 0),       # NavDP cross-attn
    (4, 24, 132, 8, 8, 48, False, 4),       # critic memory mask (first 4 cond slots hidden)
    (2, 128, 2304, 8, 8, 48, False, 0),     # former_net cross-attn
    (4, 32, 36, 6, 6, 64, False, 0),        # NextDiT cross-attn
    (4, 32, 32, 6, 6, 64, False, 0),        # NextDiT self-attn
    (2, 32, 512, 12, 12, 64, False, 0),     # QFormer cross-attn
    (2, 784, 784, 16, 16, 80, False, 0),    # Qwen ViT full-attention block
    (1, 920, 920, 28, 4, 128, True, 0),     # LLM prefill (GQA, causal)
    (2, 4, 924, 28, 4, 128, True, 0),       # latent queries on a cached prefix
    (3, 1, 17, 8, 8, 48, False, 0),         # TokenCompressor-like single query
]


@pytest.mark.parametrize("B,Lq,Lk,H,Hkv,D,causal,kv_start", ATTN_CASES)
def test_attention_dense(ops, B, Lq, Lk, H, Hkv, D, causal, kv_start):
    g = torch.Generator().manual_seed(B * 1000 + Lq + Lk + D)
    q, k, v = _rand((B, Lq, H, D), g), _rand((B, Lk, Hkv, D), g), _rand((B, Lk, Hkv, D), g)
    scale = D ** -0.5
    out = ops.attention(q, k, v, scale=scale, causal=causal, kv_start=kv_start)
    _close(out, _ref_attn(q, k, v, scale, causal, kv_start), atol=1.5e-2)


def test_attention_packed_qkv_and_broadcast_kv(ops):
    """q/k/v as strided views of one fused qkv buffer; K/V shared by groups of 4 query batches (NavDP samples)."""
    g = torch.Generator().manual_seed(77)
    B, L, H, D = 8, 24, 8, 48
    qkv = _rand((B, L, 3 * H * D), g)
    q = qkv[..., : H * D].view(B, L, H, D)
    k = qkv[..., H * D: 2 * H * D].view(B, L, H, D)
    v = qkv[..., 2 * H * D:].view(B, L, H, D)
    out = ops.attention(q, k, v, causal=True)
    _close(out, _ref_attn(q, k, v, D ** -0.5, True), atol=1.5e-2)
    mem_k, mem_v = _rand((2, 132, H, D), g), _rand((2, 132, H, D), g)
    out2 = ops.attention(q, mem_k, mem_v, kv_bdiv=4)
    ref2 = _ref_attn(q, mem_k.repeat_interleave(4, 0), mem_v.repeat_interleave(4, 0), D ** -0.5)
    _close(out2, ref2, atol=1.5e-2)


def test_attention_varlen_windows(ops):
    """Qwen2.5-VL window attention: ragged windows (64/32/16 tokens) packed on one token axis."""
    g = torch.Generator().manual_seed(99)
    lens = [64, 64, 32, 64, 16, 32, 64]
    T, H, D = sum(lens), 16, 80
    q, k, v = _rand((T, H, D), g), _rand((T, H, D), g), _rand((T, H, D), g)
    cu = torch.tensor([0] + list(torch.tensor(lens).cumsum(0)), dtype=torch.int32, device=_dev())
    out = ops.attention(q, k, v, cu_q=cu, cu_k=cu, max_q=64, max_k=64)
    ref = torch.empty_like(out, dtype=torch.float32)
    o = 0
    for n in lens:
        ref[o:o + n] = _ref_attn(q[None, o:o + n], k[None, o:o + n], v[None, o:o + n], D ** -0.5)[0]
        o += n
    _close(out, ref, atol=1.5e-2)


def test_attention_gate_and_accumulate(ops):
    g = torch.Generator().manual_seed(3)
    B, Lq, Lk, H, D = 3, 32, 36, 6, 64
    q, k, v = _rand((B, Lq, H, D), g), _rand((B, Lk, H, D), g), _rand((B, Lk, H, D), g)
    base = _rand((B, Lq, H, D), g)
    gate = torch.randn(H, generator=g).to(_dev())
    out = base.clone()
    ops.attention(q, k, v, head_gate=gate, out=out, accumulate=True)
    ref = base.float() + _ref_attn(q, k, v, D ** -0.5) * torch.tanh(gate)[None, None, :, None]
    _close(out, ref, atol=2e-2)


@pytest.mark.parametrize("rows,C", [(1000, 384), (77, 768), (300, 1280), (50, 3584), (9, 5120), (1, 8), (333, 64), (90, 256)])
@pytest.mark.parametrize("rms", [False, True])
@pytest.mark.parametrize("xf32", [False, True])
def test_norm_plain(ops, rows, C, rms, xf32):
    g = torch.Generator().manual_seed(rows + C)
    x = _rand((rows, C), g, scale=2.0, dtype=torch.float32 if xf32 else torch.bfloat16)
    gamma, beta = torch.randn(C, generator=g).to(_dev()), torch.randn(C, generator=g).to(_dev())
    if rms:
        out = ops.norm(x, gamma=gamma, eps=1e-6, rms=True)
        xf = x.float()
        ref = xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + 1e-6) * gamma
    else:
        out = ops.norm(x, gamma=gamma, beta=beta, eps=1e-6)
        ref = torch.nn.functional.layer_norm(x.float(), (C,), gamma, beta, 1e-6)
    _close(out, ref)


def test_norm_modulation_gate_base_f32(ops):
    """NextDiT forms: norm(x) * gamma * (1 + scale[b]) and base + tanh(gate[b]) * norm(x) * gamma, f32 stream in/out."""
    g = torch.Generator().manual_seed(21)
    B, T, C = 6, 32, 384
    x = _rand((B * T, C), g, dtype=torch.float32)
    base = _rand((B * T, C), g, dtype=torch.float32)
    gamma = torch.randn(C, generator=g).to(_dev())
    emb = torch.randn(B, 4 * C, generator=g).to(_dev())
    out = ops.norm(x, gamma=gamma, eps=1e-5, rms=True, mod_scale=emb[:, :C], mod_div=T)
    nrm = x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + 1e-5) * gamma
    _close(out, nrm * (1 + emb[:, :C].repeat_interleave(T, 0)))
    out32 = torch.empty_like(x)
    outb = torch.empty(B * T, C, dtype=torch.bfloat16, device=_dev())
    ops.norm(x, gamma=gamma, eps=1e-5, rms=True, gate=emb[:, C:2 * C], base=base, mod_div=T, out=outb, out32=out32)
    ref = base + torch.tanh(emb[:, C:2 * C]).repeat_interleave(T, 0) * nrm
    _close(out32, ref, rtol=1e-5, atol=1e-5)
    _close(outb, ref)
    # in place on the f32 stream (how the engines use it)
    xin = x.clone()
    ops.norm(xin, gamma=gamma, eps=1e-5, rms=True, out32=xin)
    _close(xin, nrm, rtol=1e-5, atol=1e-5)


def test_norm_rowmaps_and_pos_table(ops):
    """final ViT norm: drop the cls token (in_map), scatter frames into a [env, slot] token buffer (out_map), add a pos table."""
    g = torch.Generator().manual_seed(5)
    n, T, C, M = 6, 257, 384, 3  # 2 envs x 3 frames
    x = _rand((n * T, C), g, dtype=torch.float32)
    gamma, beta = torch.randn(C, generator=g).to(_dev()), torch.randn(C, generator=g).to(_dev())
    pos = torch.randn(M * 256, C, generator=g).to(_dev())
    nt = (M + 1) * 256
    out = torch.zeros(2 * nt, C, dtype=torch.bfloat16, device=_dev())
    ops.norm(x, gamma, beta, eps=1e-6, out=out, rows=n * 256, in_map=(256, T, 1), out_map=(M * 256, nt, 0), pos=pos)
    ref = torch.nn.functional.layer_norm(x.view(n, T, C)[:, 1:], (C,), gamma, beta, 1e-6).reshape(2, M * 256, C) + pos
    _close(out.view(2, nt, C)[:, : M * 256], ref)
    assert out.view(2, nt, C)[:, M * 256:].abs().max().item() == 0


@pytest.mark.parametrize("C,dt", [(3, torch.float32), (1, torch.float32), (3, torch.bfloat16)])
def test_patchify(ops, C, dt):
    g = torch.Generator().manual_seed(C)
    n = 3
    img = torch.rand(n, 224, 224, C, generator=g).to(dt).to(_dev())
    mean, std = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)
    out = torch.empty(n * 256, 592, dtype=torch.bfloat16, device=_dev())
    ops.patchify(img, out, mean, std)
    x = img.float().permute(0, 3, 1, 2)
    if C == 1:
        x = x.repeat(1, 3, 1, 1)
    x = (x - torch.tensor(mean, device=_dev()).view(1, 3, 1, 1)) / torch.tensor(std, device=_dev()).view(1, 3, 1, 1)
    ref = torch.nn.functional.unfold(x, kernel_size=14, stride=14).transpose(1, 2).reshape(n * 256, 588)
    _close(out[:, :588], ref, atol=1e-2)
    assert out[:, 588:].abs().max().item() == 0


def test_embed3_and_table_fill(ops):
    g = torch.Generator().manual_seed(2)
    rows, C, T = 96, 384, 24
    x = torch.randn(rows, 3, generator=g).to(_dev())
    w, b = torch.randn(C, 3, generator=g).to(_dev()), torch.randn(C, generator=g).to(_dev())
    pos = torch.randn(T, C, generator=g).to(_dev())
    out = torch.empty(rows, C, dtype=torch.float32, device=_dev())
    ops.embed3(x, w, b, out=out, pos=pos)
    ref = x @ w.t() + b + pos.repeat(rows // T, 1)
    _close(out, ref, rtol=1e-5, atol=1e-5)
    # scatter: one vector per env written to slot 2 of a [env, 5] bf16 buffer, broadcast fill of slot 0
    buf = torch.zeros(4 * 5, C, dtype=torch.bfloat16, device=_dev())
    ops.embed3(x[:4].contiguous(), w, b, out=buf, pos=pos[1:2], rows=4, out_map=(1, 5, 2))
    ops.embed3(None, None, None, out=buf, pos=pos[3:4], rows=4, out_map=(1, 5, 0))
    _close(buf.view(4, 5, C)[:, 2], x[:4] @ w.t() + b + pos[1])
    _close(buf.view(4, 5, C)[:, 0], pos[3].expand(4, C))
    assert buf.view(4, 5, C)[:, [1, 3, 4]].abs().max().item() == 0


def test_head3_ddpm_and_euler(ops):
    g = torch.Generator().manual_seed(9)
    rows, C = 200, 384
    x = _rand((rows, C), g, dtype=torch.float32)
    gamma, beta = torch.randn(C, generator=g).to(_dev()), torch.randn(C, generator=g).to(_dev())
    w, b = (torch.randn(3, C, generator=g) * C ** -0.5).to(_dev()), torch.randn(3, generator=g).to(_dev())
    s0 = torch.randn(rows, 3, generator=g).to(_dev())
    noise = torch.randn(rows, 3, generator=g).to(_dev())
    e_ref = torch.nn.functional.layer_norm(x, (C,), gamma, beta, 1e-5) @ w.t() + b
    eps_out = torch.empty(rows, 3, device=_dev())
    ops.head3(x, w, b, gamma, beta, mode=0, eps_out=eps_out)
    _close(eps_out, e_ref, rtol=1e-4, atol=1e-4)
    coef = (1.3, 0.7, 0.4, 0.55, 0.2)
    s = s0.clone()
    ops.head3(x, w, b, gamma, beta, mode=1, sample=s, noise=noise, coef=coef, clip=1.0)
    x0 = ((s0 - coef[1] * e_ref) * coef[0]).clamp(-1, 1)
    _close(s, coef[2] * x0 + coef[3] * s0 + coef[4] * noise, rtol=1e-4, atol=1e-4)
    s = s0.clone()
    ms = torch.randn(rows // 8, C, generator=g).to(_dev())
    ops.head3(x, w, b, None, None, eps=1e-6, mode=2, sample=s, coef=(-0.1, 0, 0, 0, 0), mod_scale=ms, mod_div=8)
    e2 = (torch.nn.functional.layer_norm(x, (C,), None, None, 1e-6) * (1 + ms.repeat_interleave(8, 0))) @ w.t() + b
    _close(s, s0 - 0.1 * e2, rtol=1e-4, atol=1e-4)


def test_seqpool_head_and_select_traj(ops):
    g = torch.Generator().manual_seed(13)
    B, S, T, C = 3, 32, 24, 384
    x = _rand((B * S * T, C), g, dtype=torch.float32)
    gamma, beta = torch.randn(C, generator=g).to(_dev()), torch.randn(C, generator=g).to(_dev())
    w, b = (torch.randn(1, C, generator=g) * C ** -0.5).to(_dev()), torch.randn(1, generator=g).to(_dev())
    critic = torch.empty(B * S, device=_dev())
    ops.seqpool_head(x, T, gamma, beta, w, b, critic)
    ref = (torch.nn.functional.layer_norm(x, (C,), gamma, beta, 1e-5).view(B * S, T, C).mean(1) @ w.t())[:, 0] + b
    _close(critic, ref, rtol=1e-4, atol=1e-4)
    sample = torch.randn(B, S, T, 3, generator=g).to(_dev())
    neg, pos = torch.empty(B, 8, T, 3, device=_dev()), torch.empty(B, 8, T, 3, device=_dev())
    cv = critic.view(B, S).contiguous()
    ops.select_traj(cv, sample, neg, pos)
    traj = torch.cumsum(sample / 4.0, dim=2)
    for i in range(B):
        _close(neg[i], traj[i][cv[i].argsort()[0:8]], rtol=1e-5, atol=1e-5)
        _close(pos[i], traj[i][(-cv[i]).argsort()[0:8]], rtol=1e-5, atol=1e-5)


def test_gemm_batched_broadcast_residual(ops):
    """patch-embed form: per-frame GEMM writing rows 1.. of a [frame, 257, C] f32 buffer, + a shared [256, C] table."""
    g = torch.Generator().manual_seed(31)
    n, K, C = 5, 592, 384
    a = _rand((n, 256, K), g)
    w = _rand((C, K), g, scale=K ** -0.5)
    bias = torch.randn(C, generator=g).to(_dev())
    pos = torch.randn(256, C, generator=g).to(_dev())
    x = torch.zeros(n, 257, C, dtype=torch.float32, device=_dev())
    ops.linear(a, w, bias=bias, residual=pos, out=x[:, 1:, :], batched=True)
    _close(x[:, 1:], a.float() @ w.float().t() + bias + pos, rtol=1e-4, atol=2e-3)
    assert x[:, 0].abs().max().item() == 0


def test_bad_arguments_fail_loudly(ops):
    x = torch.zeros(4, 12, dtype=torch.bfloat16, device=_dev())  # K % 8 != 0
    w = torch.zeros(8, 12, dtype=torch.bfloat16, device=_dev())
    with pytest.raises(RuntimeError, match="multiples of 8"):
        ops.linear(x, w)
    q = torch.zeros(1, 4, 2, 40, dtype=torch.bfloat16, device=_dev())
    with pytest.raises(RuntimeError, match="unsupported head dim"):
        ops.attention(q, q, q)


@pytest.mark.parametrize("cfg", [11, 14, 18, 21, 22, 26, 27, 33, 39])
@pytest.mark.parametrize("M,N,K", [(300, 260, 320), (1000, 1280, 3456), (257, 4608, 3584), (130, 132, 64)])
def test_gemm_glds_tile_configs(ops, cfg, M, N, K):
    """LDS-DMA staged kernels: ragged M/N edges (clamped rows), swizzled LDS, every epilogue term."""
    g = torch.Generator().manual_seed(cfg + M)
    x, w = _rand((M, K), g), _rand((N, K), g, scale=K ** -0.5)
    bias = torch.randn(N, generator=g).to(_dev())
    res = torch.randn(M, N, generator=g).to(_dev())
    out = ops.linear(x, w, bias=bias, act="gelu_tanh", residual=res, out_dtype=torch.float32, force_cfg=cfg)
    ref = torch.nn.functional.gelu(x.float() @ w.float().t() + bias, approximate="tanh") + res
    _close(out, ref, rtol=2e-3, atol=5e-3)


@pytest.mark.parametrize("cfg,group_m", [(11, 3), (11, 8), (14, 8), (26, 5), (27, 0), (18, 8), (21, 3), (39, 8), (33, 3)])
def test_gemm_glds_grouped_tile_order(ops, cfg, group_m):
    """Grouped tile order (group_m row-tiles per group, ragged last group, auto rule at 0) is a pure re-ordering: same result as
    row-major, bit for bit, and equal to the fp32 reference."""
    M, N, K = 2900, 6400 + 72, 256
    g = torch.Generator().manual_seed(cfg * 31 + group_m)
    x, w = _rand((M, K), g), _rand((N, K), g, scale=K ** -0.5)
    out = ops.linear(x, w, out_dtype=torch.float32, force_cfg=cfg, group_m=group_m)
    row_major = ops.linear(x, w, out_dtype=torch.float32, force_cfg=cfg, group_m=1)
    assert torch.equal(out, row_major)
    _close(out, x.float() @ w.float().t(), rtol=2e-3, atol=5e-3)


@pytest.mark.parametrize("M,N,K,glu", [(2760, 3584, 3584, False), (3680, 4608, 3584, False), (2760, 4096, 3584, True), (9408, 1280, 3456, False), (7, 4608, 3584, False)])
def test_gemm_shared_tail_selection_is_bit_equal(ops, M, N, K, glu):
    """ops.shared_tail() (force_cfg = -1: the launch runs beside another stream's GEMMs, tile quantisation not charged) only changes WHICH
    tile kernel runs - 256 x 256 where the single-stream cost model takes 192 x 256 (e.g. 2760 x 3584 x 3584) - and every tile shape
    accumulates K in the same order: the result equals the plain auto selection bit for bit (also on the weight-streaming M <= 64 path,
    which ignores the mode). Same comparison at the full prefill shapes on the device: profiles/r03v_native_gemm_sweep.log."""
    g = torch.Generator().manual_seed(M + N)
    x, w = _rand((M, K), g), _rand((N, K), g, scale=K ** -0.5)
    res = None if glu else torch.randn(M, N, generator=g).to(_dev())
    kw = dict(act="silu", glu=True) if glu else dict(residual=res, out_dtype=torch.float32)
    plain = ops.linear(x, w, **kw)
    with ops.shared_tail():
        shared = ops.linear(x, w, **kw)
    assert torch.equal(plain, shared)
    assert torch.equal(plain, ops.linear(x, w, force_cfg=-1, **kw))


@pytest.mark.parametrize("cfg", [39, 40])
@pytest.mark.parametrize("M,N,K,mode", [(2760, 4608, 3584, "bias"), (2100, 4096, 3584, "glu"), (777, 1024, 448, "glu"), (1000, 1000, 192, "res"),
                                        (300, 264, 64, "bias"), (515, 520, 128, "plain"), (2761, 3592, 320, "bf16res"),
                                        (300, 272, 64, "bias"), (515, 528, 128, "plain"), (1000, 1008, 192, "res"), (2761, 3600, 320, "bf16res")])
def test_gemm_four_wave_tile_is_bit_equal(ops, cfg, M, N, K, mode):
    """gemm_w4.hip (tile config 39: 256 x 256 on four waves of 128 x 128, asm-threaded K loop; 40: the same tile with its B fragments
    fetched from the fragment-ordered copy of W straight into registers) accumulates K in the order of the 8-wave tiles: bit-equal to the
    ping-pong kernel (cfg 18) under every epilogue, with ragged edges and 1 / 2 / 3 / 5 / 7 / 56 K stages (the loop requests two stages
    ahead: the last two stages take the path that requests nothing; cfg 40 peels its first stage as well)."""
    g = torch.Generator().manual_seed(M + K)
    x, w = _rand((M, K), g), _rand((N, K), g, scale=K ** -0.5)
    kw = {}
    if mode == "bias":
        kw = dict(bias=torch.randn(N, generator=g).to(_dev()))
    elif mode == "glu":
        kw = dict(act="silu", glu=True)
    elif mode == "res":
        kw = dict(residual=torch.randn(M, N, generator=g).to(_dev()), out_dtype=torch.float32, act="gelu_tanh", bias=torch.randn(N, generator=g).to(_dev()))
    elif mode == "bf16res":
        kw = dict(residual=_rand((M, N), g))
    ref = ops.linear(x, w, force_cfg=18, **kw)
    if cfg == 40:
        if N % 16:
            with pytest.raises(RuntimeError, match="N % 16"):
                ops.gemm_preshuffle(w)
            return
        kw["w_frag"] = ops.gemm_preshuffle(w)
    out = ops.linear(x, w, force_cfg=cfg, **kw)
    torch.cuda.synchronize()
    assert torch.equal(ref, out)
    if mode in ("bias", "plain"):
        _close(out, x.float() @ w.float().t() + (kw["bias"] if "bias" in kw else 0.0))


def test_gemm_preshuffle_layout_and_auto_selection(ops):
    """ina_gemm_preshuffle: fragment (n // 16, k // 32) is one contiguous KiB, lane (k % 32 // 8) * 16 + n % 16 holds 8 consecutive k; with the
    fragment-ordered copy at hand the auto selection runs config 40 exactly where it would run 39 (and the result does not change by a bit)."""
    from internnav_amd import _lib
    import ctypes as C

    g = torch.Generator().manual_seed(40)
    N, K = 4608, 3584
    w = _rand((N + 3, K + 8), g)[:N, :K]                                 # row-strided source
    wf = ops.gemm_preshuffle(w)
    ref = w.reshape(N // 16, 16, K // 32, 4, 8).permute(0, 2, 3, 1, 4).reshape(-1)
    assert torch.equal(wf, ref)
    x = _rand((2760, K), g)
    bias = torch.randn(N, generator=g).to(_dev())
    a = _lib.GemmArgs()
    a.A, a.W, a.C, a.M, a.N, a.K, a.lda, a.ldw, a.ldc, a.batch = x.data_ptr(), w.data_ptr(), x.data_ptr(), 2760, N, K, K, w.stride(0), N, 1
    a.out_dtype = 0
    k = C.c_int(0)
    _lib.check(_lib.lib().ina_gemm_select(C.byref(a), C.byref(k)), "select")
    assert k.value == 39
    a.Wp = wf.data_ptr()
    _lib.check(_lib.lib().ina_gemm_select(C.byref(a), C.byref(k)), "select")
    assert k.value == 40
    assert torch.equal(ops.linear(x, w, bias=bias), ops.linear(x, w, bias=bias, w_frag=wf))
    wc = w.contiguous()
    small = _rand((64, K), g)
    assert torch.equal(ops.linear(small, wc), ops.linear(small, wc, w_frag=wf))     # any other tile ignores the copy


@pytest.mark.parametrize("M,N,K,glu", [(3680, 37888, 3584, True), (8192, 16384, 3584, False), (2760, 3584, 18944, False)])
def test_gemm_frag_weights_repeatable_at_prefill_size(ops, M, N, K, glu):
    """config 40 at the sizes where its loads miss in L2 (hundreds of column tiles, nine rounds of workgroups): 25 launches, every one bit-equal
    to config 39. Regression test of an intermittent failure of the first version: reloads still in flight behind the loop landed in registers
    the epilogue already used - whole tiles wrong, only when the late loads had missed in L2, never at the small shapes of the test above."""
    g = torch.Generator().manual_seed(N)
    x, w = _rand((M, K), g), _rand((N, K), g, scale=K ** -0.5)
    wf = ops.gemm_preshuffle(w)
    kw = dict(act="silu", glu=True) if glu else {}
    ref = ops.linear(x, w, force_cfg=39, **kw)
    out = torch.empty_like(ref)
    for rep in range(25):
        out.zero_()
        ops.linear(x, w, force_cfg=40, w_frag=wf, out=out, **kw)
        torch.cuda.synchronize()
        assert torch.equal(out, ref), f"launch {rep}: {int((out != ref).sum())} elements differ"


def test_gemm_four_wave_tile_rejects_unaligned_rows(ops):
    x, w = torch.zeros(300, 64, dtype=torch.bfloat16, device=_dev()), torch.zeros(260, 64, dtype=torch.bfloat16, device=_dev())
    with pytest.raises(RuntimeError, match="39 / 40"):
        ops.linear(x, w, force_cfg=39)          # bf16 rows of 520 bytes
    with pytest.raises(RuntimeError, match="fragment-ordered copy"):
        ops.linear(x[:, :64], torch.zeros(256, 64, dtype=torch.bfloat16, device=_dev()), force_cfg=40)   # no w_frag


SKINNY = [(7, 4608, 3584), (1, 3584, 18944), (35, 3584, 3584), (64, 18816, 384), (5, 1000, 1176), (16, 256, 128), (48, 152064, 256)]


@pytest.mark.parametrize("cfg", [0, 32])
@pytest.mark.parametrize("M,N,K", SKINNY)
def test_gemm_skinny_weight_streaming(ops, M, N, K, cfg):
    """M <= 64 goes through the weight-streaming kernels (0 = auto = 32: column-owner kernel with fused epilogue): all epilogue terms, bf16 and
    f32 outputs."""
    g = torch.Generator().manual_seed(M + N + K)
    x, w = _rand((M, K), g), _rand((N, K), g, scale=K ** -0.5)
    bias, cs = torch.randn(N, generator=g).to(_dev()), torch.randn(N, generator=g).to(_dev())
    res = torch.randn(M, N, generator=g).to(_dev())
    ref = torch.nn.functional.gelu(x.float() @ w.float().t() + bias) * cs + res
    out = ops.linear(x, w, bias=bias, act="gelu", colscale=cs, residual=res, out_dtype=torch.float32, force_cfg=cfg)
    _close(out, ref, rtol=2e-3, atol=5e-3)
    out_b = ops.linear(x, w, bias=bias, force_cfg=cfg)
    _close(out_b, x.float() @ w.float().t() + bias)
    same = ops.linear(x, w, bias=bias, force_cfg=3 if N >= 128 else 4)  # tiled kernel on the same problem
    _close(same, out_b.float(), rtol=1.0 / 64, atol=3e-2)


@pytest.mark.parametrize("cfg,M", [(0, 7), (32, 7), (32, 35), (32, 64)])
def test_gemm_skinny_glu(ops, cfg, M):
    g = torch.Generator().manual_seed(66)
    K, I = 3584, 2048
    x = _rand((M, K), g)
    wg, wu = _rand((I, K), g, scale=K ** -0.5), _rand((I, K), g, scale=K ** -0.5)
    w = torch.stack([wg.view(I // 16, 16, K), wu.view(I // 16, 16, K)], dim=1).reshape(2 * I, K).contiguous()
    out = ops.linear(x, w, act="silu", glu=True, force_cfg=cfg)
    ref = torch.nn.functional.silu(x.float() @ wg.float().t()) * (x.float() @ wu.float().t())
    _close(out, ref)


@pytest.mark.parametrize("M,N,K,glu,xdt", [(7, 4608, 3584, False, torch.float32), (7, 37888, 3584, True, torch.float32), (1, 4608, 3584, False, torch.bfloat16),
                                            (16, 512, 1024, False, torch.float32), (6, 2048, 384, True, torch.bfloat16), (7, 16384, 3584, False, torch.float32)])
def test_gemm_skinny_fused_input_rmsnorm(ops, M, N, K, glu, xdt):
    """decode passes: RMSNorm fused in front of the weight-streaming GEMM (one launch instead of norm + GEMM). Against the fp32 formula
    and against the two-launch path it replaces (same bf16 operand up to the rounding of the row statistic's summation order)."""
    g = torch.Generator().manual_seed(M * 31 + N)
    x = (torch.randn(M, K, generator=g) * 3.0).to(xdt).to(_dev())
    gamma = (1.0 + 0.1 * torch.randn(K, generator=g)).to(_dev())
    w = _rand((N, K), g, scale=K ** -0.5)
    bias = None if glu else torch.randn(N, generator=g).to(_dev())
    res = None if glu else torch.randn(M, N, generator=g).to(_dev())
    xn = x.float() * torch.rsqrt(x.float().pow(2).mean(-1, keepdim=True) + 1e-6) * gamma
    y = xn.bfloat16().float() @ w.float().t()
    if glu:
        I = N // 2
        y4 = y.view(M, I // 16, 2, 16)
        ref = (torch.nn.functional.silu(y4[:, :, 0]) * y4[:, :, 1]).reshape(M, I)
        out = ops.linear(x, w, act="silu", glu=True, prenorm=(gamma, 1e-6))
        h = ops.norm(x, gamma, None, eps=1e-6, rms=True)
        two = ops.linear(h, w, act="silu", glu=True)
    else:
        ref = y + bias + res
        out = ops.linear(x, w, bias=bias, residual=res, out_dtype=torch.float32, prenorm=(gamma, 1e-6))
        h = ops.norm(x, gamma, None, eps=1e-6, rms=True)
        two = ops.linear(h, w, bias=bias, residual=res, out_dtype=torch.float32)
    _close(out, ref, rtol=4e-3, atol=1e-2)
    d = (out.float() - two.float()).abs().max().item()
    assert d <= 2e-2, d                      # identical up to a bf16 flip of single operand elements where the row statistic rounds differently
    with pytest.raises(Exception):
        ops.linear(torch.zeros(17, K, device=_dev()), w, prenorm=(gamma, 1e-6))      # built for <= 16 rows


@pytest.mark.parametrize("B,Lq,Lk", [(7, 1, 921), (1, 1, 300), (16, 1, 1024), (2, 4, 500)])
def test_decode_attention_with_rope_and_kv_append(ops, B, Lq, Lk):
    """single-token decoder passes: the attention launch rotates q, rotates the new tokens' keys into the cache, copies their values there and
    attends (ina_attn_args.rope_cos) - outputs and cache rows bit-equal to rope launch (ops.rope kv_out) + attention launch."""
    nh, nkv, hd, Smax = 28, 4, 128, 1100
    g = torch.Generator().manual_seed(B * 100 + Lq + Lk)
    qkv = _rand((B * Lq, (nh + 2 * nkv) * hd), g)
    cache = _rand((B * Smax, 2 * nkv * hd), g)
    ang = torch.rand(B * Lq, hd // 2, generator=g) * 6.28
    cos, sin = torch.cat([ang.cos(), ang.cos()], 1).to(_dev()).contiguous(), torch.cat([ang.sin(), ang.sin()], 1).to(_dev()).contiguous()
    dst = (torch.arange(B)[:, None] * Smax + (Lk - Lq) + torch.arange(Lq)[None]).reshape(-1).to(torch.int32).to(_dev())
    # two launches
    qkv_a, cache_a = qkv.clone(), cache.clone()
    ops.rope(qkv_a, cos, sin, heads=nh + nkv, D=hd, col0=0, rows=B * Lq, kv_out=cache_a, kv_dst=dst, kv_head0=nh, v_heads=nkv)
    kv_a = cache_a.view(B, Smax, 2, nkv, hd)[:, :Lk]
    out_a = ops.attention(qkv_a[:, : nh * hd].view(B, Lq, nh, hd), kv_a[:, :, 0], kv_a[:, :, 1], causal=True)
    # one launch
    assert ops.attention_rope_ok(Lq, Lk, nh, nkv, hd)
    qkv_b, cache_b = qkv.clone(), cache.clone()
    kv_b = cache_b.view(B, Smax, 2, nkv, hd)[:, :Lk]
    out_b = ops.attention(qkv_b[:, : nh * hd].view(B, Lq, nh, hd), kv_b[:, :, 0], kv_b[:, :, 1], causal=True,
                          rope=(cos, sin, qkv_b[:, nh * hd:(nh + nkv) * hd].view(B, Lq, nkv, hd), qkv_b[:, (nh + nkv) * hd:].view(B, Lq, nkv, hd)))
    torch.cuda.synchronize()
    assert torch.equal(cache_b, cache_a), f"cache rows differ in {(cache_b != cache_a).sum().item()} elements"
    assert torch.equal(out_b, out_a), (out_b.float() - out_a.float()).abs().max().item()
    assert torch.equal(qkv_b, qkv)                                       # the projection buffer is read only
    with pytest.raises(Exception, match="rope"):                         # a shape that is not the one-launch decode kernel's: refused, not ignored
        ops.attention(qkv_b[:, : nh * hd].view(B, Lq, nh, hd), kv_b[:, :100, 0], kv_b[:, :100, 1], causal=True,
                      rope=(cos, sin, qkv_b[:, nh * hd:(nh + nkv) * hd].view(B, Lq, nkv, hd), qkv_b[:, (nh + nkv) * hd:].view(B, Lq, nkv, hd)))


DECODE = [
    # B, Lq, Lk, H, Hkv, D, causal
    (7, 1, 927, 28, 4, 128, True), (7, 5, 932, 28, 4, 128, True), (3, 1, 300, 28, 4, 128, False), (2, 2, 513, 8, 8, 64, True),
    (4, 1, 1024, 16, 16, 80, True), (1, 6, 256, 28, 4, 128, True), (2, 1, 1100, 28, 4, 128, True), (2, 5, 2340, 28, 4, 128, True),     # beyond 1024 keys: the split path either way
]


@pytest.mark.parametrize("B,Lq,Lk,H,Hkv,D,causal", DECODE)
def test_attention_decode_gqa_split_kv(ops, B, Lq, Lk, H, Hkv, D, causal):
    g = torch.Generator().manual_seed(B * 100 + Lq + Lk)
    q, k, v = _rand((B, Lq, H, D), g), _rand((B, Lk, Hkv, D), g), _rand((B, Lk, Hkv, D), g)
    out = ops.attention(q, k, v, causal=causal)                 # Lk <= 1024: the one-launch kernel (4 waves walk the chunks, partials meet in LDS)
    ref = _ref_attn(q, k, v, D ** -0.5, causal)
    _close(out, ref, atol=1.5e-2)
    two = ops.attention(q, k, v, causal=causal, kernel=1)       # the split + combine pair it replaces
    _close(two, ref, atol=1.5e-2)
    assert (out.float() - two.float()).abs().max().item() <= 1.6e-2        # same chunks, another merge order: bf16 rounding of the outputs
    if D == 128 and Lk <= 1024:
        # d = 128 now runs the 8-wave kernel (K fragments straight from global memory, V^T image in two halves: round 5); kernel = 3 pins the
        # 4-wave kernel it replaced - same chunk arithmetic, eight partials instead of four
        four = ops.attention(q, k, v, causal=causal, kernel=3)
        _close(four, ref, atol=1.5e-2)
        assert (out.float() - four.float()).abs().max().item() <= 1.6e-2
    # per-sequence key lengths (ragged answers): keys beyond k_len[b] are ignored, causal offset follows k_len
    lens = torch.tensor([Lk - 3 * (i % 4) for i in range(B)], dtype=torch.int32)
    out2 = ops.attention(q, k, v, causal=causal, k_len=lens.to(_dev()))
    for b in range(B):
        n = int(lens[b])
        _close(out2[b:b + 1], _ref_attn(q[b:b + 1], k[b:b + 1, :n], v[b:b + 1, :n], D ** -0.5, causal), atol=1.5e-2)


def test_argmax_rows_first_maximum(ops):
    g = torch.Generator().manual_seed(12)
    x = torch.randn(7, 152064, generator=g).to(_dev())
    x[2, 777] = x[2, 90000] = 50.0   # tie: the first maximum wins (torch.argmax semantics used by greedy decoding)
    x[5, 152063] = 60.0
    out = torch.empty(7, dtype=torch.int32, device=_dev())
    ops.argmax_rows(x, out)
    assert out.tolist() == x.argmax(-1).tolist() and out[2].item() == 777 and out[5].item() == 152063
    y = torch.randn(3, 1001, generator=g).to(_dev())[:, 1:]   # unaligned rows, odd length
    o2 = torch.empty(3, dtype=torch.int32, device=_dev())
    ops.argmax_rows(y, o2)
    assert o2.tolist() == y.argmax(-1).tolist()


@pytest.mark.parametrize("envs,S,T,Lz", [(2, 5, 32, 36), (1, 3, 20, 64), (3, 1, 32, 5), (2, 32, 32, 36), (2, 4, 32, 20), (1, 2, 7, 48)])
def test_dit_attention_fused_qknorm_self_cross(ops, envs, S, T, Lz):
    """NextDiT attention stage in one launch vs the unfused fp32 formula: LayerNorm across heads on q1 / k1 / q2, self-attention
    inside each T-token sequence, tanh-gated cross-attention against the env's Lz condition rows; and vs the unfused op sequence
    (3 norm launches + 2 attention launches) it replaces in the engine."""
    heads, D = 6, 384
    g = torch.Generator().manual_seed(envs * 100 + S * 10 + T + Lz)
    nseq = envs * S
    x = _rand((nseq * T, 4 * D), g)
    kv2 = _rand((envs, Lz, 2, heads, 64), g)
    norms = [(1.0 + 0.2 * torch.randn(D, generator=g)).to(_dev()) for _ in range(3)]
    biases = [(0.2 * torch.randn(D, generator=g)).to(_dev()) for _ in range(3)]
    gate = torch.randn(heads, generator=g).to(_dev())
    v2t = torch.empty(envs, heads, 64, 64, dtype=torch.bfloat16, device=_dev())
    ops.dit_v2t(kv2, heads, v2t)
    out = torch.empty(nseq * T, D, dtype=torch.bfloat16, device=_dev())
    ops.dit_attention(x, out, list(zip(norms, biases)), kv2, v2t, gate, T=T, seq_per_env=S, heads=heads)

    xf = x.float().view(nseq, T, 4, D)
    ln = lambda t, i: torch.nn.functional.layer_norm(t, (D,), norms[i], biases[i], 1e-5).to(torch.bfloat16).float()
    q1, k1, v1, q2 = ln(xf[:, :, 0], 0), ln(xf[:, :, 1], 1), xf[:, :, 2], ln(xf[:, :, 3], 2)
    hv = lambda t: t.view(t.shape[0], t.shape[1], heads, 64).transpose(1, 2)
    sdpa = torch.nn.functional.scaled_dot_product_attention
    o1 = sdpa(hv(q1), hv(k1), hv(v1)).transpose(1, 2).reshape(nseq, T, D)
    kc = kv2[:, :, 0].float().repeat_interleave(S, 0).transpose(1, 2)
    vc = kv2[:, :, 1].float().repeat_interleave(S, 0).transpose(1, 2)
    o2 = sdpa(hv(q2), kc, vc).transpose(1, 2).reshape(nseq, T, heads, 64) * torch.tanh(gate).view(1, 1, heads, 1)
    ref = (o1 + o2.reshape(nseq, T, D)).view(nseq * T, D)
    _close(out, ref, atol=2e-2)

    # the unfused launch sequence
    seg = x.clone().view(nseq * T * 4, D)
    for j, i in ((0, 0), (1, 1), (3, 2)):
        ops.norm(seg, norms[i], biases[i], eps=1e-5, out=seg, rows=nseq * T, in_map=(1, 4, j), out_map=(1, 4, j))
    q5 = seg.view(nseq, T, 4, heads, 64)
    un = torch.empty_like(out)
    ops.attention(q5[:, :, 0], q5[:, :, 1], q5[:, :, 2], out=un.view(nseq, T, heads, 64))
    ops.attention(seg.view(envs, S * T, 4, heads, 64)[:, :, 3], kv2[:, :, 0], kv2[:, :, 1], head_gate=gate,
                  out=un.view(envs, S * T, heads, 64), accumulate=True)
    _close(out, un, atol=8e-3, rtol=1.0 / 128)

    # the statistics handed over by the producer of the rows (ina_dit_rowchain's seg_stats): the one-pass kernel without its own statistics pass
    seg4 = x.float().view(nseq * T, 4, D)
    mean = seg4.mean(-1)
    stats = torch.stack([mean, torch.rsqrt(seg4.var(-1, unbiased=False) + 1e-5)], -1).contiguous()
    stats[:, 2] = float("nan")                                           # the v1 segment is not normalised: its pair must never be read
    out_s = torch.full_like(out, 7.0)
    ops.dit_attention(x, out_s, list(zip(norms, biases)), kv2, v2t, gate, T=T, seq_per_env=S, heads=heads, stats=stats)
    torch.cuda.synchronize()
    _close(out_s, ref, atol=2e-2)
    _close(out_s, out, atol=8e-3, rtol=1.0 / 128)


@pytest.mark.parametrize("rms,C", [(True, 384), (False, 384), (True, 1024), (True, 96)])
def test_norm_chained_prenorm(ops, rms, C):
    """norm launch with a chained second norm on the produced residual row == the two separate launches it replaces."""
    g = torch.Generator().manual_seed(C + int(rms))
    rows, div = 300, 60
    xin = _rand((rows, C), g)
    base = torch.randn(rows, C, generator=g).to(_dev())
    g1, g2 = [(1.0 + 0.1 * torch.randn(C, generator=g)).to(_dev()) for _ in range(2)]
    mod = (0.3 * torch.randn(rows // div, 3 * C, generator=g)).to(_dev())
    gate, ms2 = mod[:, :C], mod[:, 2 * C:]
    x_a, h_a = torch.empty(rows, C, device=_dev()), torch.empty(rows, C, dtype=torch.bfloat16, device=_dev())
    ops.norm(xin, g1, None, eps=1e-5, rms=rms, gate=gate, base=base, mod_div=div, out32=x_a, out2=h_a, gamma2=g2, mod_scale2=ms2)
    x_b = torch.empty_like(x_a)
    ops.norm(xin, g1, None, eps=1e-5, rms=rms, gate=gate, base=base, mod_div=div, out32=x_b)
    h_b = ops.norm(x_b, g2, None, eps=1e-5, rms=rms, mod_scale=ms2, mod_div=div)
    assert torch.equal(x_a, x_b)
    _close(h_a, h_b, atol=1e-2, rtol=1.0 / 128)
    # in place on the base buffer (how the DiT block uses it)
    x_c = base.clone()
    ops.norm(xin, g1, None, eps=1e-5, rms=rms, gate=gate, base=x_c, mod_div=div, out32=x_c, out2=h_a, gamma2=g2, mod_scale2=ms2)
    assert torch.equal(x_c, x_b)


def test_workspace_growth_never_frees_a_buffer_a_graph_has_seen(ops):
    """library hardening: a hipGraph captured under a workspace slot keeps the scratch pointer it was captured with.
    When a later eager launch under the same slot needs more scratch, the old buffer must stay alive (retired, not freed): the graph
    replays into valid memory and reproduces its result."""
    from internnav_amd import _lib

    lib = _lib.lib()
    _lib.check(lib.ina_set_workspace_slot(5), "slot")
    g = torch.Generator().manual_seed(11)
    B, H, Hkv, D, Lk = 2, 28, 4, 128, 512
    q = _rand((B, 1, H, D), g)
    k, v = _rand((B, Lk, Hkv, D), g), _rand((B, Lk, Hkv, D), g)
    out = torch.zeros(B, 1, H, D, dtype=torch.bfloat16, device=_dev())
    klen = torch.full((B,), Lk, dtype=torch.int32, device=_dev())
    ops.attention(q, k, v, causal=True, out=out, k_len=klen, kernel=1)           # eager once: the slot's scratch exists before capture (kernel = 1: the split + combine pair, the one-launch kernel needs no scratch)
    ref = out.clone()
    retired0 = lib.ina_workspace_retired()
    graph = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    import gc
    gc.disable()                                                                 # (no collection of older graph objects inside a capture: runtime.GraphedCall)
    try:
        with torch.cuda.stream(s):
            with torch.cuda.graph(graph, stream=s):
                ops.attention(q, k, v, causal=True, out=out, k_len=klen, kernel=1)
    finally:
        gc.enable()
    torch.cuda.current_stream().wait_stream(s)
    # a much larger decode-attention launch under the same slot: needs > 32 MiB of flash-decoding partials -> the slot grows
    B2, Lk2 = 48, 8192
    q2 = _rand((B2, 1, H, D), g)
    k2, v2 = _rand((B2, Lk2, Hkv, D), g), _rand((B2, Lk2, Hkv, D), g)
    ops.attention(q2, k2, v2, causal=True, k_len=torch.full((B2,), Lk2, dtype=torch.int32, device=_dev()))
    torch.cuda.synchronize()
    assert lib.ina_workspace_retired() == retired0 + 1                          # the captured buffer was retired, not freed
    out.zero_()
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, ref)
    _lib.check(lib.ina_set_workspace_slot(0), "slot")


def test_rope_with_fused_kv_append_equals_rope_then_gather(ops):
    """decoder-layer fusion: rope(q, k) + KV-cache append (rotated k | raw v -> cache rows) in one launch == the two launches it replaces,
    bit for bit (q rotated in place, the k / v columns of the projection row are left untouched)."""
    g = torch.Generator().manual_seed(5)
    rows, nh, nkv, D = 37, 28, 4, 128
    qkv = _rand((rows, (nh + 2 * nkv) * D), g)
    cos, sin = torch.randn(rows, D, generator=g).to(_dev()), torch.randn(rows, D, generator=g).to(_dev())
    dst = torch.randperm(64, generator=g)[:rows].to(torch.int32).to(_dev())
    a = qkv.clone()
    cache_a = torch.zeros(64, 2 * nkv * D, dtype=torch.bfloat16, device=_dev())
    ops.rope(a, cos, sin, heads=nh + nkv, D=D, rows=rows)
    ops.gather_rows(a[:, nh * D:], cache_a, dst=dst)
    b = qkv.clone()
    cache_b = torch.zeros(64, 2 * nkv * D, dtype=torch.bfloat16, device=_dev())
    ops.rope(b, cos, sin, heads=nh + nkv, D=D, rows=rows, kv_out=cache_b, kv_dst=dst, kv_head0=nh, v_heads=nkv)
    assert torch.equal(cache_a, cache_b)
    assert torch.equal(a[:, : nh * D], b[:, : nh * D])                     # rotated queries
    assert torch.equal(b[:, nh * D:], qkv[:, nh * D:])                      # k / v columns untouched in the fused form


@pytest.mark.parametrize("cfg", [34, 35])
@pytest.mark.parametrize("M,N,glu", [(32, 128, False), (7168, 1536, False), (8224, 2048, True), (4128, 256, True), (65536, 1536, False)])
def test_gemm_rowpanel_k384(ops, cfg, M, N, glu):
    """row-panel kernels of the d = 384 heads (csrc/gemm_rowpanel.hip: activations as register-resident MFMA fragments, W streamed through an
    LDS ring, 32x32x16 MFMAs): against the fp32 product of the bf16 operands and against the tiled kernel they replace, on row counts with a
    partial last workgroup (M % 256 != 0) and on strided views of wider buffers (the engine's fused projection / activation buffers)."""
    g = torch.Generator().manual_seed(M + N + cfg)
    K = 384
    xw = _rand((M, K + 64), g)                  # row-strided A
    x = xw[:, :K]
    w = _rand((N, K), g, scale=K ** -0.5)
    n_out = N // 2 if glu else N
    outw = torch.zeros(M, n_out + 8, dtype=torch.bfloat16, device=_dev())
    out = outw[:, :n_out]
    if glu:
        ops.linear(x, w, act="silu", glu=True, out=out, force_cfg=cfg)
        tiled = ops.linear(x, w, act="silu", glu=True, force_cfg=22)
        wg = w.view(N // 32, 2, 16, K)[:, 0].reshape(N // 2, K)
        wu = w.view(N // 32, 2, 16, K)[:, 1].reshape(N // 2, K)
        ref = torch.nn.functional.silu(x.float() @ wg.float().t()) * (x.float() @ wu.float().t())
    else:
        ops.linear(x, w, out=out, force_cfg=cfg)
        tiled = ops.linear(x, w, force_cfg=22)
        ref = x.float() @ w.float().t()
    torch.cuda.synchronize()
    _close(out, ref)
    assert float(outw[:, n_out:].abs().max()) == 0.0, "columns beyond the output were written"
    # same bf16 result as the tiled kernel up to the rounding of a different K summation order (16- vs 32-wide MFMA steps)
    d = (out.float() - tiled.float()).abs()
    assert d.max().item() <= 2.0 ** -7 * max(1.0, ref.abs().max().item()) and (d > 0).float().mean().item() < 0.2
    if not glu and N % 384 == 0:
        # LayerNorm statistics of the 384-wide segments of the produced rows (ina_gemm_args.seg_stats: what dit_attention(stats=) consumes) - the
        # output itself must not change, the statistics are those of the fp32 rows
        st = torch.full((M, N // 384, 2), float("nan"), device=_dev())
        out2 = torch.zeros_like(outw)
        ops.linear(x, w, out=out2[:, :n_out], force_cfg=cfg, seg_stats=(st, 1e-5))
        torch.cuda.synchronize()
        assert torch.equal(out2, outw)
        seg = ref.view(M, N // 384, 384)
        _close(st[..., 0], seg.mean(-1), rtol=1e-3, atol=2e-3)
        _close(st[..., 1], torch.rsqrt(seg.var(-1, unbiased=False) + 1e-5), rtol=2e-3, atol=1e-3)
        with pytest.raises(Exception, match="seg_stats"):
            ops.linear(x, w, force_cfg=22, seg_stats=(st, 1e-5))              # any other tile: refused, not ignored


def test_gemm_rowpanel_rejects_what_it_does_not_compute(ops):
    g = torch.Generator().manual_seed(3)
    x, w = _rand((64, 384), g), _rand((256, 384), g)
    bias = torch.zeros(256, device=_dev())
    res = torch.zeros(64, 256, device=_dev())
    for kw in (dict(out_dtype=torch.float32), dict(colscale=bias), dict(residual=res, out_dtype=torch.float32), dict(bias=bias, act="silu", glu=True)):
        with pytest.raises(Exception):
            ops.linear(x, w, force_cfg=34, **kw)
    with pytest.raises(Exception):
        ops.linear(_rand((48, 384), g)[:40], w, force_cfg=35)            # M % 32 != 0
    with pytest.raises(Exception):
        ops.linear(_rand((64, 512), g), _rand((256, 512), g), force_cfg=34)   # K != 384


@pytest.mark.parametrize("cfg", [34, 35])
@pytest.mark.parametrize("M,N,act", [(49152, 1152, None), (49152, 1536, "gelu"), (2080, 1024, "relu"), (64, 128, "gelu_tanh")])
def test_gemm_rowpanel_bias_activation_epilogue(ops, cfg, M, N, act):
    """the biased projections of the nn.Transformer layers at d = 384 (NavDP decoder q|k|v N = 1152, FFN N = 1536 + GELU) on the row-panel
    kernels: bias from an LDS copy, activation in registers; bit-equal to the tiled kernel's epilogue (same K order, same fp32 epilogue math)."""
    g = torch.Generator().manual_seed(M + N)
    x, w = _rand((M, 384), g), _rand((N, 384), g, scale=384 ** -0.5)
    bias = torch.randn(N, generator=g).to(_dev())
    out = ops.linear(x, w, bias=bias, act=act, force_cfg=cfg)
    tiled = ops.linear(x, w, bias=bias, act=act, force_cfg=22)
    y = x.float() @ w.float().t() + bias
    ref = {None: lambda t: t, "gelu": torch.nn.functional.gelu, "relu": torch.relu, "gelu_tanh": lambda t: torch.nn.functional.gelu(t, approximate="tanh")}[act](y)
    _close(out, ref)
    d = (out.float() - tiled.float()).abs()
    assert d.max().item() <= 2.0 ** -7 * max(1.0, ref.abs().max().item()) and (d > 0).float().mean().item() < 0.05


@pytest.mark.parametrize("M,K1,N2,glu", [(256, 384, 2048, True), (2048, 1024, 1536, False), (1024 + 256, 384, 256, True), (512, 1024, 0, False),
                                         (768, 384, 0, False),
                                         # FFN width 1536 (the reference's block under its pinned diffusers 0.33.1): the first launch form with SwiGLU N2 = 2 x 1536
                                         (512, 384, 3072, True)])
def test_dit_rowchain(ops, M, K1, N2, glu):
    """csrc/dit_rowchain.hip: GEMM 1 (N = 384) + gated rmsnorm + residual + next pre-norm + GEMM 2 of a NextDiT block in one launch, against
    (a) the fp32 formula of the chain with the unfused chain's rounding points (bf16 projection, bf16 pre-normed operand), (b) the three
    launches it replaces. Several environments per launch (mod_div = 256: a workgroup's rows share one modulation row), more than one
    workgroup per environment, row-strided views, both weight pairs; N2 = 0: no second GEMM, H written to memory instead."""
    D, div = 384, 256
    g = torch.Generator().manual_seed(M + K1 + N2 + 4)
    aw = _rand((M, K1 + 64), g)
    a_in = aw[:, :K1]                                                    # row-strided A
    w1 = _rand((D, K1), g, scale=K1 ** -0.5)
    x0 = torch.randn(M, D, generator=g).to(_dev())
    g1, g2 = [(1.0 + 0.1 * torch.randn(D, generator=g)).to(_dev()) for _ in range(2)]
    mod = (0.5 * torch.randn(M // div, 3 * D + 16, generator=g)).to(_dev())
    gate, ms2 = mod[:, :D], mod[:, D:2 * D]
    rowb = torch.arange(M, device=_dev()) // div
    rms = lambda t: t * torch.rsqrt(t.pow(2).mean(-1, keepdim=True) + 1e-5)
    proj = (a_in.float() @ w1.float().t()).to(torch.bfloat16).float()
    x_ref = x0 + torch.tanh(gate[rowb]) * rms(proj) * g1
    h_ref = rms(x_ref) * g2 * (1.0 + ms2[rowb])
    x = x0.clone()
    if N2 == 0:
        h = torch.full((M, D), 7.0, dtype=torch.bfloat16, device=_dev())
        ops.dit_rowchain(a_in, w1, g1, x, gate=gate, gamma2=g2, mod_scale2=ms2, h=h, mod_div=div)
        torch.cuda.synchronize()
        _close(x, x_ref, rtol=2e-3, atol=5e-3)
        _close(h, h_ref, rtol=1.0 / 128, atol=1e-2)
        x2 = x0.clone()                                                  # the last block's form: no H, no second GEMM
        ops.dit_rowchain(a_in, w1, g1, x2, gate=gate, mod_div=div)
        assert torch.equal(x2, x)
        return
    w2 = _rand((N2, D), g, scale=D ** -0.5)
    n_out = N2 // 2 if glu else N2
    cw = torch.zeros(M, n_out + 8, dtype=torch.bfloat16, device=_dev())
    c2 = cw[:, :n_out]
    ops.dit_rowchain(a_in, w1, g1, x, gate=gate, gamma2=g2, mod_scale2=ms2, w2=w2, c2=c2, glu2=glu, mod_div=div)
    torch.cuda.synchronize()
    _close(x, x_ref, rtol=2e-3, atol=5e-3)
    assert float(cw[:, n_out:].abs().max()) == 0.0, "columns beyond the output were written"
    hb = h_ref.to(torch.bfloat16).float()
    if glu:
        wg = w2.view(N2 // 32, 2, 16, D)[:, 0].reshape(N2 // 2, D).float()
        wu = w2.view(N2 // 32, 2, 16, D)[:, 1].reshape(N2 // 2, D).float()
        ref2 = torch.nn.functional.silu(hb @ wg.t()) * (hb @ wu.t())
    else:
        ref2 = hb @ w2.float().t()
    # the kernel rounds U = x * gamma2 * (1 + scale) to bf16 and applies the row's 1 / rms to the fp32 accumulators: one bf16 rounding of the
    # operand either way, at a different place - the results agree to the bf16 step of the operand amplified by the K = 384 sum
    d = (c2.float() - ref2).abs()
    scale = ref2.abs().max().item()
    assert d.max().item() <= 2.0 ** -6 * max(1.0, scale) and d.mean().item() <= 2.0 ** -9 * max(1.0, ref2.abs().mean().item()) + 1e-3, (d.max().item(), d.mean().item(), scale)
    if not glu and N2 % 384 == 0:
        # LayerNorm statistics of the 384-wide segments of the produced rows, for the attention stage that consumes them
        x3, c3 = x0.clone(), torch.zeros_like(cw)
        st = torch.full((M, N2 // 384, 2), float("nan"), device=_dev())
        ops.dit_rowchain(a_in, w1, g1, x3, gate=gate, gamma2=g2, mod_scale2=ms2, w2=w2, c2=c3[:, :n_out], mod_div=div, seg_stats=st, seg_eps=1e-5)
        torch.cuda.synchronize()
        assert torch.equal(x3, x) and torch.equal(c3, cw), "the statistics epilogue changed the outputs"
        seg = c2.float().view(M, N2 // 384, 384)
        m_ref, r_ref = seg.mean(-1), torch.rsqrt(seg.var(-1, unbiased=False) + 1e-5)
        # (from the fp32 accumulators: the bf16 rounding of the 384 stored values averages out of both moments)
        assert (st[..., 0] - m_ref).abs().max().item() <= 2e-3 * max(1.0, scale), (st[..., 0] - m_ref).abs().max().item()
        assert ((st[..., 1] - r_ref).abs() / r_ref).max().item() <= 2e-3, ((st[..., 1] - r_ref).abs() / r_ref).max().item()
    # the three launches it replaces
    pj = ops.linear(a_in, w1)
    xu, hu = x0.clone(), torch.empty(M, D, dtype=torch.bfloat16, device=_dev())
    ops.norm(pj, g1, None, eps=1e-5, rms=True, gate=gate, base=xu, mod_div=div, out32=xu, out2=hu, gamma2=g2, mod_scale2=ms2)
    cu = ops.linear(hu, w2, act="silu", glu=True) if glu else ops.linear(hu, w2)
    torch.cuda.synchronize()
    _close(x, xu, rtol=1.0 / 128, atol=2e-2)
    du = (c2.float() - cu.float()).abs()
    assert du.max().item() <= 2.0 ** -5 * max(1.0, scale) and du.mean().item() <= 2.0 ** -8 * max(1.0, ref2.abs().mean().item()) + 1e-3, (du.max().item(), du.mean().item())


def test_dit_rowchain_rejects_what_it_does_not_compute(ops):
    from internnav_amd._lib import EngineError

    g = torch.Generator().manual_seed(3)
    a, w1 = _rand((256, 512), g), _rand((384, 512), g)
    x = torch.zeros(256, 384, device=_dev())
    g1 = torch.ones(384, device=_dev())
    with pytest.raises(EngineError, match="K1"):
        ops.dit_rowchain(a, w1, g1, x)
    a, w1 = _rand((200, 384), g), _rand((384, 384), g)
    with pytest.raises(EngineError, match="multiples"):
        ops.dit_rowchain(a, w1, g1, torch.zeros(200, 384, device=_dev()))


