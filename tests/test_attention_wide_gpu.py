"""The 32-rows-per-wave attention forward (csrc/attention_wide.hip) against the fp32 formula and against the 16-row kernel it replaces
on the long dense shapes (LLM prefill, ViT full-attention blocks, DINOv2): every masking mode of the contract (causal with
Lq != Lk = cached prefix, kv_start, per-sequence k_len, varlen cu_q / cu_k), ragged tails, strided views of a fused qkv buffer, both
workgroup sizes and both running-max policies. Tolerance: the bf16 rounding of P and O (the 16-row kernel's own bound, 1.5e-2
absolute on N(0,1) inputs); the two kernels are also compared with each other."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _dev():
    return torch.device("cuda:0")


def _rand(shape, gen, scale=1.0):
    return (torch.randn(shape, generator=gen, dtype=torch.float32) * scale).to(torch.bfloat16).to(_dev())


@pytest.fixture(scope="module")
def ops(built_lib):
    from internnav_amd import ops

    return ops


def _ref(q, k, v, scale, causal=False, kv_start=0, k_len=None):
    B, Lq, H, D = q.shape
    Lk, Hkv = k.shape[1], k.shape[2]
    out = torch.zeros(B, Lq, H, D, device=q.device)
    for b in range(B):
        n = Lk if k_len is None else int(k_len[b])
        qf = q[b].float().permute(1, 0, 2)
        kf = k[b, :n].float().permute(1, 0, 2).repeat_interleave(H // Hkv, dim=0)
        vf = v[b, :n].float().permute(1, 0, 2).repeat_interleave(H // Hkv, dim=0)
        s = qf @ kf.transpose(-1, -2) * scale
        i = torch.arange(Lq, device=q.device)[:, None]
        j = torch.arange(n, device=q.device)[None, :]
        ok = torch.ones(Lq, n, dtype=torch.bool, device=q.device)
        if causal:
            ok &= j <= i + (n - Lq)
        if kv_start:
            ok &= j >= kv_start
        s = s.masked_fill(~ok, float("-inf"))
        p = torch.softmax(s, dim=-1)
        p = torch.nan_to_num(p, nan=0.0)      # fully masked rows: zeros (the kernels' convention)
        out[b] = (p @ vf).permute(1, 0, 2)
    return out


def _check(out, ref, atol=1.5e-2, rtol=1.0 / 128, what=""):
    err = (out.float() - ref).abs()
    bad = int((err > atol + rtol * ref.abs()).sum())
    print(f"{what}: mean|err| {err.mean().item():.3e} max|err| {err.max().item():.3e} ref max {ref.abs().max().item():.2f}")
    assert bad == 0, f"{what}: {bad}/{err.numel()} out of tolerance, max err {err.max().item():.4g}"
    return err.mean().item()


CASES = [
    # B, Lq, Lk, H, Hkv, D, causal, kv_start
    (2, 920, 920, 28, 4, 128, True, 0),      # LLM prefill
    (2, 624, 920, 28, 4, 128, True, 0),      # prefill of a suffix on a cached prefix (prefix-KV reuse): Lq < Lk, causal
    (1, 333, 1001, 8, 2, 128, True, 0),      # ragged tails on both axes
    (2, 784, 784, 16, 16, 80, False, 0),     # Qwen ViT full-attention block
    (1, 196, 196, 16, 16, 80, False, 0),     # un-resized look-down frame
    (3, 257, 257, 6, 6, 64, False, 0),       # DINOv2 ViT-S
    (2, 130, 515, 6, 6, 64, False, 70),      # kv_start inside the second block
    (2, 128, 128, 4, 4, 128, True, 0),       # smallest eligible shape
]


@pytest.mark.parametrize("B,Lq,Lk,H,Hkv,D,causal,kv_start", CASES)
def test_wide_attention_vs_fp32_formula_and_16row_kernel(ops, B, Lq, Lk, H, Hkv, D, causal, kv_start):
    g = torch.Generator().manual_seed(B * 1000 + Lq + Lk + D)
    q, k, v = _rand((B, Lq, H, D), g), _rand((B, Lk, Hkv, D), g), _rand((B, Lk, Hkv, D), g)
    scale = D ** -0.5
    ref = _ref(q, k, v, scale, causal, kv_start)
    old = ops.attention(q, k, v, scale=scale, causal=causal, kv_start=kv_start, kernel=1)
    new = ops.attention(q, k, v, scale=scale, causal=causal, kv_start=kv_start, kernel=2)
    auto = ops.attention(q, k, v, scale=scale, causal=causal, kv_start=kv_start)
    torch.cuda.synchronize()
    assert torch.equal(auto, new), "the automatic rule runs the 32-rows-per-wave kernel on every long dense shape"
    e_old = _check(old, ref, what="16-row kernel")
    e_new = _check(new, ref, what="wide kernel")
    assert e_new <= 1.35 * e_old + 1e-5, (e_new, e_old)
    assert (new.float() - old.float()).abs().max().item() <= 3e-2


def test_wide_attention_peaked_rows_and_growing_maximum(ops):
    """Scores with a large spread (x4 inputs) and keys sorted so that the row maximum keeps growing along the key axis: exercises the
    rescale path of both running-max policies."""
    g = torch.Generator().manual_seed(5)
    B, L, H, D = 2, 512, 4, 128
    q, k, v = _rand((B, L, H, D), g, 2.0), _rand((B, L, H, D), g, 2.0), _rand((B, L, H, D), g)
    ramp = torch.linspace(0.2, 2.0, L, device=_dev())[None, :, None, None]
    k = (k.float() * ramp).to(torch.bfloat16)
    scale = D ** -0.5
    ref = _ref(q, k, v, scale, True)
    for kern in (2, 1):
        out = ops.attention(q, k, v, scale=scale, causal=True, kernel=kern)
        _check(out, ref, atol=2e-2, what=f"peaked rows, kernel={kern}")


def test_wide_attention_ragged_k_len_batch(ops):
    """Right-padded System-2 batch: per-sequence key counts (k_len), with and without the causal mask (rows that see no key: zeros)."""
    g = torch.Generator().manual_seed(11)
    B, L, H, Hkv, D = 3, 700, 28, 4, 128
    q, k, v = _rand((B, L, H, D), g), _rand((B, L, Hkv, D), g), _rand((B, L, Hkv, D), g)
    lens = torch.tensor([700, 655, 513], dtype=torch.int32, device=_dev())
    for causal in (False, True):
        out = ops.attention(q, k, v, causal=causal, k_len=lens)
        ref = _ref(q, k, v, D ** -0.5, causal, 0, lens.tolist())
        _check(out, ref, what=f"k_len, causal={causal}")


def test_wide_attention_varlen_long_sequences_in_a_fused_qkv_buffer(ops):
    """cu_q / cu_k packing with sequences on both sides of the eligibility bound, q / k / v as strided views of one qkv buffer."""
    g = torch.Generator().manual_seed(21)
    lens = [784, 196, 130, 784, 64]
    T, H, D = sum(lens), 16, 80
    qkv = _rand((T, 3, H, D), g)
    q, k, v = qkv[:, 0], qkv[:, 1], qkv[:, 2]
    cu = torch.tensor([0] + list(torch.tensor(lens).cumsum(0)), dtype=torch.int32, device=_dev())
    out = ops.attention(q, k, v, cu_q=cu, cu_k=cu, max_q=max(lens), max_k=max(lens))
    o = 0
    for n in lens:
        ref = _ref(q[None, o:o + n], k[None, o:o + n], v[None, o:o + n], D ** -0.5)[0]
        _check(out[o:o + n], ref, what=f"varlen sequence of {n}")
        o += n


def test_window_attention_on_the_wide_kernel(ops):
    """Qwen2.5-VL ViT windows (28 of 32 blocks): packed d-80 sequences of at most 64 tokens - 64, 32, 32 and 16 tokens at the ragged edges of a
    28 x 28 grid - as strided views of the fused qkv buffer. Round 4: these run on the 32-rows-per-wave kernel with 2 waves and ONE K / V
    buffer (kernel = 0 / 2); kernel = 1 pins the 16-rows-per-wave kernel. Both against the fp32 formula and against each other."""
    g = torch.Generator().manual_seed(5)
    H, D = 16, 80
    lens = [64, 64, 32, 64, 16, 32, 64, 64, 7, 64, 33, 64]
    T = sum(lens)
    qkv = _rand((T, 3, H, D), g)
    q, k, v = qkv[:, 0], qkv[:, 1], qkv[:, 2]
    cu = torch.tensor([0] + list(torch.tensor(lens).cumsum(0)), dtype=torch.int32, device=_dev())
    outs = {kern: ops.attention(q, k, v, cu_q=cu, cu_k=cu, max_q=64, max_k=64, kernel=kern) for kern in (0, 1, 2)}
    assert torch.equal(outs[0], outs[2])                     # the automatic rule takes the wide kernel for this shape
    o = 0
    for n in lens:
        ref = _ref(q[o:o + n][None], k[o:o + n][None], v[o:o + n][None], D ** -0.5)[0]
        for kern in (1, 2):
            _check(outs[kern][o:o + n], ref, what=f"window of {n} tokens, kernel {kern}")
        o += n
    assert (outs[1].float() - outs[2].float()).abs().max().item() < 3e-2
    with pytest.raises(RuntimeError, match="kernel = 2"):     # the same packing at d = 64 is outside the wide kernel's contract
        q64 = _rand((T, H, 64), g)
        ops.attention(q64, q64, q64, cu_q=cu, cu_k=cu, max_q=64, max_k=64, kernel=2)
