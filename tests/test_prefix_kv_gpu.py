"""GPU: prefix-KV reuse in the System-2 engine (DESIGN 8.5 / VERDICT r2 item 4). Between the System-2 calls of an episode the prompt starts
with the same tokens (chat template + instruction + the first history frame: np.linspace always samples frame 0,
internvla_n1_policy.py:125-133). With a causal mask the K/V of a token depend on the tokens before it only, so a later call can take
the prefix's K/V of all layers from a cache and run only the rest: the first images are not encoded, their tokens not prefilled.

Claim tested: logits, greedy tokens and the latent queries of the prefix-cached call EQUAL the full call's - bit for bit, because the
tiled GEMMs accumulate K in the same order for every tile shape and the attention kernel walks the keys in the same blocks from key 0
(the same property the per-frame ViT cache rests on, tests/test_vit_cache_gpu.py) - for prefixes taken (a) from the full call itself
and (b) from a prefill of the prefix ALONE (what an earlier call with a different suffix leaves behind), dense and ragged batches."""
import numpy as np
import pytest
import torch

from internnav_amd import synthetic as S

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
N_TEXT, N_IMG = 40, 3


@pytest.fixture(scope="module")
def eng(built_lib):
    from internnav_amd.qwen_vl import QwenVLEngine

    cfg = S.QWEN_TEST_CFG
    sd = S.qwen_state_dict(seed=8, cfg=cfg)
    return QwenVLEngine(sd, cfg, DEV, max_seqs=2, max_seq_len=1024, max_patches=2 * N_IMG * 784), cfg


def _run(eng, ids, pv, grid, n_dec=5, **kw):
    st = eng.prefill(ids, pv, grid, **kw)
    B, Sr = st["B"], st["S_run"]
    if "lens" in st:
        rows = torch.from_numpy((np.arange(B) * Sr + st["lens"] - (st["S"] - Sr) - 1).astype(np.int32)).to(DEV)
        eng._last_logits(B, Sr, None, rows_idx=rows)
    else:
        eng._last_logits(B, Sr, Sr - 1)
    logits = eng.logits[:B].clone()
    toks = eng.decode(st, n_dec).clone()
    lat = eng.latents(st, toks[:, -1:].contiguous()).clone()
    return logits, toks, lat, st


@pytest.mark.parametrize("ragged", [False, True])
def test_prefix_cached_call_equals_full_call(eng, ragged):
    eng, cfg = eng
    inp = S.qwen_inputs(2, N_IMG, seed=8, cfg=cfg, n_text=N_TEXT, n_tail=24)
    pv = inp["pixel_values"].to(DEV, torch.bfloat16)
    ids, grid = inp["input_ids"], inp["grid_thw"]
    B, S_ = ids.shape
    P = N_TEXT + 196 + 2                                     # text | <vs> image 0 <ve>  = the cached prefix
    kw = dict(seq_lens=[S_, S_ - 9]) if ragged else {}
    logits_a, toks_a, lat_a, _ = _run(eng, ids, pv, grid, **kw)
    # (a) the prefix K/V the full call itself left in the cache
    eng.prefill(ids, pv, grid, **kw)
    kv_full = [eng.export_prefix_kv(b, P) for b in range(B)]
    # (b) the K/V of a prefill of the prefix alone (image 0 only): what an EARLIER call with another suffix would have cached
    pv3 = pv.view(B, N_IMG, 784, 1176)
    eng.prefill(ids[:, :P], pv3[:, 0].reshape(-1, 1176).contiguous(), grid[::N_IMG])
    kv_alone = [eng.export_prefix_kv(b, P) for b in range(B)]
    same = all(torch.equal(a, b) for a, b in zip(kv_full, kv_alone))
    diff = max((a.float() - b.float()).abs().max().item() for a, b in zip(kv_full, kv_alone))
    print(f"prefix K/V from the full call vs from a prefill of the prefix alone: bit-equal {same}, max|diff| {diff:.3e}")
    assert diff <= 2e-2                                      # causality; bit-equality is reported, the suffix results below are what counts
    pv_rest = pv3[:, 1:].reshape(-1, 1176).contiguous()      # patches of images 1 .. N_IMG-1 only
    for name, kvs in (("own prefix", kv_full), ("prefix computed alone", kv_alone)):
        for L in eng.layers:                                 # poison the cache: the call below must not depend on stale rows
            L["kv"].zero_()
        for b in range(B):
            eng.import_prefix_kv(b, kvs[b])
        logits_b, toks_b, lat_b, st = _run(eng, ids, pv_rest, grid, prefix_len=P, **kw)
        assert st["S_run"] == S_ - P and st["plan"]["images_run"] == [1, 2, 4, 5]
        d_log = (logits_b.float() - logits_a.float()).abs().max().item()
        d_lat = (lat_b.float() - lat_a.float()).abs().max().item()
        print(f"{name}: logits max|diff| {d_log:.3e} (bit-equal {torch.equal(logits_b, logits_a)}), tokens equal {torch.equal(toks_b, toks_a)}, "
              f"latents max|diff| {d_lat:.3e} (bit-equal {torch.equal(lat_b, lat_a)})")
        assert torch.equal(toks_b, toks_a)
        if name == "own prefix":
            assert torch.equal(logits_b, logits_a) and torch.equal(lat_b, lat_a)          # exact: same K/V bits, same arithmetic on the suffix rows
        else:
            assert d_log <= 3e-2 and d_lat <= 3e-2


def test_prefix_must_end_on_an_image_boundary_and_leave_tokens(eng):
    eng, cfg = eng
    inp = S.qwen_inputs(1, 2, seed=3, cfg=cfg, n_text=N_TEXT, n_tail=8)
    with pytest.raises(AssertionError, match="image boundary"):
        eng.plan(inp["input_ids"], inp["grid_thw"], prefix_len=N_TEXT + 50)
    with pytest.raises(AssertionError):
        eng.plan(inp["input_ids"], inp["grid_thw"], prefix_len=inp["input_ids"].shape[1])
