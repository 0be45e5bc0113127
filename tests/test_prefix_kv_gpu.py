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


class _Tok:
    def __call__(self, texts, return_tensors="pt"):
        ids, i, t = [], 0, texts[0]
        cfg = S.QWEN_TEST_CFG
        special = {"<|image_pad|>": cfg["image_token_id"], "<|vision_start|>": cfg["vision_start_id"], "<|vision_end|>": cfg["vision_end_id"]}
        while i < len(t):
            for k, v in special.items():
                if t.startswith(k, i):
                    ids.append(v)
                    i += len(k)
                    break
            else:
                ids.append(ord(t[i]) % 3000)
                i += 1
        return {"input_ids": torch.tensor([ids])}

    def decode(self, ids, skip_special_tokens=True):
        return "12 34"                       # a pixel goal: every call also runs the latent queries


class _Proc:
    tokenizer = _Tok()
    image_token = "<|image_pad|>"

    def apply_chat_template(self, conv, tokenize=False, add_generation_prompt=True):
        return "".join("<|vision_start|><|image_pad|><|vision_end|>" if c["type"] == "image" else c["text"] for m in conv for c in m["content"])


def test_policy_reuses_the_episode_prefix_between_system2_calls(built_lib):
    """through InternVLAN1Net (prefix_cache=True): the first call with history stores the K/V of [template | instruction | frame 0], later
    calls (more history, the look-down continuation) hand them back; sequences and latents stay bit-identical to the uncached policy and
    the engine prefills fewer rows."""
    from internnav_amd.policy import InternVLAN1ForCausalLM, InternVLAN1Net
    from internnav_amd.preprocess import FramePreprocessor

    cfg = S.QWEN_TEST_CFG
    sd = S.materialize(S.n1_full_spec(cfg, "nextdit_async"), 5)
    model = InternVLAN1ForCausalLM(sd, cfg, "nextdit_async", device=DEV, max_envs=1, num_history=3, resize_w=280, resize_h=280, cam_w=320, cam_h=240)
    pre = FramePreprocessor(DEV, resize_w=280, resize_h=280)
    nets = [InternVLAN1Net(model, _Proc(), num_history=3, resize_w=280, resize_h=280, frame_preprocessor=pre, prefix_cache=pc) for pc in (False, True)]
    assert nets[1].prefix_cache and not nets[0].prefix_cache
    rng = np.random.default_rng(4)
    frames = [rng.integers(0, 256, (240, 320, 3), dtype=np.uint8) for _ in range(9)]
    outs, rows = [], []
    for net in nets:
        o, r = [], []
        net.step_no_infer(frames[0], None, None)
        net.step_no_infer(frames[1], None, None)
        for f, look_down in ((frames[2], False), (frames[3], True), (frames[4], False), (frames[5], False)):
            if not look_down and o:
                net.step_no_infer(frames[6 + len(o) % 3], None, None)       # the episode moves on between System-2 calls
            so = net.s2_step(f, None, None, "go to the door", None, look_down)
            assert so.output_latent is not None
            o.append((so.output_action, None if so.output_latent is None else so.output_latent.cpu()))
            r.append(model._gen["state"]["S_run"])
        outs.append(o)
        rows.append(r)
    for (a0, l0), (a1, l1) in zip(*outs):
        assert a0 == a1 and ((l0 is None and l1 is None) or torch.equal(l0, l1))
    assert rows[1][0] == rows[0][0]                       # first call of the episode: nothing cached yet (it stores the prefix)
    assert all(c < u for c, u in zip(rows[1][1:], rows[0][1:])), (rows[0], rows[1])   # later calls run only what lies behind the prefix
