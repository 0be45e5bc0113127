"""GPU: per-frame ViT-embedding cache (SURVEY.md 8f-1 second half). Exactness claim: the Qwen2.5-VL vision tower attends inside one
image only (window / per-image cu_seqlens), every other op is row-wise, and the tiled GEMM kernels accumulate K in the same order for
every tile shape - so a frame's merged embeddings do not depend on which other frames share the launch (for frames of more than 64 merged
tokens, i.e. every real camera geometry: 196 / 391 tokens). Tested BIT-EXACT:
cached == recomputed, at engine level (logits / tokens / latents) and through the policy (look-down turn re-uses the previous turn)."""
import numpy as np
import pytest
import torch

from internnav_amd import synthetic as S

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def eng(built_lib):
    from internnav_amd.qwen_vl import QwenVLEngine

    cfg = S.QWEN_TEST_CFG
    sd = S.qwen_state_dict(seed=6, cfg=cfg)
    return QwenVLEngine(sd, cfg, DEV, max_seqs=2, max_seq_len=1024, max_patches=2 * (2 * 784 + 34 * 46)), cfg


def test_cached_frames_equal_recomputed_bit_exact(eng):
    eng, cfg = eng
    inp = S.qwen_inputs(2, 2, seed=6, cfg=cfg)
    pv = inp["pixel_values"].to(DEV, torch.bfloat16)
    B, S_ = inp["input_ids"].shape
    st = eng.prefill(inp["input_ids"], pv, inp["grid_thw"], cached_embeds=[None] * 4)
    fresh = eng.fresh_image_embeds(st["plan"])
    assert sorted(fresh) == [0, 1, 2, 3] and all(v.shape == (196, cfg["t_hidden"]) for v in fresh.values())
    toks_a = eng.decode(st, 4).clone()
    lat_a = eng.latents(st, toks_a[:, -1:].contiguous()).clone()
    st = eng.prefill(inp["input_ids"], pv, inp["grid_thw"])
    eng._last_logits(B, S_, S_ - 1)
    logits_a = eng.logits[:B].clone()
    # frames 0 and 2 (the first image of each env) come from the cache; only frames 1 and 3 go through the tower, in a smaller batch
    pv4 = pv.view(4, 784, 1176)
    st = eng.prefill(inp["input_ids"], torch.cat([pv4[1], pv4[3]]), inp["grid_thw"], cached_embeds=[fresh[0], None, fresh[2], None])
    again = eng.fresh_image_embeds(st["plan"])
    assert sorted(again) == [1, 3]
    assert torch.equal(again[1], fresh[1]) and torch.equal(again[3], fresh[3])       # batch composition does not change a frame's embeddings
    eng._last_logits(B, S_, S_ - 1)
    assert torch.equal(eng.logits[:B], logits_a)
    st = eng.prefill(inp["input_ids"], torch.cat([pv4[1], pv4[3]]), inp["grid_thw"], cached_embeds=[fresh[0], None, fresh[2], None])
    toks_b = eng.decode(st, 4)
    assert torch.equal(toks_b, toks_a)
    assert torch.equal(eng.latents(st, toks_b[:, -1:].contiguous()), lat_a)
    # everything cached: no vision launch at all
    st = eng.prefill(inp["input_ids"], None, inp["grid_thw"], cached_embeds=[fresh[k] for k in range(4)])
    eng._last_logits(B, S_, S_ - 1)
    assert torch.equal(eng.logits[:B], logits_a)


class _Tok:
    def __call__(self, texts, return_tensors="pt"):
        ids, i, t = [], 0, texts[0]
        while i < len(t):
            if t.startswith("<|image_pad|>", i):
                ids.append(S.QWEN_TEST_CFG["image_token_id"])
                i += len("<|image_pad|>")
            elif t.startswith("<|vision_start|>", i):
                ids.append(S.QWEN_TEST_CFG["vision_start_id"])
                i += len("<|vision_start|>")
            else:
                ids.append(ord(t[i]) % 3000)
                i += 1
        return {"input_ids": torch.tensor([ids])}

    def decode(self, ids, skip_special_tokens=True):
        return "↓"


class _Proc:
    tokenizer = _Tok()
    image_token = "<|image_pad|>"

    def apply_chat_template(self, conv, tokenize=False, add_generation_prompt=True):
        return "".join("<|vision_start|><|image_pad|>" if c["type"] == "image" else c["text"] for m in conv for c in m["content"])


def test_policy_look_down_turn_reuses_the_previous_turns_frames(built_lib):
    from internnav_amd.policy import InternVLAN1ForCausalLM, InternVLAN1Net
    from internnav_amd.preprocess import FramePreprocessor

    cfg = S.QWEN_TEST_CFG
    sd = S.materialize(S.n1_full_spec(cfg, "nextdit_async"), 5)
    # geometry: 280 x 280 history frames (20 x 20 patches = 100 tokens), 320 x 240 camera (18 x 22 patches = 99 tokens): like the real
    # 196 / 391-token frames, every image has > 64 merged tokens, so its merger GEMMs take the tiled kernels in every batch composition
    # (a lone image of <= 64 tokens would take the weight-streaming kernel, whose cross-wave K reduction rounds differently)
    model = InternVLAN1ForCausalLM(sd, cfg, "nextdit_async", device=DEV, max_envs=1, num_history=3, resize_w=280, resize_h=280, cam_w=320, cam_h=240)
    pre = FramePreprocessor(DEV, resize_w=280, resize_h=280)
    nets = [InternVLAN1Net(model, _Proc(), num_history=3, resize_w=280, resize_h=280, frame_preprocessor=pre, vit_cache=vc) for vc in (False, True)]
    rng = np.random.default_rng(3)
    frames = [rng.integers(0, 256, (240, 320, 3), dtype=np.uint8) for _ in range(7)]
    outs = []
    for net in nets:
        for f in frames[:4]:
            net.step_no_infer(f, None, None)
        o = []
        for f, look_down in ((frames[4], False), (frames[5], True), (frames[6], False)):
            so = None
            inputs = net.build_s2_inputs(f, "go to the door", look_down)
            extra = {"cached_image_embeds": inputs["cached_image_embeds"]} if "cached_image_embeds" in inputs else {}
            seq = model.generate(input_ids=inputs["input_ids"], pixel_values=inputs["pixel_values"], image_grid_thw=inputs["image_grid_thw"],
                                 max_new_tokens=4, return_dict_in_generate=True, **extra).sequences
            if extra:
                net.update_frame_cache(inputs, model.last_image_embeds())
            lat = model.generate_latents(seq, inputs["pixel_values"], inputs["image_grid_thw"], **extra)
            net.llm_output = "↓"
            o.append((inputs["input_ids"].clone(), seq.cpu(), lat.cpu(), inputs["pixel_values"].shape[0], inputs["image_grid_thw"].tolist()))
        outs.append(o)
    for (ids0, seq0, lat0, n0, g0), (ids1, seq1, lat1, n1, g1) in zip(*outs):
        assert torch.equal(ids0, ids1) and g0 == g1
        assert torch.equal(seq0, seq1) and torch.equal(lat0, lat1)          # bit-exact with and without the cache
    n_first, n_ld, n_third = (o[3] for o in outs[1])
    full_first, full_ld, full_third = (o[3] for o in outs[0])
    assert n_first == full_first                                             # nothing cached yet
    assert n_ld < full_ld and n_ld == full_ld - full_first                   # look-down turn: only the look-down frame is encoded
    assert n_third < full_third                                              # frame 0 (always in the np.linspace sample) comes from the cache
