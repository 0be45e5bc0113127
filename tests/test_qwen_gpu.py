"""GPU parity of the System-2 engine (internnav_amd.qwen_vl) against the fixture produced by the installed transformers
Qwen2.5-VL modules + the reference's rope index and glue (tests/golden/qwen.pt; true layer widths, reduced depth/vocab).
Tolerances: bf16 operands / fp32 accumulation vs an fp32 fixture; logits are O(60) wide (unit-gain random weights), so the
check is relative to the logit scale; greedy tokens must match wherever the fixture's top-2 margin exceeds the logit error."""
from pathlib import Path

import numpy as np
import pytest
import torch

from oracle import weights as W

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def setup(built_lib):
    from internnav_amd.qwen_vl import QwenVLEngine

    gold = torch.load(Path(__file__).resolve().parent / "golden" / "qwen.pt", weights_only=True)
    cfg = W.QWEN_TEST_CFG
    sd = W.qwen_state_dict(seed=gold["seed"], cfg=cfg)
    inp = W.qwen_inputs(gold["B"], gold["n_img"], seed=gold.get("input_seed", gold["seed"]), cfg=cfg)
    eng = QwenVLEngine(sd, cfg, DEV, max_seqs=gold["B"], max_seq_len=512, max_patches=inp["pixel_values"].shape[0])
    return gold, cfg, inp, eng


def test_rope_index_matches_reference(setup):
    from internnav_amd.qwen_vl import rope_index

    gold, cfg, inp, _ = setup
    grids = [tuple(g) for g in inp["grid_thw"].tolist()]
    pos, _ = rope_index(inp["input_ids"].numpy(), grids, cfg["image_token_id"], cfg["vision_start_id"])
    assert np.array_equal(pos, gold["position_ids"].numpy().astype(np.int64))  # bit-exact integer logic vs the reference's get_rope_index_25


def test_vision_tower(setup):
    gold, cfg, inp, eng = setup
    grids = [tuple(g) for g in inp["grid_thw"].tolist()]
    emb, inv = eng.vision(inp["pixel_values"].to(DEV, torch.bfloat16), grids)
    out = emb.float().cpu()[torch.from_numpy(inv).long()][:: gold["embed_row_stride"]]
    ref = gold["image_embeds"]
    d = (out - ref).abs()
    print(f"image embeds: mean|err| {d.mean():.3e} max|err| {d.max():.3e} ref rms {ref.pow(2).mean().sqrt():.3f}")
    assert d.mean() < 1e-2 * ref.pow(2).mean().sqrt() and d.max() < 0.1 * ref.abs().max()


def test_prefill_logits_greedy_tokens_and_latents(setup):
    gold, cfg, inp, eng = setup
    pv = inp["pixel_values"].to(DEV, torch.bfloat16)
    state = eng.prefill(inp["input_ids"], pv, inp["grid_thw"])
    toks = eng.decode(state, 3)
    logits0 = None
    ref = gold["last_logits"]
    # logits of the last prompt position (recomputed: decode() overwrote eng.logits)
    st2 = eng.prefill(inp["input_ids"], pv, inp["grid_thw"])
    eng._last_logits(st2["B"], st2["S"], st2["S"] - 1)
    logits0 = eng.logits[: st2["B"]].float().cpu()
    d = (logits0 - ref).abs()
    scale = ref.std().item()
    print(f"last-position logits: mean|err| {d.mean():.3e} max|err| {d.max():.3e} logit std {scale:.2f}")
    assert d.mean() < 5e-3 * scale and d.max() < 5e-2 * scale
    gen_ref = gold["generated"][:, inp["input_ids"].shape[1]:]
    top2 = ref.topk(2, dim=-1).values
    margin = (top2[:, 0] - top2[:, 1])
    same = (toks.cpu().long() == gen_ref)
    print("greedy tokens", toks.cpu().tolist(), "reference", gen_ref.tolist(), "first-step margin", margin.tolist())
    assert bool(same[:, 0][margin > 2 * d.max()].all()), "greedy token differs although the reference margin exceeds our logit error"
    if bool(same.all()):
        # latent queries against the KV cache of generate() vs the reference's full re-run
        state = eng.prefill(inp["input_ids"], pv, inp["grid_thw"])
        toks = eng.decode(state, 3)
        lat = eng.latents(state, toks[:, -1:].contiguous())
        refl = gold["latents"]
        dl = (lat.float().cpu() - refl).abs()
        print(f"latents (cache reuse): mean|err| {dl.mean():.3e} max|err| {dl.max():.3e} ref rms {refl.pow(2).mean().sqrt():.3f}")
        assert dl.mean() < 1e-2 * refl.pow(2).mean().sqrt()
        # per-sequence placement path gives the same result when every sequence kept all its tokens
        state = eng.prefill(inp["input_ids"], pv, inp["grid_thw"])
        toks = eng.decode(state, 3)
        S = inp["input_ids"].shape[1]
        lat2 = eng.latents(state, toks[:, -1:].contiguous(), seq_lens=np.full(gold["B"], S + 2))
        assert torch.equal(lat, lat2)
        # reference-signature path (full prefill over output_ids + N_QUERY traj tokens)
        lat3 = eng.generate_latents(gold["generated"], pv, inp["grid_thw"])
        d3 = (lat3.float().cpu() - refl).abs()
        print(f"latents (full re-run): mean|err| {d3.mean():.3e} max|err| {d3.max():.3e}")
        assert d3.mean() < 1e-2 * refl.pow(2).mean().sqrt()


def test_single_token_pass_variants_agree(setup):
    """the single-token passes with the rotary embedding + KV-cache append inside the attention launch (default) and with the rope launch of their
    own are BIT-IDENTICAL - same arithmetic, same rounding points - in tokens, logits, latent queries and the appended cache rows. The variant
    with separate norm launches differs by rounding noise only (same tokens)."""
    gold, cfg, inp, eng = setup
    pv = inp["pixel_values"].to(DEV, torch.bfloat16)
    keep = (eng.fuse_decode_norm, eng.fuse_decode_rope)
    res = {}
    try:
        for name, fuse, rope in (("rope_in_attention", True, True), ("rope_launch", True, False), ("norm_launches", False, False)):
            eng.fuse_decode_norm, eng.fuse_decode_rope = fuse, rope
            state = eng.prefill(inp["input_ids"], pv, inp["grid_thw"])
            toks = eng.decode(state, 4)
            lat = eng.latents(state, toks[:, -1:].contiguous())
            S = inp["input_ids"].shape[1]
            assert S >= 256, "the fixture's prompts reach the decode attention kernel"
            kv = torch.stack([L["kv"].view(eng.B_max, eng.S_max, -1)[: state["B"], S:S + 3].float().cpu() for L in eng.layers])
            res[name] = (toks.cpu(), eng.logits[: state["B"]].float().cpu().clone(), lat.float().cpu(), kv)
    finally:
        eng.fuse_decode_norm, eng.fuse_decode_rope = keep
    base = res["rope_launch"]
    for what, x, y in zip(("tokens", "logits", "latents", "decoded K/V rows"), res["rope_in_attention"], base):
        assert torch.equal(x, y), f"rope inside the attention launch: {what} differ (max |d| {(x.float() - y.float()).abs().max().item():.3e})"
    scale = base[1].std().item()
    toks, logits, lat, _ = res["norm_launches"]
    assert torch.equal(toks, base[0])
    d = (logits - base[1]).abs()
    assert d.mean().item() < 2e-3 * scale and d.max().item() < 3e-2 * scale, (d.mean().item(), d.max().item(), scale)
    assert (lat - base[2]).abs().mean().item() < 4e-3 * base[2].pow(2).mean().sqrt().item()


def test_generate_surface(setup):
    gold, cfg, inp, eng = setup
    seqs = eng.generate(inp["input_ids"], inp["pixel_values"].to(DEV, torch.bfloat16), inp["grid_thw"], max_new_tokens=3)
    assert seqs.shape == gold["generated"].shape and seqs.dtype == torch.long
    assert torch.equal(seqs[:, : inp["input_ids"].shape[1]], inp["input_ids"])


def test_planned_launch_sequence_equals_eager_and_is_graph_capturable(setup):
    """plan() + run_s2() (host work up front, pure launch sequence) == the eager stateful API, eagerly and replayed from a hipGraph."""
    gold, cfg, inp, eng = setup
    pv = inp["pixel_values"].to(DEV, torch.bfloat16)
    B, S = inp["input_ids"].shape
    state = eng.prefill(inp["input_ids"], pv, inp["grid_thw"])
    toks_ref = eng.decode(state, 3).clone()
    lat_ref = eng.latents(state, toks_ref[:, -1:].contiguous()).clone()
    P = eng.plan(inp["input_ids"], inp["grid_thw"], n_decode=3, with_latents=True)
    toks = torch.zeros(B, 3, dtype=torch.int32, device=DEV)
    lat = torch.zeros(B, cfg["n_query"], cfg["t_hidden"], dtype=torch.bfloat16, device=DEV)
    eng.run_s2(P, pv, toks, lat)
    assert torch.equal(toks, toks_ref) and torch.equal(lat, lat_ref)
    from internnav_amd.runtime import GraphedCall

    toks.zero_()
    lat.zero_()
    g = GraphedCall(lambda pixel_values: eng.run_s2(P, pixel_values, toks, lat), {"pixel_values": pv})
    toks.zero_()
    lat.zero_()
    g()
    torch.cuda.synchronize()
    assert torch.equal(toks, toks_ref) and torch.equal(lat, lat_ref)


def test_split_prefill_two_streams_equals_joint_prefill(setup):
    """The prefill of >= 2 plain prompts as two half batches on two streams (engine.split_prefill, fork / join inside the launch
    sequence) against the joint launch sequence: same greedy tokens, logits / latents / K-V cache equal to rounding (the halves may
    pick other GEMM tile shapes than the joint batch), eagerly and replayed from ONE captured graph; the frame-cache rows as well."""
    from internnav_amd.runtime import GraphedCall

    gold, cfg, inp, eng = setup
    pv = inp["pixel_values"].to(DEV, torch.bfloat16)
    B, S = inp["input_ids"].shape
    assert B >= 2
    out = {}
    for split in (False, True):
        eng.split_prefill = split
        P = eng.plan(inp["input_ids"], inp["grid_thw"], n_decode=3, with_latents=True)
        assert ("split" in P) == split
        toks = torch.zeros(B, 3, dtype=torch.int32, device=DEV)
        lat = torch.zeros(B, cfg["n_query"], cfg["t_hidden"], dtype=torch.bfloat16, device=DEV)
        for L in eng.layers:
            L["kv"].zero_()
        eng.run_s2(P, pv, toks, lat)
        torch.cuda.synchronize()
        kv = torch.stack([L["kv"].view(eng.B_max, eng.S_max, -1)[:B, :S].clone() for L in eng.layers])
        out[split] = dict(toks=toks.clone(), lat=lat.clone(), kv=kv, logits=eng.logits[:B].clone(), emb_tok=eng.emb_tok.clone(), P=P)
    eng.split_prefill = True
    a, b = out[False], out[True]
    assert torch.equal(a["toks"], b["toks"])
    for k in ("lat", "kv", "logits", "emb_tok"):
        d = (a[k].float() - b[k].float()).abs()
        print(f"split vs joint prefill, {k}: bit-equal {torch.equal(a[k], b[k])}, max|diff| {d.max().item():.3e} (scale {a[k].float().abs().max().item():.2f})")
        assert d.max().item() <= 2e-2 * max(1.0, a[k].float().abs().max().item())
    # one captured graph with the fork / join inside
    P = b["P"]
    toks = torch.zeros(B, 3, dtype=torch.int32, device=DEV)
    lat = torch.zeros(B, cfg["n_query"], cfg["t_hidden"], dtype=torch.bfloat16, device=DEV)
    g = GraphedCall(lambda pixel_values: eng.run_s2(P, pixel_values, toks, lat), {"pixel_values": pv})
    toks.zero_()
    lat.zero_()
    g()
    torch.cuda.synchronize()
    assert torch.equal(toks, b["toks"]) and torch.equal(lat, b["lat"])


def test_facade_generate_then_latents_reuses_cache(setup):
    """InternVLAN1ForCausalLM surface: generate(...).sequences then generate_latents(output_ids, ...) as the reference's policy calls them."""
    from internnav_amd.policy import InternVLAN1ForCausalLM
    from internnav_amd import synthetic

    gold, cfg, inp, eng = setup
    spec = synthetic.n1_full_spec(cfg)
    sd = synthetic.materialize({k: v for k, v in spec.items()}, seed=gold["seed"])
    m = InternVLAN1ForCausalLM(sd, cfg, "nextdit_async", device=DEV, max_envs=gold["B"], max_seq_len=512, max_patches=inp["pixel_values"].shape[0])
    out = m.generate(input_ids=inp["input_ids"], pixel_values=inp["pixel_values"], image_grid_thw=inp["grid_thw"], max_new_tokens=3,
                     do_sample=False, use_cache=True, past_key_values=None, return_dict_in_generate=True, eos_token_id=-1).sequences
    assert torch.equal(out.cpu(), gold["generated"])
    lat = m.generate_latents(out, inp["pixel_values"], inp["grid_thw"])
    d = (lat.float().cpu() - gold["latents"]).abs()
    assert d.mean() < 1e-2 * gold["latents"].pow(2).mean().sqrt()
    img = torch.rand(gold["B"], 2, 224, 224, 3, generator=torch.Generator().manual_seed(1))
    traj = m.generate_traj(lat, img.to(DEV))
    assert traj.shape == (32 * gold["B"], 32, 3) and torch.isfinite(traj).all()


def test_lookdown_frame_ragged_windows(built_lib):
    """the un-resized look-down frame (476x644 -> 34x46 patches): ragged 112-px windows (full / half / quarter) and a 391-token image next
    to a 196-token one - vision tower, 3-D rope index and last-position logits vs the transformers / reference fixture."""
    from internnav_amd import synthetic
    from internnav_amd.qwen_vl import QwenVLEngine, rope_index

    gold = torch.load(Path(__file__).resolve().parent / "golden" / "qwen_lookdown.pt", weights_only=True)
    cfg = W.QWEN_TEST_CFG
    sd = W.qwen_state_dict(seed=gold["seed"], cfg=cfg)
    inp = synthetic.qwen_lookdown_inputs(cfg)
    grids = [tuple(g) for g in inp["grid_thw"].tolist()]
    pos, _ = rope_index(inp["input_ids"].numpy(), grids, cfg["image_token_id"], cfg["vision_start_id"])
    assert np.array_equal(pos, gold["position_ids"].numpy().astype(np.int64))
    eng = QwenVLEngine(sd, cfg, DEV, max_seqs=1, max_seq_len=1024, max_patches=inp["pixel_values"].shape[0])
    pv = inp["pixel_values"].to(DEV, torch.bfloat16)
    emb, inv = eng.vision(pv, grids)
    out = emb.float().cpu()[torch.from_numpy(inv).long()][:: gold["embed_row_stride"]]
    d = (out - gold["image_embeds"]).abs()
    rms = gold["image_embeds"].pow(2).mean().sqrt()
    print(f"look-down image embeds: mean|err| {d.mean():.3e} max|err| {d.max():.3e} ref rms {rms:.3f}")
    assert d.mean() < 1e-2 * rms
    st = eng.prefill(inp["input_ids"], pv, inp["grid_thw"])
    eng._last_logits(1, st["S"], st["S"] - 1)
    dl = (eng.logits[:1].float().cpu() - gold["last_logits"]).abs()
    print(f"look-down last logits: mean|err| {dl.mean():.3e} max|err| {dl.max():.3e}")
    assert dl.mean() < 5e-3 * gold["last_logits"].std() and dl.max() < 5e-2 * gold["last_logits"].std()


def test_ragged_answers_latents_after_early_eos(setup):
    """batched generate where one sequence hits EOS after its first token: generate_latents must place the latent queries right behind
    each sequence's own last kept token (per-sequence key lengths) == the reference's batch-1 semantics, checked against the oracle."""
    from internnav_amd.policy import InternVLAN1ForCausalLM
    from internnav_amd import synthetic
    from oracle import qwen_vl as o_q

    gold, cfg, inp, eng = setup
    S = inp["input_ids"].shape[1]
    eos = int(gold["generated"][0, S])          # first generated token of sequence 0 plays EOS: sequence 0 stops after one token
    assert int(gold["generated"][1, S]) != eos and eos not in gold["generated"][1, S:].tolist()
    sd = synthetic.materialize(synthetic.n1_full_spec(cfg), seed=gold["seed"])
    m = InternVLAN1ForCausalLM(sd, cfg, "nextdit_async", device=DEV, max_envs=2, max_seq_len=512, max_patches=inp["pixel_values"].shape[0])
    seqs = m.generate(input_ids=inp["input_ids"], pixel_values=inp["pixel_values"], image_grid_thw=inp["grid_thw"], max_new_tokens=3,
                      do_sample=False, return_dict_in_generate=True, eos_token_id=eos).sequences.cpu()
    assert seqs[0, S:].tolist() == [eos, eos, eos] and torch.equal(seqs[1], gold["generated"][1])
    lat = m.generate_latents(seqs, inp["pixel_values"], inp["grid_thw"]).float().cpu()
    qsd = {k: v for k, v in sd.items() if not k.startswith("model.traj_dit") and not k.startswith("model.rgb_")}
    per_img = inp["pixel_values"].shape[0] // 2
    with torch.no_grad():
        ref0 = o_q.generate_latents(qsd, cfg, seqs[0:1, : S + 1], inp["pixel_values"][:per_img], inp["grid_thw"][: gold["n_img"]])
        ref1 = o_q.generate_latents(qsd, cfg, seqs[1:2], inp["pixel_values"][per_img:], inp["grid_thw"][gold["n_img"]:])
    for b, ref in ((0, ref0), (1, ref1)):
        d = (lat[b] - ref[0]).abs()
        print(f"seq {b}: latents mean|err| {d.mean():.3e} max|err| {d.max():.3e}")
        assert d.mean() < 1e-2 * ref.pow(2).mean().sqrt()


def test_ragged_prompt_batch_vs_per_env_oracle(built_lib):
    """f2: prompts of DIFFERENT lengths (a 1-image prompt with a short instruction, a 2-image prompt with a long one) run as one
    right-padded batch (attention_mask) - prefill, greedy tokens and the latent queries of every sequence must equal the per-env
    oracle (the reference's batch-1 semantics). Tokens are compared wherever the oracle's top-2 margin exceeds the logit tolerance."""
    from internnav_amd import synthetic
    from internnav_amd.policy import InternVLAN1ForCausalLM
    from oracle import qwen_vl as o_q

    cfg = W.QWEN_TEST_CFG
    sd = synthetic.materialize(synthetic.n1_full_spec(cfg), seed=6)
    qsd = {k: v for k, v in sd.items() if not k.startswith("model.traj_dit") and not k.startswith("model.rgb_")}
    a = synthetic.qwen_inputs(1, 1, seed=31, cfg=cfg, n_text=10, n_tail=5)
    b = synthetic.qwen_inputs(1, 2, seed=32, cfg=cfg, n_text=40, n_tail=9)
    La, Lb = a["input_ids"].shape[1], b["input_ids"].shape[1]
    assert La < Lb
    ids = torch.zeros(2, Lb, dtype=torch.long)
    mask = torch.zeros(2, Lb, dtype=torch.long)
    ids[0, :La], mask[0, :La] = a["input_ids"][0], 1
    ids[1], mask[1] = b["input_ids"][0], 1
    pv = torch.cat([a["pixel_values"], b["pixel_values"]])
    grid = torch.cat([a["grid_thw"], b["grid_thw"]])
    m = InternVLAN1ForCausalLM(sd, cfg, "nextdit_async", device=DEV, max_envs=2, max_seq_len=768, max_patches=pv.shape[0])
    n_new = 3
    seqs = m.generate(input_ids=ids, pixel_values=pv, image_grid_thw=grid, attention_mask=mask, max_new_tokens=n_new, do_sample=False,
                      return_dict_in_generate=True).sequences.cpu()
    lat = m.generate_latents(seqs.to(DEV), pv, grid).float().cpu()
    for r, (inp, L) in enumerate(((a, La), (b, Lb))):
        assert torch.equal(seqs[r, :L], inp["input_ids"][0])
        with torch.no_grad():
            ref = o_q.generate(qsd, cfg, inp["input_ids"], inp["pixel_values"], inp["grid_thw"], n_new)
            cur, same = inp["input_ids"], True
            for j in range(n_new):                           # token by token, teacher-forced on the oracle's own continuation
                logits, _ = o_q.forward_logits(qsd, cfg, cur, inp["pixel_values"], inp["grid_thw"])
                top2 = logits[0, -1].topk(2).values
                if same and int(seqs[r, L + j]) != int(ref[0, L + j]):
                    assert float(top2[0] - top2[1]) < 0.05, f"seq {r} token {j} differs although the oracle margin is {float(top2[0] - top2[1]):.3f}"
                    same = False
                cur = ref[:, : L + j + 1]
            if same:
                rl = o_q.generate_latents(qsd, cfg, ref, inp["pixel_values"], inp["grid_thw"])
                d = (lat[r] - rl[0]).abs()
                print(f"ragged seq {r} (len {L}): tokens equal, latents mean|err| {d.mean():.3e} max|err| {d.max():.3e}")
                assert d.mean() < 1e-2 * rl.pow(2).mean().sqrt()
    # the padded batch and the two single-sequence calls agree (different GEMM tile shapes: tolerance, not bit equality)
    for r, (inp, L) in enumerate(((a, La), (b, Lb))):
        s1 = m.generate(input_ids=inp["input_ids"], pixel_values=inp["pixel_values"], image_grid_thw=inp["grid_thw"], max_new_tokens=n_new,
                        do_sample=False, return_dict_in_generate=True).sequences
        l1 = m.generate_latents(s1, inp["pixel_values"], inp["grid_thw"]).float().cpu()
        if torch.equal(s1.cpu()[0], seqs[r, : L + n_new]):
            assert (l1[0] - lat[r]).abs().mean() < 1e-2 * l1.pow(2).mean().sqrt()
