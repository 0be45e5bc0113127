"""GPU: the agent constructed from an AgentCfg-like config ALONE on a synthetic checkpoint ON DISK (safetensors shards with the
reference's parameter names + config.json + tokenizer files) - `InternVLAN1ForCausalLM.from_pretrained`, the config-only policy
constructor and the batched agent step end to end on the HIP engines - plus the capacity case of ADVICE r1 (8 history frames + the
current frame at 384 x 384 and an UN-resized 640 x 480 look-down frame in one System-2 prompt).

The checkpoint has the true layer widths at reduced depth / vocabulary (synthetic.QWEN_TEST_CFG) so it is written in seconds. The
Qwen2.5-VL AutoProcessor cannot be instantiated in this image (its video processor needs torchvision), so the checkpoint's tokenizer
is loaded with AutoTokenizer and images go through the bit-exact device pre-processor (`device_preprocess`), which needs only the
tokenizer + chat template of the processor."""
import json
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from internnav_amd import synthetic as S

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
CHAT = ("{% for message in messages %}<|im_start|>{{ message['role'] }}\n{% for c in message['content'] %}{% if c['type'] == 'image' %}"
        "<|vision_start|><|image_pad|><|vision_end|>{% else %}{{ c['text'] }}{% endif %}{% endfor %}<|im_end|>\n{% endfor %}"
        "{% if add_generation_prompt %}<|im_start|>assistant\n{% endif %}")


def _write_tokenizer(path, cfg):
    """byte-level tokenizer whose special tokens sit at the checkpoint's ids (image 4001, traj 4002, vision start/end 4003/4004, eos 4005)."""
    from tokenizers import Tokenizer, decoders, models, pre_tokenizers
    from transformers import PreTrainedTokenizerFast

    vocab = {c: i for i, c in enumerate(sorted(pre_tokenizers.ByteLevel.alphabet()))}
    tok = Tokenizer(models.BPE(vocab=vocab, merges=[], unk_token=None))
    tok.pre_tokenizer = pre_tokenizers.ByteLevel(add_prefix_space=False)
    tok.decoder = decoders.ByteLevel()
    fast = PreTrainedTokenizerFast(tokenizer_object=tok)
    fast.add_tokens([f"<|fill{i}|>" for i in range(len(vocab), cfg["image_token_id"])], special_tokens=True)
    fast.add_special_tokens({"additional_special_tokens": ["<|image_pad|>", "<|traj|>", "<|vision_start|>", "<|vision_end|>", "<|im_end|>", "<|im_start|>"]})
    assert fast.convert_tokens_to_ids(["<|image_pad|>", "<|traj|>", "<|vision_start|>", "<|vision_end|>", "<|im_end|>"]) == \
        [cfg["image_token_id"], cfg["traj_token_id"], cfg["vision_start_id"], cfg["vision_end_id"], cfg["eos_token_id"]]
    fast.eos_token = fast.pad_token = "<|im_end|>"
    fast.chat_template = CHAT
    fast.save_pretrained(str(path))


class _ScriptedDecode:
    """the checkpoint's real tokenizer, except that decode() of an ANSWER returns a scripted System-2 reply: a random-weight LLM emits
    noise, and the point of the agent test is to drive the pixel-goal / look-down / discrete-action branches through the real engines."""
    ANSWERS = ["215 206", "↑↑→", "↓", "180 122", "←", "77"]

    def __init__(self, tok):
        self._tok, self.n = tok, 0

    def __call__(self, *a, **k):
        return self._tok(*a, **k)

    def __getattr__(self, name):
        return getattr(self._tok, name)

    def decode(self, ids, skip_special_tokens=True):
        self.n += 1
        return self.ANSWERS[(self.n - 1) % len(self.ANSWERS)]


def _load_processor(model_path, scripted=False):
    from transformers import AutoTokenizer

    tok = AutoTokenizer.from_pretrained(model_path, use_fast=True)
    tok.padding_side = "left"
    if scripted:
        tok = _ScriptedDecode(tok)
    return SimpleNamespace(tokenizer=tok, image_token="<|image_pad|>",
                           apply_chat_template=lambda conv, tokenize=False, add_generation_prompt=True: tok.apply_chat_template(
                               conv, tokenize=tokenize, add_generation_prompt=add_generation_prompt))


@pytest.fixture(scope="module")
def ckpt(built_lib, tmp_path_factory):
    d = tmp_path_factory.mktemp("n1_ckpt")
    sd = S.write_checkpoint(d, S.QWEN_TEST_CFG, "nextdit_async", seed=5)
    _write_tokenizer(d, S.QWEN_TEST_CFG)
    return d, sd


def _settings(path, **kw):
    ms = {"policy_name": "InternVLAN1_Policy", "state_encoder": None, "env_num": 2, "model_path": str(path), "width": 640, "height": 480, "hfov": 79,
          "resize_w": 384, "resize_h": 384, "num_history": 8, "device": DEV, "continuous_traj": True, "infer_mode": "partial_async",
          "vis_debug": False, "device_preprocess": True}
    ms.update(kw)
    return ms


def _obs(rng, n):
    return [{"rgb": rng.integers(0, 256, (480, 640, 3), dtype=np.uint8), "depth": rng.random((480, 640, 1), dtype=np.float32) * 0.6,
             "instruction": "walk past the sofa and stop at the door"} for _ in range(n)]


def test_from_pretrained_equals_direct_construction(ckpt):
    """key-name mapping of a checkpoint on disk: the model loaded with from_pretrained computes exactly what a model built from the
    same state dict in memory computes."""
    from internnav_amd.policy import InternVLAN1ForCausalLM

    d, sd = ckpt
    cfg = S.QWEN_TEST_CFG
    inp = S.qwen_inputs(2, 2, seed=9, cfg=cfg)
    a = InternVLAN1ForCausalLM.from_pretrained(str(d), torch_dtype=torch.bfloat16, attn_implementation="flash_attention_2", device_map={"": DEV},
                                               max_envs=2, max_seq_len=512, max_patches=inp["pixel_values"].shape[0])
    b = InternVLAN1ForCausalLM(sd, cfg, "nextdit_async", device=DEV, max_envs=2, max_seq_len=512, max_patches=inp["pixel_values"].shape[0])
    assert a.config.system1 == "nextdit_async" and a.qwen.cfg == cfg
    outs = []
    for m in (a, b):
        seq = m.generate(input_ids=inp["input_ids"], pixel_values=inp["pixel_values"], image_grid_thw=inp["grid_thw"], max_new_tokens=4,
                         do_sample=False, return_dict_in_generate=True).sequences
        lat = m.generate_latents(seq, inp["pixel_values"], inp["grid_thw"])
        s1 = S.n1_nextdit_inputs(2, seed=3)
        traj = m.generate_traj(lat, s1["images"].to(DEV), None, noise=dict(x_init=s1["x_init"].to(DEV)))
        outs.append((seq.cpu(), lat.float().cpu(), traj.float().cpu()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]) and torch.equal(outs[0][2], outs[1][2])
    with pytest.raises(FileNotFoundError):
        InternVLAN1ForCausalLM.from_pretrained(str(d / "nowhere"))


@pytest.mark.parametrize("s1_cfg", [S.N1_NEXTDIT_CFG, S.N1_NEXTDIT_CFG_FFN1024, dict(S.N1_NEXTDIT_CFG, dit_layers=3, dit_ffn=768)],
                         ids=["ffn1536", "ffn1024", "3-layer-ffn768"])
def test_from_pretrained_reads_the_dit_geometry_off_the_checkpoint(built_lib, tmp_path, s1_cfg):
    """nothing about the trajectory DiT is in config.json (NextDiTCrossAttnConfig is built in code, internvla_n1_arch.py:127-131) and its FFN
    width depends on the diffusers release the checkpoint was trained under: a checkpoint written at either width - or any other geometry -
    loads with no configuration edit, and computes what a directly constructed engine with that geometry computes."""
    from internnav_amd.nextdit import NextDiTSystem1
    from internnav_amd.policy import InternVLAN1ForCausalLM

    sd = S.write_checkpoint(tmp_path, S.QWEN_TEST_CFG, "nextdit_async", seed=11, s1_cfg=s1_cfg)
    m = InternVLAN1ForCausalLM.from_pretrained(str(tmp_path), torch_dtype=torch.bfloat16, device_map={"": DEV}, max_envs=2, max_seq_len=512, max_patches=1568)
    assert m.s1.cfg == s1_cfg, (m.s1.cfg, s1_cfg)
    assert m.s1.layers[0]["w2"].shape == (s1_cfg["dit_dim"], s1_cfg["dit_ffn"]) and len(m.s1.layers) == s1_cfg["dit_layers"]
    s1 = S.n1_nextdit_inputs(2, seed=3)
    lat = s1["traj_latents"].to(DEV, torch.bfloat16)
    traj = m.generate_traj(lat, s1["images"].to(DEV), None, noise=dict(x_init=s1["x_init"].to(DEV))).float().cpu()
    direct = NextDiTSystem1({k[len("model."):]: v for k, v in sd.items() if k.startswith("model.")}, s1_cfg, DEV, max_envs=2)
    ref = direct.generate_traj(lat, s1["images"].to(DEV), s1["x_init"].to(DEV)).float().cpu()
    assert torch.isfinite(traj).all() and torch.equal(traj.view_as(ref), ref)
    # a checkpoint whose tensors disagree with each other is refused by name, not loaded into the wrong geometry
    from safetensors.torch import load_file, save_file
    f = sorted(tmp_path.glob("*.safetensors"))
    for shard in f:
        t = load_file(str(shard))
        k = "model.traj_dit.model.layers.1.feed_forward.linear_2.weight"
        if k in t:
            t[k] = t[k][:, :-128].contiguous()
            save_file(t, str(shard))
    with pytest.raises(ValueError, match="NextDiT"):
        InternVLAN1ForCausalLM.from_pretrained(str(tmp_path), torch_dtype=torch.bfloat16, device_map={"": DEV}, max_envs=2, max_seq_len=512, max_patches=1568)


def test_agent_from_config_alone_steps_two_envs(ckpt, monkeypatch):
    from internnav_amd.agent import InternVLAN1Agent
    from internnav_amd.policy import InternVLAN1Net

    d, _ = ckpt
    monkeypatch.setattr(InternVLAN1Net, "load_processor", staticmethod(lambda path: _load_processor(path, scripted=True)))
    InternVLAN1Net._shared.clear()
    agent = InternVLAN1Agent(SimpleNamespace(model_name="internvla_n1", model_settings=_settings(d)))
    assert agent.model.qwen.S_max >= 2155 + 128 and agent.model.qwen.Np_max >= 2 * 8620     # sized for the harness defaults
    agent.reset()
    rng = np.random.default_rng(0)
    seen = []
    for step in range(12):
        out = agent.step(_obs(rng, 2))
        assert len(out) == 2 and all(o["ideal_flag"] is True and len(o["action"]) == 1 and o["action"][0] in (-1, 0, 1, 2, 3) for o in out)
        json.dumps(out)
        seen.append([o["action"][0] for o in out])
        if step == 5:
            agent.reset([1])
    print("actions", seen, "S2 STOP fallbacks", agent.s2_failures)
    flat = [a for row in seen for a in row]
    assert -1 in flat and any(a in (1, 2, 3) for a in flat)       # a look-down turn (-1) and System-1 / discrete motion actions both occurred
    assert agent.s2_failures <= 4                                 # only the scripted one-number answers ("77") end in the STOP fallback
    # a second agent in the same process re-uses the loaded engines (one set per GPU process)
    n_before = len(InternVLAN1Net._shared)
    InternVLAN1Agent(SimpleNamespace(model_name="internvla_n1", model_settings=_settings(d)))
    assert len(InternVLAN1Net._shared) == n_before == 1
    InternVLAN1Net._shared.clear()


def test_batched_agent_with_prefix_cache_equals_uncached_agent(ckpt, monkeypatch):
    """model_settings['prefix_cache'] through the batched agent: three envs with instructions of different lengths (ragged System-2 batches,
    per-env prefixes of different lengths, envs with and without a stored prefix in one call, a mid-run episode reset, look-down turns):
    every action and every latent equals the uncached agent's bit for bit, and the engine prefills fewer rows once prefixes exist."""
    from internnav_amd.agent import InternVLAN1Agent
    from internnav_amd.policy import InternVLAN1Net

    d, _ = ckpt
    monkeypatch.setattr(InternVLAN1Net, "load_processor", staticmethod(lambda path: _load_processor(path, scripted=True)))
    instr = ["go to the door", "walk past the sofa and stop at the door", "turn left at the kitchen, then enter the second bedroom on the right"]
    runs = []
    for pc in (False, True):
        InternVLAN1Net._shared.clear()                       # a fresh model per agent: same sampler-noise stream for both runs
        agent = InternVLAN1Agent(SimpleNamespace(model_name="internvla_n1", model_settings=_settings(d, env_num=3, prefix_cache=pc)))
        assert all(e.policy.prefix_cache == pc for e in [agent._env(i) for i in range(3)])
        agent.reset()
        rng = np.random.default_rng(7)
        acts, lats, widths = [], [], []
        for step in range(14):
            obs = _obs(rng, 3)
            for o, t in zip(obs, instr):
                o["instruction"] = t
            out = agent.step(obs)
            acts.append([o["action"][0] for o in out])
            lats.append([None if e.s2_output.output_latent is None else e.s2_output.output_latent.float().cpu().clone() for e in agent.envs])
            g = getattr(agent.model, "_gen", None)
            widths.append(None if g is None else int(g["state"]["S_run"]))
            if step == 6:
                agent.reset([2])
        runs.append((acts, lats, widths, agent.s2_failures))
    InternVLAN1Net._shared.clear()
    (a0, l0, w0, f0), (a1, l1, w1, f1) = runs
    assert a0 == a1 and f0 == f1, (a0, a1)
    for x, y in zip(l0, l1):
        for u, v in zip(x, y):
            assert (u is None) == (v is None) and (u is None or torch.equal(u, v))
    assert any(c is not None and u is not None and c < u for c, u in zip(w1, w0)), (w0, w1)     # some System-2 call ran a narrower rectangle


def test_full_history_plus_camera_size_lookdown_frame_fits(ckpt, monkeypatch):
    """step >= 8 of an episode: 8 history + current frame (196 tokens each) + the 640 x 480 look-down frame (34 x 46 patches, 391
    tokens): 2155 image tokens / 8620 patches in ONE prompt. Round 1 asserted here (max_seq_len 2048) and the agent turned it into STOP."""
    from internnav_amd.policy import InternVLAN1Net

    d, _ = ckpt
    monkeypatch.setattr(InternVLAN1Net, "load_processor", staticmethod(_load_processor))
    from internnav_amd.policy import InternVLAN1ModelConfig

    net = InternVLAN1Net(config=InternVLAN1ModelConfig(model_cfg={"model": _settings(d)}))
    rng = np.random.default_rng(1)
    for _ in range(9):
        net.step_no_infer(rng.integers(0, 256, (480, 640, 3), dtype=np.uint8), None, None)
    frame = rng.integers(0, 256, (480, 640, 3), dtype=np.uint8)
    inputs = net.build_s2_inputs(frame, "go to the kitchen", look_down=False)
    assert inputs["image_grid_thw"].tolist() == [[1, 28, 28]] * 9
    net.llm_output = "↓"
    inputs = net.build_s2_inputs(frame, "go to the kitchen", look_down=True)
    assert inputs["image_grid_thw"].tolist() == [[1, 28, 28]] * 9 + [[1, 34, 46]]
    n_img_tok = int((inputs["input_ids"] == S.QWEN_TEST_CFG["image_token_id"]).sum())
    assert n_img_tok == 9 * 196 + 391 and inputs["pixel_values"].shape[0] == 8620
    seq = net.model.generate(input_ids=inputs["input_ids"], pixel_values=inputs["pixel_values"], image_grid_thw=inputs["image_grid_thw"],
                             max_new_tokens=128, do_sample=False, return_dict_in_generate=True).sequences
    assert seq.shape[1] > inputs["input_ids"].shape[1]
    lat = net.model.generate_latents(seq, inputs["pixel_values"], inputs["image_grid_thw"])
    assert lat.shape == (1, 4, 3584) and bool(torch.isfinite(lat.float()).all())


def test_capacity_overflow_is_a_loud_configuration_error(ckpt):
    from internnav_amd.policy import InternVLAN1ForCausalLM
    from internnav_amd.runtime import CapacityError

    d, sd = ckpt
    cfg = S.QWEN_TEST_CFG
    m = InternVLAN1ForCausalLM(sd, cfg, "nextdit_async", device=DEV, max_envs=1, max_seq_len=256, max_patches=784)
    inp = S.qwen_inputs(1, 2, seed=9, cfg=cfg)
    with pytest.raises(CapacityError):
        m.generate(input_ids=inp["input_ids"], pixel_values=inp["pixel_values"], image_grid_thw=inp["grid_thw"], max_new_tokens=4)
    with pytest.raises(NotImplementedError):            # not one of the four System-1 types of generate_traj (internvla_n1.py:359-441)
        InternVLAN1ForCausalLM(sd, cfg, "unet", device=DEV, max_envs=1, max_seq_len=256, max_patches=784)
    # the plain (non-async) 'nextdit' type: generate_traj ignores the images, classifier-free guidance weights other than 1 are accepted
    plain = InternVLAN1ForCausalLM(sd, cfg, "nextdit", device=DEV, max_envs=1, max_seq_len=256, max_patches=784)
    lat = torch.randn(1, cfg["n_query"], cfg["t_hidden"], device=DEV).to(torch.bfloat16)
    t1 = plain.generate_traj(traj_latents=lat, images_dp=None, noise=dict(x_init=torch.zeros(1, 32, 32, 3, device=DEV)))
    t2 = plain.generate_traj(traj_latents=lat, images_dp=None, guidance_scale=2.0, noise=dict(x_init=torch.zeros(1, 32, 32, 3, device=DEV)))
    assert t1.shape == (32, 32, 3) and torch.isfinite(t1).all() and torch.isfinite(t2).all() and (t1 - t2).abs().max().item() > 0
