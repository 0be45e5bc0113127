"""Drop-in model / policy surface of InternVLA-N1 on the MI355X engines.

Mirrors, with the same names, arguments and return types, what the reference's callers use:
  * `InternVLAN1ForCausalLM`  <- internnav/model/basemodel/internvla_n1/internvla_n1.py:39 : `.eval()`, `.device`,
        `.generate(**inputs, max_new_tokens, do_sample=False, ...).sequences`, `.generate_latents(output_ids, pixel_values,
        image_grid_thw)`, `.generate_traj(traj_latents, images_dp, depths_dp)`
        (call sites habitat_vln_evaluator.py:418-459, internvla_n1_policy.py:169-203, internvla_n1_agent_realworld.py:221-256)
  * `InternVLAN1Net`          <- internnav/model/basemodel/internvla_n1/internvla_n1_policy.py:26 : `reset`, `step_no_infer`,
        `s2_step`, `s1_step_latent`
  * host post-processing `traj_to_actions`, `chunk_token`, `split_and_clean`, `S1Output`, `S2Output`, `S2Input`
        <- internnav/model/utils/vln_utils.py:19-175 (numpy / python; stays on the host in the reference too).
The tokenizer / image processor (a1 in SURVEY.md 8a: HF `AutoProcessor`, third-party, host side) is injected; the kernel
boundary starts at `input_ids` / `pixel_values` / `image_grid_thw`.

Unlike the reference (one env per call) every method here also accepts a batch of environments; batch-1 calls keep the
reference's exact shapes (e.g. generate_traj returns [32*B, T, 3]).
"""
from __future__ import annotations

import copy
import itertools
import re
from collections import OrderedDict
from dataclasses import dataclass
from types import SimpleNamespace
from typing import Any, Dict, List, Optional, Tuple

import numpy as np
import torch

from . import synthetic
from .navdp import NavDPPolicyDAT
from .nextdit import NextDiTSystem1
from .qwen_vl import QwenVLEngine
from .runtime import CapacityError  # noqa: F401  (re-exported)


# ------------------------------------------------------------------------------------------------------ vln_utils mirror
def split_and_clean(text: str) -> List[str]:
    """vln_utils.py:19-33: split on <image>, drop newlines / empty parts."""
    out = []
    for part in re.split(r"(<image>)", text):
        if part == "<image>":
            out.append(part)
        else:
            c = part.replace("\n", "").strip()
            if c:
                out.append(c)
    return out


def chunk_token(dp_actions) -> List[int]:
    """vln_utils.py:36-60: per-waypoint discretisation (0 stop, 1 forward, 2 left, 3 right)."""
    out = []
    for xyyaw in dp_actions:
        x, yaw = float(xyyaw[0]), float(xyyaw[-1])
        if x < 0.05 and abs(yaw) < 0.05:
            out.append(0)
        elif abs(x / 0.25) >= abs(yaw * 12 / np.pi):
            out.append(1)
        else:
            out.append(3 if yaw < 0 else 2)
    return out


def traj_to_actions(dp_actions, use_discrate_action: bool = True):
    """vln_utils.py:63-136. dp_actions [S, T, 3] (x4-scaled increments): un-normalise IN PLACE (the reference mutates its input,
    :129), cumulative sum, mean over the S samples, greedy pure-pursuit discretisation (0.25 m steps, 15 degree turns, lookahead 4)."""
    dp_actions[:, :, :2] /= 4.0
    d = dp_actions.float().cpu().numpy() if isinstance(dp_actions, torch.Tensor) else np.asarray(dp_actions, dtype=np.float64)
    B, T = d.shape[:2]
    xy = np.zeros((B, T + 1, 2))
    xy[:, 1:] = np.cumsum(d[:, :, :2], axis=1)
    trajectory = xy.mean(axis=0)
    if not use_discrate_action:
        return trajectory
    actions: List[int] = []
    yaw, pos, goal = 0.0, trajectory[0], trajectory[-1]
    turn = np.deg2rad(15)

    def norm_angle(a):
        return (a + np.pi) % (2 * np.pi) - np.pi

    while np.linalg.norm(pos - goal) > 0.2:
        nearest = int(np.argmin(np.linalg.norm(trajectory - pos, axis=1)))
        target = trajectory[min(nearest + 4, len(trajectory) - 1)]
        tdir = target - pos
        if np.linalg.norm(tdir) < 1e-6:
            break
        n_turns = int(round(norm_angle(np.arctan2(tdir[1], tdir[0]) - yaw) / turn))
        if n_turns > 0:
            actions += [2] * n_turns
        elif n_turns < 0:
            actions += [3] * (-n_turns)
        yaw = norm_angle(yaw + n_turns * turn)
        nxt = pos + 0.25 * np.array([np.cos(yaw), np.sin(yaw)])
        if np.linalg.norm(nxt - goal) > np.linalg.norm(pos - goal):
            break
        actions.append(1)
        pos = nxt
    return actions


@dataclass
class S2Input:
    idx: Optional[int] = -1
    instruction: Optional[str] = None
    rgb: Optional[np.ndarray] = None
    depth: Optional[np.ndarray] = None
    pose: Optional[Any] = None
    look_down: Optional[bool] = False
    should_infer: Optional[bool] = False


@dataclass
class S2Output:
    idx: Optional[int] = -1
    is_infering: Optional[bool] = False
    output_action: Optional[Any] = None
    output_trajectory: Optional[np.ndarray] = None
    output_pixel: Optional[np.ndarray] = None
    output_latent: Optional[torch.Tensor] = None
    rgb_memory: Optional[np.ndarray] = None
    depth_memory: Optional[np.ndarray] = None

    def validate(self) -> bool:
        return sum(x is not None for x in (self.output_action, self.output_pixel, self.output_latent)) > 0 and self.idx >= 0


@dataclass
class S1Output:
    idx: Optional[List[int]] = None
    vis_image: Optional[np.ndarray] = None


# ------------------------------------------------------------------------------------------------------ model facade
def smart_resize(height: int, width: int, factor: int = 28, min_pixels: int = 56 * 56, max_pixels: int = 14 * 14 * 4 * 1280):
    """transformers image_processing_qwen2_vl.smart_resize (the processor's target size; restated in preprocess.py for the device path)."""
    import math

    h_bar, w_bar = round(height / factor) * factor, round(width / factor) * factor
    if h_bar * w_bar > max_pixels:
        beta = math.sqrt((height * width) / max_pixels)
        h_bar, w_bar = max(factor, math.floor(height / beta / factor) * factor), max(factor, math.floor(width / beta / factor) * factor)
    elif h_bar * w_bar < min_pixels:
        beta = math.sqrt(min_pixels / (height * width))
        h_bar, w_bar = math.ceil(height * beta / factor) * factor, math.ceil(width * beta / factor) * factor
    return h_bar, w_bar


def s2_capacity(num_history: int = 8, resize_w: int = 384, resize_h: int = 384, cam_w: int = 640, cam_h: int = 480,
                text_tokens: int = 512, max_new_tokens: int = 128, n_query: int = 4):
    """(max_seq_len, max_patches_per_seq) of the longest System-2 prompt the reference's callers can build: num_history history frames
    + the current frame at the resized size, + one look-down frame fed UN-resized at camera size (internvla_n1_policy.py:113-116,140),
    the chat text, the answer and the latent queries. Default harness settings: 9 x 196 + 391 image tokens + text = 2.8 K tokens."""
    rh, rw = smart_resize(resize_h, resize_w)
    ch, cw = smart_resize(cam_h, cam_w)
    p_hist, p_cam = (rh // 14) * (rw // 14), (ch // 14) * (cw // 14)
    patches = (num_history + 1) * p_hist + p_cam
    seq = patches // 4 + 2 * (num_history + 2) + text_tokens + max_new_tokens + n_query
    return (seq + 127) // 128 * 128, patches


def qwen_cfg_from_hf(cfgj: dict) -> dict:
    """engine configuration from a checkpoint's config.json (Qwen2.5-VL layout of transformers 4.51: text keys at the top level;
    newer exports nest them under text_config) - defaults are the Qwen2.5-VL-7B values InternVLA-N1 ships with."""
    t = dict(cfgj)
    t.update(cfgj.get("text_config") or {})
    v = cfgj.get("vision_config") or {}
    d = synthetic.QWEN_N1_CFG
    rope = t.get("rope_theta") or (t.get("rope_parameters") or {}).get("rope_theta") or d["rope_theta"]
    eos = t.get("eos_token_id", d["eos_token_id"])
    return dict(
        v_hidden=v.get("hidden_size", d["v_hidden"]), v_heads=v.get("num_heads", d["v_heads"]), v_inter=v.get("intermediate_size", d["v_inter"]),
        v_depth=v.get("depth", d["v_depth"]), v_fullatt=tuple(v.get("fullatt_block_indexes", d["v_fullatt"])), v_window=v.get("window_size", d["v_window"]),
        v_patch=v.get("patch_size", d["v_patch"]), v_out=v.get("out_hidden_size", d["v_out"]),
        t_hidden=t.get("hidden_size", d["t_hidden"]), t_inter=t.get("intermediate_size", d["t_inter"]), t_heads=t.get("num_attention_heads", d["t_heads"]),
        t_kv_heads=t.get("num_key_value_heads", d["t_kv_heads"]), t_layers=t.get("num_hidden_layers", d["t_layers"]), vocab=t.get("vocab_size", d["vocab"]),
        rope_theta=float(rope), n_query=cfgj.get("n_query", d["n_query"]),
        image_token_id=cfgj.get("image_token_id", d["image_token_id"]), traj_token_id=cfgj.get("traj_token_id", d["traj_token_id"]),
        vision_start_id=cfgj.get("vision_start_token_id", d["vision_start_id"]), vision_end_id=cfgj.get("vision_end_token_id", d["vision_end_id"]),
        eos_token_id=int(eos[0] if isinstance(eos, (list, tuple)) else eos))


class InternVLAN1ForCausalLM:
    """HF-style model object backed by the HIP engines (no nn.Module, no CPU fallback)."""

    def __init__(self, weights, qwen_cfg: dict, system1: str = "nextdit_async", s1_cfg: Optional[dict] = None,
                 device="cuda:0", max_envs: int = 16, max_seq_len: Optional[int] = None, max_patches: Optional[int] = None,
                 max_s2_seqs: Optional[int] = None, num_history: int = 8, resize_w: int = 384, resize_h: int = 384,
                 cam_w: int = 640, cam_h: int = 480):
        """Engine capacity defaults to the longest prompt the reference's harness can build for (num_history, resize, camera size):
        see `s2_capacity`; exceeding it raises `CapacityError` (never a silent STOP)."""
        self.device = torch.device(device)
        if system1 not in ("nextdit_async", "navdp_async", "nextdit", "navdp"):
            # the four System-1 types of generate_traj (internvla_n1.py:359-441): 'nextdit' [+ 'async'] and 'navdp' [+ 'async']
            raise NotImplementedError(f"system1={system1!r}: known types are 'nextdit_async' (DualVLN), 'navdp_async', 'nextdit', 'navdp'")
        self.config = SimpleNamespace(system1=system1, n_query=qwen_cfg["n_query"], hidden_size=qwen_cfg["t_hidden"],
                                      image_token_id=qwen_cfg["image_token_id"])
        n_s2 = max_s2_seqs or max_envs   # System-2 runs on micro-batches of the envs whose plan expired (agent / bench schedule)
        cap_seq, cap_patches = s2_capacity(num_history, resize_w, resize_h, cam_w, cam_h, n_query=qwen_cfg["n_query"])
        self.qwen = QwenVLEngine(weights, qwen_cfg, device, max_seqs=n_s2, max_seq_len=max_seq_len or cap_seq,
                                 max_patches=max_patches or n_s2 * cap_patches)
        if "nextdit" in system1:
            # the DiT's geometry (width, depth, heads, FFN width) is not in config.json - NextDiTCrossAttnConfig is constructed in code
            # (internvla_n1_arch.py:127-131) and its FFN width depends on the diffusers release (synthetic.lumina_ffn_width): read it off
            # the checkpoint's tensor shapes
            s1w = _Prefixed(weights, "model.")
            self.s1 = NextDiTSystem1(s1w, s1_cfg or synthetic.n1_nextdit_cfg_from_weights(s1w), device, max_envs, use_async="async" in system1)
        else:
            self.s1 = NavDPPolicyDAT(_Prefixed(weights, "model.navdp."), s1_cfg or synthetic.N1_NAVDP_CFG, device, max_envs, use_async="async" in system1)
        self._noise_gen = torch.Generator(device=self.device).manual_seed(0)

    # ---- construction
    @classmethod
    def from_pretrained(cls, path, torch_dtype=torch.bfloat16, attn_implementation: str = "flash_attention_2", device_map=None, **kw):
        """reference call: InternVLAN1ForCausalLM.from_pretrained(path, torch_dtype=bf16, attn_implementation=..., device_map={"": dev}).
        Loads every *.safetensors shard under `path` (HF checkpoint layout, the reference's parameter names); config.json supplies the
        Qwen2.5-VL dimensions / token ids, system1 and n_query (InternVLAN1ModelConfig fields, internvla_n1.py:22-29)."""
        import json
        from pathlib import Path

        from safetensors import safe_open

        p = Path(path)
        files = sorted(p.glob("*.safetensors"))
        if not files:
            raise FileNotFoundError(f"no *.safetensors under {p}: InternVLA-N1 checkpoints are HF safetensors shards")
        cfgj = json.loads((p / "config.json").read_text()) if (p / "config.json").exists() else {}
        device = (device_map or {"": "cuda:0"})[""]
        weights = _ShardedCheckpoint(files)
        system1 = cfgj.get("system1", "nextdit_async")
        s1_cfg = None
        if "nextdit" in system1:
            probe = "model.traj_dit.model.layers.0.feed_forward.linear_1.weight"
            if probe not in weights:
                raise KeyError(f"checkpoint {p} has no {probe}: not an InternVLA-N1 '{system1}' checkpoint")
            s1_cfg = synthetic.n1_nextdit_cfg_from_weights(_Prefixed(weights, "model."))
        spec = synthetic.n1_full_spec(qwen_cfg_from_hf(cfgj), system1, s1_cfg=s1_cfg)
        missing = [k for k in spec if k not in weights]
        if missing:
            raise KeyError(f"checkpoint {p} lacks {len(missing)} parameters the engines need, e.g. {missing[:4]}")
        bad = [(k, weights.shape_of(k), shp) for k, (shp, _) in spec.items() if k.startswith("model.traj_dit.") and tuple(weights.shape_of(k)) != tuple(shp)]
        if bad:
            raise ValueError(f"checkpoint {p}: {len(bad)} NextDiT tensors do not have the shapes its own geometry implies, e.g. {bad[:3]}")
        return cls(weights, qwen_cfg_from_hf(cfgj), system1=system1, s1_cfg=s1_cfg, device=device, **kw)

    def eval(self):
        return self

    def get_n_query(self):
        return self.config.n_query

    def get_system1_type(self):
        return self.config.system1

    # ---- System 2
    def generate(self, input_ids=None, pixel_values=None, image_grid_thw=None, attention_mask=None, max_new_tokens: int = 128,
                 do_sample: bool = False, use_cache: bool = True, past_key_values=None, return_dict_in_generate: bool = False,
                 decode_chunk: int = 8, eos_token_id=None, cached_image_embeds: Optional[list] = None, prefix_kv: Optional[list] = None,
                 export_prefix: Optional[list] = None, **_):
        """greedy decoding (do_sample=False is the only mode the reference uses, internvla_n1_policy.py:169-176). Decodes in chunks of
        `decode_chunk` device-side steps and stops once every sequence has emitted EOS. Sequences are right-filled with EOS.
        cached_image_embeds (extension, per-frame ViT cache): one entry per image of the batch, None = encode it (its patches are in
        pixel_values), else the embeddings an earlier call returned through `last_image_embeds()`.
        prefix_kv (extension, prefix-KV reuse): one entry per sequence - None, or the K/V bf16 [layers, P, 1024] of the sequence's first P
        prompt tokens as an earlier call exported them (the prompt must start with the same P tokens: system prompt + instruction +
        first history frame between the System-2 calls of an episode). Those tokens are not run and their images not encoded;
        pixel_values may still hold every image's patches (the cached images' rows are skipped). Exact: causal attention.
        export_prefix: one entry per sequence - 0, or the number of leading prompt tokens whose K/V to keep (`last_prefix_kv()`)."""
        assert not do_sample, "the reference only decodes greedily"
        eos = self.qwen.cfg["eos_token_id"] if eos_token_id is None else eos_token_id
        pv = pixel_values.to(self.device, torch.bfloat16) if pixel_values is not None and pixel_values.numel() else None
        B, S = input_ids.shape
        # ragged batch (extension; the reference is batch 1): prompts RIGHT-padded to a common length, real lengths from attention_mask
        plens = np.full(B, S, dtype=np.int64) if attention_mask is None else np.asarray(attention_mask.cpu().long().sum(1), dtype=np.int64)
        if attention_mask is not None:
            am = attention_mask.cpu().long()
            assert bool((am[:, 1:] <= am[:, :-1]).all()), "ragged System-2 batches are right-padded (mask = 1...1 0...0)"
        pl = 0
        if prefix_kv is not None and any(k is not None for k in prefix_kv):
            assert len(prefix_kv) == B and cached_image_embeds is None, "prefix_kv: one entry per sequence (not combined with cached_image_embeds)"
            pl = np.asarray([0 if k is None else int(k.shape[1]) for k in prefix_kv], dtype=np.int64)
            if int(pl.max()) + S - int(pl.min()) + max_new_tokens + self.qwen.latent_q.shape[0] > self.qwen.S_max:
                # prefixes of different lengths widen the rectangle the engine runs (every sequence's rows start behind ITS prefix): if that
                # no longer fits the cache, run this call without the prefix caches - same results, only the saving is lost
                prefix_kv, pl = [None] * B, np.zeros(B, dtype=np.int64)
            for b, k in enumerate(prefix_kv):
                if k is not None:
                    self.qwen.import_prefix_kv(b, k)
            # drop the patch rows of the images whose tokens are cached K/V
            skip = self.qwen.images_in_prefix(input_ids, image_grid_thw, pl)
            if pv is not None and any(skip):
                n_rows = [int(t * h * w) for t, h, w in image_grid_thw.tolist()]
                assert pv.shape[0] == sum(n_rows), "prefix_kv: pixel_values must hold the patches of every image of the batch"
                off = np.concatenate([[0], np.cumsum(n_rows)])
                keep = [pv[off[i]:off[i + 1]] for i, s in enumerate(skip) if not s]
                pv = torch.cat(keep, 0) if keep else None
        state = self.qwen.prefill(input_ids, pv, image_grid_thw, cached_embeds=cached_image_embeds, seq_lens=plens if attention_mask is not None else None,
                                  prefix_len=pl)
        self._prefix_out = {}
        if export_prefix is not None:
            for b, n in enumerate(export_prefix):
                if n:
                    self._prefix_out[b] = self.qwen.export_prefix_kv(b, int(n))
        self._fresh = self.qwen.fresh_image_embeds(state["plan"]) if cached_image_embeds is not None else {}
        chunks, n = [], 0
        while n < max_new_tokens:
            k = min(decode_chunk, max_new_tokens - n)
            t = self.qwen.decode(state, k + 1 if n else k)  # later chunks re-emit the pending token first
            t = t[:, 1:] if n else t
            chunks.append(t.cpu().long())
            n += k
            if bool((torch.cat(chunks, dim=1) == eos).any(dim=1).all()):
                break
        toks = torch.cat(chunks, dim=1)
        lens = []
        for b in range(toks.shape[0]):
            hit = (toks[b] == eos).nonzero()
            e = int(hit[0]) + 1 if hit.numel() else toks.shape[1]
            toks[b, e:] = eos
            lens.append(e)
        self._gen = dict(state=state, tokens=toks, lens=np.asarray(lens), prompt_lens=plens)
        # every row = its own prompt, then its answer, right-filled with EOS to the common width S + n
        ids_cpu = input_ids.cpu().long()
        seqs = torch.full((B, S + toks.shape[1]), eos, dtype=torch.long)
        for b in range(B):
            L = int(plens[b])
            seqs[b, :L] = ids_cpu[b, :L]
            seqs[b, L:L + toks.shape[1]] = toks[b]
        seqs = seqs.to(self.device)
        return SimpleNamespace(sequences=seqs) if return_dict_in_generate else seqs

    def last_prefix_kv(self) -> Dict[int, torch.Tensor]:
        """sequence index of the last generate() call -> the prefix K/V it was asked to export (`export_prefix`)."""
        return getattr(self, "_prefix_out", {})

    def last_image_embeds(self) -> Dict[int, torch.Tensor]:
        """image index (position in the last generate() call's image list) -> merged embeddings bf16 [tokens, 3584] of every image that
        call encoded; only filled when generate() was given `cached_image_embeds` (i.e. by callers that keep a frame cache)."""
        return getattr(self, "_fresh", {})

    def generate_latents(self, output_ids, pixel_values, image_grid_thw, cached_image_embeds: Optional[list] = None, rows=None):
        """[B, N_QUERY, 3584] hidden states of the latent trajectory queries (internvla_n1.py:320-347). When called right after
        generate() on its own output (the reference's only usage, internvla_n1_policy.py:191) the KV cache is reused: the queries
        are placed behind each sequence's last kept token; otherwise the full prompt is re-run.
        rows (extension for the batched agent): the batch rows whose answer was a pixel goal - only their latents are returned
        ([len(rows), N_QUERY, 3584]); the pass itself streams the weights once whatever the number of rows."""
        if rows is not None:
            return self.generate_latents(output_ids, pixel_values, image_grid_thw, cached_image_embeds)[torch.as_tensor(list(rows), dtype=torch.long, device=self.device)]
        g = getattr(self, "_gen", None)
        if g is not None and output_ids.shape[0] == g["tokens"].shape[0]:
            pl, lens, toks = g["prompt_lens"], g["lens"], g["tokens"]
            oc = output_ids.cpu().long()
            nt = toks.shape[1]
            if all(int(pl[b]) + nt <= oc.shape[1] and torch.equal(oc[b, int(pl[b]):int(pl[b]) + nt], toks[b]) for b in range(toks.shape[0])):
                last = torch.stack([toks[b, lens[b] - 1] for b in range(toks.shape[0])]).to(self.device, torch.int32).view(-1, 1)
                return self.qwen.latents(g["state"], last.contiguous(), seq_lens=pl + lens - 1)
        pv = pixel_values.to(self.device, torch.bfloat16) if pixel_values is not None and pixel_values.numel() else None
        return self.qwen.generate_latents(output_ids, pv, image_grid_thw, cached_embeds=cached_image_embeds)

    # ---- System 1
    def generate_traj(self, traj_latents, images_dp, depths_dp=None, predict_step_nums: int = 32, guidance_scale: float = 1.0,
                      num_inference_steps: int = 10, num_sample_trajs: int = 32, noise: Optional[dict] = None):
        """[32*B, T, 3] sampled trajectories (internvla_n1.py:349-441). `noise` = dict(x_init[, step_noise]) makes the sampler
        reproducible / checkable; by default it is drawn on the device like the reference's randn_tensor / torch.randn."""
        B = traj_latents.shape[0]
        s1 = self.s1
        S, T = s1.S, s1.T
        x_init = noise["x_init"] if noise else torch.randn(B, S, T, 3, device=self.device, generator=self._noise_gen)
        lat = traj_latents.to(self.device, torch.bfloat16)
        if isinstance(s1, NextDiTSystem1):
            assert num_inference_steps == s1.cfg["num_inference_steps"] and predict_step_nums == T and num_sample_trajs == S
            # without 'async' the images are not part of the condition (internvla_n1.py:382-383): whatever the caller passes is ignored
            img = images_dp.to(self.device) if (s1.use_async and torch.is_tensor(images_dp)) else None
            out = s1.generate_traj(lat, img, x_init, guidance_scale=guidance_scale)
        else:
            K = s1.cfg["num_train_timesteps"]
            sn = noise["step_noise"] if noise else torch.randn(K, B, S, T, 3, device=self.device, generator=self._noise_gen)
            if s1.use_async:
                out = s1.predict_pointgoal_action_async(lat, images_dp.to(self.device), depths_dp.to(self.device), x_init, sn)
            else:                                       # internvla_n1.py:438-439: predict_pointgoal_action(traj_latents) - no images
                out = s1.predict_pointgoal_action(lat, x_init, sn)
        return out.reshape(B * S, T, 3).clone()


class _Prefixed:
    """view of a weight mapping under a key prefix (checkpoints carry the System-1 modules under `model.`)."""

    def __init__(self, base, prefix):
        self.base, self.prefix = base, prefix

    def __getitem__(self, k):
        return self.base[self.prefix + k]

    def get(self, k, default=None):
        try:
            return self.base[self.prefix + k]
        except KeyError:
            return default

    def __contains__(self, k):
        return (self.prefix + k) in self.base

    def shape_of(self, k):
        b = self.base
        return tuple(b.shape_of(self.prefix + k)) if hasattr(b, "shape_of") else tuple(b[self.prefix + k].shape)


class _ShardedCheckpoint:
    """key -> tensor over a list of safetensors shards, read lazily (one tensor on the host at a time: the engines repack and upload
    each parameter as they read it, so a 16 GB checkpoint never sits in host memory twice)."""

    def __init__(self, files):
        from safetensors import safe_open

        self._where = {}
        for f in files:
            with safe_open(str(f), framework="pt", device="cpu") as sf:
                for k in sf.keys():
                    self._where[k] = str(f)

    def __contains__(self, k):
        return k in self._where

    def keys(self):
        return self._where.keys()

    def __getitem__(self, k):
        from safetensors import safe_open

        with safe_open(self._where[k], framework="pt", device="cpu") as sf:
            return sf.get_tensor(k)

    def shape_of(self, k):
        """tensor shape from the shard header (no tensor data is read)."""
        from safetensors import safe_open

        with safe_open(self._where[k], framework="pt", device="cpu") as sf:
            return tuple(sf.get_slice(k).get_shape())


# ------------------------------------------------------------------------------------------------------ policy wrapper
class InternVLAN1ModelConfig:
    """`InternVLAN1ModelConfig` as the agent layer uses it (internvla_n1_agent.py:40-43: `policy_config(model_cfg={'model': ModelCfg.model_dump()})`):
    a holder of `model_cfg`; the HF PretrainedConfig machinery of the reference's class (internvla_n1.py:22-29) is not needed here."""
    model_type = "internvla_n1"

    def __init__(self, model_cfg: Optional[dict] = None, **kwargs):
        self.model_cfg = model_cfg or {"model": {}}
        for k, v in kwargs.items():
            setattr(self, k, v)


class InternVLAN1Net:
    """`InternVLAN1Net` of the reference (internvla_n1_policy.py:26-215) for ONE environment's episode state; the batched agent
    (internnav_amd/agent.py) holds one instance per env that share a single model and batches their model calls."""

    PROMPT = ("You are an autonomous navigation assistant. Your task is to <instruction>. Where should you go next to stay on track? "
              "Please output the next waypoint's coordinates in the image. Please output STOP when you have successfully completed the task.")
    CONJUNCTION = "you can see "
    ACTIONS2IDX = OrderedDict({"STOP": [0], "↑": [1], "←": [2], "→": [3], "↓": [5]})

    _shared: Dict[Any, Any] = {}   # (model_path, device) -> (model, processor): one set of engines per GPU process, shared by all envs

    def __init__(self, config=None, processor=None, num_history: int = 8, resize_w: int = 384, resize_h: int = 384,
                 continuous_traj: bool = True, frame_preprocessor=None, model: Optional[InternVLAN1ForCausalLM] = None,
                 vit_cache: bool = False, prefix_cache: bool = False):
        """Two ways in, both ending in (model, processor, episode state):
          * the reference's: `InternVLAN1Net(config=InternVLAN1ModelConfig(model_cfg={'model': model_settings}))`
            (internvla_n1_agent.py:39-43, internvla_n1_policy.py:29-48) - loads the checkpoint at model_settings['model_path'] on
            model_settings['device'] with `InternVLAN1ForCausalLM.from_pretrained`, and the HF tokenizer / processor from the same
            path (a1 stays on the host); engines are sized from num_history / resize / camera size / env_num;
          * `InternVLAN1Net(model, processor, ...)` with an already-built model (tests, bench, the batched agent's per-env states).
        frame_preprocessor: an internnav_amd.preprocess.FramePreprocessor - the PIL resizes and the HF image processor then run on
        the device from raw uint8 frames (bit-exact with the host path); the processor is only used for the chat template and
        its tokenizer."""
        if model is None and config is not None and not hasattr(config, "model_cfg"):
            model, config = config, None                      # legacy positional form: InternVLAN1Net(model, processor, ...)
        if model is None:
            ms = dict(config.model_cfg["model"])
            num_history, resize_w, resize_h = ms.get("num_history", num_history), ms.get("resize_w", resize_w), ms.get("resize_h", resize_h)
            continuous_traj = ms.get("continuous_traj", continuous_traj)
            model, processor = self._load(ms)
            if frame_preprocessor is None and ms.get("device_preprocess", False):
                from .preprocess import FramePreprocessor

                frame_preprocessor = FramePreprocessor(model.device, resize_w=resize_w, resize_h=resize_h)
        self.model_config = SimpleNamespace(num_history=num_history, resize_w=resize_w, resize_h=resize_h, continuous_traj=continuous_traj)
        self.model, self.processor, self.pre = model, processor, frame_preprocessor
        # per-frame ViT cache (needs the device pre-processor): the embeddings of the frames of the previous System-2
        # call are kept, so the look-down turn (which re-sends every image of the turn before, internvla_n1_policy.py:140-147) and
        # re-sampled history frames (frame 0 is in every np.linspace sample) skip the vision tower. Exact: the tower attends per image.
        self.vit_cache = bool(vit_cache or (config is not None and dict(config.model_cfg["model"]).get("vit_cache", False))) and frame_preprocessor is not None
        # prefix-KV reuse (model_settings['prefix_cache']): every System-2 prompt of an episode after the first starts with the same tokens -
        # chat template, instruction, "These are your historical observations:" and history frame 0 (np.linspace always samples it,
        # internvla_n1_policy.py:125-133); the look-down turn repeats the whole previous prompt. Their K/V of all 28 layers are kept per env
        # (17 MB for 296 tokens) and handed back to generate(): those tokens are not prefilled, frame 0 is not encoded. Exact (causal mask).
        self.prefix_cache = bool(prefix_cache or (config is not None and dict(config.model_cfg["model"]).get("prefix_cache", False))) \
            and hasattr(getattr(model, "qwen", None), "export_prefix_kv") and not self.vit_cache
        self.tokenizer = getattr(processor, "tokenizer", None)
        self.num_history, self.resize_w, self.resize_h, self.continuous_traj = num_history, resize_w, resize_h, continuous_traj
        self.device = model.device
        self.reset()

    @classmethod
    def _load(cls, ms: dict):
        """model + processor for a model_settings dict, loaded once per (checkpoint, device) and shared afterwards."""
        key = (str(ms["model_path"]), str(ms.get("device", "cuda:0")))
        if key not in cls._shared:
            n_env = int(ms.get("env_num", 1) or 1)
            model = InternVLAN1ForCausalLM.from_pretrained(
                ms["model_path"], torch_dtype=torch.bfloat16, attn_implementation="flash_attention_2", device_map={"": ms.get("device", "cuda:0")},
                max_envs=max(n_env, int(ms.get("max_envs", 1))), max_s2_seqs=ms.get("max_s2_seqs"), num_history=ms.get("num_history", 8),
                resize_w=ms.get("resize_w", 384), resize_h=ms.get("resize_h", 384), cam_w=ms.get("width", 640), cam_h=ms.get("height", 480))
            cls._shared[key] = (model.eval(), cls.load_processor(ms["model_path"]))
        return cls._shared[key]

    @staticmethod
    def load_processor(model_path):
        """the HF processor + tokenizer of the checkpoint, exactly as the reference loads them (internvla_n1_policy.py:40-43):
        third-party host-side pre-processing."""
        from transformers import AutoProcessor, AutoTokenizer

        processor = AutoProcessor.from_pretrained(model_path)
        processor.tokenizer = AutoTokenizer.from_pretrained(model_path, use_fast=True)
        processor.tokenizer.padding_side = "left"
        return processor

    def spawn(self) -> "InternVLAN1Net":
        """a fresh episode state on the same model / processor (the batched agent keeps one per environment)."""
        return InternVLAN1Net(processor=self.processor, num_history=self.num_history, resize_w=self.resize_w, resize_h=self.resize_h,
                              continuous_traj=self.continuous_traj, frame_preprocessor=self.pre, model=self.model, vit_cache=self.vit_cache,
                              prefix_cache=self.prefix_cache)

    def eval(self):
        return self

    def reset(self):
        self.rgb_list, self.depth_list, self.pose_list = [], [], []
        self.episode_idx = 0
        self.conversation_history = []
        self.llm_output = ""
        self.input_images = []
        self.input_keys = []          # frame identity of every input image: index into rgb_list, or "look_down"
        self._emb_cache = {}          # frame key -> (embeds bf16 [tokens, H], grid) of the frames of the last System-2 call
        self._prefix = None           # (token ids of the cached prompt prefix, K/V bf16 [layers, P, 1024]) of this episode

    def parse_actions(self, output: str) -> List[int]:
        regex = re.compile("|".join(re.escape(a) for a in self.ACTIONS2IDX))
        return list(itertools.chain.from_iterable(self.ACTIONS2IDX[m] for m in regex.findall(output)))

    def _to_image(self, rgb, resize: bool):
        if self.pre is not None:   # device path: uint8 [H, W, 3] tensors instead of PIL images
            frame = torch.from_numpy(np.ascontiguousarray(np.asarray(rgb)[..., :3], dtype=np.uint8)).to(self.pre.device)
            return self.pre.resize(frame[None], self.resize_w, self.resize_h)[0] if resize else frame
        from PIL import Image

        image = Image.fromarray(rgb).convert("RGB")
        return image.resize((self.resize_w, self.resize_h)) if resize else image

    def step_no_infer(self, rgb, depth, pose):
        self.rgb_list.append(self._to_image(rgb, True))
        self.episode_idx += 1

    def build_s2_inputs(self, rgb, instruction: str, look_down: bool = False):
        """steps 1-2 of s2_step (internvla_n1_policy.py:110-165): history sampling, prompt, chat template, processor call."""
        image = self._to_image(rgb, not look_down)
        if not look_down:
            self.rgb_list.append(image)
            self.conversation_history = []
            text = self.PROMPT.replace("<instruction>.", instruction)
            if self.episode_idx == 0:
                history_id = []
            else:
                history_id = np.unique(np.linspace(0, self.episode_idx - 1, self.num_history, dtype=np.int32)).tolist()
                text += f" These are your historical observations: {('<image>' + chr(10)) * len(history_id)}."
            self.input_images = [self.rgb_list[i] for i in sorted(history_id)] + self.rgb_list[-1:]
            self.input_keys = sorted(history_id) + [len(self.rgb_list) - 1]
            img_id = 0
            self.episode_idx += 1
        else:
            self.input_images.append(image)
            self.input_keys = list(self.input_keys) + ["look_down"]
            img_id = -1
            assert self.llm_output != "", "Last llm_output should not be empty when look down"
            text = ""
            self.conversation_history.append({"role": "assistant", "content": [{"type": "text", "text": self.llm_output}]})
        text += f" {self.CONJUNCTION}<image>."
        content = []
        for part in split_and_clean(text):
            if part == "<image>":
                content.append({"type": "image", "image": self.input_images[img_id]})
                img_id += 1
            else:
                content.append({"type": "text", "text": part})
        self.conversation_history.append({"role": "user", "content": content})
        chat = self.processor.apply_chat_template(self.conversation_history, tokenize=False, add_generation_prompt=True)
        if self.pre is None:
            return self.processor(text=[chat], images=self.input_images, return_tensors="pt")
        # device pre-processing: pixel_values / grid from the raw frames; the text side restates Qwen2VLProcessor.__call__ - every
        # image placeholder is expanded to grid.prod() / merge^2 image tokens before tokenisation
        cached = None
        if self.vit_cache:
            hit = [self._emb_cache.get(k) if k != "look_down" else None for k in self.input_keys]
            cached = [h[0] if h is not None else None for h in hit]
            fresh = [im for im, h in zip(self.input_images, hit) if h is None]
            if fresh:
                pixel_values, fgrid = self.pre.processor_pixel_values(fresh)
            else:
                pixel_values, fgrid = torch.empty(0, 1176, dtype=torch.bfloat16, device=self.pre.device), torch.empty(0, 3, dtype=torch.int64)
            it = iter(fgrid.tolist())
            grid = torch.tensor([h[1] if h is not None else next(it) for h in hit], dtype=torch.int64)
        else:
            pixel_values, grid = self.pre.processor_pixel_values(self.input_images)
        tok = getattr(self.processor, "image_token", "<|image_pad|>")
        merge2 = self.pre.merge ** 2
        parts = chat.split(tok)
        assert len(parts) == len(self.input_images) + 1, "chat template and image list disagree on the number of images"
        expanded = parts[0]
        for g, rest in zip(grid.tolist(), parts[1:]):
            expanded += tok * (g[0] * g[1] * g[2] // merge2) + rest
        enc = self.processor.tokenizer([expanded], return_tensors="pt")
        out = {"input_ids": enc["input_ids"], "pixel_values": pixel_values, "image_grid_thw": grid}
        if cached is not None:
            out["cached_image_embeds"] = cached
        return out

    def prefix_request(self, inputs) -> Tuple[Optional[torch.Tensor], int]:
        """(prefix K/V to hand to generate() or None, number of leading tokens to export after the call or 0) for the prompt in `inputs`.
        The reusable prefix ends with the <|vision_end|> of the first image when that image is episode frame 0."""
        if not self.prefix_cache or not self.input_keys or self.input_keys[0] != 0:
            return None, 0
        ids = inputs["input_ids"][0]
        ve = (ids == self.model.qwen.cfg["vision_end_id"]).nonzero()
        if ve.numel() == 0:
            return None, 0
        P = int(ve[0]) + 1
        if P >= ids.shape[0]:
            return None, 0
        if self._prefix is not None and self._prefix[0].shape[0] == P and torch.equal(self._prefix[0], ids[:P].cpu()):
            return self._prefix[1], 0
        return None, P

    def store_prefix(self, inputs, kv: torch.Tensor):
        self._prefix = (inputs["input_ids"][0, : kv.shape[1]].cpu().clone(), kv)

    def update_frame_cache(self, inputs, fresh: Dict[int, torch.Tensor]):
        """keep the embeddings of exactly the frames of this call (bounded: <= num_history + 2 frames of 196 - 391 tokens)."""
        if not self.vit_cache:
            return
        grids = inputs["image_grid_thw"].tolist()
        new = {}
        for k, key in enumerate(self.input_keys):
            if key == "look_down":
                continue
            c = inputs["cached_image_embeds"][k]
            e = c if c is not None else fresh.get(k)
            if e is not None:
                new[key] = (e, grids[k])
        self._emb_cache = new

    def finish_s2(self, inputs, output_ids, latents_fn) -> S2Output:
        """steps 3-4 of s2_step (internvla_n1_policy.py:177-197): decode text, pixel goal -> latents, else discrete actions."""
        self.llm_output = self.processor.tokenizer.decode(output_ids[0][inputs["input_ids"].shape[1]:], skip_special_tokens=True)
        out = S2Output()
        if re.search(r"\d", self.llm_output):
            coord = [int(c) for c in re.findall(r"\d+", self.llm_output)]
            out.output_pixel = np.array([int(coord[1]), int(coord[0])])   # a one-number answer raises IndexError like the reference (:187) -> the agent's retry path
            out.output_latent = latents_fn()      # the batched agent passes a marker here and fills the latents of all pixel-goal rows in one call
        else:
            out.output_action = self.parse_actions(self.llm_output)
            if not out.output_action:
                # neither a pixel goal nor a single action token: the reference's agent would index an empty list in its main thread
                # (internvla_n1_agent.py:282, an uncaught IndexError); here it is an S2 failure like any other -> retry once, then STOP
                raise ValueError(f"System-2 answer holds neither a pixel goal nor an action: {self.llm_output!r}")
        return out

    def s2_step(self, rgb, depth, pose, instruction, intrinsic, look_down: bool = False) -> S2Output:
        inputs = self.build_s2_inputs(rgb, instruction, look_down)
        extra = {"cached_image_embeds": inputs["cached_image_embeds"]} if "cached_image_embeds" in inputs else {}
        kv, want = self.prefix_request(inputs)
        if kv is not None or want:
            extra.update(prefix_kv=[kv], export_prefix=[want])
        ids = self.model.generate(input_ids=inputs["input_ids"], pixel_values=inputs["pixel_values"], image_grid_thw=inputs["image_grid_thw"],
                                  max_new_tokens=128, do_sample=False, use_cache=True, past_key_values=None, return_dict_in_generate=True, **extra).sequences
        if "cached_image_embeds" in extra:
            self.update_frame_cache(inputs, self.model.last_image_embeds())
        if want:
            self.store_prefix(inputs, self.model.last_prefix_kv()[0])
        extra = {k: v for k, v in extra.items() if k == "cached_image_embeds"}
        return self.finish_s2(inputs, ids, lambda: self.model.generate_latents(ids, inputs["pixel_values"], inputs["image_grid_thw"], **extra))

    def s1_step_latent(self, rgb, depth, latent) -> S1Output:
        dp_actions = self.model.generate_traj(traj_latents=latent, images_dp=rgb, depths_dp=depth)
        return self.actions_from_traj(dp_actions)

    def actions_from_traj(self, dp_actions) -> S1Output:
        if self.continuous_traj:
            action_list = traj_to_actions(dp_actions)
        else:
            action_list = chunk_token(dp_actions[np.random.choice(dp_actions.shape[0])])
        return S1Output(idx=[x for x in action_list if x != 0][:4])
