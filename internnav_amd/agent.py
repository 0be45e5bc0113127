"""Batched drop-in for the reference's `InternVLAN1Agent` (internnav/agent/internvla_n1_agent.py:21-407).

Same plugin contract: constructed from an AgentCfg-like object, `step(obs: list[dict]) -> list[dict]`,
`reset(reset_index: list[int] | None)`; every returned dict is {'action': [int], 'ideal_flag': True} (JSON serialisable, as the
HTTP AgentServer requires). Differences, all behind that contract:
  * one agent drives MANY environments: `obs` may hold any number of envs (the reference takes obs[0] only, :246);
  * the S2 daemon thread + sleep polling (:133-208, 270-274) is replaced by a synchronous batched schedule: all envs whose plan
    expired run System-2 in ONE batched generate (grouped by prompt length), then all envs holding a latent run System-1 in ONE
    batched generate_traj. The reference's main thread blocks until S2 finishes anyway, so per-env results are unchanged;
  * an S2 failure of an env (any exception out of its s2_step, e.g. an unparsable pixel goal) resets that env's policy and retries
    ONCE with look_down=False, then falls back to STOP ([0]) - the reference's handler (:168-189). Configuration / engine errors
    (`CapacityError`, `EngineError`) are NOT policy failures: they propagate to the caller instead of ending episodes silently.
    A failure of a whole batched generate() is first re-run env by env, so only envs that fail on their own are reset.
Pinned by execution: tests/test_agent_trace.py replays tests/golden/agent_trace.json - 8 scripted scenarios run through the reference's
own InternVLAN1Agent.step + InternVLAN1Net (oracle/make_golden_agent.py) - and demands the same action every step and the same chat
text / image bytes / System-1 inputs at every model call. Two documented divergences, both where the reference itself breaks:
  * an answer with neither digits nor arrows: the reference's s2_step returns output_action=[] and its main thread raises IndexError
    out of step() (:282); here it counts as an S2 failure (retry once, then STOP);
  * both attempts of the FIRST System-2 call of an episode failing: the reference's main thread then waits forever (the STOP it
    stores keeps idx = -1, so S2Output.validate() never turns true, :272-274); here the env STOPs.
The per-env dual-system state machine (sync / partial_async cadence, look-down turn, action queue, dual_forward_step
accounting) follows internvla_n1_agent.py:210-241 and :243-356 line by line.

Registration: `internnav_amd.register_all()` replaces the 'internvla_n1' entry of the reference's registry (Agent.register raises on
duplicates, internnav/agent/base.py:33-34) and the 'InternVLAN1_Policy' / 'NavDP_Policy' branches of get_policy / get_config - see
INTEGRATION.md. `Agent.init(cfg)` -> `InternVLAN1Agent(cfg)` then works from the config alone, as in the reference.
"""
from __future__ import annotations

import copy
from types import SimpleNamespace
from typing import Any, Dict, List, Optional

import numpy as np
import torch

from ._lib import EngineError
from .policy import InternVLAN1ModelConfig, InternVLAN1Net, S1Output, S2Output
from .runtime import CapacityError

_FATAL = (CapacityError, EngineError, MemoryError, KeyboardInterrupt)
_PENDING = object()      # marker for "this env's answer is a pixel goal, its latents come with the chunk's generate_latents call"


class _EnvState:
    def __init__(self, policy: InternVLAN1Net):
        self.policy = policy
        self.reset()

    def reset(self):
        self.episode_step = 0
        self.look_down = False
        self.s2_output = S2Output()
        self.s1_output = S1Output()
        self.dual_forward_step = 0
        self.sys1_infer_times = 0
        self.policy.reset()


class InternVLAN1Agent:
    def __init__(self, config, model=None, processor=None, policy_factory=None, frame_preprocessor=None):
        """config: AgentCfg (or a dict / object with `model_settings`). From the config ALONE - the way `Agent.init(cfg)` and the
        AgentServer construct agents (internnav/agent/base.py:40-45) - the agent does what the reference's does (:30-43):
        `policy = get_policy(policy_name)(config=get_config(policy_name)(model_cfg={'model': model_settings}))`, which loads the
        checkpoint at model_settings['model_path'] on model_settings['device'] and the HF processor from the same path.
        model_settings read: model_path, device, policy_name, infer_mode ('sync' | 'partial_async'), sys2_max_forward_step, num_history,
        resize_w/h, width/height (camera), continuous_traj, env_num (engine capacity), device_preprocess (optional).
        Tests / bench may pass a built `model` + `processor`, or a `policy_factory() -> InternVLAN1Net`, instead."""
        self.config = config
        ms = getattr(config, "model_settings", None) or (config.get("model_settings") if isinstance(config, dict) else {}) or {}
        self.mode = ms.get("infer_mode", "sync")
        self.sys2_max_forward_step = ms.get("sys2_max_forward_step", 8)
        self.sys1_depth_threshold = 5.0
        self.sys1_forward_step = 4
        self._ms = ms
        self._first = None
        if policy_factory is None:
            if model is not None:
                assert processor is not None, "a pre-built model needs its processor"
                first = InternVLAN1Net(model, processor, num_history=ms.get("num_history", 8), resize_w=ms.get("resize_w", 384),
                                       resize_h=ms.get("resize_h", 384), continuous_traj=ms.get("continuous_traj", True),
                                       frame_preprocessor=frame_preprocessor)
            else:
                from . import get_config, get_policy

                name = ms.get("policy_name") or "InternVLAN1_Policy"
                first = get_policy(name)(config=get_config(name)(model_cfg={"model": dict(ms)}))
                first.eval()
                model, frame_preprocessor = first.model, frame_preprocessor or first.pre
            self._first = first
            if self.mode == "sync" and getattr(getattr(model, "config", None), "system1", "").endswith("_async"):
                # the sync branch feeds the raw camera frame / 10000 x depth to generate_traj (:334), which the checkpoints' async
                # heads cannot take; the reference's own configs run these checkpoints with infer_mode='partial_async'
                raise ValueError(f"infer_mode='sync' cannot drive a '{model.config.system1}' checkpoint: set model_settings['infer_mode']='partial_async'")

            def policy_factory():
                if self._first is not None:
                    p, self._first = self._first, None
                    return p
                return first.spawn()
        self._factory = policy_factory
        self.pre = frame_preprocessor
        self.model = model
        self.envs: List[_EnvState] = []
        self.episode_idx = 0
        self.s2_failures = 0     # env-turns that ended in the STOP fallback (visible to the operator; the reference only prints)

    # ------------------------------------------------------------------------------------------------ plugin surface
    def reset(self, reset_index: Optional[List[int]] = None):
        if reset_index is None:
            for e in self.envs:
                e.reset()
            self.episode_idx = -1
        else:
            for i in reset_index:
                self._env(i).reset()
            self.episode_idx += 1

    def step(self, obs: List[Dict[str, Any]]) -> List[Dict[str, Any]]:
        n = len(obs)
        envs = [self._env(i) for i in range(n)]
        # ---- System 2 for every env whose plan expired (internvla_n1_agent.py:253-268)
        need = [i for i, e in enumerate(envs) if self._should_infer_s2(e) or e.look_down]
        for i, e in enumerate(envs):
            if i in need:
                e.dual_forward_step = 0
            else:
                e.policy.step_no_infer(obs[i]["rgb"], obs[i]["depth"], None)
        if need:
            self._run_s2([(envs[i], obs[i]) for i in need])
        # ---- System 1 for every env that holds a latent and no queued action (:294-337)
        out: List[Optional[List[int]]] = [None] * n
        s1_jobs = []
        for i, e in enumerate(envs):
            so = e.s2_output
            if so.output_action is not None:
                a = so.output_action[0]
                so.output_action = so.output_action[1:] or None
                if a == 5:  # look down: next frame must run S2 on the look-down image (:284-292)
                    e.look_down = True
                    so.output_action = so.output_pixel = so.output_latent = None
                    out[i] = [-1]
                    e.sys1_infer_times = 0
                else:
                    e.look_down = False
                    if e.sys1_infer_times > 0:
                        e.dual_forward_step += 1
                    out[i] = [a]
            else:
                e.look_down = False
                assert so.output_latent is not None, f"S2 output should be either action or latent, but got neither! {so}"
                s1_jobs.append(i)
        if s1_jobs:
            self._run_s1([(envs[i], obs[i]) for i in s1_jobs])
            for i in s1_jobs:
                e = envs[i]
                idx = e.s1_output.idx
                out[i] = [-1] if idx == [] else [idx[0]]
                so = e.s2_output
                so.output_action = (idx[1:] or None) if len(idx) > 1 else None
                so.output_pixel = None
                if self.mode == "sync":
                    so.output_latent = None
                else:
                    if len(idx) < self.sys1_forward_step and len(idx) + e.dual_forward_step < self.sys2_max_forward_step:
                        e.dual_forward_step = self.sys2_max_forward_step - len(idx)  # already reached the pixel goal (:341-345)
                    e.sys1_infer_times += 1
                    e.dual_forward_step += 1
        for e in envs:
            e.episode_step += 1
        return [{"action": a, "ideal_flag": True} for a in out]

    # ------------------------------------------------------------------------------------------------ internals
    def _env(self, i: int) -> _EnvState:
        while len(self.envs) <= i:
            self.envs.append(_EnvState(self._factory()))
        return self.envs[i]

    def _should_infer_s2(self, e: _EnvState) -> bool:
        """should_infer_s2 (internvla_n1_agent.py:210-241)."""
        if e.episode_step == 0:
            return True
        so = e.s2_output
        if self.mode == "sync":
            return so.output_action is None
        if self.mode == "partial_async":
            if e.dual_forward_step >= self.sys2_max_forward_step:
                return True
            return so.output_action is None and so.output_pixel is None and so.output_latent is None
        raise ValueError(f"Invalid mode: {self.mode}")

    def _run_s2(self, jobs, retry: bool = True):
        """batched System-2 over the envs whose plan expired. Prompts are built per env (host) and run as ragged batches in chunks of
        the engine's capacity. An env whose turn fails is reset and retried once without look-down, then STOPs (reference :156-189);
        fatal errors propagate."""
        built, failed = [], []
        for e, o in jobs:
            try:
                inputs = e.policy.build_s2_inputs(o["rgb"], o["instruction"], e.look_down if retry else False)
                built.append((e, o, inputs))
            except _FATAL:
                raise
            except Exception as ex:  # noqa: BLE001
                failed.append((e, o, ex))
        # ONE ragged batch per engine-capacity chunk: prompts of different lengths (instruction length, number of history frames, a
        # camera-size look-down frame) are right-padded to the longest of the chunk - causal attention makes the padding invisible to
        # the real tokens (QwenVLEngine.prefill seq_lens). Sorted by length so a chunk pads as little as possible.
        built.sort(key=lambda it: it[2]["input_ids"].shape[1])
        for items_all in ([built] if built else []):
            model = items_all[0][0].policy.model
            cap = getattr(getattr(model, "qwen", None), "B_max", None) or len(items_all)
            singles = []
            chunks = [items_all[c0:c0 + cap] for c0 in range(0, len(items_all), cap)]
            while chunks or singles:
                if not chunks:
                    chunks, singles = singles, []
                items = chunks.pop(0)
                try:
                    lens = [int(it[2]["input_ids"].shape[1]) for it in items]
                    ids = torch.zeros(len(items), max(lens), dtype=torch.long)
                    mask = torch.zeros(len(items), max(lens), dtype=torch.long)
                    for r, (it, L) in enumerate(zip(items, lens)):
                        ids[r, :L], mask[r, :L] = it[2]["input_ids"][0], 1
                    ragged = {"attention_mask": mask} if min(lens) != max(lens) else {}
                    pv = torch.cat([it[2]["pixel_values"] for it in items], 0)
                    grid = torch.cat([it[2]["image_grid_thw"] for it in items], 0)
                    extra = {}
                    if all("cached_image_embeds" in it[2] for it in items):        # per-frame ViT cache of the policies (device pre-processing)
                        extra["cached_image_embeds"] = [c for it in items for c in it[2]["cached_image_embeds"]]
                    # prefix-KV reuse of the policies: per env the K/V of its episode's constant prompt prefix (or a request to keep them)
                    pref = [it[0].policy.prefix_request(it[2]) if hasattr(it[0].policy, "prefix_request") else (None, 0) for it in items]
                    gen_extra = dict(extra)
                    if any(kv is not None or want for kv, want in pref):
                        gen_extra.update(prefix_kv=[kv for kv, _ in pref], export_prefix=[want for _, want in pref])
                    seqs = model.generate(input_ids=ids, pixel_values=pv, image_grid_thw=grid, max_new_tokens=128, do_sample=False,
                                          use_cache=True, past_key_values=None, return_dict_in_generate=True, **ragged, **gen_extra).sequences
                    if any(want for _, want in pref):
                        kept = model.last_prefix_kv()
                        for r, (it, (_, want)) in enumerate(zip(items, pref)):
                            if want:
                                it[0].policy.store_prefix(it[2], kept[r])
                    if extra:
                        fresh, o = model.last_image_embeds(), 0
                        for e, _, inputs in items:
                            n = len(inputs["cached_image_embeds"])
                            e.policy.update_frame_cache(inputs, {k - o: v for k, v in fresh.items() if o <= k < o + n})
                            o += n
                except _FATAL:
                    raise
                except Exception as ex:  # noqa: BLE001
                    if len(items) > 1:
                        # the reference's handler is per env (:156-189): one bad env must not wipe the episode history of the others
                        # that happened to share its batch -> re-run this chunk's envs one at a time, only those that fail alone fail
                        singles += [[it] for it in items]
                    else:
                        failed += [(e, o, ex) for e, o, _ in items]
                    continue
                # pixel-goal rows get their latents from ONE generate_latents call over the chunk (reference: one call per env, only
                # when the answer holds a pixel goal, internvla_n1_policy.py:183-192)
                pending = []
                for k, (e, o, inputs) in enumerate(items):
                    try:
                        so = e.policy.finish_s2(inputs, seqs[k:k + 1], lambda: _PENDING)
                    except _FATAL:
                        raise
                    except Exception as ex:  # noqa: BLE001 - e.g. IndexError on a one-number pixel goal (internvla_n1_policy.py:187)
                        failed.append((e, o, ex))
                        continue
                    so.idx = e.episode_step
                    so.rgb_memory, so.depth_memory = o["rgb"], o["depth"]
                    so.is_infering = False
                    e.s2_output = so
                    if so.output_latent is _PENDING:
                        pending.append((k, so))
                if pending:
                    try:
                        lat = model.generate_latents(seqs, pv, grid, rows=[k for k, _ in pending], **extra)
                    except _FATAL:
                        raise
                    except Exception as ex:  # noqa: BLE001 - the envs waiting for latents fail TOGETHER and take the reference's path
                        # (reset, one retry, then STOP); nobody keeps the _PENDING sentinel a later System-1 call would consume
                        for k, so in pending:
                            e, o, _ = items[k]
                            e.s2_output = S2Output(idx=e.episode_step, is_infering=False)
                            failed.append((e, o, ex))
                        pending = []
                    for j, (_, so) in enumerate(pending):
                        so.output_latent = lat[j:j + 1]
        if not failed:
            return
        for e, o, ex in failed:
            print(f"s2 infer error: {type(ex).__name__}: {ex}")
            e.policy.reset()
        if retry:
            self._run_s2([(e, o) for e, o, _ in failed], retry=False)     # second attempt: look_down=False (:172-180)
        else:
            for e, o, _ in failed:
                self.s2_failures += 1
                e.s2_output = S2Output(idx=e.episode_step, output_action=[0], rgb_memory=o["rgb"], depth_memory=o["depth"])

    def _prep_s1(self, rgb, depth):
        """224x224 RGB in 0..1 and depth x10 clipped at 5 m (internvla_n1_agent.py:309-321)."""
        from PIL import Image

        r = np.array(Image.fromarray(rgb).resize((224, 224))) / 255.0
        d = np.array(Image.fromarray(depth[:, :, 0]).resize((224, 224))) * 10.0
        d[d > self.sys1_depth_threshold] = self.sys1_depth_threshold
        return r, d

    def _prep_s1_device(self, jobs):
        """the look-down pairs of all jobs in one device batch (frame pre-processor): PIL 8-bit / float bicubic resize to 224 x 224,
        rgb / 255.0 and depth x10 clipped - the same values as _prep_s1, computed from the raw frames on the device."""
        dev = self.pre.device
        rgb = np.stack([np.asarray(f)[..., :3] for e, o in jobs for f in (e.s2_output.rgb_memory, o["rgb"])]).astype(np.uint8, copy=False)
        dep = np.stack([np.asarray(d)[:, :, 0] for e, o in jobs for d in (e.s2_output.depth_memory, o["depth"])]).astype(np.float32, copy=False)
        r = self.pre.unit_lut[self.pre.resize(torch.from_numpy(np.ascontiguousarray(rgb)).to(dev), 224, 224).long()]
        d = self.pre.s1_depth(torch.from_numpy(np.ascontiguousarray(dep)).to(dev), 224, 10.0, self.sys1_depth_threshold)
        n = len(jobs)
        return r.view(n, 2, 224, 224, 3), d.view(n, 2, 224, 224, 1)

    def _run_s1(self, jobs):
        model = jobs[0][0].policy.model
        cap = getattr(getattr(model, "s1", None), "b_max", None)
        if cap and len(jobs) > cap:      # more envs than the System-1 engine was sized for: run it in engine-sized chunks
            for c0 in range(0, len(jobs), cap):
                self._run_s1(jobs[c0:c0 + cap])
            return
        if self.pre is not None and self.mode != "sync" and len({np.asarray(f).shape for e, o in jobs for f in (e.s2_output.rgb_memory, o["rgb"])}) == 1:
            rgb_t, dep_t = self._prep_s1_device(jobs)
            lats = [e.s2_output.output_latent for e, _ in jobs]
            traj = model.generate_traj(traj_latents=torch.cat(lats, 0), images_dp=rgb_t.to(model.device), depths_dp=dep_t.to(model.device))
            S = traj.shape[0] // len(jobs)
            for k, (e, _) in enumerate(jobs):
                e.s1_output = e.policy.actions_from_traj(traj[k * S:(k + 1) * S])
            return
        rgbs, depths, lats = [], [], []
        for e, o in jobs:
            so = e.s2_output
            if self.mode != "sync":
                pr, pd = self._prep_s1(so.rgb_memory, so.depth_memory)
                cr, cd = self._prep_s1(o["rgb"], o["depth"])
                rgbs.append(np.stack([pr, cr]))
                depths.append(np.stack([pd, cd])[..., None])
            else:  # sync mode feeds the raw frame (reference :334)
                rgbs.append(np.asarray(o["rgb"])[None])
                depths.append((np.asarray(o["depth"]) * 10000.0)[None])
            lats.append(so.output_latent)
        rgb_t = torch.from_numpy(np.stack(rgbs)).to(model.device, torch.float32)
        dep_t = torch.from_numpy(np.stack(depths)).to(model.device, torch.float32)
        traj = model.generate_traj(traj_latents=torch.cat(lats, 0), images_dp=rgb_t, depths_dp=dep_t)
        S = traj.shape[0] // len(jobs)
        for k, (e, _) in enumerate(jobs):
            e.s1_output = e.policy.actions_from_traj(traj[k * S:(k + 1) * S])


def register(agent_registry, name: str = "internvla_n1"):
    """Install this agent under `name` in the reference's registry class (internnav.agent.base.Agent), replacing the PyTorch one.
    `internnav_amd.register_all()` does this and the model factories in one call."""
    agent_registry.agents[name] = InternVLAN1Agent
    return InternVLAN1Agent
