"""Batched drop-in for the reference's `InternVLAN1Agent` (internnav/agent/internvla_n1_agent.py:21-407).

Same plugin contract: constructed from an AgentCfg-like object, `step(obs: list[dict]) -> list[dict]`,
`reset(reset_index: list[int] | None)`; every returned dict is {'action': [int], 'ideal_flag': True} (JSON serialisable, as the
HTTP AgentServer requires). Differences, all behind that contract:
  * one agent drives MANY environments: `obs` may hold any number of envs (the reference takes obs[0] only, :246);
  * the S2 daemon thread + sleep polling (:133-208, 270-274) is replaced by a synchronous batched schedule: all envs whose plan
    expired run System-2 in ONE batched generate (grouped by prompt length), then all envs holding a latent run System-1 in ONE
    batched generate_traj. The reference's main thread blocks until S2 finishes anyway, so per-env results are unchanged;
  * S2 failures never raise: the env falls back to STOP ([0]) like the reference's handler (:182-189).
The per-env dual-system state machine (sync / partial_async cadence, look-down turn, action queue, dual_forward_step
accounting) follows internvla_n1_agent.py:210-241 and :243-356 line by line.

Registration: `register(Agent)` replaces the 'internvla_n1' entry of the reference's registry (Agent.register raises on duplicates,
internnav/agent/base.py:33-34) - see INTEGRATION.md.
"""
from __future__ import annotations

import copy
from types import SimpleNamespace
from typing import Any, Dict, List, Optional

import numpy as np
import torch

from .policy import InternVLAN1Net, S1Output, S2Output


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
        """config: AgentCfg-like (attribute or dict `model_settings` with mode / infer_mode, sys2_max_forward_step, num_history,
        resize_w/h, continuous_traj, device ...). `model` + `processor` (or a `policy_factory() -> InternVLAN1Net`) supply the policy.
        frame_preprocessor (internnav_amd.preprocess.FramePreprocessor): System-2 image pre-processing on the device."""
        self.config = config
        ms = getattr(config, "model_settings", None) or (config.get("model_settings") if isinstance(config, dict) else {}) or {}
        self.mode = ms.get("infer_mode", "sync")
        self.sys2_max_forward_step = ms.get("sys2_max_forward_step", 8)
        self.sys1_depth_threshold = 5.0
        self.sys1_forward_step = 4
        self._ms = ms
        if policy_factory is None:
            assert model is not None and processor is not None, "pass a model + processor, or a policy_factory"

            def policy_factory():
                return InternVLAN1Net(model, processor, num_history=ms.get("num_history", 8), resize_w=ms.get("resize_w", 384),
                                      resize_h=ms.get("resize_h", 384), continuous_traj=ms.get("continuous_traj", True),
                                      frame_preprocessor=frame_preprocessor)
        self._factory = policy_factory
        self.pre = frame_preprocessor
        self.model = model
        self.envs: List[_EnvState] = []
        self.episode_idx = 0

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

    def _run_s2(self, jobs):
        """batched System-2: envs are grouped by prompt length (the engine batches equal-length sequences); failures -> STOP."""
        built = []
        for e, o in jobs:
            try:
                inputs = e.policy.build_s2_inputs(o["rgb"], o["instruction"], e.look_down)
                built.append((e, o, inputs))
            except Exception as ex:  # noqa: BLE001 - the agent must never raise out of step() (reference :156-189)
                self._s2_fail(e, o, ex)
        groups: Dict[Any, list] = {}
        for item in built:
            ids = item[2]["input_ids"]
            groups.setdefault((ids.shape[1], tuple(map(tuple, item[2]["image_grid_thw"].tolist()))), []).append(item)
        for (_, _), items in groups.items():
            try:
                model = items[0][0].policy.model
                ids = torch.cat([it[2]["input_ids"] for it in items], 0)
                pv = torch.cat([it[2]["pixel_values"] for it in items], 0)
                grid = torch.cat([it[2]["image_grid_thw"] for it in items], 0)
                seqs = model.generate(input_ids=ids, pixel_values=pv, image_grid_thw=grid, max_new_tokens=128, do_sample=False,
                                      use_cache=True, past_key_values=None, return_dict_in_generate=True).sequences
                lat_cache = {}

                def latents():
                    if "v" not in lat_cache:
                        lat_cache["v"] = model.generate_latents(seqs, pv, grid)
                    return lat_cache["v"]

                for k, (e, o, inputs) in enumerate(items):
                    so = e.policy.finish_s2(inputs, seqs[k:k + 1], lambda k=k: latents()[k:k + 1])
                    so.idx = e.episode_step
                    so.rgb_memory, so.depth_memory = o["rgb"], o["depth"]
                    so.is_infering = False
                    e.s2_output = so
            except Exception as ex:  # noqa: BLE001
                for e, o, _ in items:
                    self._s2_fail(e, o, ex)

    def _s2_fail(self, e: _EnvState, o, ex):
        print(f"[internnav_amd.agent] S2 failed for an env ({type(ex).__name__}: {ex}); emitting STOP")
        e.policy.reset()
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
    """Install this agent under `name` in the reference's registry class (internnav.agent.base.Agent), replacing the PyTorch one."""
    agent_registry.agents[name] = InternVLAN1Agent
    return InternVLAN1Agent
