"""`InternVLAN1AsyncAgent` of the reference's real-world deployment (internnav/agent/internvla_n1_agent_realworld.py:24-300) on the HIP
engines: the GPU-less-testable functional driver of the dual system.

Same surface: `InternVLAN1AsyncAgent(args)` with args.{device, model_path, resize_w, resize_h, num_history, plan_step_gap},
`reset()`, `step(rgb, depth, pose, instruction, intrinsic, look_down=False) -> S2Output`, `step_s2`, `step_s1`, `step_no_infer`,
`trajectory_tovw`. One call of `step` = one camera frame: System-2 runs when the plan is older than PLAN_STEP_GAP frames, on a look-down
frame, or when nothing is pending (:127-139); otherwise the frame only extends the history. A discrete answer is returned once as
`output_action`; a pixel goal leaves a latent behind and every following frame returns System-1's continuous trajectory
(`traj_to_actions(..., use_discrate_action=False)`, :141-162) until the next System-2 call.
Differences: prompt building / decoding is `InternVLAN1Net`'s (the two reference classes duplicate that code, :167-290 vs
internvla_n1_policy.py:110-197); the debug JPEG / text dumps of the reference (:113,170-173,270-271) are not written.
"""
from __future__ import annotations

import copy
from types import SimpleNamespace

import numpy as np
import torch

from .policy import InternVLAN1ModelConfig, InternVLAN1Net, S2Output, traj_to_actions


class InternVLAN1AsyncAgent:
    def __init__(self, args, model=None, processor=None, frame_preprocessor=None):
        g = (lambda k, d=None: getattr(args, k, d)) if not isinstance(args, dict) else (lambda k, d=None: args.get(k, d))
        self.device = torch.device(g("device", "cuda:0"))
        kw = dict(num_history=g("num_history", 8), resize_w=g("resize_w", 384), resize_h=g("resize_h", 384))
        if model is None:
            ms = dict(model_path=g("model_path"), device=str(self.device), env_num=1, device_preprocess=g("device_preprocess", False),
                      vit_cache=g("vit_cache", False), **kw)
            self.net = InternVLAN1Net(config=InternVLAN1ModelConfig(model_cfg={"model": ms}))
        else:
            self.net = InternVLAN1Net(model, processor, frame_preprocessor=frame_preprocessor, **kw)
        self.model, self.processor = self.net.model, self.net.processor
        self.resize_w, self.resize_h, self.num_history = kw["resize_w"], kw["resize_h"], kw["num_history"]
        self.PLAN_STEP_GAP = g("plan_step_gap", 8)
        self.reset()

    # ---- state of the reference object, kept on the shared policy
    @property
    def episode_idx(self):
        return self.net.episode_idx

    @property
    def rgb_list(self):
        return self.net.rgb_list

    @property
    def llm_output(self):
        return self.net.llm_output

    def reset(self):
        self.net.reset()
        self.last_s2_idx = -100
        self.output_action = None
        self.output_latent = None
        self.output_pixel = None
        self.pixel_goal_rgb = None
        self.pixel_goal_depth = None

    def parse_actions(self, output):
        return self.net.parse_actions(output)

    def step_no_infer(self, rgb, depth, pose):
        self.net.step_no_infer(rgb, depth, pose)

    def trajectory_tovw(self, trajectory, kp: float = 1.0):
        """:118-123"""
        subgoal = trajectory[-1]
        linear_vel, angular_vel = kp * np.linalg.norm(subgoal[:2]), kp * subgoal[2]
        return np.clip(linear_vel, 0, 0.5), np.clip(angular_vel, -0.5, 0.5)

    def step(self, rgb, depth, pose, instruction, intrinsic, look_down: bool = False) -> S2Output:
        """:125-164"""
        out = S2Output()
        no_output = self.output_action is None and self.output_latent is None
        if (self.episode_idx - self.last_s2_idx > self.PLAN_STEP_GAP) or look_down or no_output:
            self.output_action, self.output_latent, self.output_pixel = self.step_s2(rgb, depth, pose, instruction, intrinsic, look_down)
            self.last_s2_idx = self.episode_idx
            out.output_pixel = self.output_pixel
            self.pixel_goal_rgb = copy.deepcopy(rgb)
            self.pixel_goal_depth = copy.deepcopy(depth)
        else:
            self.step_no_infer(rgb, depth, pose)
        if self.output_action is not None:
            out.output_action = copy.deepcopy(self.output_action)
            self.output_action = None
        elif self.output_latent is not None:
            rgbs, depths = self._s1_inputs(rgb, depth)
            trajectories = self.step_s1(self.output_latent, rgbs, depths)
            out.output_trajectory = traj_to_actions(trajectories, use_discrate_action=False)
        return out

    def _s1_inputs(self, rgb, depth):
        """the look-down pair at 224 x 224, RGB / 255 (:142-159; depth is resized but not rescaled in this class)."""
        pre = self.net.pre
        if pre is not None:
            fr = torch.from_numpy(np.stack([np.asarray(self.pixel_goal_rgb)[..., :3], np.asarray(rgb)[..., :3]]).astype(np.uint8)).to(pre.device)
            rgbs = pre.unit_lut[pre.resize(fr.contiguous(), 224, 224).long()].unsqueeze(0)
            d2 = np.stack([np.asarray(self.pixel_goal_depth, dtype=np.float32).reshape(np.asarray(self.pixel_goal_depth).shape[:2]),
                           np.asarray(depth, dtype=np.float32).reshape(np.asarray(depth).shape[:2])])
            depths = pre.resize_f32(torch.from_numpy(d2).to(pre.device).contiguous(), 224, 224).unsqueeze(0).unsqueeze(-1)
            return rgbs, depths
        from PIL import Image

        def r(x):
            return np.array(Image.fromarray(x).resize((224, 224)))

        def d(x):
            x = np.asarray(x, dtype=np.float32)
            return np.array(Image.fromarray(x.reshape(x.shape[:2])).resize((224, 224)))

        rgbs = torch.stack([torch.from_numpy(r(self.pixel_goal_rgb) / 255), torch.from_numpy(r(rgb) / 255)]).unsqueeze(0).to(self.device)
        depths = torch.stack([torch.from_numpy(d(self.pixel_goal_depth)), torch.from_numpy(d(depth))]).unsqueeze(0).unsqueeze(-1).to(self.device)
        return rgbs, depths

    def step_s2(self, rgb, depth, pose, instruction, intrinsic, look_down: bool = False):
        """-> (action_seq, traj_latents, pixel_goal) exactly one of action_seq / (latents, pixel) set (:166-290)."""
        so = self.net.s2_step(rgb, depth, pose, instruction, intrinsic, look_down)
        if so.output_latent is not None:
            return None, so.output_latent, [int(so.output_pixel[0]), int(so.output_pixel[1])]
        return so.output_action, None, None

    def step_s1(self, latent, rgb, depth):
        """:292-294"""
        return self.model.generate_traj(latent, rgb, depth)
