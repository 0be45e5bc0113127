#!/usr/bin/env python
"""Benchmark of the MI355X policy engine: policy steps / sec / node (BASELINE.json metric).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--envs 64] [--workload n1_dual|navdp_s1]
  N > 1 is launched by the driver as  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Workloads (config.workload):
  n1_dual_b64 (default) - the configuration BASELINE.json's metric is quoted on: InternVLA-N1 full dual system (ViT + Qwen2.5-VL-7B
      System-2 @ 1 Hz + NextDiT System-1 @ 10 Hz), 64 parallel episodes per GPU (512 per 8-GPU node), 4 x RGB frames (392x392 after
      the HF processor = 784 patches each) + 64-token instruction -> S = 920 prompt tokens, 8 greedy answer tokens + 4 latent queries
      per System-2 call, 2 look-down frames of 224x224 + 32 samples x 10 flow-matching steps per System-1 call. One bench step = ONE
      policy step of every env of the rank: System-1 for all 64 envs, System-2 for the 6-7 envs whose plan expires this step
      (every env exactly once per 10 steps), i.e. 64 policy steps at the nominal 1 : 10 cadence.
  navdp_s1_b64 - BASELINE config #2: NavDP System-1 only (NavDPNet, 10 DDPM steps x 32 samples + critic ranking, 8 RGB + 1 depth frame).
Weights are seeded random at the true architecture shapes (7.6 B + 0.68 B + S1 parameters, drawn on the device; no checkpoint is
available offline), inputs synthetic and resident in HBM before the timed region. Episodes are independent, so ranks shard envs
with no data-path exchange ("scaling": "weak"); the only collective is the all_gather of per-env actions over RCCL/xGMI per step.

Timed region: W warm-up steps, then exactly K steps bracketed by barrier + device synchronise, max over ranks. Each engine call
replays a hipGraph: System-2 ViT + prefill of the 6- or 7-env micro-batch; then its decode + latent-query passes on the main stream
concurrently with System-1 of the 57-58 other envs on a side stream; then System-1 of the micro-batch's envs (same results as the
single-stream order, checked before the timed region). Noise draw, micro-batch gather/scatter, D2H of the trajectories and the host
post-processing (traj_to_actions; for the side-stream envs it runs while the GPU finishes the main stream) are inside the step.
After the timed region rank 0 runs ONE instrumented eager pass with per-launch HIP events (ina_prof_*) to attribute time and
algorithmic FLOPs to kernel classes for the "roofline" object, and (N = 1 only) times the CPU oracle on a bounded sample.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC (RCCL / cross-process tensors) on this driver

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

PEAK_BF16_TFLOPS = 2500.0  # dense MFMA bf16 peak of MI355X (guides/MI355X_MICROARCH.md: ~2.5 PF dense)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--envs", type=int, default=64, help="environments per GPU")
    ap.add_argument("--workload", default="n1_dual", choices=["n1_dual", "navdp_s1", "unet1d_s1", "sft"],
                    help="n1_dual = the BASELINE metric's configuration (default); navdp_s1 = config #2; unet1d_s1 = the diffusion-policy UNet head; "
                         "sft = config #5 (the SFT step, bench_sft.py: its own flags pass through)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of hipGraph replay (debug)")
    ap.add_argument("--overlap-at", choices=["start", "decode"], default="decode",
                    help="n1_dual: side-stream System-1 starts with the System-2 micro-batch, or only once its prefill is done (decode phase)")
    ap.add_argument("--priority", choices=["none", "decode", "s1", "main", "side-low"], default="none",
                    help="n1_dual experiment: high-priority stream for the System-2 decode graph, for the side-stream System-1, for the whole "
                         "main chain (prefill, decode, System-1 of the System-2 envs), or the lowest priority for the side stream")
    ap.add_argument("--no-overlap", action="store_true", help="n1_dual: run System-2 and System-1 back to back on one stream")
    ap.add_argument("--vit-cache", action="store_true",
                    help="n1_dual: per-frame ViT cache variant - the first history frame of every env (frame 0, present in every np.linspace history "
                         "sample of the reference) comes from the cache, 3 of the 4 frames are encoded; algorithmic FLOPs are accounted accordingly")
    ap.add_argument("--prefix-kv", action="store_true",
                    help="n1_dual: prefix-KV reuse variant - the K/V of system prompt + instruction + first history frame (296 of the 920 prompt "
                         "tokens, identical between the System-2 calls of an episode) come from a per-env cache; the call encodes 3 of 4 frames and "
                         "prefills 624 tokens per env. Exact (causal mask); algorithmic FLOPs are accounted accordingly. Reported next to the headline.")
    ap.add_argument("--s1-early-images", action="store_true",
                    help="n1_dual experiment: the look-down frames of the System-2 envs are encoded (DINOv2, MemoryEncoder, QFormer) on the side stream "
                         "at the start of the concurrent phase instead of after the decode chain")
    ap.add_argument("--s1-split", action="store_true",
                    help="n1_dual experiment: the side-stream System-1 call (envs keeping their plan) as two half batches on two streams "
                         "(76.1 -> 69.4 ms alone, the step does not move: 282.2 / 282.3 vs 283.0 - the main chain is the critical one)")
    ap.add_argument("--no-split-prefill", action="store_true",
                    help="n1_dual: System-2 prefill as ONE launch sequence instead of two half micro-batches on two streams")
    ap.add_argument("--no-fuse-decode-norm", action="store_true", help="n1_dual: separate RMSNorm launches in the decode passes (round-2 chain)")
    ap.add_argument("--fuse-rownorm", action="store_true",
                    help="n1_dual: NextDiT attn2.to_out / linear_2 as row-block GEMMs with the gated-norm + residual + next-pre-norm epilogue")
    ap.add_argument("--no-raw-frames", action="store_true",
                    help="n1_dual: start the timed step at resident pixel_values / 224x224 frames (round-1 boundary) instead of raw uint8 640x480 camera frames")
    a, rest = ap.parse_known_args()
    if rest and a.workload != "sft":
        ap.error(f"unrecognized arguments: {' '.join(rest)}")
    a.rest = rest
    return a


def calibration_gemm(dev, seconds: float = 0.5):
    """sustained TF/s of THIS box on one fixed compute-bound launch (8192^3 bf16 through the library's own tiled GEMM), so the line tells
    box-to-box clock / power spread (265-281 steps/s across round-2 boxes on unchanged code) apart from code changes."""
    from internnav_amd import ops

    x = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
    w = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16) * 0.01
    out = torch.empty(8192, 8192, device=dev, dtype=torch.bfloat16)
    for _ in range(3):
        ops.linear(x, w, out=out)
    torch.cuda.synchronize()
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        for _ in range(10):
            ops.linear(x, w, out=out)
        torch.cuda.synchronize()
        n += 10
    dt = time.perf_counter() - t0
    return {"gemm_8192_tflops": round(n * 2 * 8192 ** 3 / dt / 1e12, 1), "launches": n, "seconds": round(dt, 2)}


# ------------------------------------------------------------------------------------------------------------ workloads
class NavDPS1:
    """BASELINE config #2."""

    def __init__(self, a, dev, rank):
        from internnav_amd import flops, synthetic
        from internnav_amd.navdp import NavDPNet

        self.cfg = cfg = synthetic.NAVDPNET_CFG
        self.B = B = a.envs
        self.name = f"navdp_s1_b{B}"
        self.desc = {"policy": "NavDPNet (BASELINE config #2: NavDP System-1 only)", "samples_per_env": cfg["sample_num"],
                     "ddpm_steps": cfg["num_train_timesteps"], "frames_per_env": f"{cfg['memory_size']} rgb + 1 depth @224x224"}
        self.net = NavDPNet(synthetic.navdpnet_state_dict(seed=0), cfg, dev, max_envs=B)
        self.g = g = torch.Generator(device=dev).manual_seed(1000 * rank + 7)
        self.inp = dict(
            goal=torch.randn(B, 3, device=dev, generator=g) * 3.0,
            images=torch.rand(B, cfg["memory_size"], 224, 224, 3, device=dev, generator=g),
            depths=torch.rand(B, 1, 224, 224, 1, device=dev, generator=g) * 5.0,
            x_init=torch.randn(B, cfg["sample_num"], cfg["predict_size"], 3, device=dev, generator=g),
            step_noise=torch.randn(cfg["num_train_timesteps"], B, cfg["sample_num"], cfg["predict_size"], 3, device=dev, generator=g))
        self.f_alg = flops.navdpnet_flops_per_env(cfg)["total"]
        self.graph = None
        self.action_shape = (B, 8, cfg["predict_size"], 3)

    def _call(self, goal, images, depths, x_init, step_noise):
        return self.net.predict_pointgoal_batch_action_vel(goal, images, depths, x_init, step_noise)

    def capture(self):
        from internnav_amd import runtime

        self.graph = runtime.GraphedCall(self._call, self.inp)

    def step(self, i):
        self.inp["x_init"].normal_(generator=self.g)
        self.inp["step_noise"].normal_(generator=self.g)
        neg, pos = self.graph() if self.graph else self._call(**self.inp)
        self.last_out = pos
        return pos

    def instrumented(self):
        self._call(**self.inp)

    def step_output_for_check(self):
        return self.last_out

    def cpu_baseline(self):
        from internnav_amd import synthetic
        from oracle import navdp as o_navdp  # cpu_baseline leg only

        cores = min(64, os.cpu_count() or 1)
        torch.set_num_threads(cores)
        sd = synthetic.navdpnet_state_dict(0)
        inp = synthetic.navdpnet_inputs(1, 0)
        fn = lambda: o_navdp.navdpnet_pointgoal(sd, inp["goal"], inp["images"], inp["depths"], inp["x_init"], inp["step_noise"], self.cfg)  # noqa: E731
        t32, t16 = _median_time(fn, 3), _median_time(fn, 3, autocast=True)
        dt = min(t32, t16)
        return {"value": round(1.0 / dt, 4), "unit": "policy steps/s", "cores": cores, "cpu": _cpu_model(), "kind": "port",
                "seconds": {"fp32": round(t32, 2), "bf16_autocast": round(t16, 2)},
                "sample": "1 env x 1 policy step (9 ViT-S frames + 10 DDPM steps x 32 samples + critic), torch CPU, batch-1 as the reference executes; "
                          "1 warm-up + median of 3 runs each for fp32 and bf16-autocast, value = the faster"}


class UNet1DS1:
    """diffusion-policy ConditionalUnet1D + 10 DDIM steps as a System-1 head, 64 envs x 32 samples per call."""

    def __init__(self, a, dev, rank):
        from internnav_amd import flops, synthetic
        from internnav_amd.unet1d import UNet1DHead

        self.cfg = cfg = synthetic.UNET1D_CFG
        self.B = B = a.envs
        self.name = f"unet1d_s1_b{B}"
        self.desc = {"policy": "ConditionalUnet1D (diffusion-policy, down [256,512,1024], kernel 5, FiLM) + DDIM", "samples_per_env": cfg["sample_num"],
                     "ddim_steps": cfg["num_inference_steps"], "condition": f"one {cfg['global_cond_dim']}-d vector per env"}
        self.net = UNet1DHead(synthetic.materialize(synthetic.unet1d_spec(cfg), 0), cfg, dev, max_envs=B)
        self.g = g = torch.Generator(device=dev).manual_seed(1000 * rank + 7)
        self.inp = dict(global_cond=torch.randn(B, cfg["global_cond_dim"], device=dev, generator=g),
                        x_init=torch.randn(B, cfg["sample_num"], cfg["predict_size"], cfg["input_dim"], device=dev, generator=g))
        self.f_alg = flops.unet1d_flops_per_env(cfg)["total"]
        self.graph = None
        self.action_shape = (B, cfg["sample_num"], cfg["predict_size"], cfg["input_dim"])

    def _call(self, global_cond, x_init):
        return self.net.sample_traj(global_cond, x_init)

    def capture(self):
        from internnav_amd import runtime

        self.graph = runtime.GraphedCall(self._call, self.inp)

    def step(self, i):
        self.inp["x_init"].normal_(generator=self.g)
        out = self.graph() if self.graph else self._call(**self.inp)
        self.last_out = out
        return out

    def instrumented(self):
        self._call(**self.inp)

    def step_output_for_check(self):
        return self.last_out

    def cpu_baseline(self):
        from internnav_amd import synthetic
        from oracle import unet1d as o_u  # cpu_baseline leg only

        cores = min(64, os.cpu_count() or 1)
        torch.set_num_threads(cores)
        sd = synthetic.materialize(synthetic.unet1d_spec(self.cfg), 0)
        inp = synthetic.unet1d_inputs(1, 0, self.cfg)
        fn = lambda: o_u.ddim_sample(sd, inp["global_cond"], inp["x_init"], self.cfg["num_train_timesteps"], self.cfg["num_inference_steps"])  # noqa: E731
        t32, t16 = _median_time(fn, 3), _median_time(fn, 3, autocast=True)
        dt = min(t32, t16)
        return {"value": round(1.0 / dt, 4), "unit": "policy steps/s", "cores": cores, "cpu": _cpu_model(), "kind": "port",
                "seconds": {"fp32": round(t32, 2), "bf16_autocast": round(t16, 2)},
                "sample": "1 env x 1 policy step (32 samples x 10 DDIM steps of the UNet), torch CPU; 1 warm-up + median of 3 runs each for fp32 and bf16-autocast"}


class N1Dual:
    """InternVLA-N1 full dual system at the nominal cadence (the configuration the BASELINE metric is quoted on)."""

    N_IMG, GRID, N_INSTR, N_DECODE, CADENCE = 4, (1, 28, 28), 64, 8, 10

    def __init__(self, a, dev, rank):
        from internnav_amd import flops, synthetic
        from internnav_amd.policy import InternVLAN1ForCausalLM, traj_to_actions

        self.traj_to_actions = traj_to_actions
        self.a = a
        self.dev, self.B = dev, a.envs
        B = self.B
        self.name = f"n1_dual_b{B}"
        qcfg, scfg = synthetic.QWEN_N1_CFG, synthetic.N1_NEXTDIT_CFG
        self.qcfg, self.scfg = qcfg, scfg
        per = self.GRID[1] * self.GRID[2]
        # prompt layout (HF chat template shape): 34 template tokens | 64 instruction tokens | 4 x (<vs> 196 x <img> <ve>) | 30 tail tokens
        self.S = 34 + self.N_INSTR + self.N_IMG * (per // 4 + 2) + 30
        self.mb = [B // self.CADENCE + (1 if j < B % self.CADENCE else 0) for j in range(self.CADENCE)]   # micro-batch sizes, sum = B
        self.mb_start = np.concatenate([[0], np.cumsum(self.mb)])
        mmax = max(self.mb)
        spec = synthetic.n1_full_spec(qcfg, "nextdit_async")
        weights = synthetic.LazyDeviceWeights(spec, dev, seed=0)
        self.model = InternVLAN1ForCausalLM(weights, qcfg, "nextdit_async", scfg, device=dev, max_envs=B, max_seq_len=1024,
                                            max_patches=mmax * self.N_IMG * per, max_s2_seqs=mmax)
        if getattr(a, "fuse_rownorm", False):
            self.model.s1.fuse_rownorm = True
        if getattr(a, "no_fuse_decode_norm", False):
            self.model.qwen.fuse_decode_norm = False
        if getattr(a, "no_split_prefill", False):
            self.model.qwen.split_prefill = False
        g = self.g = torch.Generator(device=dev).manual_seed(1000 * rank + 7)
        lim = qcfg["image_token_id"] - 16
        ids = torch.randint(0, lim, (B, self.S), device=dev, generator=g)
        o = 34 + self.N_INSTR
        for k in range(self.N_IMG):
            ids[:, o] = qcfg["vision_start_id"]
            ids[:, o + 1:o + 1 + per // 4] = qcfg["image_token_id"]
            ids[:, o + 1 + per // 4] = qcfg["vision_end_id"]
            o += per // 4 + 2
        self.ids = ids
        self.pixel_values = torch.randn(B, self.N_IMG * per, 1176, device=dev, generator=g, dtype=torch.float32).to(torch.bfloat16)
        self.grid = torch.tensor([list(self.GRID)] * self.N_IMG)
        self.images_dp = torch.rand(B, 2, 224, 224, 3, device=dev, generator=g).to(torch.bfloat16)
        # raw camera frames (the metric's input: 4 x 640x480 RGB per env) resident in HBM; every step runs them through the
        # bit-exact device pre-processor (PIL bicubic 640x480 -> 384x384 -> 392x392, rescale / normalise / patchify for the System-2
        # micro-batch; 640x480 -> 224x224, / 255 for the System-1 look-down pair = frames 0 and 3), one batched launch per stage
        self.raw = not getattr(a, "no_raw_frames", False)
        if self.raw:
            from internnav_amd.preprocess import FramePreprocessor

            self.pre = FramePreprocessor(dev, resize_w=384, resize_h=384)
            self.frames_u8 = torch.randint(0, 256, (B, self.N_IMG, 480, 640, 3), device=dev, generator=g, dtype=torch.uint8)
            self.s1_sel = torch.tensor([0, self.N_IMG - 1], device=dev)
            self.s1_raw = torch.empty(B, 2, 480, 640, 3, dtype=torch.uint8, device=dev)
        self.latent_table = torch.randn(B, qcfg["n_query"], qcfg["t_hidden"], device=dev, generator=g).to(torch.bfloat16)
        self.x_init = torch.randn(B, scfg["sample_num"], scfg["predict_size"], 3, device=dev, generator=g)
        self.desc = {"policy": "InternVLA-N1 dual system (Qwen2.5-VL-7B S2 + NextDiT-async S1), nominal cadence 1 S2 : 10 S1",
                     "input": ("raw uint8 640x480 RGB frames resident in HBM, device pre-processing (PIL-exact resize, HF rescale/normalise/patchify) inside the timed step"
                               if self.raw else "pre-processed pixel_values / 224x224 frames resident in HBM"),
                     "s2": f"{self.N_IMG} frames x 784 patches + {self.N_INSTR}-token instruction, S={self.S}, {self.N_DECODE} greedy tokens + 4 latent queries",
                     "s1": "2 look-down frames @224x224, 32 samples x 10 flow-matching steps", "s2_microbatches_per_10_steps": self.mb}
        self.desc["s2_prefill"] = ("two half micro-batches on two streams (fork / join inside the captured launch sequence), GEMM tiles selected in the shared-tail mode (force_cfg = -1)"
                                   if self.model.qwen.split_prefill else "one launch sequence")
        self.vit_cache = bool(getattr(a, "vit_cache", False)) and self.raw
        self.prefix_kv = bool(getattr(a, "prefix_kv", False)) and self.raw
        assert not (self.vit_cache and self.prefix_kv), "--prefix-kv already covers frame 0 (its tokens are cached K/V): use one of the two"
        self.prefix_len = (34 + self.N_INSTR + per // 4 + 2) if self.prefix_kv else 0     # template + instruction + <vs> frame 0 <ve>
        n_fresh = self.N_IMG - 1 if (self.vit_cache or self.prefix_kv) else self.N_IMG
        f2 = flops.s2_call_flops(self.S, [self.GRID] * n_fresh, self.N_DECODE, qcfg, prefix_len=self.prefix_len)   # cached frames / tokens cost no FLOPs
        f1 = flops.nextdit_s1_flops_per_env(scfg)
        self.f_alg = f1["total"] + f2["total"] / self.CADENCE
        self.f_parts = {"s1_per_env": f1["total"], "s2_per_call": f2["total"]}
        # static S2 buffers per micro-batch size
        q = self.model.qwen
        self.s2 = {}
        if self.vit_cache:
            # frame-0 embeddings of every env, computed once (they would have been produced by the env's first System-2 call)
            self.emb0 = torch.empty(B, per // 4, qcfg["t_hidden"], dtype=torch.bfloat16, device=dev)
            for lo in range(0, B, mmax):
                hi = min(B, lo + mmax)
                pv0, _ = self.pre.qwen_pixel_values(self.frames_u8[lo:hi, 0].contiguous())
                emb, inv = q.vision(pv0, [self.GRID] * (hi - lo))
                self.emb0[lo:hi].copy_(emb[torch.from_numpy(inv).to(dev).long()].view(hi - lo, per // 4, -1))
            self.desc["vit_cache"] = "frame 0 of every env from the per-frame ViT cache (3 of 4 frames encoded per System-2 call)"
        if self.prefix_kv:
            # prefix K/V of every env, computed once by a prefill of the prefix alone (they would have been left behind by the env's
            # previous System-2 call): bf16 [B, layers, prefix_len, 1024] = 17 MB per env
            pl = self.prefix_len
            self.kv_prefix = torch.empty(B, qcfg["t_layers"], pl, q.kv_w, dtype=torch.bfloat16, device=dev)
            for lo in range(0, B, mmax):
                hi = min(B, lo + mmax)
                pv0, _ = self.pre.qwen_pixel_values(self.frames_u8[lo:hi, 0].contiguous())
                q.prefill(ids[lo:hi, :pl].cpu(), pv0, torch.cat([self.grid[:1]] * (hi - lo)))
                for k in range(hi - lo):
                    self.kv_prefix[lo + k].copy_(q.export_prefix_kv(k, pl))
            self.desc["prefix_kv"] = (f"K/V of the first {pl} prompt tokens (template + instruction + frame 0) of every env from the prefix cache: "
                                      f"{self.S - pl} tokens prefilled and 3 of 4 frames encoded per System-2 call")
        for m in sorted(set(self.mb)):
            cache0 = torch.empty(m, per // 4, qcfg["t_hidden"], dtype=torch.bfloat16, device=dev) if self.vit_cache else None
            cached = [c for k in range(m) for c in ([cache0[k]] + [None] * (self.N_IMG - 1))] if self.vit_cache else None
            P = q.plan(ids[:m].cpu(), torch.cat([self.grid] * m), n_decode=self.N_DECODE, with_latents=True, cached_embeds=cached,
                       prefix_len=self.prefix_len)
            self.s2[m] = dict(P=P, cache0=cache0, pv=torch.empty(m * n_fresh * per, 1176, dtype=torch.bfloat16, device=dev),
                              toks=torch.zeros(m, self.N_DECODE, dtype=torch.int32, device=dev),
                              lat=torch.zeros(m, qcfg["n_query"], qcfg["t_hidden"], dtype=torch.bfloat16, device=dev), graph=None)
        self.s1_graph = None
        self.action_shape = (B, 4)
        self.actions = torch.zeros(B, 4, dtype=torch.int32, device=dev)
        # Two-stream schedule (same results, different order): System-1 of the envs that do NOT run System-2 this step is independent
        # of it and is launched on a side stream, concurrently with the System-2 micro-batch (whose decode passes are launch-latency /
        # HBM bound and leave the MFMA pipes idle); System-1 of the 6-7 System-2 envs follows System-2 on the main stream, on a second
        # small engine instance (own workspace).
        self.overlap = not getattr(a, "no_overlap", False) and not a.no_graph
        if self.overlap:
            from internnav_amd.nextdit import NextDiTSystem1
            from internnav_amd.policy import _Prefixed

            self.s1_small = NextDiTSystem1(_Prefixed(weights, "model."), scfg, dev, max_envs=mmax, fuse_rownorm=bool(getattr(a, "fuse_rownorm", False)))
            self.side = torch.cuda.Stream(device=dev)
            nA = B - min(self.mb)
            # experiment (--s1-split): the side-stream call as two half batches on two streams (a second engine instance with its own
            # buffers): the MFMA-bound GEMM launches of one half run beside the HBM-bound norm / attention launches of the other (57
            # envs: 76.1 -> 69.4 ms alone, profiles/r03t_two_stream_s1_probe.log) - neutral in the step, where the main chain is critical
            self.s1_split = bool(getattr(a, "s1_split", False))
            if self.s1_split:
                self.s1_half = NextDiTSystem1(_Prefixed(weights, "model."), scfg, dev, max_envs=nA // 2, fuse_rownorm=bool(getattr(a, "fuse_rownorm", False)))
                self.side2 = torch.cuda.Stream(device=dev)
                self.gA2 = {}
                self.desc["s1_side_call"] = "two half batches on two streams"
            self.latA, self.imgA, self.xA = (torch.empty((nA,) + tuple(t.shape[1:]), dtype=t.dtype, device=dev)
                                             for t in (self.latent_table, self.images_dp, self.x_init))
            self.latB, self.imgB, self.xB = (torch.empty((mmax,) + tuple(t.shape[1:]), dtype=t.dtype, device=dev)
                                             for t in (self.latent_table, self.images_dp, self.x_init))
            self.idxA = [torch.tensor([e for e in range(B) if not (int(self.mb_start[j]) <= e < int(self.mb_start[j]) + self.mb[j])],
                                      device=dev) for j in range(self.CADENCE)]
            self.traj = torch.empty(B, scfg["sample_num"], scfg["predict_size"], 3, device=dev)
            self.hostA = torch.empty(nA, scfg["sample_num"], scfg["predict_size"], 3).pin_memory()
            self.hostB = torch.empty(mmax, scfg["sample_num"], scfg["predict_size"], 3).pin_memory()
            self.idxA_host = [t.tolist() for t in self.idxA]
            self.gA, self.gB, self.gP, self.gD, self.gBimg = {}, {}, {}, {}, {}
            self.overlap_at = a.overlap_at
            self.ev, self.ev2, self.ev3 = torch.cuda.Event(), torch.cuda.Event(), torch.cuda.Event()
            # the decode / latent-query passes are chains of short kernels: on a high-priority stream their workgroups are dispatched
            # ahead of the queued workgroups of the concurrent System-1 kernels instead of waiting behind them
            self.hi = torch.cuda.Stream(device=dev, priority=-1) if getattr(a, "priority", "none") == "decode" else None
            if getattr(a, "priority", "none") == "s1":
                self.side = torch.cuda.Stream(device=dev, priority=-1)
            if getattr(a, "priority", "none") == "side-low":
                try:
                    low = max(torch.cuda.Stream.priority_range())
                except Exception:  # noqa: BLE001
                    low = 1
                self.side = torch.cuda.Stream(device=dev, priority=low)
                self.desc["side_stream_priority"] = low
            self.mainhi = torch.cuda.Stream(device=dev, priority=-1) if getattr(a, "priority", "none") == "main" else None

    def _ingest_s2(self, lo, m, dst):
        """System-2 images of envs [lo, lo + m): raw frames -> pixel_values of the micro-batch (or the round-1 resident tensor)."""
        if self.prefix_kv:
            self.model.qwen.import_prefix_kv_batch(self.kv_prefix[lo:lo + m])           # 28 strided device copies, inside the timed step
            pv, _ = self.pre.qwen_pixel_values(self.frames_u8[lo:lo + m, 1:].reshape(m * (self.N_IMG - 1), 480, 640, 3))
            dst.copy_(pv)
        elif self.vit_cache:
            self.s2[m]["cache0"].copy_(self.emb0[lo:lo + m])
            pv, _ = self.pre.qwen_pixel_values(self.frames_u8[lo:lo + m, 1:].reshape(m * (self.N_IMG - 1), 480, 640, 3))
            dst.copy_(pv)
        elif self.raw:
            pv, _ = self.pre.qwen_pixel_values(self.frames_u8[lo:lo + m].reshape(m * self.N_IMG, 480, 640, 3))
            dst.copy_(pv)
        else:
            dst.copy_(self.pixel_values[lo:lo + m].reshape(-1, 1176))

    def _ingest_s1(self):
        """System-1 look-down pairs of every env (goal frame = frame 0, current = last frame) -> images_dp bf16 [B, 2, 224, 224, 3]."""
        if self.raw:
            torch.index_select(self.frames_u8, 1, self.s1_sel, out=self.s1_raw)
            self.images_dp.copy_(self.pre.s1_frames(self.s1_raw.view(self.B * 2, 480, 640, 3)).view(self.B, 2, 224, 224, 3))

    def _s2_call(self, m):
        s = self.s2[m]
        self.model.qwen.run_s2(s["P"], s["pv"], s["toks"], s["lat"])

    def _s1_call(self):
        return self.model.s1.generate_traj(self.latent_table, self.images_dp, self.x_init)

    def capture(self):
        from internnav_amd import runtime

        for m, s in self.s2.items():
            self._ingest_s2(0, m, s["pv"])
            s["graph"] = runtime.GraphedCall(lambda m=m: self._s2_call(m), {})
        self.s1_graph = runtime.GraphedCall(lambda: self._s1_call(), {})
        if self.overlap:
            for m in sorted(set(self.mb)):
                nA = self.B - m
                n1 = nA - nA // 2 if self.s1_split else nA          # envs [0, n1) on the first engine, [n1, nA) on the second
                self.gA[nA] = runtime.GraphedCall(lambda n1=n1: self.model.s1.generate_traj(self.latA[:n1], self.imgA[:n1], self.xA[:n1]), {}, workspace_slot=1)
                if self.s1_split:
                    self.gA2[nA] = runtime.GraphedCall(lambda n1=n1, nA=nA: self.s1_half.generate_traj(self.latA[n1:nA], self.imgA[n1:nA], self.xA[n1:nA]), {},
                                                       workspace_slot=4)
                early = bool(getattr(self.a, "s1_early_images", False))
                self.gB[m] = runtime.GraphedCall(lambda m=m: self.s1_small.generate_traj(self.latB[:m], self.imgB[:m], self.xB[:m], images_encoded=early), {}, workspace_slot=2)
                if early:
                    self.gBimg[m] = runtime.GraphedCall(lambda m=m: self.s1_small.encode_images(m, self.imgB[:m]), {}, workspace_slot=3)
                if self.overlap_at == "decode":
                    s, q = self.s2[m], self.model.qwen
                    self.gP[m] = runtime.GraphedCall(lambda s=s: q.run_prefill(s["P"], s["pv"]), {})

                    def dec(s=s):
                        q.run_decode(s["P"], s["toks"])
                        q.run_latents(s["P"], s["lat"])
                    self.gD[m] = runtime.GraphedCall(dec, {})

    def check_overlap(self):
        """same step run with both schedules from the same state and noise: the trajectories must agree (different batch splits select
        different GEMM tile kernels, so the comparison is to bf16 tolerance, not bit-exact)."""
        self.freeze_noise = True
        lat0 = self.latent_table.clone()
        ov = self.overlap
        self.overlap = False
        self.step(0)
        t1 = self.last_traj.clone()
        self.latent_table.copy_(lat0)
        self.overlap = ov
        self.step(0)
        t2 = self.last_traj.clone()
        self.latent_table.copy_(lat0)
        self.freeze_noise = False
        return float((t1 - t2).abs().max().item()), float(t1.abs().max().item())

    def _finish(self, traj):
        self.last_traj = traj
        t = traj.cpu()                                        # [B, 32, 32, 3] -> host post-processing of the reference (vln_utils)
        acts = np.zeros((self.B, 4), dtype=np.int32)
        for b in range(self.B):
            al = [x for x in self.traj_to_actions(t[b]) if x != 0][:4]
            acts[b, :len(al)] = al
        self.actions.copy_(torch.from_numpy(acts))
        return self.actions

    def step_overlapped(self, i):
        j = i % self.CADENCE
        m, lo = self.mb[j], int(self.mb_start[j])
        nA, idx = self.B - m, self.idxA[j]
        s = self.s2[m]
        main = torch.cuda.current_stream()
        if not getattr(self, "freeze_noise", False):
            self.x_init.normal_(generator=self.g)
        self._ingest_s1()
        # side stream: System-1 for the envs keeping their current plan (latents of earlier System-2 calls)
        torch.index_select(self.latent_table, 0, idx, out=self.latA[:nA])
        torch.index_select(self.images_dp, 0, idx, out=self.imgA[:nA])
        torch.index_select(self.x_init, 0, idx, out=self.xA[:nA])
        late = self.overlap_at == "decode"
        split = self.s1_split
        n1 = nA - nA // 2 if split else nA
        trajA2 = None
        if not late:
            self.side.wait_stream(main)
            with torch.cuda.stream(self.side):
                trajA = self.gA[nA]()
            if split:
                self.side2.wait_stream(main)
                with torch.cuda.stream(self.side2):
                    trajA2 = self.gA2[nA]()
        # main stream: System-2 micro-batch, then System-1 for exactly those envs
        s["P"]["ids"].copy_(self.ids[lo:lo + m, self.prefix_len:].reshape(-1).to(torch.int32))
        self._ingest_s2(lo, m, s["pv"])
        if late:
            self.gP[m]()
            self.ev.record(main)
            with torch.cuda.stream(self.side):
                self.side.wait_event(self.ev)
                if self.gBimg:                       # look-down frames of the System-2 envs: encoded ahead of the latents, off the main chain
                    self.imgB[:m].copy_(self.images_dp[lo:lo + m])
                    self.gBimg[m]()
                    self.ev3.record(self.side)
                trajA = self.gA[nA]()
            if split:
                with torch.cuda.stream(self.side2):
                    self.side2.wait_event(self.ev)
                    trajA2 = self.gA2[nA]()
            if self.hi is not None:
                with torch.cuda.stream(self.hi):
                    self.hi.wait_event(self.ev)
                    self.gD[m]()
                    self.ev2.record(self.hi)
                main.wait_event(self.ev2)
            else:
                self.gD[m]()
        else:
            s["graph"]()
        self.latent_table[lo:lo + m].copy_(s["lat"])
        self.latB[:m].copy_(s["lat"])
        if self.gBimg and late:
            main.wait_event(self.ev3)
        else:
            self.imgB[:m].copy_(self.images_dp[lo:lo + m])
        self.xB[:m].copy_(self.x_init[lo:lo + m])
        trajB = self.gB[m]()
        # host post-processing (vln_utils.traj_to_actions, as the reference does per env) of the side-stream envs runs while the main
        # stream is still busy with the System-2 decode passes and the System-1 call of the System-2 envs
        acts = np.zeros((self.B, 4), dtype=np.int32)
        with torch.cuda.stream(self.side):
            self.hostA[:n1].copy_(trajA, non_blocking=True)
        if split:
            with torch.cuda.stream(self.side2):
                self.hostA[n1:nA].copy_(trajA2, non_blocking=True)
            self.side2.synchronize()
        self.side.synchronize()
        for k, b in enumerate(self.idxA_host[j]):
            al = [x for x in self.traj_to_actions(self.hostA[k]) if x != 0][:4]
            acts[b, :len(al)] = al
        self.hostB[:m].copy_(trajB, non_blocking=True)
        main.synchronize()
        for k in range(m):
            al = [x for x in self.traj_to_actions(self.hostB[k]) if x != 0][:4]
            acts[lo + k, :len(al)] = al
        if getattr(self, "freeze_noise", False):   # schedule check: keep the assembled trajectories
            self.traj.index_copy_(0, idx, torch.cat([trajA, trajA2]) if split else trajA)
            self.traj[lo:lo + m].copy_(trajB)
            self.last_traj = self.traj
        self.actions.copy_(torch.from_numpy(acts))
        return self.actions

    def step(self, i):
        if self.overlap:
            if getattr(self, "mainhi", None) is not None:      # experiment: the whole main chain on a high-priority stream
                self.mainhi.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(self.mainhi):
                    out = self.step_overlapped(i)
                torch.cuda.current_stream().wait_stream(self.mainhi)
                return out
            return self.step_overlapped(i)
        j = i % self.CADENCE
        m, lo = self.mb[j], int(self.mb_start[j])
        s = self.s2[m]
        # System-2 for the envs whose plan expires this step: gather their prompt / frames, run, scatter the latents back
        s["P"]["ids"].copy_(self.ids[lo:lo + m, self.prefix_len:].reshape(-1).to(torch.int32))
        self._ingest_s2(lo, m, s["pv"])
        self._ingest_s1()
        s["graph"]() if s["graph"] else self._s2_call(m)
        self.latent_table[lo:lo + m].copy_(s["lat"])
        # System-1 for every env
        if not getattr(self, "freeze_noise", False):
            self.x_init.normal_(generator=self.g)
        traj = self.s1_graph() if self.s1_graph else self._s1_call()
        return self._finish(traj)

    def step_output_for_check(self):
        return self.actions

    def instrumented(self):
        m = max(self.mb)
        q = self.model.qwen
        keep, q.split_prefill = q.split_prefill, False      # one stream: the per-launch HIP events time kernels that do not overlap
        try:
            self._s2_call(m)
        finally:
            q.split_prefill = keep
        self._s1_call()
        return m

    def cpu_baseline(self):
        """reference PyTorch path on the host cores = the CPU oracle, on a bounded sample of one env's policy step: the complete
        System-1 call, and for System-2 two ViT blocks (1 window + 1 full) on 3136 patches and two decoder layers on S = 920 tokens at
        the true widths, scaled linearly to the 32 / 28 layers of one call; combined at the 1 : 10 cadence."""
        from internnav_amd import synthetic
        from oracle import nextdit as o_nd  # cpu_baseline leg only
        from oracle import qwen_vl as o_q

        cores = min(64, os.cpu_count() or 1)
        torch.set_num_threads(cores)
        sd = synthetic.n1_nextdit_state_dict(0)
        inp = synthetic.n1_nextdit_inputs(1, 0)
        cfg = dict(self.qcfg, v_depth=2, v_fullatt=(1,), t_layers=2, vocab=8)
        sdq = synthetic.materialize({k: v for k, v in synthetic.qwen_spec(cfg).items()}, 0)
        pv = torch.randn(self.N_IMG * 784, 1176)
        x = torch.randn(1, self.S, self.qcfg["t_hidden"])
        pos = torch.arange(self.S).view(1, 1, -1).expand(3, 1, -1)
        legs = {"s1_call": lambda: o_nd.generate_traj(sd, inp["traj_latents"], inp["images"], inp["x_init"]),
                "s2_vit_2_blocks": lambda: o_q.vision_tower(pv, [self.GRID] * self.N_IMG, sdq, cfg),
                "s2_prefill_2_layers": lambda: o_q.decoder_stack(x, pos, sdq, cfg)}
        out = {}
        for dtype in ("fp32", "bf16"):
            sec = {}
            for name, fn in legs.items():
                if dtype == "bf16" and name == "s1_call":
                    runs = 2           # bounded sample: the bf16 System-1 leg is the slowest on hosts without AMX
                else:
                    runs = 3
                sec[name] = _median_time(fn, runs, autocast=dtype == "bf16")
            t_vit = sec["s2_vit_2_blocks"] * self.qcfg["v_depth"] / 2
            t_llm = sec["s2_prefill_2_layers"] * self.qcfg["t_layers"] / 2
            t_step = sec["s1_call"] + (t_vit + t_llm) / self.CADENCE      # prefill only: decode + latent queries add < 2 % of the FLOPs
            out[dtype] = {"policy_steps_per_s": round(1.0 / t_step, 4),
                          "seconds": {"s1_call": round(sec["s1_call"], 2), "s2_vit_scaled": round(t_vit, 2), "s2_prefill_scaled": round(t_llm, 2)}}
        best = max(out.values(), key=lambda v: v["policy_steps_per_s"])
        return {"value": best["policy_steps_per_s"], "unit": "policy steps/s", "cores": cores, "cpu": _cpu_model(), "kind": "port",
                "fp32": out["fp32"], "bf16_autocast": out["bf16"],
                "sample": "1 env: full System-1 call + (2 of 32 ViT blocks on 3136 patches and 2 of 28 decoder layers on S=920, scaled linearly) combined at "
                          "1 S2 : 10 S1; torch CPU, batch-1 as the reference executes; each leg: 1 warm-up + median of 3 runs (2 for the bf16 System-1 leg); "
                          "value = the faster of fp32 and bf16-autocast"}


def _median_time(fn, runs: int, autocast: bool = False) -> float:
    """wall-clock of fn(): one untimed warm-up, then the median of `runs` runs (>= 3 runs after warm-up)."""
    ts = []
    with torch.no_grad(), torch.autocast(device_type="cpu", dtype=torch.bfloat16, enabled=autocast):
        fn()
        for _ in range(runs):
            t0 = time.time()
            fn()
            ts.append(time.time() - t0)
    return sorted(ts)[len(ts) // 2]


def _cpu_model() -> str:
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def pmc_traffic(workload):
    """HBM bytes per launch of the dominant kernel from the PMC passes committed under profiles/ (rocprofv3 cannot run inside the
    bench): average FETCH_SIZE (x2 gfx950 correction) + WRITE_SIZE per GEMM launch of the same workload, or None if not collected."""
    f = ROOT / "profiles" / "pmc_traffic.json"
    if not f.exists():
        return None
    t = json.loads(f.read_text()).get(workload)
    if not t:
        return None
    return {"bytes_per_launch": round((t["read_MB_per_launch_x2_corrected"] + t["write_MB_per_launch"]) * 1e6), "kernel": t["kernel"], "source": t["source"]}


# ------------------------------------------------------------------------------------------------------------ driver
def main():
    a = parse()
    from internnav_amd.dist import maybe_self_spawn

    maybe_self_spawn(str(Path(__file__).resolve()), a.gpus)      # `python bench.py --gpus N` without a launcher starts its own N ranks
    if a.workload == "sft":                                       # config #5 in the same JSON shape (bench_sft.py)
        import bench_sft

        sys.argv = [str(ROOT / "bench_sft.py"), "--gpus", str(a.gpus), "--steps", str(a.steps), "--warmup", str(a.warmup)] + \
                   (["--no-cpu-baseline"] if a.no_cpu_baseline else []) + a.rest
        return bench_sft.main()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    from internnav_amd import runtime

    arch = runtime.require_gfx950()
    dist = None
    if world > 1:
        import torch.distributed as dist

        from internnav_amd.dist import pin_host_threads

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)  # "nccl" is RCCL on ROCm
        pin_host_threads(local_rank, int(os.environ.get("LOCAL_WORLD_SIZE", world)))   # own slice of the host cores per rank
    wl = {"n1_dual": N1Dual, "navdp_s1": NavDPS1, "unet1d_s1": UNet1DS1}[a.workload](a, dev, rank)
    if not a.no_graph:
        wl.capture()
    gathered = torch.empty((world * wl.action_shape[0],) + tuple(wl.action_shape[1:]), device=dev,
                           dtype=torch.int32 if a.workload == "n1_dual" else torch.float32) if world > 1 else None

    def step(i):
        out = wl.step(i)
        if world > 1:
            dist.all_gather_into_tensor(gathered, out)  # per-env action outputs to every rank (RCCL over xGMI)

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    overlap_check = None
    if getattr(wl, "overlap", False):
        overlap_check = wl.check_overlap()
    for i in range(a.warmup):
        step(i)
    sync()
    t0 = time.perf_counter()
    for i in range(a.steps):
        step(a.warmup + i)
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        # the exchanged actions are really everybody's: this rank's slice equals its own last output, every slice is a valid action table
        mine = wl.step_output_for_check()
        assert torch.equal(gathered[rank * wl.B:(rank + 1) * wl.B], mine), "all_gather: own slice differs from the local actions"
        if a.workload == "n1_dual":
            assert int(gathered.min()) >= 0 and int(gathered.max()) <= 3, "all_gather: action ids outside {0..3}"
        chk = torch.stack([gathered[r * wl.B:(r + 1) * wl.B].double().sum() for r in range(world)])
        ref = chk.clone()
        dist.broadcast(ref, src=0)
        assert torch.equal(chk, ref), "all_gather: ranks hold different gathered tensors"
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    value = world * wl.B * a.steps / dt

    if rank == 0:
        calib = calibration_gemm(dev)
        # ---- roofline: one instrumented eager pass, HIP events around every launch on the launch stream
        runtime.prof_enable(True)
        extra = wl.instrumented()
        torch.cuda.synchronize()
        prof = runtime.prof_read()
        per_kernel = runtime.prof_read_gemm_kernels()
        runtime.prof_enable(False)               # (also clears the tally)
        gm = prof["gemm"]
        blend = gm["flops"] / (gm["ms"] * 1e-3) / 1e12 if gm["ms"] > 0 else 0.0
        # the dominant kernel = the named kernel with the largest share of the step's GEMM time
        dom_name, dom = max(per_kernel.items(), key=lambda kv: kv[1]["ms"]) if per_kernel else ("none", dict(ms=0.0, launches=0, flops=0.0))
        achieved = dom["flops"] / (dom["ms"] * 1e-3) / 1e12 if dom["ms"] > 0 else 0.0
        step_tflops = (value / world) * wl.f_alg / 1e12
        roofline = {
            "bound": "mfma", "kernel": dom_name,
            "achieved": round(achieved, 1), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": round(achieved / PEAK_BF16_TFLOPS, 4),
            "launches": dom["launches"], "avg_launch_us": round(dom["ms"] * 1e3 / max(dom["launches"], 1), 2),
            "algorithmic_tflop_per_launch": round(dom["flops"] / max(dom["launches"], 1) / 1e12, 4),
            "traffic": pmc_traffic(a.workload),
            "gemm_class_blend": {"what": "all tiled MFMA GEMM launches of the instrumented pass (every tile config)", "achieved": round(blend, 1),
                                 "frac": round(blend / PEAK_BF16_TFLOPS, 4)},
            "per_kernel": {k: {"launches": v["launches"], "ms": round(v["ms"], 3), "tflop": round(v["flops"] / 1e12, 3),
                               "tflops": round(v["flops"] / max(v["ms"], 1e-9) / 1e9, 1)} for k, v in per_kernel.items()},
            "instrumented_pass": {"what": "one System-1 call over all envs" + (f" + one System-2 call over {extra} envs" if extra else ""),
                                  "gemm_launches": gm["launches"], "gemm_avg_launch_us": round(gm["ms"] * 1e3 / max(gm["launches"], 1), 2),
                                  "gemm_tflop": round(gm["flops"] / 1e12, 3),
                                  "kernel_class_ms": {k: round(v["ms"], 3) for k, v in prof.items()},
                                  "kernel_class_tflops": {k: round(v["flops"] / max(v["ms"], 1e-9) / 1e9, 1) for k, v in prof.items()},
                                  "kernel_class_algorithmic_GBps": {k: round(v["bytes"] / max(v["ms"], 1e-9) / 1e6, 1) for k, v in prof.items()}},
            "whole_step": {"algorithmic_tflop_per_env_step": round(wl.f_alg / 1e12, 4),
                           "achieved_tflops_per_gpu": round(step_tflops, 1), "frac": round(step_tflops / PEAK_BF16_TFLOPS, 4)},
        }
        line = {
            "metric": "policy steps/sec/node", "value": round(value, 2), "unit": "policy steps/s", "n_gpus": world,
            "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic (seeded random weights at the true shapes, synthetic camera frames / prompts)",
            "rccl_ranks": world,
            "config": dict({"workload": wl.name, "envs_per_gpu": wl.B, "parallelism": f"dp{world}", "calibration": calib,
                            "launch": "eager" if a.no_graph else "hipGraph replay",
                            "schedule": (("S2 ViT+prefill, then S2 decode+latent queries || S1(non-S2 envs) on a side stream, then S1(S2 envs)"
                                          if getattr(wl, "overlap_at", "") == "decode" else
                                          "S1(non-S2 envs) on a side stream || S2 micro-batch, then S1(S2 envs)")
                                         if getattr(wl, "overlap", False) else "single stream"),
                            "device": arch}, **wl.desc),
            "roofline": roofline,
        }
        if overlap_check is not None:
            line["config"]["schedule_check"] = {"max_abs_diff_vs_single_stream": round(overlap_check[0], 5), "traj_abs_max": round(overlap_check[1], 3)}
        # the CPU port of the reference path is timed on rank 0 of a single-GPU run only (contract); the key is always present
        line["cpu_baseline"] = wl.cpu_baseline() if (world == 1 and not a.no_cpu_baseline) else None
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
