"""One SFT step of InternVLA-N1 (`system1 = nextdit_async`) as the reference trains it (BASELINE config #5):
`InternVLAN1ForCausalLM.forward(labels=...)` + HF Trainer optimiser step, on the HIP kernels.

Batch layout = `DataCollatorForSupervisedDataset` output (internvla_n1_lerobot_dataset.py:1150-1280): `input_ids` [B, S] right-padded with
the N_QUERY `<traj>` tokens appended behind each sample's own tokens (`t_s_pos[b]` = where they start, :1168-1183), `pixel_values` /
`image_grid_thw` of the Qwen processor, `traj_images` [B, T, 224, 224, 3], `traj_poses` [B, T, 32, 3], `video_frame_num` [B].

Step = frozen System-2 forward of the tokens before `t_s_pos` (ViT + LLM prefill, ragged batch, KV cache kept)  ->  latent-query rows
(`sft_llm.LatentQueryGrad`)  ->  System-1 loss and gradients (`sft.NextDiTSftHead`)  ->  latent-query backward  ->  gradient reduction
over the data-parallel ranks (ONE flat bucket; all-reduce, or reduce-scatter + sharded update + all-gather = ZeRO-2)  ->  fused
clip + AdamW. Optimiser / schedule: adamw_torch, lr 1e-4 -> cosine_with_min_lr 1e-5, warm-up ratio 0.003, weight decay 0, clip 1.0
(train_dual_system.sh:72-77).
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import numpy as np
import torch

from . import train_ops as T
from .qwen_vl import QwenVLEngine
from .sft import NavDPSftHead, NextDiTSftHead
from .sft_llm import LatentQueryGrad

LQ = "latent_queries"


def cosine_with_min_lr(step: int, total_steps: int, warmup_steps: int, lr: float, min_lr: float) -> float:
    """transformers.get_cosine_with_min_lr_schedule_with_warmup (num_cycles 0.5) evaluated at optimiser step `step` (0-based)."""
    if step < warmup_steps:
        return lr * step / max(1, warmup_steps)
    progress = (step - warmup_steps) / max(1, total_steps - warmup_steps)
    factor = 0.5 * (1.0 + math.cos(math.pi * 2.0 * 0.5 * progress))
    rate = min_lr / lr
    return lr * max(0.0, factor * (1 - rate) + rate)


def _splitmix64(x: int) -> int:
    x = (x + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
    x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
    x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
    return x ^ (x >> 31)


def shard_bounds(numel: int, world: int, rank: int, align: int = 1024):
    """[lo, hi) of rank's slice of a flat buffer whose length is a multiple of `align`: equal slices of whole `align` blocks, the last
    ranks may be shorter / empty (ZeRO-2 partition of the flat gradient / moment buffers)."""
    blocks = numel // align
    per = (blocks + world - 1) // world
    lo = min(blocks, rank * per) * align
    hi = min(blocks, (rank + 1) * per) * align
    return lo, hi


class InternVLAN1SftTrainer:
    def __init__(self, engine: QwenVLEngine, s1_state_dict: Dict[str, torch.Tensor], device, total_steps: int = 1000, lr: float = 1e-4,
                 min_lr: float = 1e-5, warmup_ratio: float = 0.003, weight_decay: float = 0.0, max_grad_norm: float = 1.0,
                 betas=(0.9, 0.999), eps: float = 1e-8, process_group=None, zero2: bool = False, system1: str = "nextdit_async",
                 s1_cfg: Optional[dict] = None, dropout: float = 0.1, seed: int = 0, graph_s1: bool = False, graph_prefix: bool = False):
        """system1: 'nextdit_async' / 'nextdit' (flow-matching loss on the NextDiT head, with / without the memory tokens of the goal / current
        frame pair in the condition, internvla_n1.py:234-258) or 'navdp_async' (epsilon loss on the NavDP head; `s1_cfg` =
        its hyper-parameters, synthetic.N1_NAVDP_CFG; the batch then also carries `traj_depths` [B, T, 224, 224] in metres).
        graph_s1: the System-1 loss + backward of a micro-batch (a few thousand fixed-shape launches of the tape, launch-bound when issued
        one by one) is captured into a hipGraph per batch geometry and replayed; dropout masks stay fresh through a device-side seed word
        (`ina_*_args.drop_salt`) that is rewritten before every replay."""
        self.engine, self.device = engine, torch.device(device)
        self.graph_s1 = bool(graph_s1)
        # graph_prefix: the frozen prefix (ViT + ragged prefill, ~1 100 launches: the step is host-bound on them) is captured into a hipGraph per
        # prompt GEOMETRY (batch, padded length, per-sequence lengths, image-token layout, grids) the second time that geometry is seen, and
        # replayed on the new token ids / pixels afterwards; a geometry seen once runs eagerly (a data loader that never repeats one pays nothing)
        self.graph_prefix = bool(graph_prefix)
        self._prefix_graphs: Dict[tuple, dict] = {}
        self._cap_slot = 0            # library workspace slot of a prefix captured now (3 while the prefetch stream issues it)
        self._prefix_seen: Dict[tuple, int] = {}
        self.max_prefix_graphs = 4
        self._s1_graphs: Dict[tuple, tuple] = {}
        self.max_s1_graphs = 2
        self._salt = torch.zeros(1, dtype=torch.int32, device=self.device) if self.device.type == "cuda" else None
        nq, H = engine.latent_q.shape
        sd = dict(s1_state_dict)
        sd[LQ] = sd.get(LQ, engine.latent_q.float().view(1, nq, H).cpu())
        self.system1, self.seed = system1, seed
        # dropout: the reference trains in module.train() mode, where MemoryEncoder / QFormer (nextdit) and former_net / decoder / drop
        # (navdp) apply p = 0.1; masks come from a counter hash seeded per (seed, rank, micro-step)
        if system1 in ("nextdit_async", "nextdit"):
            # 'nextdit' (internvla_n1.py:256-258): the same flow-matching loss with the projected trajectory hidden states as the ONLY condition
            self.head = NextDiTSftHead(sd, device, n_query=nq, extra_trainable=(LQ,), dropout=dropout, use_async=system1 == "nextdit_async")
        elif system1 == "navdp_async":
            assert s1_cfg is not None, "navdp_async needs the NavDP hyper-parameters (s1_cfg)"
            self.head = NavDPSftHead(sd, device, s1_cfg, n_query=nq, extra_trainable=(LQ,), dropout=dropout)
        else:
            # plain 'navdp' has no loss in the reference either (internvla_n1.py:287-303: only its `async` sub-branch computes one)
            raise NotImplementedError(f"SFT for system1={system1!r}: the reference's forward(labels=...) defines a loss for nextdit, nextdit_async and navdp_async")
        self.P = self.head.P
        self.lq = LatentQueryGrad(engine)
        # prefetch pipeline (prefetch()): two engines over the same weights, the frozen prefix of the NEXT micro-batch runs in the idle one
        self._engines, self._lqs, self._cur = [engine], [self.lq], 0
        self._pf = None               # (batch object, engine slot, prefill state, completion event) of the prefix in flight
        self._pf_stream = None
        self.prefetch_first = False   # A/B switch: issue the next prefix BEFORE this step's own launches (the first version of the pipeline)
        # pipelined steps issue their own launches (latent-query rows, System-1 graph, update: ~2 000 short kernels, 61 ms alone) on a HIGH-priority
        # stream: beside the prefetched prefix (256 x 256 GEMM tiles that hold a CU for ~0.4 ms each, 76 ms alone) every short kernel otherwise queues
        # behind the prefix's pending tiles for the CUs that free up. With the prefix as one graph replay: 124.8 -> 113.2 ms per step
        # (profiles/r06z_sft_priority_ab.txt); capturing the System-1 graph ON a high-priority stream instead is much worse (141.8 / 186.7 ms)
        self.priority_step = True
        self._hp_stream = None
        self.total_steps, self.lr, self.min_lr = total_steps, lr, min_lr
        self.warmup_steps = math.ceil(total_steps * warmup_ratio)
        self.wd, self.max_norm, self.betas, self.eps = weight_decay, max_grad_norm, betas, eps
        self.pg, self.zero2 = process_group, zero2
        self.world = torch.distributed.get_world_size(process_group) if self._dist() else 1
        self.rank = torch.distributed.get_rank(process_group) if self._dist() else 0
        self.grad_norm = torch.zeros(1, dtype=torch.float32, device=self.device)
        self.step_idx = 0
        self.micro_idx = 0
        if self.world > 1 and self.zero2:
            # ZeRO-2 (the reference's deepspeed zero2.json, train_dual_system.sh): optimiser state lives on its owner rank only
            self.P.shard_moments(*shard_bounds(self.P.numel, self.world, self.rank))
        # the draws the reference makes inside forward (torch.randn / torch.rand, internvla_n1.py:261-264, navdp.py:163-175) come from
        # trainer-owned generators, seeded per (seed, rank) and saved with the checkpoint: a resumed run continues the same streams
        s = _splitmix64(_splitmix64(seed) ^ (self.rank + 1))
        self.gen_dev = torch.Generator(device=self.device).manual_seed(s & 0x7FFFFFFFFFFFFFFF)
        self.gen_cpu = torch.Generator().manual_seed((s >> 1) & 0x7FFFFFFFFFFFFFFF)

    @classmethod
    def from_pretrained(cls, path, device="cuda:0", max_seqs: int = 2, max_seq_len: int = 4096, max_patches: int = 2 * 10 * 784, **kw):
        """trainer on an InternVLA-N1 checkpoint directory (HF safetensors shards + config.json, the layout `from_pretrained` of the
        reference reads, internvla_n1_trainer.py:141-160): the frozen Qwen2.5-VL goes to the engine, the System-1 modules under `model.`
        (`model.navdp.` for navdp_async) and `model.latent_queries` become the trainable store. Capacity arguments size the engine's
        buffers (per-device batch, longest prompt incl. the <traj> tokens, patches of one batch)."""
        import json
        from pathlib import Path

        from . import synthetic
        from .policy import _ShardedCheckpoint, qwen_cfg_from_hf

        p = Path(path)
        files = sorted(p.glob("*.safetensors"))
        if not files:
            raise FileNotFoundError(f"no *.safetensors under {p}: InternVLA-N1 checkpoints are HF safetensors shards")
        cfgj = json.loads((p / "config.json").read_text()) if (p / "config.json").exists() else {}
        qcfg = qwen_cfg_from_hf(cfgj)
        system1 = cfgj.get("system1", "nextdit_async")
        weights = _ShardedCheckpoint(files)
        engine = QwenVLEngine(weights, qcfg, device, max_seqs=max_seqs, max_seq_len=max_seq_len, max_patches=max_patches)
        frozen = set(synthetic.qwen_spec(qcfg))                          # visual.*, model.layers.*, embed / norm / lm_head, latent_queries
        pre = "model.navdp." if "navdp" in system1 else "model."
        sd = {k[len(pre):]: weights[k] for k in weights.keys() if k.startswith(pre) and k not in frozen}
        if not sd:
            raise KeyError(f"checkpoint {p} holds no System-1 parameters under '{pre}'")
        sd[LQ] = weights["model.latent_queries"]
        if "navdp" in system1:
            kw.setdefault("s1_cfg", synthetic.N1_NAVDP_CFG)
        return cls(engine, sd, device, system1=system1, **kw)

    def _dist(self) -> bool:
        return torch.distributed.is_available() and torch.distributed.is_initialized()

    # ---------------------------------------------------------------------------------------------------------------- one step
    def forward_backward(self, batch: dict, noise: Optional[torch.Tensor] = None, t_index: Optional[torch.Tensor] = None,
                         loss_scale: float = 1.0, state: Optional[dict] = None) -> torch.Tensor:
        """loss of the micro-batch; gradients are accumulated into the flat store. Gradient accumulation over k micro-batches = k calls with
        loss_scale = 1 / k (HF Trainer divides the loss by gradient_accumulation_steps before backward) and one optimizer_step()."""
        loss, dh = self._loss_and_dh(batch, noise, t_index, loss_scale, state)
        self._lq_backward(dh)
        return loss

    def _lq_backward(self, dh: torch.Tensor):
        nq = self.engine.latent_q.shape[0]
        self.P.grad(LQ).view(nq, -1).add_(self.lq.backward(dh))

    def _loss_and_dh(self, batch: dict, noise, t_index, loss_scale: float, state: Optional[dict]):
        """frozen prefix (prefetched or now) -> latent-query rows -> System-1 loss and gradients; returns (loss, d loss / d latent hidden states)."""
        dev = self.device
        B = batch["input_ids"].shape[0]
        if state is None:
            state = self._acquire_prefix(batch)      # may switch self.engine / self.lq to the twin that holds the prefetched prefix
        hq = self.lq.forward(state)
        Tn = batch["traj_images"].shape[1]
        if noise is None:            # internvla_n1.py:261-264 / navdp.py:163-175 (the reference draws inside forward)
            noise = torch.randn(B * Tn, *batch["traj_poses"].shape[2:], device=dev, generator=self.gen_dev)
        if t_index is None:
            if self.system1 in ("nextdit_async", "nextdit"):
                t_index = (torch.rand(B * Tn, generator=self.gen_cpu) * 1000).long()
            else:
                t_index = torch.randint(0, self.head.cfg["num_train_timesteps"], (B * Tn,), generator=self.gen_cpu)
        if self.graph_s1:
            loss, dh = self._s1_graphed(hq, batch, noise, t_index, loss_scale)
        elif self.system1 in ("nextdit_async", "nextdit"):
            loss, dh = self.head.loss_and_grads(hq, batch["traj_images"].to(dev), batch["traj_poses"], batch["video_frame_num"], noise, t_index,
                                                loss_scale=loss_scale, seed=self._mask_seed())
        else:
            loss, dh = self.head.loss_and_grads(hq, batch["traj_images"].to(dev), batch["traj_depths"].to(dev), batch["traj_poses"],
                                                batch["video_frame_num"], noise, t_index, loss_scale=loss_scale, seed=self._mask_seed())
        return loss, dh

    def _prefix(self, e: QwenVLEngine, batch: dict) -> dict:
        """frozen System-2 forward of the tokens before `t_s_pos` (ViT + ragged prefill) on engine e; the KV cache it leaves is the
        activation checkpoint of the step. Depends on no trainable tensor (the <traj> rows are not part of the prefix)."""
        ids = batch["input_ids"]
        t_s_pos = np.asarray(batch["t_s_pos"], dtype=np.int64)
        S0 = int(t_s_pos.max())
        prefix = ids[:, :S0].clone()
        for b in range(ids.shape[0]):   # right-pad rows of shorter samples: the <traj> tokens / padding behind t_s_pos are not part of the prefix
            prefix[b, t_s_pos[b]:] = 0
        pv = batch["pixel_values"].to(self.device, torch.bfloat16)
        if not self.graph_prefix:
            return e.prefill(prefix, pv, batch["image_grid_thw"], seq_lens=t_s_pos)
        return self._prefix_graphed(e, prefix, pv, batch["image_grid_thw"], t_s_pos)

    def _prefix_graphed(self, e: QwenVLEngine, prefix: torch.Tensor, pv: torch.Tensor, grid, t_s_pos) -> dict:
        from .runtime import GraphedCall

        cfg = e.cfg
        idn = prefix.cpu().numpy()
        layout = np.zeros(idn.shape, dtype=np.uint8)              # what the plan reads of the token ids: image / vision-start / traj positions
        for bit, name in ((1, "image_token_id"), (2, "vision_start_id"), (4, "traj_token_id")):
            layout |= (idn == cfg[name]).astype(np.uint8) * bit
        g = grid.cpu().numpy() if isinstance(grid, torch.Tensor) else np.asarray(grid)
        # (the workspace slot baked into the captured launches is part of the key: the main-stream and the prefetch path capture their own graphs)
        key = (id(e), self._cap_slot, idn.shape, layout.tobytes(), g.tobytes(), tuple(int(x) for x in t_s_pos), tuple(pv.shape))
        ent = self._prefix_graphs.get(key)
        if ent is None:
            self._prefix_seen[key] = self._prefix_seen.get(key, 0) + 1
            if self._prefix_seen[key] < 2:
                if len(self._prefix_seen) > 64:
                    self._prefix_seen.clear()
                return e.prefill(prefix, pv, grid, seq_lens=t_s_pos)
            P = e.plan(prefix, grid)
            pv_static = torch.empty_like(pv)
            pv_static.copy_(pv)
            graph = GraphedCall(lambda: e.run_prefill(P, pv_static), {}, warmup=1, workspace_slot=self._cap_slot)
            ent = dict(P=P, pv=pv_static, graph=graph, state=e.prefill_state(P, prefix, grid, t_s_pos))
            while len(self._prefix_graphs) >= self.max_prefix_graphs:
                self._prefix_graphs.pop(next(iter(self._prefix_graphs)))
            self._prefix_graphs[key] = ent
        else:
            self._prefix_graphs[key] = self._prefix_graphs.pop(key)      # most recently used last
        if ent.get("done") is not None:      # a replay of this graph on ANOTHER stream may still read the static inputs: order the overwrite behind it
            torch.cuda.current_stream().wait_event(ent["done"])
        ent["P"]["ids"].copy_(prefix.reshape(-1).to(torch.int32))
        ent["pv"].copy_(pv)
        ent["graph"]()
        if prefix.is_cuda or pv.is_cuda:
            ent["done"] = torch.cuda.Event()
            ent["done"].record(torch.cuda.current_stream())
        st = dict(ent["state"])
        st["next_pos"] = st["next_pos"].copy()
        if "lens" in st:
            st["lens"] = st["lens"].copy()
        return st

    def _acquire_prefix(self, batch: dict) -> dict:
        """prefill state of this micro-batch's frozen prefix: the prefetched one (the trainer then switches to the engine that holds its KV
        cache, once the prefetch stream's work has landed) or a prefill on the current stream, now."""
        if self._pf is not None and self._pf[0] is batch:
            _, self._cur, state, ev = self._pf
            self._pf = None
            torch.cuda.current_stream().wait_event(ev)
            self.engine, self.lq = self._engines[self._cur], self._lqs[self._cur]
            return state
        return self._prefix(self.engine, batch)

    def prefetch(self, batch: dict, after: Optional["torch.cuda.Event"] = None):
        """Start the frozen prefix of a FUTURE micro-batch now, on a second stream and in a second engine over the same weights
        (QwenVLEngine.twin: own activations and KV cache). The prefix reads no trainable tensor, so it may run beside the System-1 loss /
        backward, the latent-query rows and the optimiser launch of the current micro-batch (the MFMA-bound prefill fills the gaps of
        those launch- and HBM-bound chains); `forward_backward(batch)` picks the result up when it is handed the same batch object.
        Data-loader style use: `training_step(batch_i, next_batch=batch_i+1)`."""
        from . import _lib

        if self._pf is not None and self._pf[0] is batch:
            return
        if len(self._engines) == 1:
            self._engines.append(self.engine.twin())
            self._lqs.append(LatentQueryGrad(self._engines[1]))
            self._pf_stream = torch.cuda.Stream(device=self.device)
        slot = 1 - self._cur
        main = torch.cuda.current_stream()
        # the idle engine's cache was last read by the previous step's latent-query backward: wait for the current stream up to here - or, when the
        # caller has ALREADY issued part of this step (training_step), only up to the event it recorded before doing so
        if after is not None:
            self._pf_stream.wait_event(after)
        else:
            self._pf_stream.wait_stream(main)
        _lib.check(_lib.lib().ina_set_workspace_slot(3), "set_workspace_slot")      # library scratch of launches issued beside the main stream's
        self._cap_slot = 3
        try:
            with torch.cuda.stream(self._pf_stream):
                state = self._prefix(self._engines[slot], batch)
                ev = torch.cuda.Event()
                ev.record(self._pf_stream)
        finally:
            self._cap_slot = 0
            _lib.check(_lib.lib().ina_set_workspace_slot(0), "set_workspace_slot")
        self._pf = (batch, slot, state, ev)

    def drop_prefetch(self):
        """forget a prefix in flight (after waiting for it): the next forward_backward prefills on the current stream again."""
        if self._pf is not None:
            self._pf[3].synchronize()
            self._pf = None

    def _s1_graphed(self, hq: torch.Tensor, batch: dict, noise: torch.Tensor, t_index: torch.Tensor, loss_scale: float):
        """System-1 loss + backward as ONE hipGraph replay. The tape's launch sequence depends on the batch geometry only, so it is captured
        once per (shapes, loss_scale) over static device buffers; a step copies its inputs into them, writes this micro-step's mask seed
        into the device word the mask kernels add to their (baked) site seeds, and replays. Gradients accumulate into the flat store exactly
        as in the eager path (the warm-up runs of the capture are undone)."""
        from .runtime import GraphedCall

        dev, nav = self.device, self.system1 == "navdp_async"
        names = ["traj_images", "traj_poses"] + (["traj_depths"] if nav else [])
        key = (tuple(hq.shape), tuple(noise.shape), float(loss_scale)) + tuple(tuple(batch[n].shape) for n in names)
        ent = self._s1_graphs.get(key)
        if ent is None:
            st = {n: torch.empty(batch[n].shape, dtype=torch.float32, device=dev) for n in names}
            st["hq"] = torch.empty_like(hq)
            st["noise"] = torch.empty(noise.shape, dtype=torch.float32, device=dev)
            st["t"] = torch.empty(t_index.shape, dtype=torch.int64, device=dev)
            st["vfn"] = torch.empty(batch["video_frame_num"].shape, dtype=torch.int64, device=dev)

            def stage():
                st["hq"].copy_(hq)
                st["noise"].copy_(noise)
                st["t"].copy_(t_index)
                st["vfn"].copy_(torch.as_tensor(batch["video_frame_num"]))
                for n in names:
                    st[n].copy_(batch[n])
            stage()

            def run():
                if nav:
                    return self.head.loss_and_grads(st["hq"], st["traj_images"], st["traj_depths"], st["traj_poses"], st["vfn"], st["noise"], st["t"],
                                                    loss_scale=loss_scale, seed=0, salt=self._salt)
                return self.head.loss_and_grads(st["hq"], st["traj_images"], st["traj_poses"], st["vfn"], st["noise"], st["t"],
                                                loss_scale=loss_scale, seed=0, salt=self._salt)
            keep = self.P.g32.clone()                    # the capture's warm-up runs execute the tape for real: undo their accumulation
            g = GraphedCall(run, {}, warmup=1)
            self.P.g32.copy_(keep)
            del keep
            ent = (g, st, stage)
            # every entry pins a captured activation pool + static input copies (GBs): keep the two most recently used geometries only
            # (a ragged last batch beside the regular one); anything older is released and re-captured if it ever comes back
            while len(self._s1_graphs) >= self.max_s1_graphs:
                self._s1_graphs.pop(next(iter(self._s1_graphs)))
            self._s1_graphs[key] = ent
        else:
            self._s1_graphs[key] = self._s1_graphs.pop(key)      # most recently used last
            g, st, _ = ent
            st["hq"].copy_(hq)
            st["noise"].copy_(noise)
            st["t"].copy_(t_index)
            st["vfn"].copy_(torch.as_tensor(batch["video_frame_num"]))
            for n in names:
                st[n].copy_(batch[n])
        self._salt.fill_(self._mask_seed())
        loss, dh = g()
        # the graph's outputs live in its private pool and are overwritten by the next replay: hand out copies, as the eager path hands out
        # fresh tensors (a gradient-accumulation loop may keep the k micro-batch losses and reduce them afterwards - ADVICE r4)
        return loss.clone(), dh.clone()

    def _mask_seed(self) -> int:
        """dropout-mask seed of this micro-step: (seed, rank, micro_idx) hashed together (a linear mix collides across ranks / steps)."""
        self.micro_idx += 1
        return _splitmix64(_splitmix64(_splitmix64(self.seed) ^ (self.rank + 1)) ^ self.micro_idx) & 0x7FFFFFFF

    def reduce_gradients(self):
        """sum the flat gradient bucket over the data-parallel ranks (the 1 / world average is folded into the AdamW launch)."""
        if self.world == 1:
            return
        if self.zero2:
            lo, hi = shard_bounds(self.P.numel, self.world, self.rank)
            per = shard_bounds(self.P.numel, self.world, 0)[1]
            padded = self.P.g32 if per * self.world == self.P.numel else torch.cat([self.P.g32, self.P.g32.new_zeros(per * self.world - self.P.numel)])
            out = torch.empty(per, dtype=torch.float32, device=self.device)
            torch.distributed.reduce_scatter_tensor(out, padded, group=self.pg)
            self.P.g32[lo:hi].copy_(out[: hi - lo])
        else:
            torch.distributed.all_reduce(self.P.g32, group=self.pg)

    def optimizer_step(self) -> float:
        lr = cosine_with_min_lr(self.step_idx, self.total_steps, self.warmup_steps, self.lr, self.min_lr)
        P = self.P
        if self.world > 1 and self.zero2:
            lo, hi = shard_bounds(P.numel, self.world, self.rank)
            parts = T.sumsq_parts(P.g32[lo:hi]) if hi > lo else torch.zeros(1024, dtype=torch.float32, device=self.device)
            torch.distributed.all_reduce(parts, group=self.pg)          # global gradient norm from the shards' partial sums
            P.step_count += 1
            if hi > lo:
                T.adamw(P.p32[lo:hi], P.g32[lo:hi], P.m, P.v, lr, self.betas[0], self.betas[1], self.eps, self.wd, P.step_count,
                        p_bf16=P.p16[lo:hi], sumsq_parts=parts, max_norm=self.max_norm, grad_scale=1.0 / self.world, norm_out=self.grad_norm)
            P.g32.zero_()
            per = shard_bounds(P.numel, self.world, 0)[1]
            for buf in (P.p32, P.p16):                                  # all-gather the updated master / working weights
                full = torch.empty(per * self.world, dtype=buf.dtype, device=self.device)
                mine = torch.zeros(per, dtype=buf.dtype, device=self.device)
                mine[: hi - lo].copy_(buf[lo:hi])
                torch.distributed.all_gather_into_tensor(full, mine, group=self.pg)
                buf.copy_(full[: P.numel])
            P.version += 1
        else:
            P.adamw_step(lr, self.betas, self.eps, self.wd, self.max_norm, grad_scale=1.0 / self.world, norm_out=self.grad_norm)
        self.engine.latent_q.copy_(P.w16(LQ).view(self.engine.latent_q.shape))
        self.step_idx += 1
        return lr

    # ---------------------------------------------------------------------------------------------------------------- checkpoints
    def state_dict(self, prefix: str = "model.") -> Dict[str, torch.Tensor]:
        """TRAINED tensors only (fp32 master copies) under the names they have in an InternVLA-N1 checkpoint: the System-1 modules and
        `latent_queries` are attributes of `InternVLAN1Model`, i.e. `model.<name>` (navdp_async: `model.navdp.<name>`). Frozen tensors
        (the Qwen2.5-VL weights, navdp's rgb_model, DINOv2 mask_token) are not in it: an export merges this over the checkpoint it
        started from."""
        p1 = prefix + ("navdp." if self.system1 == "navdp_async" else "")
        return {(prefix + k if k == LQ else p1 + k): v for k, v in self.P.state_dict().items()}

    def _gather_flat(self, shard: torch.Tensor) -> torch.Tensor:
        """all-gather of a ZeRO-2 shard into the full flat buffer (padded like the p32 / p16 all-gather of optimizer_step)."""
        per = shard_bounds(self.P.numel, self.world, 0)[1]
        mine = torch.zeros(per, dtype=shard.dtype, device=self.device)
        mine[: shard.numel()].copy_(shard)
        full = torch.empty(per * self.world, dtype=shard.dtype, device=self.device)
        torch.distributed.all_gather_into_tensor(full, mine, group=self.pg)
        return full[: self.P.numel]

    def checkpoint(self) -> dict:
        """resume point: master weights, Adam moments (ZeRO-2: all-gathered from their owner ranks - a COLLECTIVE, every rank calls it),
        optimiser / schedule / mask counters and the state of this rank's noise / time-step generators. Weights, moments and counters
        are the same on every rank; `rng` is per rank (each rank saves its own file; a rank that loads ANOTHER rank's file re-seeds its streams
        from (seed, rank, step_idx) - fresh draws, never a replay of the steps before the checkpoint)."""
        sharded = (self.P.m_lo, self.P.m_hi) != (0, self.P.numel)
        return dict(store=self.P.checkpoint(self._gather_flat if sharded else None), step_idx=self.step_idx, micro_idx=self.micro_idx,
                    system1=self.system1, rng=dict(rank=self.rank, dev=self.gen_dev.get_state().cpu(), cpu=self.gen_cpu.get_state()))

    def save_checkpoint(self, path: str):
        """`checkpoint()` written with torch.save (collective under ZeRO-2: all ranks call it, each with its own path or only rank 0 writing
        what it got back from `checkpoint()`)."""
        torch.save(self.checkpoint(), path)

    def load_checkpoint(self, path: str):
        ck = torch.load(path, map_location="cpu", weights_only=True)
        if ck["system1"] != self.system1:
            raise ValueError(f"checkpoint of a {ck['system1']} head, trainer built for {self.system1}")
        self.P.load_checkpoint(ck["store"])
        self.step_idx, self.micro_idx = int(ck["step_idx"]), int(ck["micro_idx"])
        if "rng" in ck and int(ck["rng"]["rank"]) == self.rank:
            self.gen_dev.set_state(ck["rng"]["dev"])
            self.gen_cpu.set_state(ck["rng"]["cpu"])
        else:
            # another rank's file (the "rank 0 writes" flow): this rank's own generator state was not saved, and the constructor-seeded
            # streams would REPLAY the draws of steps 0 .. step_idx - 1. Start fresh streams keyed by (seed, rank, step_idx):
            # never the draws of an earlier step, the same on every resume from this checkpoint.
            s_ = _splitmix64(_splitmix64(_splitmix64(self.seed) ^ (self.rank + 1)) ^ (self.step_idx + 1))
            self.gen_dev.manual_seed(s_ & 0x7FFFFFFFFFFFFFFF)
            self.gen_cpu.manual_seed((s_ >> 1) & 0x7FFFFFFFFFFFFFFF)
        self.engine.latent_q.copy_(self.P.w16(LQ).view(self.engine.latent_q.shape))

    def training_step(self, batch: dict, noise=None, t_index=None, next_batch: Optional[dict] = None) -> torch.Tensor:
        """next_batch: the micro-batch of the following step, if the caller already has it (a data loader does): its frozen prefix is
        started on the prefetch stream (prefetch()). Host issue order of a pipelined step: this step's latent-query rows and System-1 graph
        first (≈ 50 ms of GPU work from ≈ 10 ms of host work), THEN the ≈ 1 100 eager launches of the next prefix (they only have to wait for
        the state of the stream at the start of this step), then the latent-query backward and the update - so neither stream waits for the
        host while the other's launches are being issued."""
        if next_batch is None:
            loss = self.forward_backward(batch, noise, t_index)
            self.reduce_gradients()
            self.optimizer_step()
            return loss
        caller = torch.cuda.current_stream()
        own = caller
        if self.priority_step and self.device.type == "cuda":
            if self._hp_stream is None:
                self._hp_stream = torch.cuda.Stream(device=self.device, priority=-1)
            own = self._hp_stream
            own.wait_stream(caller)
        with torch.cuda.stream(own):
            start = torch.cuda.Event()
            start.record(own)
            state = self._acquire_prefix(batch)
            if self.prefetch_first:
                self.prefetch(next_batch)
            loss, dh = self._loss_and_dh(batch, noise, t_index, 1.0, state)
            if not self.prefetch_first:
                self.prefetch(next_batch, after=start)
            self._lq_backward(dh)
            self.reduce_gradients()
            self.optimizer_step()
        if own is not caller:
            caller.wait_stream(own)          # the caller's stream sees the step as if it had run there
            if isinstance(loss, torch.Tensor) and loss.is_cuda:
                loss.record_stream(caller)
        return loss
