"""diffusion-policy `ConditionalUnet1D` + DDIM sampling as a System-1 trajectory head on the gfx950 op library (SURVEY.md 8f-3;
BASELINE.json north_star: "third_party/diffusion-policy UNet", "DDIM").

Reference: the vendored module internnav/model/encoder/diffusion_policy/model/diffusion/conditional_unet1d.py:14-241
(ConditionalResidualBlock1D with FiLM `cond_predict_scale`, down [256, 512, 1024], kernel 5, GroupNorm 8, Mish), conv1d_components.py
(Downsample1d = Conv1d(3, stride 2, pad 1), Upsample1d = ConvTranspose1d(4, 2, 1)), sampling loop as
diffusion_policy/policy/diffusion_unet_lowdim_policy.py with diffusers' DDIMScheduler (config/train_diffusion_unet_ddim_lowdim_workspace.yaml).
No InternNav policy instantiates this network; it is offered as an alternative head: `sample(global_cond [B, 384], x_init [B, S, T, 3])`,
one condition vector per environment shared by its S samples.

MI355X design: every convolution is an IMPLICIT GEMM on the tiled MFMA kernels. Activations are channels-last and sequence-padded
(row (b, t) = b * (T_l + 2 p_l) + p_l + t, pads p = 8 / 4 / 2 at T / T/2 / T/4 so that T_l + 2 p_l halves exactly per level): the
k * C_in window of output row (b, t) is then k * C_in CONTIGUOUS bf16 values of the padded buffer, so the im2col matrix is just an
overlapping strided view (row stride C_in, or 2 C_in for the stride-2 down-convolution) handed to `ina_gemm_bf16` - nothing is
gathered or copied. ConvTranspose1d(4, 2, 1) = two phase GEMMs over 2-row windows writing the even / odd output rows (ldc = 2 C).
Channel concatenation on the up path = two GEMMs accumulating through the fp32 residual input. GroupNorm + Mish + FiLM + residual
is one launch per Conv1dBlock (csrc/unet1d.hip). The FiLM projections of all 14 blocks are hoisted: their timestep half is input
independent (tabulated per DDIM step at load, fp32 on the host), their condition half is ONE GEMM per call.
"""
from __future__ import annotations

import math
from typing import Dict, List

import numpy as np
import torch

from . import ops


def _mish(x):
    return x * torch.tanh(torch.nn.functional.softplus(x))


class UNet1DHead:
    def __init__(self, state_dict: Dict[str, torch.Tensor], cfg: dict, device="cuda:0", max_envs: int = 64):
        dev = torch.device(device)
        bf, f32 = torch.bfloat16, torch.float32
        sd = state_dict
        self.cfg, self.device, self.b_max = cfg, dev, max_envs
        self.S, self.T, self.D = cfg["sample_num"], cfg["predict_size"], cfg["input_dim"]
        dims = list(cfg["down_dims"])
        assert len(dims) == 3 and self.T % 4 == 0 and cfg["kernel_size"] == 5, "built for 3 levels, kernel 5 (the vendored DDIM config)"
        self.dims, self.k, self.G = dims, cfg["kernel_size"], cfg["n_groups"]
        self.Ts = [self.T, self.T // 2, self.T // 4]
        self.pads = [8, 4, 2]
        self.Tp = [t + 2 * p for t, p in zip(self.Ts, self.pads)]
        Bs = max_envs * self.S
        self.Bs_max = Bs

        def conv_w(key, cin_pad=None):
            """Conv1d weight [Co, Ci, k] -> GEMM weight [Co, k * Ci] in window order (tap-major, channels inside)."""
            w = sd[key].float()
            if cin_pad and w.shape[1] < cin_pad:
                w = torch.cat([w, w.new_zeros(w.shape[0], cin_pad - w.shape[1], w.shape[2])], 1)
            return w.permute(0, 2, 1).reshape(w.shape[0], -1).to(device=dev, dtype=bf).contiguous()

        def f(key):
            return sd[key].to(device=dev, dtype=f32).contiguous()

        # ---- residual blocks in execution order; FiLM projections concatenated
        self.blocks: List[dict] = []
        film_w, film_b = [], []
        off = 0

        def res(p, ci, co, split=None, cin_pad=None):
            nonlocal off
            b = dict(ci=ci, co=co, film_off=off, split=split,
                     g0=(f(p + ".blocks.0.block.1.weight"), f(p + ".blocks.0.block.1.bias")), g1=(f(p + ".blocks.1.block.1.weight"), f(p + ".blocks.1.block.1.bias")),
                     b0=f(p + ".blocks.0.block.0.bias"), w1=conv_w(p + ".blocks.1.block.0.weight"), b1=f(p + ".blocks.1.block.0.bias"))
            w0 = sd[p + ".blocks.0.block.0.weight"]
            if split:   # input = channel concatenation [x | skip]: one GEMM per part
                b["w0"] = [conv_w_t(w0[:, :split]), conv_w_t(w0[:, split:])]
            else:
                b["w0"] = [conv_w(p + ".blocks.0.block.0.weight", cin_pad)]
            if (p + ".residual_conv.weight") in sd:
                rw = sd[p + ".residual_conv.weight"].float()[:, :, 0]
                if cin_pad and rw.shape[1] < cin_pad:
                    rw = torch.cat([rw, rw.new_zeros(rw.shape[0], cin_pad - rw.shape[1])], 1)
                parts = [rw[:, :split], rw[:, split:]] if split else [rw]
                b["rw"] = [x.to(device=dev, dtype=bf).contiguous() for x in parts]
                b["rb"] = f(p + ".residual_conv.bias")
            film_w.append(sd[p + ".cond_encoder.1.weight"].float())
            film_b.append(sd[p + ".cond_encoder.1.bias"].float())
            off += 2 * co
            self.blocks.append(b)
            return b

        def conv_w_t(w):
            return w.float().permute(0, 2, 1).reshape(w.shape[0], -1).to(device=dev, dtype=bf).contiguous()

        self.cin0 = 8                                                           # the 3 waypoint channels padded to one 16-byte chunk
        self.down = []
        chans = [self.cin0] + dims
        for i in range(3):
            r1 = res(f"down_modules.{i}.0", chans[i] if i else cfg["input_dim"], dims[i], cin_pad=self.cin0 if i == 0 else None)
            r2 = res(f"down_modules.{i}.1", dims[i], dims[i])
            ds = None
            if i < 2:
                ds = (conv_w(f"down_modules.{i}.2.conv.weight"), f(f"down_modules.{i}.2.conv.bias"))
            self.down.append((r1, r2, ds))
        self.mid = [res(f"mid_modules.{i}", dims[2], dims[2]) for i in range(2)]
        self.up = []
        for i, (di, do) in enumerate([(dims[1], dims[2]), (dims[0], dims[1])]):
            r1 = res(f"up_modules.{i}.0", 2 * do, di, split=do)
            r2 = res(f"up_modules.{i}.1", di, di)
            wt = sd[f"up_modules.{i}.2.conv.weight"].float()                     # ConvTranspose1d [in, out, 4]: out[2m] = x[m-1] W3 + x[m] W1, out[2m+1] = x[m] W2 + x[m+1] W0
            even = torch.cat([wt[:, :, 3].t(), wt[:, :, 1].t()], 1).to(device=dev, dtype=bf).contiguous()
            odd = torch.cat([wt[:, :, 2].t(), wt[:, :, 0].t()], 1).to(device=dev, dtype=bf).contiguous()
            self.up.append((r1, r2, (even, odd, f(f"up_modules.{i}.2.conv.bias"))))
        self.fin_w, self.fin_b = conv_w("final_conv.0.block.0.weight"), f("final_conv.0.block.0.bias")
        self.fin_g = (f("final_conv.0.block.1.weight"), f("final_conv.0.block.1.bias"))
        w_out = sd["final_conv.1.weight"].float()[:, :, 0]                       # [3, 256] -> 4 output columns (N % 4 == 0)
        self.out_w = torch.cat([w_out, w_out.new_zeros(4 - w_out.shape[0], w_out.shape[1])], 0).to(device=dev, dtype=bf).contiguous()
        self.out_b = torch.cat([sd["final_conv.1.bias"].float(), torch.zeros(4 - w_out.shape[0])]).to(dev).contiguous()
        # ---- FiLM hoisting: emb = Linear(Mish([temb_t | gcond])) = W[:, :dsed] mish(temb_t) + b  (per step, host fp32)  +  W[:, dsed:] mish(gcond) (per call)
        Wf, bfv = torch.cat(film_w, 0), torch.cat(film_b, 0)
        dsed = cfg["dsed"]
        self.film_ld = Wf.shape[0]
        self.film_wg = Wf[:, dsed:].to(device=dev, dtype=bf).contiguous()
        from math import log

        self.sched = _ddim_tables(cfg["num_train_timesteps"], cfg["num_inference_steps"])
        half = dsed // 2
        e = torch.exp(torch.arange(half, dtype=f32) * -(log(10000) / (half - 1)))
        ts = torch.tensor(self.sched["timesteps"], dtype=f32)
        emb = torch.cat([(ts[:, None] * e[None]).sin(), (ts[:, None] * e[None]).cos()], -1)
        h = _mish(emb @ sd["diffusion_step_encoder.1.weight"].float().t() + sd["diffusion_step_encoder.1.bias"].float())
        temb = h @ sd["diffusion_step_encoder.3.weight"].float().t() + sd["diffusion_step_encoder.3.bias"].float()
        self.film_step = (_mish(temb) @ Wf[:, :dsed].t() + bfv).to(dev).contiguous()       # [steps, film_ld]
        self.film_env = torch.empty(max_envs, self.film_ld, dtype=f32, device=dev)
        self.gc_act = torch.empty(max_envs, cfg["global_cond_dim"], dtype=bf, device=dev)
        # ---- activations: per level four padded buffers at the level's width + three at the up path's narrower width, conv / residual scratch
        R = [Bs * tp for tp in self.Tp]
        self.R = R

        def buf(level, c):
            return torch.zeros(R[level], c, dtype=bf, device=dev)

        self.xin = buf(0, self.cin0)
        self.P = [[buf(l, dims[l]) for _ in range(4)] for l in range(3)]
        self.U = {2: [buf(2, dims[1]) for _ in range(3)], 1: [buf(1, dims[0]) for _ in range(3)]}   # up-path blocks: 512 ch at T/4, 256 ch at T/2
        self.Dn = {1: buf(1, dims[0]), 2: buf(2, dims[1])}                                            # Downsample1d keeps the channel count
        # conv outputs (row (b, t) at b * Tp + t). bf16: fp32 outputs (supported by gn_mish) were measured at 78.8 vs 55.0 ms per 64-env
        # call for mean |err| 1.50e-3 vs 1.53e-3 - the error budget is the bf16 activations between the blocks over 10 DDIM steps
        self._tmp = [torch.empty(R[l] * dims[l], dtype=bf, device=dev) for l in range(3)]
        self._rtmp = [torch.empty(R[l] * dims[l], dtype=bf, device=dev) for l in range(3)]          # 1x1 residual-conv outputs (padded indexing)
        self._p32 = [torch.empty(R[l] * dims[l], dtype=f32, device=dev) for l in range(3)]          # first half of a concatenated-input GEMM
        self.eps = torch.zeros(R[0], 4, dtype=f32, device=dev)
        self.sample = torch.empty(Bs * self.T, self.D, dtype=f32, device=dev)

    # ------------------------------------------------------------------------------------------------ building blocks
    def _scr(self, pool, level: int, c: int):
        return pool[level][: self.R[level] * c].view(self.R[level], c)

    def _nrows(self, level: int, nseq: int) -> int:
        """rows up to the last valid output (b = nseq - 1, t = T - 1) in the b * Tp + t indexing."""
        return (nseq - 1) * self.Tp[level] + self.Ts[level]

    def _win(self, src: torch.Tensor, k: int, rows: int, lead: int, stride: int = 1):
        """overlapping window view of a padded buffer: row r = the k * C consecutive values starting at padded row stride * r + lead."""
        Cc = src.shape[1]
        return src.as_strided((rows, k * Cc), (stride * Cc, 1), src.storage_offset() + lead * Cc)

    def _conv(self, srcs, ws, bias, level: int, nseq: int, out: torch.Tensor):
        """Conv1d(k = 5, pad 2) over the channel concatenation of `srcs` -> out rows (b, t) at b * Tp + t (bf16)."""
        rows, lead = self._nrows(level, nseq), self.pads[level] - self.k // 2
        if len(srcs) == 1:
            ops.linear(self._win(srcs[0], self.k, rows, lead), ws[0], bias=bias, out=out[:rows])
        else:
            p32 = self._scr(self._p32, level, ws[0].shape[0])[:rows]
            ops.linear(self._win(srcs[0], self.k, rows, lead), ws[0], bias=bias, out=p32)
            ops.linear(self._win(srcs[1], self.k, rows, lead), ws[1], residual=p32, out=out[:rows])

    def _res_block(self, b: dict, srcs, level: int, nseq: int, mid: torch.Tensor, dst: torch.Tensor, step: int):
        """ConditionalResidualBlock1D (:14-66): conv -> GN -> Mish -> FiLM -> conv -> GN -> Mish -> + residual_conv(x), result in dst."""
        T, p, Tp, co = self.Ts[level], self.pads[level], self.Tp[level], b["co"]
        tmp = self._scr(self._tmp, level, co)
        self._conv(srcs, b["w0"], b["b0"], level, nseq, tmp)
        ops.gn_mish(tmp, mid, b["g0"][0], b["g0"][1], nseq, T, p, Tp, self.G, film_env=self.film_env, film_step=self.film_step[step],
                    film_off=b["film_off"], seq_per_env=self.S)
        self._conv([mid], [b["w1"]], b["b1"], level, nseq, tmp)
        if "rw" in b:        # 1x1 convolution of the block input, evaluated on every padded row (padded indexing, like dst)
            rows = nseq * Tp
            res = self._scr(self._rtmp, level, co)
            if len(srcs) == 1:
                ops.linear(srcs[0][:rows], b["rw"][0], bias=b["rb"], out=res[:rows])
            else:
                p32 = self._scr(self._p32, level, co)[:rows]
                ops.linear(srcs[0][:rows], b["rw"][0], bias=b["rb"], out=p32)
                ops.linear(srcs[1][:rows], b["rw"][1], residual=p32, out=res[:rows])
        else:
            res = srcs[0]
        ops.gn_mish(tmp, dst, b["g1"][0], b["g1"][1], nseq, T, p, Tp, self.G, residual=res)

    def _forward(self, nseq: int, step: int):
        """one noise prediction (ConditionalUnet1D.forward :189-241): xin (padded bf16 sample) -> eps f32 [rows, 4], padded indexing of level 0."""
        P, U = self.P, self.U
        x = self.xin
        for l, (r1, r2, ds) in enumerate(self.down):
            self._res_block(r1, [x], l, nseq, P[l][1], P[l][2], step)
            self._res_block(r2, [P[l][2]], l, nseq, P[l][1], P[l][3], step)          # P[l][3] = the level's skip connection h_l
            x = P[l][3]
            if ds is not None:
                # Downsample1d = Conv1d(3, stride 2, pad 1): the window of output (b, t') starts at padded row 2 t' + p - 1; output row
                # r' = b * Tp / 2 + t' is padded row r' + p_next of the next level (Tp_next = Tp / 2 by construction of the pads)
                rows = (nseq - 1) * self.Tp[l + 1] + self.Ts[l + 1]
                dst = self.Dn[l + 1]
                ops.linear(self._win(x, 3, rows, self.pads[l] - 1, stride=2), ds[0], bias=ds[1], out=dst[self.pads[l + 1]: self.pads[l + 1] + rows])
                ops.pad_rows(dst, nseq, self.Ts[l + 1], self.pads[l + 1])
                x = dst
        self._res_block(self.mid[0], [P[2][3]], 2, nseq, P[2][1], P[2][0], step)
        self._res_block(self.mid[1], [P[2][0]], 2, nseq, P[2][1], P[2][2], step)
        x = P[2][2]
        for i, (r1, r2, (w_even, w_odd, ub)) in enumerate(self.up):
            l = 2 - i
            self._res_block(r1, [x, P[l][3]], l, nseq, U[l][1], U[l][0], step)       # torch.cat((x, h.pop()), dim=1) as two accumulating GEMMs
            self._res_block(r2, [U[l][0]], l, nseq, U[l][1], U[l][2], step)
            # Upsample1d = ConvTranspose1d(4, 2, 1): out[2m] = x[m-1] W3 + x[m] W1, out[2m+1] = x[m] W2 + x[m+1] W0 - two GEMMs over 2-row
            # windows; window row r = b * Tp + m produces padded row 2 r + 2 p + phase of level l - 1 (Tp doubles, the pad doubles)
            y, dst, p = U[l][2], P[l - 1][0], self.pads[l]
            rows, Cn = self._nrows(l, nseq), P[l - 1][0].shape[1]
            for phase, w in ((0, w_even), (1, w_odd)):
                o = dst.as_strided((rows, Cn), (2 * Cn, 1), dst.storage_offset() + (2 * p + phase) * Cn)
                ops.linear(self._win(y, 2, rows, p - 1 + phase), w, out=o)
            ops.pad_rows(dst, nseq, self.Ts[l - 1], self.pads[l - 1], bias=ub)
            x = dst
        # final_conv: Conv1dBlock(256, 256, 5) then Conv1d(256, 3, 1)
        tmp = self._scr(self._tmp, 0, self.dims[0])
        self._conv([x], [self.fin_w], self.fin_b, 0, nseq, tmp)
        ops.gn_mish(tmp, P[0][1], self.fin_g[0], self.fin_g[1], nseq, self.Ts[0], self.pads[0], self.Tp[0], self.G)
        rows = nseq * self.Tp[0]
        ops.linear(P[0][1][:rows], self.out_w, bias=self.out_b, out=self.eps[:rows])

    # ------------------------------------------------------------------------------------------------ sampler
    def sample_traj(self, global_cond: torch.Tensor, x_init: torch.Tensor, use_clipped_model_output: bool = False) -> torch.Tensor:
        """global_cond bf16|f32 [B, G]; x_init f32 [B, S, T, D] (the initial noise) -> f32 [B, S, T, D] after the DDIM steps.
        `use_clipped_model_output` is the keyword diffusers' DDIMScheduler.step takes; the vendored policy passes none
        (diffusion_unet_lowdim_policy.py:87-91: `step(model_output, t, trajectory, generator=generator, **kwargs)`), i.e. False."""
        B = global_cond.shape[0]
        assert B <= self.b_max and tuple(x_init.shape) == (B, self.S, self.T, self.D)
        nseq = B * self.S
        # condition half of every block's FiLM projection: one GEMM per call on Mish(global_cond)
        ops.pool_act(global_cond.to(torch.float32).contiguous(), self.gc_act[:B], T=1, act="mish")
        ops.linear(self.gc_act[:B], self.film_wg, out=self.film_env[:B])
        self.sample[: nseq * self.T].copy_(x_init.reshape(nseq * self.T, self.D))
        xin3 = self.xin[: nseq * self.Tp[0]].view(nseq, self.Tp[0], self.cin0)
        xin3[:, self.pads[0]: self.pads[0] + self.T, : self.D].copy_(x_init.reshape(nseq, self.T, self.D))     # data movement: noise into the padded input
        for i, t in enumerate(self.sched["timesteps"]):
            self._forward(nseq, i)
            ops.ddim_step(self.eps, self.sample, self.xin, nseq, self.T, self.D, self.pads[0], self.sched["coefs"][i], clip=1.0,
                          use_clipped_model_output=use_clipped_model_output)
        return self.sample[: nseq * self.T].view(B, self.S, self.T, self.D)


def _ddim_tables(n_train: int, n_inf: int) -> dict:
    """DDIMScheduler tables (diffusers semantics of the vendored config: squaredcos_cap_v2, leading spacing, set_alpha_to_one)."""
    def alpha_bar(t):
        return math.cos((t + 0.008) / 1.008 * math.pi / 2) ** 2

    betas = np.array([min(1 - alpha_bar((i + 1) / n_train) / alpha_bar(i / n_train), 0.999) for i in range(n_train)], dtype=np.float32)
    ac = np.cumprod((1.0 - betas).astype(np.float32)).astype(np.float32)
    ratio = n_train // n_inf
    ts = (np.arange(0, n_inf) * ratio).round()[::-1].astype(np.int64)
    coefs = []
    for t in ts:
        a_t = float(ac[t])
        prev = int(t) - ratio
        a_p = float(ac[prev]) if prev >= 0 else 1.0
        coefs.append((1.0 / math.sqrt(a_t), math.sqrt(1 - a_t), math.sqrt(a_p), math.sqrt(1 - a_p)))
    return dict(timesteps=[int(t) for t in ts], coefs=coefs)
