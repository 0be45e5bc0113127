"""SFT step of the InternVLA-N1 `nextdit_async` System-1 (BASELINE config #5) on the gfx950 kernels.

What the reference does in `InternVLAN1ForCausalLM.forward(labels=...)` (internvla_n1.py:222-286) under HF Trainer (bf16 weights,
adamw_torch, clip 1.0; train_dual_system.sh:72-77) with the freeze map of internvla_n1_trainer.py:78-122 (ViT / merger / LLM frozen;
action_encoder, action_decoder, traj_dit, cond_projector, memory_encoder, rgb_resampler, rgb_model and latent_queries trained):
flow-matching loss on the NextDiT output, backward through the System-1 modules, AdamW. This module is that step without torch.autograd:
a define-by-run tape whose nodes launch the HIP kernels of libinternnav_amd.so (forward GEMM / attention / norm kernels of the inference
engine, backward kernels of csrc/train.hip and csrc/attention_bwd.hip), one flat fp32 master / gradient / moment buffer for all
trainable parameters (one fused AdamW launch, one reduce-scatter-able gradient bucket), bf16 working weights.

MI355X-first differences from the reference's execution (same mathematics, same gradients):
  * the goal frame of a sample is the first frame of its sub-goal sequence (internvla_n1.py:237): DINOv2 runs once per distinct frame
    (B*T images) instead of twice per (sample, sub-goal) pair (2*B*T), and the pairing is a token gather;
  * the gradient of `latent_queries` goes through the frozen LLM only along the n_query rows that depend on them (causal attention:
    every earlier row and its K / V are constants), reusing the KV cache of the forward pass - see `LlmRowsBackward`;
  * torch is used for memory, streams and shape plumbing (views, gathers of a few hundred KB); every FLOP of consequence is a kernel here.

Numerics: activations bf16, residual streams / modulation vectors / parameter gradients fp32 (the reference: pure bf16 autocast-free
DeepSpeed bf16). Dropout (p = 0.1 inside the nn.Transformer layers of MemoryEncoder / QFormer, internvla_n1_arch.py:77-81,105) is applied
when `dropout > 0` with a counter-based mask; parity tests run with dropout 0 like `model.eval()` gradients.
"""
from __future__ import annotations

import math
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import torch

from . import ops
from . import train_ops as T

BF, F32 = torch.bfloat16, torch.float32
RESNET_MEAN, RESNET_STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)


# ---------------------------------------------------------------------------------------------------------------- parameters
class ParamStore:
    """All trainable tensors in ONE flat fp32 master buffer (+ gradient, Adam moments, bf16 working copy): a single fused optimiser
    launch and a single gradient bucket for the data-parallel reduction. Views keep the reference's state-dict names and shapes."""

    ALIGN = 8

    def __init__(self, tensors: Dict[str, torch.Tensor], device, trainable: bool = True):
        """trainable=False: weights only (fp32 + bf16 views under the same names), no gradient / moment buffers - frozen modules."""
        self.trainable = trainable
        self.index: Dict[str, Tuple[int, Tuple[int, ...]]] = {}
        off = 0
        for k, t in tensors.items():
            self.index[k] = (off, tuple(t.shape))
            off += (t.numel() + self.ALIGN - 1) // self.ALIGN * self.ALIGN
        self.numel = (off + 1023) // 1024 * 1024
        self.p32 = torch.zeros(self.numel, dtype=F32, device=device)
        for k, t in tensors.items():
            o, _ = self.index[k]
            self.p32[o:o + t.numel()].copy_(t.detach().reshape(-1).to(device=device, dtype=F32))
        if trainable:
            self.g32 = torch.zeros(self.numel, dtype=F32, device=device)
            self.m = torch.zeros(self.numel, dtype=F32, device=device)
            self.v = torch.zeros(self.numel, dtype=F32, device=device)
        self.p16 = self.p32.to(BF)
        self.step_count = 0
        self.version = 0          # bumped by every optimiser step: transposed weight copies are cached per version

    def _view(self, buf, name):
        o, shape = self.index[name]
        n = 1
        for d in shape:
            n *= d
        return buf[o:o + n].view(shape)

    def w16(self, name):
        return self._view(self.p16, name)

    def w32(self, name):
        return self._view(self.p32, name)

    def grad(self, name):
        return self._view(self.g32, name)

    def __contains__(self, name):
        return name in self.index

    def state_dict(self) -> Dict[str, torch.Tensor]:
        return {k: self.w32(k).clone() for k in self.index}

    def zero_grad(self):
        self.g32.zero_()

    def adamw_step(self, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, max_norm=1.0, grad_scale=1.0, norm_out=None):
        """clip_grad_norm_(max_norm) + torch.optim.AdamW step on the whole store in one launch; gradients are zeroed."""
        self.step_count += 1
        parts = T.sumsq_parts(self.g32) if max_norm > 0 else None
        T.adamw(self.p32, self.g32, self.m, self.v, lr, betas[0], betas[1], eps, weight_decay, self.step_count, p_bf16=self.p16,
                sumsq_parts=parts, max_norm=max_norm, grad_scale=grad_scale, norm_out=norm_out, zero_grad=True)
        self.version += 1


# ---------------------------------------------------------------------------------------------------------------- tape
class Var:
    """a 2-D activation [rows, C] (bf16 or f32) and, after backward, its gradient (same shape and dtype)."""
    __slots__ = ("v", "g", "own", "req")

    def __init__(self, v: torch.Tensor, req: bool = True):
        assert v.dim() == 2 and v.stride(1) == 1
        self.v, self.g, self.own, self.req = v, None, True, req


def _acc(var: Var, g: torch.Tensor, own: bool = True):
    """var.g += g. `own` = the caller hands the buffer over (it may be updated in place later). The first contribution is kept as it
    comes (bf16 or f32); from the second one on the sum lives in fp32 - a value with many consumers (the condition tokens of 16 decoder
    layers, the adaLN embedding of 12 DiT blocks) would otherwise round its gradient to bf16 after every addition."""
    if not var.req:
        return
    assert g.shape == var.v.shape, f"gradient {tuple(g.shape)} for a value {tuple(var.v.shape)}"
    if var.g is None:
        var.g, var.own = g, own
    elif var.own and var.g.dtype == F32:
        T.affine(g, out=var.g, accumulate=True)
    else:
        var.g, var.own = T.affine(g, base=var.g, out_dtype=F32), True


def _bf(t: torch.Tensor) -> torch.Tensor:
    return t if t.dtype == BF else T.affine(t, out_dtype=BF)


class _Params:
    """name -> views over a trainable store and an optional frozen one; `trains(name)` says whether the tensor gets a gradient."""

    def __init__(self, P: ParamStore, frozen: Optional[ParamStore] = None):
        self.P, self.F = P, frozen
        self.version = 0

    def _s(self, name):
        if name in self.P.index:
            return self.P
        assert self.F is not None and name in self.F.index, f"unknown parameter {name}"
        return self.F

    def trains(self, name) -> bool:
        return name in self.P.index

    def w16(self, name):
        return self._s(name).w16(name)

    def w32(self, name):
        return self._s(name).w32(name)

    def grad(self, name):
        return self.P.grad(name)


class Tape:
    def __init__(self, P: ParamStore, frozen: Optional[ParamStore] = None):
        self.store = P
        self.P = _Params(P, frozen)
        self.nodes: List[Callable[[], None]] = []
        self._wt: Dict[Tuple[str, Optional[Tuple[int, int]]], Tuple[int, torch.Tensor]] = {}

    def backward(self):
        for fn in reversed(self.nodes):
            fn()
        self.nodes.clear()

    # ---- parameter access helpers
    def _rows(self, t, rows):
        return t if rows is None else t[rows[0]:rows[1]]

    def _w_transposed(self, w, rows):
        key = (w, rows)
        hit = self._wt.get(key)
        if hit is None or hit[0] != self.store.version:
            hit = (self.store.version, T.transpose(self._rows(self.P.w16(w), rows)))
            self._wt[key] = hit
        return hit[1]

    # ---- ops
    def linear(self, x: Var, w: str, b: Optional[str] = None, rows: Optional[Tuple[int, int]] = None, residual: Optional[Var] = None,
               out_dtype=None) -> Var:
        """y = x @ W[rows].T + b[rows] (+ residual); nn.Linear. W is a 2-D parameter view with K % 8 == 0. Output dtype: `out_dtype`,
        else the residual's, else bf16."""
        P = self.P
        W = self._rows(P.w16(w), rows)
        bias = self._rows(P.w32(b), rows) if b else None
        xb = _bf(x.v)
        if out_dtype is None:
            out_dtype = residual.v.dtype if residual is not None else BF
        train = P.trains(w)
        y = Var(ops.linear(xb, W, bias=bias, residual=None if residual is None else residual.v, out_dtype=out_dtype),
                req=train or x.req or (residual is not None and residual.req))
        if not y.req:
            return y

        def bwd():
            dy = y.g
            if dy is None:
                return
            dyb = _bf(dy)
            if residual is not None:
                _acc(residual, dy, own=False)
            if x.req:
                assert W.shape[0] % 8 == 0, "dX GEMM needs an output width that is a multiple of 8"
                _acc(x, ops.linear(dyb, self._w_transposed(w, rows), out_dtype=x.v.dtype))
            if train:
                gW = self._rows(P.grad(w), rows)
                ops.linear(T.transpose(dyb), T.transpose(xb), out=gW, residual=gW)
                if b:
                    T.colsum(dy, out=self._rows(P.grad(b), rows), accumulate=True)
        self.nodes.append(bwd)
        return y

    def norm(self, x: Var, gamma: Optional[str], beta: Optional[str], eps: float, rms: bool = False, out_dtype=BF) -> Var:
        P = self.P
        g32 = P.w32(gamma) if gamma else None
        b32 = P.w32(beta) if beta else None
        train = bool(gamma) and P.trains(gamma)
        if out_dtype == BF:
            y = Var(ops.norm(x.v, g32, b32, eps=eps, rms=rms), req=train or x.req)
        else:
            o = torch.empty(x.v.shape, dtype=F32, device=x.v.device)
            ops.norm(x.v, g32, b32, eps=eps, rms=rms, out32=o)
            y = Var(o, req=train or x.req)
        if not y.req:
            return y

        def bwd():
            dy = y.g
            if dy is None:
                return
            dx, xhat = T.norm_bwd(x.v, dy, g32, eps, rms, want_xhat=train, dx_dtype=x.v.dtype)
            if train:
                T.colsum(dy, xhat, out=P.grad(gamma), accumulate=True)
                if beta:
                    T.colsum(dy, out=P.grad(beta), accumulate=True)
            _acc(x, dx)
        self.nodes.append(bwd)
        return y

    def act(self, x: Var, kind: str) -> Var:
        y = Var(T.act_fwd(x.v, kind), req=x.req)
        if not y.req:
            return y

        def bwd():
            if y.g is not None:
                _acc(x, T.act_bwd(x.v, y.g, kind, out_dtype=x.v.dtype))
        self.nodes.append(bwd)
        return y

    def glu(self, a: Var, b: Var) -> Var:
        """silu(a) * b (SwiGLU of LuminaFeedForward)."""
        y = Var(T.glu_fwd(a.v, b.v))

        def bwd():
            if y.g is not None:
                da, db = T.glu_bwd(a.v, b.v, y.g)
                _acc(a, da)
                _acc(b, db)
        self.nodes.append(bwd)
        return y

    def add(self, a: Var, b: Var) -> Var:
        y = Var(T.affine(a.v, base=b.v, out_dtype=F32 if F32 in (a.v.dtype, b.v.dtype) else BF), req=a.req or b.req)

        def bwd():
            if y.g is not None:
                _acc(a, y.g, own=False)
                _acc(b, y.g, own=False)
        self.nodes.append(bwd)
        return y

    def modulate(self, x: Var, s: Var, f: str, div: int, base: Optional[Var] = None, out_dtype=None) -> Var:
        """y = x * f(s[r // div]) (+ base);  f in {'id', 'one_plus', 'tanh'} - adaLN scale / tanh gates of the NextDiT block."""
        y = Var(T.affine(x.v, scale=s.v, s_div=div, s_f=f, base=None if base is None else base.v,
                         out_dtype=out_dtype or (base.v.dtype if base is not None else x.v.dtype)))

        def bwd():
            dy = y.g
            if dy is None:
                return
            if base is not None:
                _acc(base, dy, own=False)
            if x.req:
                _acc(x, T.affine(dy, scale=s.v, s_div=div, s_f=f, out_dtype=x.v.dtype))
            if s.req:
                ds = T.colsum(dy, x.v, group_rows=div)
                if f == "tanh":
                    ds = T.act_bwd(s.v, ds, "tanh", out_dtype=F32)
                _acc(s, ds)
        self.nodes.append(bwd)
        return y

    def col_scale(self, x: Var, gname: str, base: Optional[Var] = None, tanh_heads: int = 0) -> Var:
        """y = x * gamma[c] (+ base): LayerScale (dinov2_layers/layer_scale.py); tanh_heads = H: gamma[c] = tanh(gate[c // (C/H)])
        (the per-head cross-attention gate of LuminaNextDiTBlock, nextdit_traj.py:166)."""
        P = self.P
        Cd = x.v.shape[1]
        if tanh_heads:
            gvec = P.w32(gname).tanh().repeat_interleave(Cd // tanh_heads).view(1, Cd).contiguous()
        else:
            gvec = P.w32(gname).view(1, Cd)
        rows = x.v.shape[0]
        train = P.trains(gname)
        y = Var(T.affine(x.v, scale=gvec, s_div=rows, base=None if base is None else base.v,
                         out_dtype=base.v.dtype if base is not None else x.v.dtype), req=train or x.req or (base is not None and base.req))
        if not y.req:
            return y

        def bwd():
            dy = y.g
            if dy is None:
                return
            if base is not None:
                _acc(base, dy, own=False)
            if x.req:
                _acc(x, T.affine(dy, scale=gvec, s_div=rows, out_dtype=x.v.dtype))
            if not train:
                return
            dg = T.colsum(dy, x.v).view(-1)
            if tanh_heads:
                g = P.w32(gname)
                P.grad(gname).add_(dg.view(tanh_heads, -1).sum(1) * (1 - g.tanh() ** 2))
            else:
                P.grad(gname).add_(dg)
        self.nodes.append(bwd)
        return y

    def add_table(self, x: Var, tname: str, mod: int, out_dtype=None) -> Var:
        """y = x + table[r % mod]  (learned positional embeddings, broadcast over the batch)."""
        P = self.P
        Cd = x.v.shape[1]
        tab = P.w32(tname).reshape(-1, Cd)[:mod].contiguous()
        train = P.trains(tname)
        y = Var(T.affine(x.v, tab=tab, out_dtype=out_dtype or x.v.dtype), req=train or x.req)
        if not y.req:
            return y

        def bwd():
            dy = y.g
            if dy is None:
                return
            _acc(x, dy, own=False)
            if train:
                n = dy.shape[0] // mod
                T.colsum(dy.contiguous().view(n, -1), out=P.grad(tname).view(-1, Cd)[:mod].view(1, -1), accumulate=True)
        self.nodes.append(bwd)
        return y

    def attention(self, q: Tuple[Var, int], k: Tuple[Var, int], v: Tuple[Var, int], B: int, Lq: int, Lk: int, H: int, D: int,
                  causal: bool = False) -> Var:
        """softmax(q k^T / sqrt(D)) v; q / k / v are (source, first column) pairs: column blocks of width H*D of 2-D activations
        (a packed qkv projection, or separate ones)."""
        Cd = H * D

        def view(src, L):
            var, c0 = src
            return var.v[:, c0:c0 + Cd].unflatten(0, (B, L)).unflatten(-1, (H, D))
        qv, kv, vv = view(q, Lq), view(k, Lk), view(v, Lk)
        o = ops.attention(qv, kv, vv, causal=causal)
        y = Var(o.view(B * Lq, Cd), req=q[0].req or k[0].req or v[0].req)
        if not y.req:
            return y

        def bwd():
            do = y.g
            if do is None:
                return
            do = _bf(do)
            if not do.is_contiguous():
                do = do.contiguous()
            bufs: Dict[int, torch.Tensor] = {}
            for var, _ in (q, k, v):
                if id(var) not in bufs:
                    used = sum(Cd for s, _c in (q, k, v) if s is var)
                    mk = torch.empty if used == var.v.shape[1] else torch.zeros
                    bufs[id(var)] = mk(var.v.shape, dtype=BF, device=var.v.device)

            def gview(src, L):
                var, c0 = src
                return bufs[id(var)][:, c0:c0 + Cd].unflatten(0, (B, L)).unflatten(-1, (H, D))
            T.attention_bwd(qv, kv, vv, o, do.view(B, Lq, H, D), causal=causal, dq=gview(q, Lq), dk=gview(k, Lk), dv=gview(v, Lk))
            done = set()
            for var, _ in (q, k, v):
                if id(var) not in done:
                    done.add(id(var))
                    _acc(var, bufs[id(var)])
        self.nodes.append(bwd)
        return y

    def cols(self, x: Var, c0: int, c1: int) -> Var:
        """column slice view; its gradient lands in the matching columns of x's gradient."""
        y = Var(x.v[:, c0:c1], req=x.req)

        def bwd():
            if y.g is None or not x.req:
                return
            if x.g is None or not x.own or x.g.dtype != F32:
                full = torch.zeros(x.v.shape, dtype=F32, device=x.v.device)
                if x.g is not None:
                    T.affine(x.g, out=full)
                x.g, x.own = full, True
            T.affine(y.g, out=x.g[:, c0:c1], accumulate=True)
        self.nodes.append(bwd)
        return y

    def mean_tokens(self, x: Var, L: int) -> Var:
        """[N*L, C] -> f32 [N, C] mean over the L tokens of each sequence."""
        y = Var(T.colsum(x.v, group_rows=L, scale=1.0 / L))

        def bwd():
            if y.g is not None:
                _acc(x, (y.g / L).repeat_interleave(L, 0))
        self.nodes.append(bwd)
        return y

    # ---- shape plumbing (torch views / copies of small tensors)
    def repeat_seq(self, x: Var, B: int, L: int, times: int) -> Var:
        """[B*L, C] -> [B*times*L, C]: each sequence repeated `times` (internvla_n1.py:229)."""
        Cd = x.v.shape[1]
        y = Var(x.v.view(B, 1, L, Cd).expand(B, times, L, Cd).reshape(-1, Cd))

        def bwd():
            if y.g is not None:
                _acc(x, y.g.float().view(B, times, L * Cd).sum(1).view(B * L, Cd))
        self.nodes.append(bwd)
        return y

    def cat_tokens(self, parts: Sequence[Tuple[Var, int]], N: int) -> Var:
        """concatenate along the token axis: parts = [(Var [N*L_i, C], L_i)] -> [N * sum(L_i), C]."""
        Cd = parts[0][0].v.shape[1]
        y = Var(torch.cat([p.v.view(N, L, Cd) for p, L in parts], dim=1).reshape(-1, Cd))
        Ls = [L for _, L in parts]

        def bwd():
            if y.g is None:
                return
            g = y.g.view(N, sum(Ls), Cd)
            o = 0
            for (p, L) in parts:
                _acc(p, g[:, o:o + L].reshape(N * L, Cd))
                o += L
        self.nodes.append(bwd)
        return y

    def cat_cols(self, a: Var, b: Var) -> Var:
        ca = a.v.shape[1]
        y = Var(torch.cat([a.v, b.v.to(a.v.dtype)], dim=1))

        def bwd():
            if y.g is not None:
                _acc(a, y.g[:, :ca], own=False)
                _acc(b, y.g[:, ca:], own=False)
        self.nodes.append(bwd)
        return y

    def pair_goal_current(self, feat: Var, B: int, Tn: int, L: int) -> Var:
        """[B*Tn*L, C] tokens of every frame -> [B*Tn*2L, C]: (goal = frame 0 of the sample, current frame t) per sub-goal
        (internvla_n1.py:236-239 with the goal frame encoded once)."""
        Cd = feat.v.shape[1]
        f = feat.v.view(B, Tn, L, Cd)
        y = Var(torch.cat([f[:, :1].expand(B, Tn, L, Cd), f], dim=2).reshape(-1, Cd))

        def bwd():
            if y.g is None:
                return
            g = y.g.view(B, Tn, 2, L, Cd)
            d = g[:, :, 1].float().clone()
            d[:, 0] += g[:, :, 0].float().sum(1)
            _acc(feat, d.view(-1, Cd))
        self.nodes.append(bwd)
        return y


# ---------------------------------------------------------------------------------------------------------------- DINOv2 ViT-S
class DinoTrain:
    """DinoVisionTransformer.get_intermediate_layers(x)[0] with gradients (dinov2.py:298-322, block.py:82-107 - the training branch
    equals the eval branch at drop_path 0); patch-embed conv = GEMM on im2col rows, bicubic pos-embed resampling = sparse row mix."""

    D, DEPTH, HEADS, PATCH, KPAD = 384, 12, 6, 14, 592

    def __init__(self, prefix: str, device, img_size: int = 224, mean=RESNET_MEAN, std=RESNET_STD, channels: int = 3):
        """mean / std: input normalisation fused into the im2col kernel; channels = 1: a depth map replicated to 3 channels
        (navdp_backbone.py:274-279). Parameters are looked up on the tape (trainable or frozen store)."""
        self.p, self.mean, self.std, self.channels = prefix, mean, std, channels
        g = img_size // self.PATCH
        self.L = g * g
        # the interpolation is linear in pos_embed: recover its sparse matrix once by resampling an identity basis on the host
        import torch.nn.functional as Fn
        n_src = 37
        w0 = g + 0.1
        eye = torch.eye(n_src * n_src, dtype=torch.float32).view(n_src * n_src, 1, n_src, n_src)
        A = Fn.interpolate(eye, scale_factor=(w0 / n_src, w0 / n_src), mode="bicubic", antialias=False)    # [1369, 1, g, g]
        A = A.view(n_src * n_src, self.L).t().contiguous()                                                  # [L, 1369]
        self.fwd_idx, self.fwd_coef = self._ell(A, device)
        self.bwd_idx, self.bwd_coef = self._ell(A.t().contiguous(), device)

    @staticmethod
    def _ell(A: torch.Tensor, device):
        nz = (A != 0)
        taps = int(nz.sum(1).max().item())
        idx = torch.full((A.shape[0], taps), -1, dtype=torch.int32)
        coef = torch.zeros(A.shape[0], taps, dtype=torch.float32)
        for r in range(A.shape[0]):
            c = nz[r].nonzero().flatten()
            idx[r, :c.numel()] = c.to(torch.int32)
            coef[r, :c.numel()] = A[r, c]
        return idx.to(device), coef.to(device)

    def forward(self, tape: Tape, frames: torch.Tensor) -> Var:
        """frames [n, 224, 224, C] -> Var [n * 256, 384] bf16 patch tokens after the final LayerNorm (cls dropped)."""
        P, p, D, L = tape.P, self.p, self.D, self.L
        n = frames.shape[0]
        dev = frames.device
        Tt = L + 1
        assert frames.shape[-1] == self.channels
        patches = torch.empty(n * L, self.KPAD, dtype=BF, device=dev)
        ops.patchify(frames.contiguous(), patches, self.mean, self.std, self.PATCH)
        wname = p + "patch_embed.proj.weight"
        train = P.trains(wname)
        wpad = torch.zeros(D, self.KPAD, dtype=BF, device=dev)
        wpad[:, :588] = P.w16(wname).view(D, 588)
        pos_src = P.w32(p + "pos_embed").view(-1, D)                                   # [1370, 384]
        pos = T.sparse_rows(pos_src[1:].contiguous(), self.fwd_idx, self.fwd_coef)     # [L, 384] f32
        x = torch.empty(n, Tt, D, dtype=F32, device=dev)
        ops.linear(patches.view(n, L, self.KPAD), wpad, bias=P.w32(p + "patch_embed.proj.bias"), residual=pos, out=x[:, 1:, :], batched=True)
        x[:, 0, :] = P.w32(p + "cls_token").view(1, D) + pos_src[:1]
        xv = x_embed = Var(x.view(n * Tt, D), req=train)

        def bwd_embed():
            dx = x_embed.g
            if dx is None:
                return
            d3 = dx.view(n, Tt, D)
            dtok = _bf(d3[:, 1:, :].reshape(n * L, D))
            gw = torch.zeros(D, self.KPAD, dtype=F32, device=dev)
            ops.linear(T.transpose(dtok), T.transpose(patches), out=gw, residual=gw)
            P.grad(wname).view(D, 588).add_(gw[:, :588])
            T.colsum(dtok, out=P.grad(p + "patch_embed.proj.bias"), accumulate=True)
            dpos_all = T.colsum(d3.reshape(n, Tt * D)).view(Tt, D)                     # summed over the frames
            gpos = P.grad(p + "pos_embed").view(-1, D)
            gpos[:1] += dpos_all[:1]
            P.grad(p + "cls_token").view(1, D).add_(dpos_all[:1])
            T.sparse_rows(dpos_all[1:].contiguous(), self.bwd_idx, self.bwd_coef, out=gpos[1:], accumulate=True)
        if train:
            tape.nodes.append(bwd_embed)

        for i in range(self.DEPTH):
            b = f"{p}blocks.{i}"
            h = tape.norm(xv, b + ".norm1.weight", b + ".norm1.bias", 1e-6)
            qkv = tape.linear(h, b + ".attn.qkv.weight", b + ".attn.qkv.bias")
            att = tape.attention((qkv, 0), (qkv, D), (qkv, 2 * D), n, Tt, Tt, self.HEADS, D // self.HEADS)
            y = tape.linear(att, b + ".attn.proj.weight", b + ".attn.proj.bias")
            xv = tape.col_scale(y, b + ".ls1.gamma", base=xv)
            h = tape.norm(xv, b + ".norm2.weight", b + ".norm2.bias", 1e-6)
            h = tape.act(tape.linear(h, b + ".mlp.fc1.weight", b + ".mlp.fc1.bias"), "gelu_erf")
            y = tape.linear(h, b + ".mlp.fc2.weight", b + ".mlp.fc2.bias")
            xv = tape.col_scale(y, b + ".ls2.gamma", base=xv)
        out = tape.norm(xv, p + "norm.weight", p + "norm.bias", 1e-6)
        # drop the cls token
        tok = Var(out.v.view(n, Tt, D)[:, 1:, :].reshape(n * L, D), req=out.req)

        def bwd_drop():
            if tok.g is None:
                return
            g = torch.zeros(n, Tt, D, dtype=tok.g.dtype, device=dev)
            g[:, 1:, :] = tok.g.view(n, L, D)
            _acc(out, g.view(n * Tt, D))
        tape.nodes.append(bwd_drop)
        return tok


# ---------------------------------------------------------------------------------------------------------------- nn.Transformer layers
def _mha(tape: Tape, xq: Var, xkv: Var, p: str, B: int, Lq: int, Lk: int, H: int, d: int, residual: Optional[Var] = None,
         causal: bool = False) -> Var:
    """nn.MultiheadAttention(batch_first=True): packed in_proj, SDPA, out_proj (+ the residual of the surrounding layer, fp32 sum)."""
    w, b = p + ".in_proj_weight", p + ".in_proj_bias"
    if xq is xkv:
        qkv = tape.linear(xq, w, b)
        att = tape.attention((qkv, 0), (qkv, d), (qkv, 2 * d), B, Lq, Lk, H, d // H, causal=causal)
    else:
        qp = tape.linear(xq, w, b, rows=(0, d))
        kvp = tape.linear(xkv, w, b, rows=(d, 3 * d))
        att = tape.attention((qp, 0), (kvp, 0), (kvp, d), B, Lq, Lk, H, d // H, causal=causal)
    return tape.linear(att, p + ".out_proj.weight", p + ".out_proj.bias", residual=residual, out_dtype=F32 if residual is not None else None)


def encoder_layer(tape: Tape, x: Var, p: str, B: int, L: int, H: int, d: int) -> Var:
    """nn.TransformerEncoderLayer(batch_first=True, norm_first=False, activation=relu), dropout off."""
    x = tape.norm(_mha(tape, x, x, p + ".self_attn", B, L, L, H, d, residual=x), p + ".norm1.weight", p + ".norm1.bias", 1e-5)
    ff = tape.linear(tape.act(tape.linear(x, p + ".linear1.weight", p + ".linear1.bias"), "relu"), p + ".linear2.weight", p + ".linear2.bias",
                     residual=x, out_dtype=F32)
    return tape.norm(ff, p + ".norm2.weight", p + ".norm2.bias", 1e-5)


def decoder_layer(tape: Tape, x: Var, mem: Var, p: str, B: int, Lq: int, Lm: int, H: int, d: int) -> Var:
    """nn.TransformerDecoderLayer(batch_first=True, norm_first=False, activation=relu), no masks, dropout off."""
    x = tape.norm(_mha(tape, x, x, p + ".self_attn", B, Lq, Lq, H, d, residual=x), p + ".norm1.weight", p + ".norm1.bias", 1e-5)
    x = tape.norm(_mha(tape, x, mem, p + ".multihead_attn", B, Lq, Lm, H, d, residual=x), p + ".norm2.weight", p + ".norm2.bias", 1e-5)
    ff = tape.linear(tape.act(tape.linear(x, p + ".linear1.weight", p + ".linear1.bias"), "relu"), p + ".linear2.weight", p + ".linear2.bias",
                     residual=x, out_dtype=F32)
    return tape.norm(ff, p + ".norm3.weight", p + ".norm3.bias", 1e-5)


# ---------------------------------------------------------------------------------------------------------------- NextDiT
def timestep_embedding(t: torch.Tensor, dim: int = 256) -> torch.Tensor:
    """diffusers Timesteps(256, flip_sin_to_cos=True, downscale_freq_shift=0): [cos | sin] of t * 10000^(-i/128)."""
    half = dim // 2
    freqs = torch.exp(-math.log(10000) * torch.arange(half, dtype=F32, device=t.device) / half)
    a = t[:, None].float() * freqs[None]
    return torch.cat([torch.cos(a), torch.sin(a)], dim=-1)


def sinusoidal_positions(Tn: int, dim: int, device) -> torch.Tensor:
    """SinusoidalPositionalEncoding(dim)(arange(T)) (internvla_n1_arch.py:50-73)."""
    half = dim // 2
    exponent = -torch.arange(half, dtype=F32, device=device) * (torch.log(torch.tensor(10000.0)) / half)
    freqs = torch.arange(Tn, dtype=F32, device=device).unsqueeze(-1) * exponent.exp()
    return torch.cat([torch.sin(freqs), torch.cos(freqs)], dim=-1).contiguous()


def _dit_attention(tape: Tape, xq: Var, xkv: Var, p: str, N: int, Lq: int, Lk: int, H: int = 6, d: int = 384) -> Var:
    """diffusers Attention + LuminaAttnProcessor2_0: qk LayerNorm across heads, no rotary, no mask, no out projection."""
    q = tape.norm(tape.linear(xq, p + ".to_q.weight"), p + ".norm_q.weight", p + ".norm_q.bias", 1e-5)
    k = tape.norm(tape.linear(xkv, p + ".to_k.weight"), p + ".norm_k.weight", p + ".norm_k.bias", 1e-5)
    v = tape.linear(xkv, p + ".to_v.weight")
    return tape.attention((q, 0), (k, 0), (v, 0), N, Lq, Lk, H, d // H)


def dit_block(tape: Tape, x: Var, enc: Var, silu_temb: Var, p: str, N: int, Tn: int, Lc: int) -> Var:
    """LuminaNextDiTBlock.forward (nextdit_traj.py:121-178); x f32 [N*Tn, 384], enc bf16 [N*Lc, 384], silu_temb [N, 384]."""
    D = 384
    emb = tape.linear(silu_temb, p + ".norm1.linear.weight", p + ".norm1.linear.bias", out_dtype=F32)       # [N, 1536]
    scale_msa, gate_msa, scale_mlp, gate_mlp = (tape.cols(emb, i * D, (i + 1) * D) for i in range(4))
    nh = tape.modulate(tape.norm(x, p + ".norm1.norm.weight", None, 1e-5, rms=True), scale_msa, "one_plus", Tn)
    sa = _dit_attention(tape, nh, nh, p + ".attn1", N, Tn, Tn)
    ca = _dit_attention(tape, nh, tape.norm(enc, p + ".norm1_context.weight", None, 1e-5, rms=True), p + ".attn2", N, Tn, Lc)
    mixed = tape.col_scale(ca, p + ".gate", base=sa, tanh_heads=6)
    h = tape.linear(mixed, p + ".attn2.to_out.0.weight")
    x = tape.modulate(tape.norm(h, p + ".norm2.weight", None, 1e-5, rms=True), gate_msa, "tanh", Tn, base=x)
    y = tape.modulate(tape.norm(x, p + ".ffn_norm1.weight", None, 1e-5, rms=True), scale_mlp, "one_plus", Tn)
    y = tape.linear(tape.glu(tape.linear(y, p + ".feed_forward.linear_1.weight"), tape.linear(y, p + ".feed_forward.linear_3.weight")),
                    p + ".feed_forward.linear_2.weight")
    return tape.modulate(tape.norm(y, p + ".ffn_norm2.weight", None, 1e-5, rms=True), gate_mlp, "tanh", Tn, base=x)


def traj_dit(tape: Tape, x: Var, timestep: torch.Tensor, z: Var, N: int, Tn: int, Lc: int, p: str = "traj_dit.model.", n_layers: int = 12) -> Var:
    """LuminaNextDiT2DModel.forward (nextdit_traj.py:299-368): x f32 [N*Tn, 384], timestep f32 [N], z bf16 [N*Lc, 768] -> bf16 [N*Tn, 384]."""
    enc = tape.linear(tape.act(tape.linear(z, p + "caption_projection.linear_1.weight", p + "caption_projection.linear_1.bias"), "gelu_tanh"),
                      p + "caption_projection.linear_2.weight", p + "caption_projection.linear_2.bias")
    te_in = Var(timestep_embedding(timestep).to(BF), req=False)
    te = tape.linear(tape.act(tape.linear(te_in, p + "time_caption_embed.timestep_embedder.linear_1.weight",
                                          p + "time_caption_embed.timestep_embedder.linear_1.bias"), "silu"),
                     p + "time_caption_embed.timestep_embedder.linear_2.weight", p + "time_caption_embed.timestep_embedder.linear_2.bias",
                     out_dtype=F32)
    pool = tape.mean_tokens(enc, Lc)
    cap = tape.linear(tape.norm(pool, p + "time_caption_embed.caption_embedder.0.weight", p + "time_caption_embed.caption_embedder.0.bias", 1e-5),
                      p + "time_caption_embed.caption_embedder.1.weight", p + "time_caption_embed.caption_embedder.1.bias", residual=te)
    silu_temb = tape.act(cap, "silu")                       # FP32SiLU(temb), shared by every block's adaLN linear and norm_out
    for i in range(n_layers):
        x = dit_block(tape, x, enc, silu_temb, f"{p}layers.{i}", N, Tn, Lc)
    scale = tape.linear(silu_temb, p + "norm_out.linear_1.weight", p + "norm_out.linear_1.bias", out_dtype=F32)
    x = tape.modulate(tape.norm(x, None, None, 1e-6), scale, "one_plus", Tn)
    return tape.linear(x, p + "norm_out.linear_2.weight", p + "norm_out.linear_2.bias")


# ---------------------------------------------------------------------------------------------------------------- the S1 loss
S1_TRAINABLE_PREFIXES = ("action_encoder", "action_decoder", "traj_dit", "cond_projector", "memory_encoder", "rgb_resampler", "rgb_model")
"""internvla_n1_trainer.py:104-117 (system1 = nextdit*): these + latent_queries receive gradients; everything else is frozen."""


class NextDiTSftHead:
    """Loss + gradients of the nextdit_async branch of InternVLAN1ForCausalLM.forward(labels=...) (internvla_n1.py:222-286)."""

    def __init__(self, sd: Dict[str, torch.Tensor], device, n_query: int = 4, extra_trainable: Sequence[str] = ()):
        """extra_trainable: further tensors of `sd` to keep in the same flat store (`latent_queries`, whose gradient the caller adds)."""
        keep = {k: v for k, v in sd.items() if (k.startswith(S1_TRAINABLE_PREFIXES) and not k.endswith("mask_token")) or k in extra_trainable}
        self.P = ParamStore(keep, device)
        self.device = device
        self.n_query = n_query
        self.dino = DinoTrain("rgb_model.", device)
        self.pos = sinusoidal_positions(32, 384, device)

    def loss_and_grads(self, hidden_q: torch.Tensor, traj_images: torch.Tensor, traj_poses: torch.Tensor, video_frame_num: torch.Tensor,
                       noise: torch.Tensor, t_index: torch.Tensor, num_train_timesteps: int = 1000, loss_scale: float = 1.0):
        """hidden_q bf16 [B, n_query, 3584] (final-norm LLM states of the trajectory tokens); traj_images [B, T, 224, 224, 3] in 0..1;
        traj_poses f32 [B, T, 32, 3]; video_frame_num int [B]; noise f32 [B*T, 32, 3]; t_index int [B*T] (the reference's `indices`).
        Gradients are ACCUMULATED into the parameter store; returns (loss f32 [1], d loss / d hidden_q bf16 [B, n_query, 3584])."""
        P, dev, nq = self.P, self.device, self.n_query
        B, Tn = traj_images.shape[:2]
        N, Ta = B * Tn, traj_poses.shape[2]
        tape = Tape(P)
        hq = Var(hidden_q.reshape(B * nq, -1).to(BF).contiguous())
        # -- condition tokens
        feat = self.dino.forward(tape, traj_images.reshape(N, *traj_images.shape[2:]).to(dev))          # [N*256, 384]
        dp_feat = tape.pair_goal_current(feat, B, Tn, 256)                                               # [N*512, 384]
        mem = tape.add_table(dp_feat, "memory_encoder.memory_pos", 512)
        for i in range(3):
            mem = encoder_layer(tape, mem, f"memory_encoder.encoder.layers.{i}", N, 512, 6, 384)
        memory_feat = tape.cat_cols(dp_feat, mem)                                                        # [N*512, 768]
        qtok = Var((P.w32("rgb_resampler.query_tokens") + P.w32("rgb_resampler.query_pos")).to(BF).unsqueeze(0).expand(N, -1, -1).reshape(N * 32, 768))

        def bwd_queries():
            if qtok.g is not None:
                g = qtok.g.float().view(N, -1).sum(0).view(32, 768)
                P.grad("rgb_resampler.query_tokens").add_(g)
                P.grad("rgb_resampler.query_pos").add_(g)
        tape.nodes.append(bwd_queries)
        q = qtok
        for i in range(3):
            q = decoder_layer(tape, q, memory_feat, f"rgb_resampler.decoder.layers.{i}", N, 32, 512, 12, 768)
        lat = tape.repeat_seq(hq, B, nq, Tn)                                                             # [N*nq, 3584]
        lat = tape.linear(tape.act(tape.linear(lat, "cond_projector.0.weight", "cond_projector.0.bias"), "gelu_tanh"),
                          "cond_projector.2.weight", "cond_projector.2.bias")
        Lc = 32 + nq
        latents = tape.cat_tokens([(q, 32), (lat, nq)], N)                                               # [N*Lc, 768]
        # -- noisy trajectory (flow matching, internvla_n1.py:259-275)
        poses = traj_poses.reshape(N, Ta, 3).to(device=dev, dtype=F32)
        timesteps = (num_train_timesteps - t_index.to(dev)).to(F32)              # FlowMatchEulerDiscreteScheduler().timesteps[indices]
        sig = (timesteps / num_train_timesteps).view(N, 1, 1)
        noise = noise.to(device=dev, dtype=F32).view(N, Ta, 3)
        noisy = ((1 - sig) * poses + sig * noise).reshape(N * Ta, 3).contiguous()
        x0 = Var(T.small_linear(noisy, P.w32("action_encoder.weight"), P.w32("action_encoder.bias"), tab=self.pos[:Ta].contiguous()))

        def bwd_encoder():
            dy = x0.g
            if dy is None:
                return
            gw = P.grad("action_encoder.weight")
            for kk in range(3):
                T.colsum(dy, noisy[:, kk], out=gw[:, kk], x2_bcast=True, out_cs=3, accumulate=True)
            T.colsum(dy, out=P.grad("action_encoder.bias"), accumulate=True)
        tape.nodes.append(bwd_encoder)
        pred = traj_dit(tape, x0, timesteps, latents, N, Ta, Lc)                                         # bf16 [N*Ta, 384]
        out = T.small_linear(pred.v, P.w32("action_decoder.weight"), P.w32("action_decoder.bias"))       # f32 [N*Ta, 3]
        target = (noise - poses).reshape(N * Ta, 3).contiguous()
        mask = (torch.arange(Tn, device=dev)[None, :] < video_frame_num.to(dev)[:, None]).to(F32).reshape(N).contiguous()
        loss, dout = T.mse_masked(out, target, mask, Ta, loss_scale=loss_scale)
        # action_decoder backward
        gw = P.grad("action_decoder.weight")
        for nn_ in range(3):
            T.colsum(pred.v, dout[:, nn_], out=gw[nn_], x2_bcast=True, accumulate=True)
        T.colsum(dout, out=P.grad("action_decoder.bias"), accumulate=True)
        pred.g = T.small_linear(dout, P.w32("action_decoder.weight"), out_dtype=BF, w_transposed=True)
        tape.backward()
        dh = hq.g if hq.g is not None else torch.zeros_like(hq.v)
        return loss, dh.view(B, nq, -1)


# ---------------------------------------------------------------------------------------------------------------- NavDP head (navdp_async)
def decoder_layer_prenorm(tape: Tape, x: Var, mem: Var, p: str, B: int, Lq: int, Lm: int, H: int, d: int, causal: bool) -> Var:
    """nn.TransformerDecoderLayer(batch_first=True, norm_first=True, activation='gelu'), causal tgt_mask, no memory mask, dropout off;
    x is the fp32 residual stream."""
    h = tape.norm(x, p + ".norm1.weight", p + ".norm1.bias", 1e-5)
    x = _mha(tape, h, h, p + ".self_attn", B, Lq, Lq, H, d, residual=x, causal=causal)
    h = tape.norm(x, p + ".norm2.weight", p + ".norm2.bias", 1e-5)
    x = _mha(tape, h, mem, p + ".multihead_attn", B, Lq, Lm, H, d, residual=x)
    h = tape.norm(x, p + ".norm3.weight", p + ".norm3.bias", 1e-5)
    return tape.linear(tape.act(tape.linear(h, p + ".linear1.weight", p + ".linear1.bias"), "gelu_erf"), p + ".linear2.weight", p + ".linear2.bias",
                       residual=x, out_dtype=F32)


def ddpm_alphas_cumprod(num_train_timesteps: int) -> torch.Tensor:
    """diffusers DDPMScheduler(beta_schedule='squaredcos_cap_v2'): betas_for_alpha_bar (cosine, max_beta 0.999) -> cumprod(1 - beta), fp32."""
    def alpha_bar(t):
        return math.cos((t + 0.008) / 1.008 * math.pi / 2) ** 2
    betas = [min(1 - alpha_bar((i + 1) / num_train_timesteps) / alpha_bar(i / num_train_timesteps), 0.999) for i in range(num_train_timesteps)]
    return torch.cumprod(1.0 - torch.tensor(betas, dtype=F32), dim=0)


def sinusoidal_pos_emb(x: torch.Tensor, dim: int) -> torch.Tensor:
    """SinusoidalPosEmb (navdp_backbone.py:9-21)."""
    half = dim // 2
    e = torch.exp(torch.arange(half, dtype=F32, device=x.device) * -(math.log(10000) / (half - 1)))
    e = x[:, None].float() * e[None, :]
    return torch.cat((e.sin(), e.cos()), dim=-1)


IMAGENET_MEAN_BF16 = (0.484375, 0.45703125, 0.40625)          # DAT_RGBD_Patch_Backbone keeps its constants in bf16 (navdp_backbone.py:119-127)
IMAGENET_STD_BF16 = (0.2294921875, 0.2236328125, 0.224609375)


class NavDPSftHead:
    """Loss + gradients of the navdp_async branch of InternVLAN1ForCausalLM.forward(labels=...) (internvla_n1.py:287-303) =
    NavDP_Policy_DPT_CriticSum_DAT.forward_vlm_traj (internvla_n1/navdp.py:291-312): epsilon-prediction MSE of the 16-layer causal
    decoder, trained with everything in `navdp.*` except the RGB DINOv2 (internvla_n1_trainer.py:118-121)."""

    def __init__(self, sd: Dict[str, torch.Tensor], device, cfg: dict, n_query: int = 4, extra_trainable: Sequence[str] = ()):
        frozen = {k: v for k, v in sd.items() if "rgb_model" in k and not k.endswith("mask_token")}
        keep = {k: v for k, v in sd.items() if k not in frozen and not k.endswith("mask_token") and (k in extra_trainable or not k.startswith(("model.", "visual.", "lm_head")))}
        self.P = ParamStore(keep, device)
        self.F = ParamStore(frozen, device, trainable=False)
        self.device, self.cfg, self.n_query = device, cfg, n_query
        self.rgb = DinoTrain("rgbd_encoder.rgb_model.", device, mean=IMAGENET_MEAN_BF16, std=IMAGENET_STD_BF16)
        self.depth = DinoTrain("rgbd_encoder.depth_model.", device, mean=(0.0, 0.0, 0.0), std=(1.0, 1.0, 1.0), channels=1)
        self.acp = ddpm_alphas_cumprod(cfg["num_train_timesteps"]).to(device)

    def loss_and_grads(self, hidden_q: torch.Tensor, traj_images: torch.Tensor, traj_depths: torch.Tensor, traj_poses: torch.Tensor,
                       video_frame_num: torch.Tensor, noise: torch.Tensor, timesteps: torch.Tensor, loss_scale: float = 1.0):
        """hidden_q bf16 [B, n_query, 3584]; traj_images [B, T, 224, 224, 3] in 0..1; traj_depths [B, T, 224, 224] metres; traj_poses f32
        [B, T, P, 3]; noise f32 [B*T, P, 3]; timesteps int [B*T] in [0, num_train_timesteps) (the reference's sample_noise draws).
        Gradients are ACCUMULATED into the store; returns (loss f32 [1], d loss / d hidden_q bf16 [B, n_query, 3584])."""
        P, dev, nq, cfg = self.P, self.device, self.n_query, self.cfg
        D, H, M = cfg["token_dim"], cfg["heads"], cfg["memory_size"]
        B, Tn = traj_images.shape[:2]
        N, Ta = B * Tn, traj_poses.shape[2]
        tape = Tape(P, self.F)
        hq = Var(hidden_q.reshape(B * nq, -1).to(BF).contiguous())
        # ---- goal token: vlm_embed_mlp + 1-query compressor
        h = tape.repeat_seq(hq, B, nq, Tn)
        h = tape.act(tape.linear(h, "vlm_embed_mlp.0.weight", "vlm_embed_mlp.0.bias"), "relu")
        h = tape.act(tape.linear(h, "vlm_embed_mlp.2.weight", "vlm_embed_mlp.2.bias"), "relu")
        h = tape.linear(h, "vlm_embed_mlp.4.weight", "vlm_embed_mlp.4.bias")
        h = tape.add_table(h, "goal_compressor.token_positional_encoding.position_embedding.weight", nq)
        gq = Var((P.w32("goal_compressor.target_embedding.weight") + P.w32("goal_compressor.query_positional_encoding.position_embedding.weight")[:1])
                 .to(BF).expand(N, D).contiguous())

        def bwd_gq():
            if gq.g is not None:
                g = gq.g.float().sum(0, keepdim=True)
                P.grad("goal_compressor.target_embedding.weight").add_(g)
                P.grad("goal_compressor.query_positional_encoding.position_embedding.weight")[:1].add_(g)
        tape.nodes.append(bwd_gq)
        goal = _mha(tape, gq, h, "goal_compressor.cross_attention", N, 1, nq, H, D)                      # [N, D]
        # ---- RGB-D memory tokens (goal frame = frame 0 of the sample; every distinct frame is encoded once)
        rgb = self.rgb.forward(tape, traj_images.reshape(N, *traj_images.shape[2:]).to(dev))           # frozen: no backward
        dep = self.depth.forward(tape, traj_depths.reshape(N, *traj_depths.shape[2:4], 1).to(dev).float())
        tok = tape.cat_tokens([(tape.pair_goal_current(rgb, B, Tn, 256), 512), (tape.pair_goal_current(dep, B, Tn, 256), 512)], N)
        tok = tape.add_table(tok, "rgbd_encoder.former_pe.weight", 2 * M * 256)
        fq = Var(P.w16("rgbd_encoder.former_query.weight").unsqueeze(0).expand(N, -1, -1).reshape(N * M * 16, 384))

        def bwd_fq():
            if fq.g is not None:
                P.grad("rgbd_encoder.former_query.weight").add_(fq.g.float().view(N, -1).sum(0).view(M * 16, 384))
        tape.nodes.append(bwd_fq)
        q = fq
        for i in range(2):
            q = decoder_layer(tape, q, tok, f"rgbd_encoder.former_net.layers.{i}", N, M * 16, 2 * M * 256, 8, 384)
        rgbd = tape.linear(q, "rgbd_encoder.project_layer.weight", "rgbd_encoder.project_layer.bias")   # [N * M*16, D]
        # ---- condition and noisy actions (sample_noise, navdp.py:163-175)
        ts = timesteps.to(dev).long()
        te = Var(sinusoidal_pos_emb(ts, D).to(BF), req=False)
        Lc = 2 + M * 16
        cond = tape.add_table(tape.cat_tokens([(te, 1), (goal, 1), (rgbd, M * 16)], N), "cond_pos_embed", Lc)
        x = traj_poses.reshape(N, Ta, 3).to(device=dev, dtype=F32)
        noise = noise.to(device=dev, dtype=F32).view(N, Ta, 3)
        a = self.acp[ts].view(N, 1, 1)
        noisy = (a.sqrt() * x + (1 - a).sqrt() * noise).reshape(N * Ta, 3).contiguous()
        x0 = Var(T.small_linear(noisy, P.w32("input_embed.weight"), P.w32("input_embed.bias")))

        def bwd_embed():
            dy = x0.g
            if dy is None:
                return
            gw = P.grad("input_embed.weight")
            for kk in range(3):
                T.colsum(dy, noisy[:, kk], out=gw[:, kk], x2_bcast=True, out_cs=3, accumulate=True)
            T.colsum(dy, out=P.grad("input_embed.bias"), accumulate=True)
        tape.nodes.append(bwd_embed)
        y = tape.add_table(x0, "out_pos_embed", Ta)
        taps = getattr(self, "taps", None)
        if taps is not None:
            taps.append(("cond", cond.v.float().clone()))
            taps.append(("x0", y.v.float().clone()))
        for i in range(cfg["temporal_depth"]):
            y = decoder_layer_prenorm(tape, y, cond, f"decoder.layers.{i}", N, Ta, Lc, H, D, causal=True)
            if taps is not None:
                taps.append((f"layer{i}", y.v.float().clone()))
        pred = tape.norm(y, "layernorm.weight", "layernorm.bias", 1e-5)
        out = T.small_linear(pred.v, P.w32("action_head.weight"), P.w32("action_head.bias"))           # f32 [N*Ta, 3]
        self.last_prediction = out
        if taps is not None:
            taps.append(("lnout", pred.v.float().clone()))
        mask = (torch.arange(Tn, device=dev)[None, :] < video_frame_num.to(dev)[:, None]).to(F32).reshape(N).contiguous()
        loss, dout = T.mse_masked(out, noise.reshape(N * Ta, 3).contiguous(), mask, Ta, loss_scale=loss_scale)
        gw = P.grad("action_head.weight")
        for nn_ in range(3):
            T.colsum(pred.v, dout[:, nn_], out=gw[nn_], x2_bcast=True, accumulate=True)
        T.colsum(dout, out=P.grad("action_head.bias"), accumulate=True)
        pred.g = T.small_linear(dout, P.w32("action_head.weight"), out_dtype=BF, w_transposed=True)
        tape.backward()
        dh = hq.g if hq.g is not None else torch.zeros_like(hq.v)
        return loss, dh.view(B, nq, -1)
