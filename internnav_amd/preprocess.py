"""On-device frame pre-processing: the host work the reference does with PIL and the HF image processor before
every policy call, as bit-exact device kernels fed with raw uint8 camera frames.

  reference host path                                              here
  ---------------------------------------------------------------  --------------------------------------------------------------
  Image.fromarray(rgb).resize((resize_w, resize_h))                 resize_u8 (W pass, H pass) with PIL's own coefficient tables
      internvla_n1_policy.py:105-116 (PIL bicubic, 8-bit)
  processor(images=...): smart_resize -> PIL bicubic -> rescale     resize_u8 again + qwen_patchify_u8 (3 x 256 table computed
      1/255 -> normalize (CLIP mean/std) -> patchify                with the processor's arithmetic: float64 rescale, fp32 normalize)
      internvla_n1_policy.py:163-165, transformers
      image_processing_qwen2_vl.py
  np.array(Image.fromarray(rgb).resize((224, 224))) / 255.0         resize_u8 + u8_lut
      internvla_n1_agent.py:309-317

The tables are the specification PIL publishes in libImaging/Resample.c (precompute_coeffs, bicubic_filter with a = -0.5,
normalize_coeffs_8bpc with 22 fractional bits); tests compare the device output with PIL itself, byte for byte."""
from __future__ import annotations

import math
from typing import Dict, Tuple

import numpy as np
import torch

from . import ops

PRECISION_BITS = 32 - 8 - 2
CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def _bicubic(x: float, a: float = -0.5) -> float:
    x = abs(x)
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def pil_bicubic_tables(in_size: int, out_size: int, fixed_point: bool = True) -> Tuple[np.ndarray, np.ndarray]:
    """(bounds int32 [out, 2], coefs [out, ksize]) of Pillow's bicubic resample from in_size to out_size samples: int32 22-bit fixed
    point for 8-bit images (normalize_coeffs_8bpc), the float64 weights themselves for mode "F" images (fixed_point=False)."""
    scale = filterscale = in_size / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = 2.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    kk = np.zeros((out_size, ksize), dtype=np.float64)
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        ww = 0.0
        for x in range(xmax):
            w = _bicubic((x + xmin - center + 0.5) * ss)
            kk[xx, x] = w
            ww += w
        if ww != 0.0:
            kk[xx, :xmax] /= ww
        bounds[xx] = (xmin, xmax)
    if not fixed_point:
        return bounds, kk
    fixed = np.where(kk < 0, (-0.5 + kk * (1 << PRECISION_BITS)).astype(np.int64), (0.5 + kk * (1 << PRECISION_BITS)).astype(np.int64))
    return bounds, fixed.astype(np.int32)


def smart_resize(height: int, width: int, factor: int = 28, min_pixels: int = 56 * 56, max_pixels: int = 14 * 14 * 4 * 1280) -> Tuple[int, int]:
    """transformers image_processing_qwen2_vl.smart_resize (round = Python's banker's rounding, as there)."""
    h_bar, w_bar = round(height / factor) * factor, round(width / factor) * factor
    if h_bar * w_bar > max_pixels:
        beta = math.sqrt((height * width) / max_pixels)
        h_bar = max(factor, math.floor(height / beta / factor) * factor)
        w_bar = max(factor, math.floor(width / beta / factor) * factor)
    elif h_bar * w_bar < min_pixels:
        beta = math.sqrt(min_pixels / (height * width))
        h_bar, w_bar = math.ceil(height * beta / factor) * factor, math.ceil(width * beta / factor) * factor
    return h_bar, w_bar


def qwen_normalize_table(mean=CLIP_MEAN, std=CLIP_STD, rescale_factor: float = 1 / 255) -> np.ndarray:
    """f32 [3, 256]: the processor's rescale (float64 multiply, cast to fp32) then normalize ((x - mean) / std in fp32) per byte value."""
    v = (np.arange(256, dtype=np.float64) * rescale_factor).astype(np.float32)
    m, s = np.asarray(mean, dtype=np.float32), np.asarray(std, dtype=np.float32)
    return np.stack([(v - m[c]) / s[c] for c in range(3)]).astype(np.float32)


class FramePreprocessor:
    """Device pipeline for raw uint8 frames [n, H, W, 3]. Tables are cached per (in_size, out_size)."""

    def __init__(self, device="cuda:0", resize_w: int = 384, resize_h: int = 384, min_pixels: int = 56 * 56, max_pixels: int = 14 * 14 * 4 * 1280,
                 patch_size: int = 14, merge_size: int = 2, temporal_patch_size: int = 2):
        self.device = torch.device(device)
        self.resize_w, self.resize_h = resize_w, resize_h
        self.min_pixels, self.max_pixels = min_pixels, max_pixels
        self.ps, self.merge, self.tdup = patch_size, merge_size, temporal_patch_size
        self._tables: Dict[Tuple[int, int], Tuple[torch.Tensor, torch.Tensor]] = {}
        self._tables_f: Dict[Tuple[int, int], Tuple[torch.Tensor, torch.Tensor]] = {}
        self.qwen_lut = torch.from_numpy(qwen_normalize_table()).to(self.device)
        self.unit_lut = torch.from_numpy((np.arange(256, dtype=np.float64) / 255.0).astype(np.float32)).to(self.device)  # np.array(img) / 255.0

    def _table(self, n_in: int, n_out: int):
        key = (n_in, n_out)
        if key not in self._tables:
            b, k = pil_bicubic_tables(n_in, n_out)
            self._tables[key] = (torch.from_numpy(b).to(self.device), torch.from_numpy(k).to(self.device))
        return self._tables[key]

    def resize(self, frames: torch.Tensor, w: int, h: int) -> torch.Tensor:
        """PIL Image.resize((w, h)) (bicubic, 8-bit) of every frame: horizontal pass, then vertical pass."""
        assert frames.dtype == torch.uint8 and frames.dim() == 4 and frames.is_contiguous()
        n, H, W, Cc = frames.shape
        x = frames
        if W != w:
            out = torch.empty(n, H, w, Cc, dtype=torch.uint8, device=frames.device)
            x = ops.resize_u8(x, out, *self._table(W, w), axis=2)
        if H != h:
            out = torch.empty(n, h, x.shape[2], Cc, dtype=torch.uint8, device=frames.device)
            x = ops.resize_u8(x, out, *self._table(H, h), axis=1)
        return x

    def resize_f32(self, frames: torch.Tensor, w: int, h: int) -> torch.Tensor:
        """PIL Image.resize((w, h)) of mode "F" images (float32 [n, H, W]): double accumulation, horizontal then vertical pass."""
        assert frames.dtype == torch.float32 and frames.dim() == 3 and frames.is_contiguous()

        def table(n_in, n_out):
            if (n_in, n_out) not in self._tables_f:
                b, k = pil_bicubic_tables(n_in, n_out, fixed_point=False)
                self._tables_f[(n_in, n_out)] = (torch.from_numpy(b).to(self.device), torch.from_numpy(k).to(self.device))
            return self._tables_f[(n_in, n_out)]

        n, H, W = frames.shape
        x = frames
        if W != w:
            x = ops.resize_f32(x, torch.empty(n, H, w, dtype=torch.float32, device=frames.device), *table(W, w), axis=2)
        if H != h:
            x = ops.resize_f32(x, torch.empty(n, h, x.shape[2], dtype=torch.float32, device=frames.device), *table(H, h), axis=1)
        return x

    def s1_depth(self, depth: torch.Tensor, size: int = 224, scale: float = 10.0, clip: float = 5.0) -> torch.Tensor:
        """depth f32 [n, H, W] (metres / 10) -> f32 [n, size, size]: np.array(Image.fromarray(d).resize((size, size))) * 10.0, values
        above the threshold set to it (internvla_n1_agent.py:313-316)."""
        x = self.resize_f32(depth, size, size) * scale
        return torch.where(x > clip, torch.full_like(x, clip), x)

    def qwen_pixel_values(self, frames: torch.Tensor):
        """raw frames -> (pixel_values bf16 [n * gh * gw, 1176], image_grid_thw int64 [n, 3]) exactly as
        processor(images=[Image.fromarray(f).resize((resize_w, resize_h)) for f in frames]) followed by the policy's bf16 cast."""
        x = self.resize(frames, self.resize_w, self.resize_h)
        hb, wb = smart_resize(self.resize_h, self.resize_w, self.ps * self.merge, self.min_pixels, self.max_pixels)
        x = self.resize(x, wb, hb)
        n, gh, gw = x.shape[0], hb // self.ps, wb // self.ps
        pv = torch.empty(n * gh * gw, 3 * self.tdup * self.ps * self.ps, dtype=torch.bfloat16, device=x.device)
        ops.qwen_patchify_u8(x, pv, self.qwen_lut, self.ps, self.merge, self.tdup)
        return pv, torch.tensor([[1, gh, gw]] * n, dtype=torch.int64)

    def processor_pixel_values(self, frames):
        """the HF image processor alone, for a LIST of uint8 frames [H, W, 3] that may differ in size (history frames already at
        resize_w x resize_h, the look-down frame at camera size): per frame smart_resize -> PIL bicubic -> rescale / normalize /
        patchify, rows concatenated in list order -> (pixel_values bf16 [sum gh*gw, 1176], image_grid_thw int64 [n, 3])."""
        pvs, grids = [], []
        for f in frames:
            H, W = int(f.shape[0]), int(f.shape[1])
            hb, wb = smart_resize(H, W, self.ps * self.merge, self.min_pixels, self.max_pixels)
            x = self.resize(f[None].contiguous(), wb, hb)
            gh, gw = hb // self.ps, wb // self.ps
            pv = torch.empty(gh * gw, 3 * self.tdup * self.ps * self.ps, dtype=torch.bfloat16, device=x.device)
            ops.qwen_patchify_u8(x, pv, self.qwen_lut, self.ps, self.merge, self.tdup)
            pvs.append(pv)
            grids.append([1, gh, gw])
        return torch.cat(pvs, 0), torch.tensor(grids, dtype=torch.int64)

    def s1_frames(self, frames: torch.Tensor, size: int = 224) -> torch.Tensor:
        """raw frames -> bf16 [n, size, size, 3] in 0..1: np.array(Image.fromarray(f).resize((size, size))) / 255.0."""
        x = self.resize(frames, size, size)
        out = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
        return ops.u8_lut(x, out, self.unit_lut)
