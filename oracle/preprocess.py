"""ORACLE (test infrastructure - never imported by the product path): numpy restatement of the frame pre-processing the reference does
on the host, byte / integer arithmetic restated from the published algorithms of its third-party dependencies:

  * PIL `Image.resize((w, h))` for 8-bit RGB (default BICUBIC): Pillow libImaging/Resample.c - precompute_coeffs, bicubic_filter
    (a = -0.5), normalize_coeffs_8bpc (PRECISION_BITS = 22), ImagingResampleHorizontal_8bpc then ImagingResampleVertical_8bpc.
    Called by the reference at internvla_n1_policy.py:105-116 and internvla_n1_agent.py:309-320.
  * HF Qwen2VLImageProcessor (requirements pin transformers==4.51.0): smart_resize, resize (PIL bicubic), rescale (float64 multiply
    -> fp32), normalize ((x - mean) / std in fp32), patch layout - image_processing_qwen2_vl.py / image_transforms.py; called at
    internvla_n1_policy.py:163-165.

Pinned: tests/test_preprocess.py compares it with PIL itself (Pillow ships in the image) byte for byte on random frames, and with
tests/golden/preprocess.pt = outputs of PIL + the installed transformers Qwen2VLImageProcessorPil (oracle/make_golden.py)."""
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2
CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def bicubic_filter(x, a=-0.5):
    """Resample.c bicubic_filter."""
    x = abs(x)
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def precompute_coeffs(in_size, out_size):
    """Resample.c precompute_coeffs + normalize_coeffs_8bpc for the full box [0, in_size)."""
    scale = filterscale = in_size / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = 2.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds, coefs = [], []
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        xmin = 0 if xmin < 0 else xmin
        xmax = int(center + support + 0.5)
        xmax = in_size if xmax > in_size else xmax
        xmax -= xmin
        k = [bicubic_filter((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for w in k:
            ww += w
        if ww != 0.0:
            k = [w / ww for w in k]
        k = [int(-0.5 + w * (1 << PRECISION_BITS)) if w < 0 else int(0.5 + w * (1 << PRECISION_BITS)) for w in k]
        bounds.append((xmin, xmax))
        coefs.append(k + [0] * (ksize - xmax))
    return bounds, coefs


def resample_axis(img, out_size, axis):
    """one 8-bit pass (ImagingResampleHorizontal_8bpc / Vertical_8bpc) along `axis` of a uint8 array."""
    x = np.moveaxis(img, axis, 0).astype(np.int64)
    bounds, coefs = precompute_coeffs(x.shape[0], out_size)
    out = np.empty((out_size,) + x.shape[1:], dtype=np.uint8)
    for xx, ((xmin, xmax), k) in enumerate(zip(bounds, coefs)):
        ss = np.full(x.shape[1:], 1 << (PRECISION_BITS - 1), dtype=np.int64)
        for j in range(xmax):
            ss += x[xmin + j] * k[j]
        out[xx] = np.clip(ss >> PRECISION_BITS, 0, 255)
    return np.moveaxis(out, 0, axis)


def pil_resize(img, w, h):
    """Image.fromarray(img).resize((w, h)) for uint8 [H, W, C]: horizontal pass first, then vertical (ImagingResample)."""
    if img.shape[1] != w:
        img = resample_axis(img, w, 1)
    if img.shape[0] != h:
        img = resample_axis(img, h, 0)
    return img


def resample_axis_f32(img, out_size, axis):
    """one pass of PIL's 32-bit float resample (mode "F", ImagingResampleHorizontal_32bpc / Vertical_32bpc): double weights (not
    quantised), double accumulation in tap order, rounded to float once."""
    x = np.moveaxis(img, axis, 0).astype(np.float64)
    scale = filterscale = x.shape[0] / out_size
    filterscale = max(filterscale, 1.0)
    support = 2.0 * filterscale
    out = np.empty((out_size,) + x.shape[1:], dtype=np.float32)
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), x.shape[0]) - xmin
        k = [bicubic_filter((j + xmin - center + 0.5) * (1.0 / filterscale)) for j in range(xmax)]   # Resample.c multiplies by ss = 1 / filterscale
        ww = 0.0
        for w in k:
            ww += w
        if ww != 0.0:
            k = [w / ww for w in k]
        ss = np.zeros(x.shape[1:], dtype=np.float64)
        for j in range(xmax):
            ss = ss + x[xmin + j] * k[j]
        out[xx] = ss.astype(np.float32)
    return np.moveaxis(out, 0, axis)


def pil_resize_f32(img, w, h):
    """Image.fromarray(img).resize((w, h)) for float32 [H, W] (mode "F")."""
    if img.shape[1] != w:
        img = resample_axis_f32(img, w, 1)
    if img.shape[0] != h:
        img = resample_axis_f32(img, h, 0)
    return img


def s1_depth(depth, size=224, threshold=5.0):
    """np.array(Image.fromarray(d).resize((size, size))) * 10.0 with values above the threshold set to it (internvla_n1_agent.py:313-316)."""
    out = []
    for d in depth:
        x = pil_resize_f32(np.asarray(d, dtype=np.float32), size, size) * 10.0
        x[x > threshold] = threshold
        out.append(x)
    return np.stack(out)


def smart_resize(height, width, factor=28, min_pixels=56 * 56, max_pixels=14 * 14 * 4 * 1280):
    h_bar, w_bar = round(height / factor) * factor, round(width / factor) * factor
    if h_bar * w_bar > max_pixels:
        beta = math.sqrt((height * width) / max_pixels)
        h_bar = max(factor, math.floor(height / beta / factor) * factor)
        w_bar = max(factor, math.floor(width / beta / factor) * factor)
    elif h_bar * w_bar < min_pixels:
        beta = math.sqrt(min_pixels / (height * width))
        h_bar, w_bar = math.ceil(height * beta / factor) * factor, math.ceil(width * beta / factor) * factor
    return h_bar, w_bar


def qwen_pixel_values(frames, resize_w, resize_h, min_pixels=56 * 56, max_pixels=14 * 14 * 4 * 1280, ps=14, merge=2, tdup=2):
    """uint8 frames [n, H, W, 3] -> (pixel_values f32 [n*gh*gw, 3*tdup*ps*ps], grid_thw [n, 3]) as the policy + HF processor produce them."""
    mean, std = np.asarray(CLIP_MEAN, dtype=np.float32), np.asarray(CLIP_STD, dtype=np.float32)
    rows, grids = [], []
    for f in frames:
        x = pil_resize(np.asarray(f), resize_w, resize_h)                       # policy: Image.resize((resize_w, resize_h))
        hb, wb = smart_resize(resize_h, resize_w, ps * merge, min_pixels, max_pixels)
        x = pil_resize(x, wb, hb)                                               # processor: resize(resample=BICUBIC)
        x = (x.astype(np.float64) * (1 / 255)).astype(np.float32)               # rescale (image_transforms.rescale)
        x = (x - mean) / std                                                    # normalize
        x = np.repeat(x.transpose(2, 0, 1)[None], tdup, axis=0)                 # channels first, last frame repeated to T
        gh, gw = hb // ps, wb // ps
        p = x.reshape(1, tdup, 3, gh // merge, merge, ps, gw // merge, merge, ps).transpose(0, 3, 6, 4, 7, 2, 1, 5, 8)
        rows.append(p.reshape(gh * gw, 3 * tdup * ps * ps))
        grids.append((1, gh, gw))
    return np.concatenate(rows, 0), np.asarray(grids, dtype=np.int64)


def s1_frames(frames, size=224):
    """np.array(Image.fromarray(f).resize((size, size))) / 255.0 (float64), internvla_n1_agent.py:309-317."""
    return np.stack([pil_resize(np.asarray(f), size, size) / 255.0 for f in frames])
