"""Frame pre-processing: PIL 8-bit bicubic resize + HF Qwen2-VL rescale / normalize / patchify + the System-1 0..1
frames, byte / integer work -> BIT-EXACT bar. Oracle = oracle/preprocess.py (numpy), pinned against tests/golden/preprocess.pt
(outputs of PIL and the installed transformers image processor, oracle/make_golden.py) and against PIL itself when it is importable;
the device kernels (internnav_amd.preprocess.FramePreprocessor) are compared with both."""
from pathlib import Path

import numpy as np
import pytest
import torch

from oracle import preprocess as o_pp

GOLD = Path(__file__).resolve().parent / "golden" / "preprocess.pt"
DEV = "cuda:0"


@pytest.fixture(scope="module")
def gold():
    return torch.load(GOLD, weights_only=True)


def test_oracle_reproduces_pil_and_hf_processor_fixture(gold):
    frames = gold["frames"].numpy()
    rw, rh = gold["resize_w"], gold["resize_h"]
    for f, r in zip(frames, gold["resized"].numpy()):
        assert np.array_equal(o_pp.pil_resize(f, rw, rh), r)
    for e in gold["extra"]:
        assert np.array_equal(o_pp.pil_resize(e["image"].numpy(), e["w"], e["h"]), e["out"].numpy())
    pv, grid = o_pp.qwen_pixel_values(frames, rw, rh)
    assert np.array_equal(pv, gold["pixel_values"].numpy()) and np.array_equal(grid, gold["image_grid_thw"].numpy())
    assert np.array_equal(o_pp.s1_frames(frames, gold["s1_size"]), gold["s1"].numpy())
    assert np.array_equal(o_pp.s1_depth(gold["depth"].numpy(), gold["s1_size"]), gold["s1_depth"].numpy())      # PIL mode "F" path


def test_oracle_matches_live_pil_at_camera_size():
    Image = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(7)
    f = rng.integers(0, 256, (480, 640, 3), dtype=np.uint8)
    a = np.array(Image.fromarray(f).resize((384, 384)))
    assert np.array_equal(o_pp.pil_resize(f, 384, 384), a)
    assert o_pp.smart_resize(384, 384) == (392, 392) and o_pp.smart_resize(480, 640) == (476, 644)
    assert np.array_equal(o_pp.pil_resize(a, 392, 392), np.array(Image.fromarray(a).resize((392, 392), resample=Image.BICUBIC)))
    assert np.array_equal(o_pp.pil_resize(f, 224, 224), np.array(Image.fromarray(f).resize((224, 224))))
    d = (rng.random((480, 640), dtype=np.float32) * 0.7).astype(np.float32)
    assert np.array_equal(o_pp.pil_resize_f32(d, 224, 224), np.array(Image.fromarray(d).resize((224, 224))))


def test_product_tables_are_the_oracle_tables():
    """the host side of the product (coefficient tables, smart_resize, normalisation table) equals the oracle's restatement."""
    from internnav_amd import preprocess as pp

    for n_in, n_out in [(640, 384), (480, 384), (384, 392), (640, 224), (80, 56), (17, 40), (33, 33), (5, 64)]:
        b, k = pp.pil_bicubic_tables(n_in, n_out)
        ob, ok = o_pp.precompute_coeffs(n_in, n_out)
        assert np.array_equal(b, np.asarray(ob, dtype=np.int32)) and np.array_equal(k, np.asarray(ok, dtype=np.int32))
    bf, kf = pp.pil_bicubic_tables(640, 224, fixed_point=False)
    assert kf.dtype == np.float64 and np.array_equal(bf, pp.pil_bicubic_tables(640, 224)[0]) and np.allclose(kf.sum(1), 1.0, atol=1e-12)
    for hw in [(384, 384), (480, 640), (30, 30), (3000, 4000), (56, 57)]:
        assert pp.smart_resize(*hw) == o_pp.smart_resize(*hw)
    v = (np.arange(256, dtype=np.float64) * (1 / 255)).astype(np.float32)
    lut = pp.qwen_normalize_table()
    for c in range(3):
        assert np.array_equal(lut[c], (v - np.float32(o_pp.CLIP_MEAN[c])) / np.float32(o_pp.CLIP_STD[c]))


@pytest.mark.gpu
def test_device_resize_is_bit_exact(built_lib, gold):
    from internnav_amd.preprocess import FramePreprocessor

    pre = FramePreprocessor(DEV, resize_w=gold["resize_w"], resize_h=gold["resize_h"])
    out = pre.resize(gold["frames"].to(DEV), gold["resize_w"], gold["resize_h"])
    assert torch.equal(out.cpu(), gold["resized"])
    for e in gold["extra"]:
        o = pre.resize(e["image"][None].contiguous().to(DEV), e["w"], e["h"])
        assert torch.equal(o[0].cpu(), e["out"])
    # camera-sized frames, both passes down- and up-scaling, against the oracle (and PIL when present)
    rng = np.random.default_rng(3)
    f = rng.integers(0, 256, (2, 480, 640, 3), dtype=np.uint8)
    for (w, h) in [(384, 384), (224, 224), (644, 476), (700, 500)]:
        dev = pre.resize(torch.from_numpy(f).to(DEV), w, h).cpu().numpy()
        ref = np.stack([o_pp.pil_resize(x, w, h) for x in f])
        assert np.array_equal(dev, ref), (w, h)
    try:
        from PIL import Image

        assert np.array_equal(pre.resize(torch.from_numpy(f).to(DEV), 384, 384).cpu().numpy()[0], np.array(Image.fromarray(f[0]).resize((384, 384))))
    except ImportError:
        pass


@pytest.mark.gpu
def test_device_qwen_pixel_values_and_s1_frames_are_bit_exact(built_lib, gold):
    from internnav_amd.preprocess import FramePreprocessor

    pre = FramePreprocessor(DEV, resize_w=gold["resize_w"], resize_h=gold["resize_h"])
    pv, grid = pre.qwen_pixel_values(gold["frames"].to(DEV))
    assert torch.equal(grid, gold["image_grid_thw"].to(torch.int64))
    assert torch.equal(pv.cpu(), gold["pixel_values"].to(torch.bfloat16))          # the policy casts the processor output to bf16
    s1 = pre.s1_frames(gold["frames"].to(DEV), gold["s1_size"])
    assert torch.equal(s1.cpu(), gold["s1"].to(torch.bfloat16))
    dd = pre.s1_depth(gold["depth"].to(DEV), gold["s1_size"])
    assert torch.equal(dd.cpu(), gold["s1_depth"])                                   # float resample: bit-exact fp32
    rng0 = np.random.default_rng(9)
    big = (rng0.random((2, 480, 640), dtype=np.float32) * 0.7).astype(np.float32)
    assert np.array_equal(pre.s1_depth(torch.from_numpy(big).to(DEV)).cpu().numpy(), o_pp.s1_depth(big))
    # full geometry of the deployment: 480x640 camera frame -> 384x384 -> 392x392 -> 784 patches x 1176
    rng = np.random.default_rng(5)
    f = rng.integers(0, 256, (1, 480, 640, 3), dtype=np.uint8)
    full = FramePreprocessor(DEV)
    pv2, grid2 = full.qwen_pixel_values(torch.from_numpy(f).to(DEV))
    ref, gref = o_pp.qwen_pixel_values(f, 384, 384)
    assert grid2.tolist() == gref.tolist() == [[1, 28, 28]]
    assert torch.equal(pv2.cpu(), torch.from_numpy(ref).to(torch.bfloat16))


class _OraclePre:
    """CPU stand-in with the FramePreprocessor interface (numpy oracle instead of the device kernels): exercises the policy wiring."""
    device, merge = torch.device("cpu"), 2

    def resize(self, frames, w, h):
        return torch.from_numpy(np.stack([o_pp.pil_resize(f.numpy(), w, h) for f in frames]))

    def processor_pixel_values(self, frames):
        pvs, grids = [], []
        for f in frames:
            H, W = f.shape[:2]
            pv, g = o_pp.qwen_pixel_values([f.numpy()], W, H)      # resize_w/h = the frame's own size: only the processor's resize acts
            pvs.append(torch.from_numpy(pv).to(torch.bfloat16))
            grids.append(g[0].tolist())
        return torch.cat(pvs, 0), torch.tensor(grids, dtype=torch.int64)


class _Tok:
    IMG = "<|image_pad|>"

    def __call__(self, texts, return_tensors="pt"):
        ids = []
        for part in texts[0].replace(self.IMG, "\\x00").split("\\x00"):
            ids += [ord(c) % 500 for c in part] + [1001]
        return {"input_ids": torch.tensor([ids[:-1]])}

    def decode(self, ids, skip_special_tokens=True):
        return "".join(chr(int(i)) for i in ids)


class _Proc:
    image_token, tokenizer = _Tok.IMG, _Tok()

    def apply_chat_template(self, conv, tokenize=False, add_generation_prompt=True):
        return "".join("<|vision_start|>" + _Tok.IMG + "<|vision_end|>" if c["type"] == "image" else c["text"] for t in conv for c in t["content"])


def test_policy_device_preprocess_wiring_matches_host_path():
    """InternVLAN1Net with a frame pre-processor: history frames are resized once on arrival, the look-down frame goes in at camera
    size, every image placeholder expands to grid.prod() / 4 image tokens (Qwen2VLProcessor.__call__), and pixel_values equal the
    host path (PIL resize + HF processor arithmetic) bit for bit."""
    from types import SimpleNamespace

    from internnav_amd.policy import InternVLAN1Net

    net = InternVLAN1Net(SimpleNamespace(device=torch.device("cpu")), _Proc(), num_history=3, resize_w=56, resize_h=56, frame_preprocessor=_OraclePre())
    rng = np.random.default_rng(11)
    frames = [rng.integers(0, 256, (60, 80, 3), dtype=np.uint8) for _ in range(4)]
    for f in frames[:3]:
        net.step_no_infer(f, None, None)
    inp = net.build_s2_inputs(frames[3], "go to the door")
    n_hist = len(np.unique(np.linspace(0, 2, 3, dtype=np.int32)))
    assert inp["image_grid_thw"].tolist() == [[1, 4, 4]] * (n_hist + 1)
    assert int((inp["input_ids"] == 1001).sum()) == 4 * (n_hist + 1)                      # 16 patches / merge^2 per image
    ref_pv, _ = o_pp.qwen_pixel_values(frames, 56, 56)
    assert torch.equal(inp["pixel_values"], torch.from_numpy(ref_pv).to(torch.bfloat16))
    # look-down frame: appended at camera size (60 x 80 -> smart_resize 56 x 84 = 4 x 6 patches)
    net.llm_output = "↓"
    look = rng.integers(0, 256, (60, 80, 3), dtype=np.uint8)
    inp2 = net.build_s2_inputs(look, "go to the door", look_down=True)
    assert inp2["image_grid_thw"].tolist()[-1] == [1, 4, 6] and int((inp2["input_ids"] == 1001).sum()) == 4 * (n_hist + 1) + 6
    ref_look, _ = o_pp.qwen_pixel_values([look], 80, 60)
    assert torch.equal(inp2["pixel_values"][-24:], torch.from_numpy(ref_look).to(torch.bfloat16))


@pytest.mark.gpu
def test_device_processor_pixel_values_mixed_sizes_in_the_policy(built_lib):
    """the wired path on the device: history frames resized on arrival + camera-sized look-down frame through InternVLAN1Net."""
    from types import SimpleNamespace

    from internnav_amd.policy import InternVLAN1Net
    from internnav_amd.preprocess import FramePreprocessor

    pre = FramePreprocessor(DEV, resize_w=56, resize_h=56)
    net = InternVLAN1Net(SimpleNamespace(device=torch.device(DEV)), _Proc(), num_history=3, resize_w=56, resize_h=56, frame_preprocessor=pre)
    rng = np.random.default_rng(11)
    frames = [rng.integers(0, 256, (60, 80, 3), dtype=np.uint8) for _ in range(4)]
    for f in frames[:3]:
        net.step_no_infer(f, None, None)
    inp = net.build_s2_inputs(frames[3], "go to the door")
    ref_pv, ref_grid = o_pp.qwen_pixel_values(frames, 56, 56)
    assert inp["image_grid_thw"].tolist() == ref_grid.tolist()
    assert torch.equal(inp["pixel_values"].cpu(), torch.from_numpy(ref_pv).to(torch.bfloat16))
    net.llm_output = "↓"
    look = rng.integers(0, 256, (60, 80, 3), dtype=np.uint8)
    inp2 = net.build_s2_inputs(look, "go to the door", look_down=True)
    ref_look, _ = o_pp.qwen_pixel_values([look], 80, 60)
    assert inp2["image_grid_thw"].tolist()[-1] == [1, 4, 6]
    assert torch.equal(inp2["pixel_values"][-24:].cpu(), torch.from_numpy(ref_look).to(torch.bfloat16))


def test_agent_s1_device_preprocess_equals_host_path():
    """InternVLAN1Agent._prep_s1_device (one batch for all jobs) == the per-frame host path _prep_s1 (PIL + numpy), bit for bit."""
    pytest.importorskip("PIL.Image")
    from types import SimpleNamespace

    from internnav_amd.agent import InternVLAN1Agent

    class Pre(_OraclePre):
        unit_lut = torch.from_numpy((np.arange(256, dtype=np.float64) / 255.0).astype(np.float32))

        def s1_depth(self, depth, size=224, scale=10.0, clip=5.0):
            return torch.from_numpy(o_pp.s1_depth(depth.numpy(), size, clip))

    ag = InternVLAN1Agent({"model_settings": {"infer_mode": "partial_async"}}, policy_factory=lambda: None, frame_preprocessor=Pre())
    rng = np.random.default_rng(21)
    jobs = []
    for _ in range(2):
        mk = lambda: (rng.integers(0, 256, (48, 64, 3), dtype=np.uint8), (rng.random((48, 64, 1), dtype=np.float32) * 0.8).astype(np.float32))
        (r0, d0), (r1, d1) = mk(), mk()
        jobs.append((SimpleNamespace(s2_output=SimpleNamespace(rgb_memory=r0, depth_memory=d0)), {"rgb": r1, "depth": d1}))
    rgb_t, dep_t = ag._prep_s1_device(jobs)
    for k, (e, o) in enumerate(jobs):
        for j, (rgb, dep) in enumerate(((e.s2_output.rgb_memory, e.s2_output.depth_memory), (o["rgb"], o["depth"]))):
            r, d = ag._prep_s1(rgb, dep)
            assert torch.equal(rgb_t[k, j], torch.from_numpy(r).to(torch.float32))
            assert torch.equal(dep_t[k, j, ..., 0], torch.from_numpy(d))


def test_agent_step_with_frame_preprocessor_end_to_end():
    """the batched agent stepping with the device pre-processing interface (oracle-backed stand-in on CPU): System-2 prompt building,
    grouping, discrete actions, pixel goal -> latent -> System-1 with the batched look-down pre-processing; same actions as without."""
    pytest.importorskip("PIL.Image")
    from types import SimpleNamespace

    from internnav_amd.agent import InternVLAN1Agent

    class Pre(_OraclePre):
        unit_lut = torch.from_numpy((np.arange(256, dtype=np.float64) / 255.0).astype(np.float32))

        def s1_depth(self, depth, size=224, scale=10.0, clip=5.0):
            return torch.from_numpy(o_pp.s1_depth(depth.numpy(), size, clip))

    class Model:
        device = torch.device("cpu")

        def __init__(self, answers):
            self.answers, self.s1_inputs, self.pv = list(answers), [], []

        def generate(self, input_ids=None, pixel_values=None, image_grid_thw=None, **kw):
            assert pixel_values.shape[0] == int(image_grid_thw.prod(1).sum())      # (host path: fp32 from the processor, device path: bf16)
            self.pv.append(pixel_values.to(torch.bfloat16).cpu())
            B = input_ids.shape[0]
            ans = [self.answers.pop(0) for _ in range(B)]
            n = max(len(a) for a in ans)
            toks = torch.tensor([[ord(c) for c in a] + [0] * (n - len(a)) for a in ans])
            return SimpleNamespace(sequences=torch.cat([input_ids, toks], 1))

        def generate_latents(self, seqs, pv, grid, rows=None):
            return torch.zeros(seqs.shape[0] if rows is None else len(rows), 4, 8)

        def generate_traj(self, traj_latents=None, images_dp=None, depths_dp=None):
            self.s1_inputs.append((images_dp.clone(), depths_dp.clone()))
            t = torch.zeros(traj_latents.shape[0] * 32, 32, 3)
            t[:, :, 0] = 1.0
            return t

    rng = np.random.default_rng(33)
    obs = [{"rgb": rng.integers(0, 256, (60, 80, 3), dtype=np.uint8), "depth": (rng.random((60, 80, 1), dtype=np.float32) * 0.6).astype(np.float32),
            "instruction": ins} for ins in ("go to the door", "go to the wall")]
    outs, s1_in, pvs = [], [], []
    for pre in (None, Pre()):
        model = Model(["↑←", "12 34"])
        ag = InternVLAN1Agent({"model_settings": {"infer_mode": "partial_async", "resize_w": 56, "resize_h": 56}}, model=model, processor=_Proc(),
                              frame_preprocessor=pre)
        if pre is None:   # host path needs a processor call: give the fake one the HF behaviour through the oracle
            def call(text, images, return_tensors="pt", _tok=_Proc.tokenizer):
                fr = [np.array(im) for im in images]
                pvs, grids = zip(*[o_pp.qwen_pixel_values([f], f.shape[1], f.shape[0]) for f in fr])
                grid = np.concatenate(grids)
                parts = text[0].split(_Tok.IMG)
                exp = parts[0] + "".join(_Tok.IMG * int(g.prod() // 4) + r for g, r in zip(grid, parts[1:]))
                return {"input_ids": _tok([exp])["input_ids"], "pixel_values": torch.from_numpy(np.concatenate(pvs)), "image_grid_thw": torch.from_numpy(grid)}
            ag_proc = type("P", (_Proc,), {"__call__": staticmethod(call)})()
            ag = InternVLAN1Agent({"model_settings": {"infer_mode": "partial_async", "resize_w": 56, "resize_h": 56}}, model=model, processor=ag_proc)
        ag.reset()
        outs.append([o["action"] for o in ag.step(obs)])
        s1_in.append(model.s1_inputs)
        pvs.append(torch.cat(model.pv))
    assert outs[0] == outs[1] == [[1], [1]]
    assert torch.equal(pvs[0], pvs[1])                                                       # System-2 pixel_values identical on both paths
    (ri, di), (rd, dd) = s1_in[0][0], s1_in[1][0]
    assert torch.equal(ri.float(), rd.float()) and torch.equal(di.float(), dd.float())      # System-1 inputs identical on both paths


def test_smart_resize_matches_installed_transformers():
    hf = pytest.importorskip("transformers.models.qwen2_vl.image_processing_pil_qwen2_vl")
    from internnav_amd import preprocess as pp

    rng = np.random.default_rng(0)
    cases = [(480, 640), (384, 384), (30, 30), (3000, 4000), (56, 57), (28, 2000), (224, 224)] + [tuple(int(v) for v in rng.integers(20, 2500, 2)) for _ in range(200)]
    for h, w in cases:
        if max(h, w) / min(h, w) > 200:
            continue
        ref = tuple(hf.smart_resize(h, w, 28, 56 * 56, 14 * 14 * 4 * 1280))
        assert pp.smart_resize(h, w) == ref == o_pp.smart_resize(h, w), (h, w)
