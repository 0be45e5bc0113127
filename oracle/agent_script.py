"""TEST INFRASTRUCTURE ONLY - scripted model / processor / tokenizer, seeded observations and scenarios for the a13 trace.

SURVEY.md 8 row a13 (agent rollout step + policy host logic) is integer / text work: history sampling, prompt text, look-down
conversation continuation, pixel / arrow parsing, the S2 / S1 cadence and the retry path. `oracle/make_golden_agent.py` runs the
REFERENCE'S OWN `InternVLAN1Agent.step` (internnav/agent/internvla_n1_agent.py:243-407, S2 thread :133-208) on top of the reference's
own `InternVLAN1Net` (internnav/model/basemodel/internvla_n1/internvla_n1_policy.py:26-215) with the objects of this file standing in
for the checkpoint (model), `AutoProcessor` and `AutoTokenizer`, and commits everything those objects SAW (chat text, image bytes
digests, System-1 input digests, generate kwargs) plus the action of every step as `tests/golden/agent_trace.json`.
`tests/test_agent_trace.py` replays the same scenarios through `internnav_amd.agent.InternVLAN1Agent` (batched, 3 staggered envs) with
the same scripted objects and demands equality of every record.

Nothing here computes anything the product ships; the scripted model answers from a per-scenario script.
"""
from __future__ import annotations

import hashlib
from types import SimpleNamespace
from typing import Dict, List

import numpy as np
import torch

IMAGE_PAD_ID = 0x110001          # ids beyond the unicode range are the "special tokens" of the scripted tokenizer
EOS_ID = 0x110002
IMAGE_PAD = "<|image_pad|>"
TAG = "[scn-%s]"


# ------------------------------------------------------------------------------------------------ tokenizer / processor
class ScriptedTokenizer:
    """characters <-> code points; `<|image_pad|>` is one special id. Deterministic and invertible, which is all the trace needs."""
    padding_side = "right"

    def encode(self, text: str) -> List[int]:
        ids, i = [], 0
        while i < len(text):
            if text.startswith(IMAGE_PAD, i):
                ids.append(IMAGE_PAD_ID)
                i += len(IMAGE_PAD)
            else:
                ids.append(ord(text[i]))
                i += 1
        return ids

    def __call__(self, texts, return_tensors="pt", **_):
        assert len(texts) == 1
        ids = torch.tensor([self.encode(texts[0])], dtype=torch.long)
        return {"input_ids": ids, "attention_mask": torch.ones_like(ids)}

    def decode(self, ids, skip_special_tokens=True):
        out = []
        for i in (int(x) for x in ids):
            if i == IMAGE_PAD_ID:
                if not skip_special_tokens:
                    out.append(IMAGE_PAD)
            elif i == 0 or i >= 0x110000:
                continue
            else:
                out.append(chr(i))
        return "".join(out)


def image_digest(img) -> str:
    """sha1 over size + raw bytes of a PIL image, an HxWx3 uint8 array or a uint8 tensor (the device pre-processor's frames)."""
    if isinstance(img, torch.Tensor):
        arr = img.detach().cpu().numpy()
    elif isinstance(img, np.ndarray):
        arr = img
    else:
        assert img.mode == "RGB", img.mode
        arr = np.asarray(img)
    arr = np.ascontiguousarray(arr, dtype=np.uint8)
    return hashlib.sha1(repr(arr.shape).encode() + arr.tobytes()).hexdigest()


def tensor_digest(x) -> str:
    """sha1 of the float32 values (shape-free: the reference feeds [1, 2, 224, 224, C] float64, a batched caller feeds slices)."""
    a = np.ascontiguousarray(torch.as_tensor(np.asarray(x) if not isinstance(x, torch.Tensor) else x).detach().cpu().to(torch.float32).numpy())
    return hashlib.sha1(a.tobytes()).hexdigest()


def _digest_rows(hexd: str) -> torch.Tensor:
    """a sha1 hex digest as 4 'patch rows' of 8 float32 values (what the scripted processor puts into pixel_values)."""
    b = bytes.fromhex(hexd) + bytes(12)
    return torch.tensor(list(b), dtype=torch.float32).view(4, 8)


def _rows_digest(rows: torch.Tensor) -> str:
    return bytes(int(v) for v in rows.reshape(-1)[:20].tolist()).hex()


class ScriptedProcessor:
    """`AutoProcessor` stand-in with the Qwen2.5-VL chat template written out (system prompt, <|im_start|>role ... <|im_end|>, one
    <|vision_start|><|image_pad|><|vision_end|> per image). `__call__` keeps ONE pad token per image, grid (1, 2, 2), and encodes each
    image's digest into its 4 pixel_values rows so the model side can name the images it was handed."""
    image_token = IMAGE_PAD

    def __init__(self):
        self.tokenizer = ScriptedTokenizer()

    def apply_chat_template(self, conversation, tokenize=False, add_generation_prompt=True):
        assert tokenize is False
        out = []
        for k, m in enumerate(conversation):
            if k == 0 and m["role"] != "system":
                out.append("<|im_start|>system\nYou are a helpful assistant.<|im_end|>\n")
            out.append(f"<|im_start|>{m['role']}\n")
            if isinstance(m["content"], str):
                out.append(m["content"])
            else:
                for c in m["content"]:
                    if c["type"] == "image":
                        out.append("<|vision_start|>" + IMAGE_PAD + "<|vision_end|>")
                    else:
                        out.append(c["text"])
            out.append("<|im_end|>\n")
        if add_generation_prompt:
            out.append("<|im_start|>assistant\n")
        return "".join(out)

    def __call__(self, text, images, return_tensors="pt"):
        from transformers import BatchFeature

        assert len(text) == 1 and text[0].count(IMAGE_PAD) == len(images)
        enc = self.tokenizer(text)
        pv = torch.cat([_digest_rows(image_digest(im)) for im in images], 0)
        grid = torch.tensor([[1, 2, 2]] * len(images), dtype=torch.long)
        return BatchFeature({"input_ids": enc["input_ids"], "attention_mask": enc["attention_mask"], "pixel_values": pv, "image_grid_thw": grid})


# ------------------------------------------------------------------------------------------------ trajectories
def make_traj(kind: str) -> torch.Tensor:
    """[32, 32, 3] x4-scaled increments (what generate_traj returns for one env) that the reference's traj_to_actions turns into a
    known action list: 'forward' -> 8 forward steps, 'short' -> 2, 'none' -> [], 'left' / 'right' -> turns first."""
    t = torch.zeros(32, 32, 3)
    if kind == "forward":
        t[:, :, 0] = 4 * 2.0 / 32
    elif kind == "short":
        t[:, :, 0] = 4 * 0.5 / 32
    elif kind == "none":
        t[:, :, 0] = 4 * 0.1 / 32
    elif kind == "left":
        t[:, :, 0] = 4 * 1.0 / 32
        t[:, :, 1] = 4 * 1.0 / 32
    elif kind == "right":
        t[:, :, 0] = 4 * 1.5 / 32
        t[:, :, 1] = -4 * 0.9 / 32
    else:
        raise KeyError(kind)
    # the 32 samples differ a little (the mean is what counts, vln_utils.py:131)
    t += 0.01 * torch.linspace(-1, 1, 32).view(32, 1, 1)
    return t


# ------------------------------------------------------------------------------------------------ scripted model
class ScriptedModel(torch.nn.Module):
    """stands in for InternVLAN1ForCausalLM behind BOTH agents. Answers / trajectories come from `scripts[scenario]`, the scenario is
    read from the tag in the prompt text (S2) or from the latent it handed out (S1). Every call is logged per scenario in `tape`."""

    def __init__(self, scripts: Dict[str, dict], system1: str = "nextdit_async"):
        super().__init__()
        self.anchor = torch.nn.Parameter(torch.zeros(1), requires_grad=False)      # gives PreTrainedModel.device something to find
        self.config = SimpleNamespace(system1=system1, n_query=4)
        self.tok = ScriptedTokenizer()
        self.load(scripts)

    def load(self, scripts):
        self.scripts = {k: dict(answers=list(v["answers"]), trajs=list(v["trajs"])) for k, v in scripts.items()}
        self.names = sorted(self.scripts)
        self.tape: Dict[str, list] = {k: [] for k in self.names}
        self.lat_count = {k: 0 for k in self.names}
        self.exhausted = False

    def eval(self):
        return self

    @property
    def device(self):
        return self.anchor.device

    def _scenario(self, text: str) -> str:
        hits = [n for n in self.names if TAG % n in text]
        assert len(hits) == 1, f"prompt names {hits} scenarios: {text[:200]!r}"
        return hits[0]

    def _rows(self, input_ids, attention_mask, pixel_values, image_grid_thw):
        """per row: (ids list without padding, image digests)."""
        B = input_ids.shape[0]
        out, p = [], 0
        for b in range(B):
            L = int(attention_mask[b].sum()) if attention_mask is not None else input_ids.shape[1]
            ids = [int(v) for v in input_ids[b, :L]]
            n_img = sum(1 for v in ids if v == IMAGE_PAD_ID)
            digs = [_rows_digest(pixel_values[p + 4 * k: p + 4 * k + 4]) for k in range(n_img)]
            p += 4 * n_img
            out.append((ids, digs))
        assert pixel_values is None or p == pixel_values.shape[0], "pixel_values rows and image placeholders disagree"
        assert image_grid_thw is None or image_grid_thw.shape[0] * 4 == p
        return out

    def generate(self, input_ids=None, pixel_values=None, image_grid_thw=None, attention_mask=None, **kw):
        rows = self._rows(input_ids.cpu(), attention_mask.cpu() if attention_mask is not None else None, pixel_values.cpu(), image_grid_thw)
        kwargs = {k: (v if not isinstance(v, torch.Tensor) else "tensor") for k, v in sorted(kw.items())}
        outs = []
        for ids, digs in rows:
            text = self.tok.decode(ids, skip_special_tokens=False)
            scn = self._scenario(text)
            if not self.scripts[scn]["answers"]:
                self.exhausted = True      # both agents swallow exceptions out of generate(): the harness checks this flag instead
                raise RuntimeError(f"script of scenario {scn} ran out of answers")
            ans = self.scripts[scn]["answers"].pop(0)
            self.tape[scn].append({"kind": "s2", "text": text, "images": digs, "kwargs": kwargs, "answer": repr(ans) if isinstance(ans, Exception) else ans})
            if isinstance(ans, Exception):
                raise ans
            outs.append(ids + [ord(c) for c in ans] + [EOS_ID])
        W = max(len(o) for o in outs)
        # like HF generate on a right-padded batch of our agent: every row = its own prompt + answer, right-filled with EOS
        seqs = torch.tensor([o + [EOS_ID] * (W - len(o)) for o in outs], dtype=torch.long)
        return SimpleNamespace(sequences=seqs)

    def generate_latents(self, output_ids, pixel_values, image_grid_thw, rows=None, **_):
        """`rows` (internnav_amd extension for batched callers): only these rows of the batch need latents; returns [len(rows), 4, 8]."""
        B = output_ids.shape[0]
        keep = list(range(B)) if rows is None else [int(r) for r in rows]
        lat = torch.zeros(B, 4, 8)
        p = 0
        for b in range(B):
            ids = [int(v) for v in output_ids[b]]
            n_img = sum(1 for v in ids if v == IMAGE_PAD_ID)
            if b in keep:
                text = self.tok.decode(ids, skip_special_tokens=False)
                scn = self._scenario(text)
                digs = [_rows_digest(pixel_values.cpu()[p + 4 * k: p + 4 * k + 4]) for k in range(n_img)]
                self.lat_count[scn] += 1
                self.tape[scn].append({"kind": "latents", "text": text, "images": digs})
                lat[b, 0, 0], lat[b, 0, 1] = self.names.index(scn), self.lat_count[scn]
            p += 4 * n_img
        return lat[keep]

    def generate_traj(self, traj_latents=None, images_dp=None, depths_dp=None, **_):
        B = traj_latents.shape[0]
        out = []
        for b in range(B):
            scn = self.names[int(traj_latents[b, 0, 0])]
            if not self.scripts[scn]["trajs"]:
                self.exhausted = True
                raise RuntimeError(f"script of scenario {scn} ran out of trajectories")
            kind = self.scripts[scn]["trajs"].pop(0)
            img = images_dp[b] if B > 1 or getattr(images_dp, "ndim", 0) == 5 else images_dp
            dep = depths_dp[b] if B > 1 or getattr(depths_dp, "ndim", 0) == 5 else depths_dp
            self.tape[scn].append({"kind": "s1", "latent": int(traj_latents[b, 0, 1]), "rgb": tensor_digest(img), "depth": tensor_digest(dep), "traj": kind})
            out.append(make_traj(kind))
        return torch.cat(out, 0)


# ------------------------------------------------------------------------------------------------ observations / scenarios
def make_obs(scenario: str, step: int, instruction: str, tag: str = None):
    """seeded 640x480 RGB-D observation of (scenario, step): smooth-ish RGB so the bicubic resizes see real gradients, depth in 0..1
    with part of it beyond the 5 m System-1 clip (x10 > 5)."""
    seed = int(hashlib.sha1(f"{scenario}:{step}".encode()).hexdigest()[:8], 16)
    rng = np.random.default_rng(seed)
    base = rng.integers(0, 256, size=(15, 20, 3), dtype=np.uint8).repeat(32, 0).repeat(32, 1)
    noise = rng.integers(-20, 21, size=(480, 640, 3))
    rgb = np.clip(base.astype(np.int64) + noise, 0, 255).astype(np.uint8)
    depth = (rng.random((480, 640, 1), dtype=np.float32) * 0.9).astype(np.float32)
    return {"rgb": rgb, "depth": depth, "instruction": f"{instruction} {TAG % (tag or scenario)}"}


# answers: consumed one per generate() row of the scenario; trajs: one per generate_traj() row. `raises_at`: the reference's step()
# raises there (recorded as such), `extra_answers`: answers only OUR agent consumes at that point (documented divergence).
SCENARIOS = {
    # pixel goals only: S2 -> latent -> S1 every 4 steps, S2 again when dual_forward_step reaches sys2_max_forward_step
    "pixel": dict(mode="partial_async", steps=44, instruction="walk past the sofa and stop at the door",
                  answers=["123 456", "200 310", "5 7", "640 480", "12 345", "99 100", "321 123", "44 55"],
                  trajs=["forward", "left", "forward", "right", "forward", "forward", "left", "right", "forward", "forward", "forward", "forward",
                         "forward", "forward", "forward", "forward"]),
    # discrete arrows, then STOP
    "arrows": dict(mode="partial_async", steps=40, instruction="turn left at the kitchen",
                   answers=["↑↑←", "→↑", "↑", "←←←↑", "↑↑↑↑↑↑", "→", "↑↑", "←↑→↑", "↑↑↑", "↑", "→→", "↑↑↑↑", "STOP", "↑", "STOP", "↑↑", "↑", "↑", "↑", "↑"],
                   trajs=[]),
    # look-down turn: the un-resized frame joins the previous images, the conversation continues with the assistant's answer
    "lookdown": dict(mode="partial_async", steps=42, instruction="go down the stairs",
                     answers=["↑↓", "150 260", "↓", "↑↑", "↓", "77 88", "301 17", "↑↓↑", "9 9", "↑", "↓", "↑", "↑", "↑", "↑", "↑", "↑", "↑", "↑", "↑"],
                     trajs=["forward", "forward", "short", "forward", "forward", "forward", "forward", "forward", "forward", "forward"]),
    # short / empty System-1 plans: "already reached the pixel goal" accounting and the -1 action
    "short": dict(mode="partial_async", steps=40, instruction="enter the bedroom",
                  answers=["10 20", "30 40", "50 60", "70 80", "90 100", "110 120", "130 140", "150 160", "170 180", "190 200", "210 220", "230 240",
                           "250 260", "270 280", "290 300", "310 320"],
                  trajs=["short", "none", "forward", "short", "short", "none", "none", "forward", "forward", "short", "forward", "none", "forward",
                         "forward", "short", "forward", "forward", "forward", "forward", "forward"]),
    # a one-number pixel goal (IndexError in s2_step, internvla_n1_policy.py:187) on a LATER call: reset + retry without look-down;
    # first the retry succeeds, later both attempts fail -> STOP ([0]) and a fresh episode history afterwards
    "retry": dict(mode="partial_async", steps=40, instruction="find the red chair",
                  answers=["↑", "42", "↑→", "↑", "42", "7", "↑↑", "1 2", "↑", "5", "↑", "↑", "↑", "↑", "↑", "↑", "↑", "↑", "↑", "↑", "↑", "↑", "↑", "↑", "↑", "↑",
                           "↑", "↑", "↑", "↑", "↑", "↑", "↑", "↑", "↑", "↑", "↑", "↑"],
                  trajs=["forward", "forward", "forward", "forward"]),
    # the look-down turn itself fails (one number) -> policy reset, retry as a normal first turn
    "lookdown_retry": dict(mode="partial_async", steps=40, instruction="cross the hallway",
                           answers=["↓", "5", "100 200", "↑", "↓", "33 44", "↑↑", "↓", "8", "↑", "↑", "↑", "↑", "↑", "↑", "↑", "↑", "↑", "↑", "↑", "↑", "↑",
                                    "↑", "↑", "↑", "↑", "↑", "↑", "↑", "↑", "↑", "↑"],
                           trajs=["forward", "forward", "forward", "forward", "forward", "forward", "forward", "forward"]),
    # an answer with neither digits nor arrows: the reference's s2_step returns output_action=[] (policy :196-198) and its main thread
    # dies with IndexError at internvla_n1_agent.py:282. Recorded as `raises`; internnav_amd treats it as an S2 failure (retry -> STOP).
    "garbage": dict(mode="partial_async", steps=41, instruction="wait by the window",
                    answers=["↑↑↑", "12 21", "↑", "60 61", "↑↑", "↑", "71 17", "↑", "↑", "↑", "↑", "↑", "↑", "↑", "↑", "hello there"],
                    trajs=["forward", "forward", "forward", "forward", "forward", "forward", "forward", "forward"],
                    extra_answers=["still nothing"]),
    # infer_mode 'sync': S2 whenever the action queue is empty, S1 on the raw frame (internvla_n1_agent.py:334), latent dropped after one use
    "sync": dict(mode="sync", steps=40, instruction="follow the corridor",
                 answers=["11 22", "↑↑", "33 44", "55 66", "↑", "77 88", "99 11", "22 33", "44 55", "66 77", "88 99", "10 20", "30 40", "↑", "↑", "↑", "↑", "↑"],
                 trajs=["forward", "short", "left", "forward", "none", "forward", "forward", "forward", "forward", "forward", "forward", "forward"]),
}
