"""TEST INFRASTRUCTURE ONLY - loads classes of the real reference (InternRobotics/InternNav at /root/reference) on CPU.

Only `oracle/make_golden.py` (run in the build container, where /root/reference is mounted) imports this file. It is
how the CPU restatement in `oracle/*.py` is pinned against the reference itself: the reference modules are
instantiated with the seeded state-dicts of `oracle/weights.py`, executed on seeded inputs, and their outputs are
committed as fixtures under `tests/golden/`. Nothing on the product path, and nothing that runs on the GPU box,
imports this file (the reference tree does not exist there).

The reference cannot be imported as-is in this image: package `__init__`s pull in gym / torchvision /
cv2 / diffusers / a transformers-4.51-only symbol. We therefore
  * pre-seed `sys.modules` with bare namespace packages for `internnav`, `internnav.model`, ... so no `__init__` runs,
  * provide empty stubs for `cv2` and `torchvision.transforms.Compose` (used only by DepthAnythingV2.infer_image),
  * put the vendored `diffusion_policy` (internnav/model/encoder/diffusion_policy) on sys.path,
  * provide a minimal fake `diffusers` whose scheduler classes are the restatements in `oracle/schedulers.py` and whose
    Lumina building blocks are the restatements in `oracle/diffusers_blocks.py` (diffusers==0.33.1 is a third-party,
    un-vendored dependency: requirements/internvla_n1.txt:3) - those pieces stay "parity unpinned", see DESIGN.md.
"""
from __future__ import annotations

import importlib
import sys
import types
from pathlib import Path

REF = Path("/root/reference")


def available() -> bool:
    return (REF / "internnav" / "model").is_dir()


def _ns(name: str, path: Path | None = None) -> types.ModuleType:
    m = sys.modules.get(name)
    if m is None:
        m = types.ModuleType(name)
        m.__path__ = [str(path)] if path is not None else []
        sys.modules[name] = m
    return m


_done = False


def setup() -> None:
    global _done
    if _done:
        return
    if not available():
        raise RuntimeError("/root/reference is not mounted: golden fixtures can only be regenerated in the build container")
    import transformers  # noqa: F401  (must be imported BEFORE the torchvision stub exists: its availability probe rejects spec-less modules)
    from transformers import PretrainedConfig, PreTrainedModel  # noqa: F401

    r = REF / "internnav"
    _ns("internnav", r)
    _ns("internnav.model", r / "model")
    _ns("internnav.model.encoder", r / "model" / "encoder")
    _ns("internnav.model.basemodel", r / "model" / "basemodel")
    _ns("internnav.model.basemodel.internvla_n1", r / "model" / "basemodel" / "internvla_n1")
    _ns("internnav.model.basemodel.navdp", r / "model" / "basemodel" / "navdp")
    _ns("internnav.model.utils", r / "model" / "utils")
    _ns("internnav.configs", r / "configs")
    # cv2 / torchvision are only touched by DepthAnythingV2.infer_image (never called here)
    if "cv2" not in sys.modules:
        cv2 = types.ModuleType("cv2")
        cv2.INTER_CUBIC, cv2.INTER_AREA, cv2.INTER_LINEAR, cv2.INTER_NEAREST = 2, 3, 1, 0
        sys.modules["cv2"] = cv2
    if "torchvision" not in sys.modules:
        tv = types.ModuleType("torchvision")
        tvt = types.ModuleType("torchvision.transforms")
        tvt.Compose = lambda fns: fns
        tv.transforms = tvt
        sys.modules["torchvision"] = tv
        sys.modules["torchvision.transforms"] = tvt
    enc = str(r / "model" / "encoder")
    if enc not in sys.path:
        sys.path.insert(0, enc)  # vendored diffusion_policy (SinusoidalPosEmb)
    # fake diffusers: schedulers + Lumina blocks restated in oracle/
    if "diffusers" not in sys.modules:
        from . import diffusers_blocks as blk
        from . import schedulers as sch

        blk.install_fake_diffusers(sch)
    _done = True


def dinov2_vits():
    """reference DINOv2('vits') constructor (depth_anything_v2/dinov2.py:399-411)."""
    setup()
    mod = importlib.import_module("internnav.model.encoder.depth_anything.depth_anything_v2.dinov2")
    return mod.DINOv2("vits")


def navdp_backbone_module():
    setup()
    return importlib.import_module("internnav.model.encoder.navdp_backbone")


def navdp_policy_module():
    """internnav/model/basemodel/navdp/navdp_policy.py with its config imports stubbed (pydantic cfg classes unused here)."""
    setup()
    for name in ("internnav.configs.model", "internnav.configs.model.base_encoders", "internnav.configs.trainer",
                 "internnav.configs.trainer.exp"):
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.__path__ = []
            sys.modules[name] = m
    sys.modules["internnav.configs.model.base_encoders"].ModelCfg = lambda **kw: types.SimpleNamespace(**kw)
    sys.modules["internnav.configs.trainer.exp"].ExpCfg = dict
    mod = importlib.import_module("internnav.model.basemodel.navdp.navdp_policy")

    class _CpuTorch:
        """navdp_policy.py:74 hard-codes torch.device(f"cuda:{local_rank}"): route that one call to the CPU."""

        def __getattr__(self, name):
            return getattr(torch, name)

        @staticmethod
        def device(*_a, **_k):
            return torch.device("cpu")

    import torch

    mod.torch = _CpuTorch()
    return mod


def n1_navdp_module():
    setup()
    return importlib.import_module("internnav.model.basemodel.internvla_n1.navdp")


def n1_arch_module():
    setup()
    return importlib.import_module("internnav.model.basemodel.internvla_n1.internvla_n1_arch")


def nextdit_module():
    setup()
    return importlib.import_module("internnav.model.basemodel.internvla_n1.nextdit_crossattn_traj")
