"""CPU oracle of the InternVLA-N1 / NavDP policy hot path - TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A plain fp32 PyTorch-CPU restatement (explicit tensor math, no nn.Module of the reference, no GPU) of the arithmetic
the reference executes on the path SURVEY.md section 8 scopes. Every function cites the reference file:line it
follows. Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this package; the
product package `internnav_amd` never does (it has no CPU fallback and fails loudly without its HIP library).

Pinning status (details in DESIGN.md "Oracle"):
  * in-tree reference code (DINOv2 ViT-S, RGB-D backbones, NavDPNet, N1 NavDP head, MemoryEncoder, QFormer, NextDiT
    block wiring): PINNED against the reference's own modules executed in the build container
    (`oracle/make_golden.py` -> `tests/golden/*.pt`); the reference itself has no golden vectors.
  * diffusers==0.33.1 pieces (DDPMScheduler, FlowMatchEulerDiscreteScheduler, Lumina* blocks) are an un-vendored
    third-party dependency absent from this image: restated from the published algorithm -> "parity unpinned".
  * transformers Qwen2.5-VL arithmetic: pinned against the installed transformers (5.x; reference pins 4.51.0, same math).
"""
