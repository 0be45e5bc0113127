"""Host side of the op front end (internnav_amd/ops.py) without a GPU: what reaches ina_gemm_bf16 for a given call.

A stand-in for the shared library records the argument struct of every GEMM call (test code only; the product path always binds
libinternnav_amd.so, `_lib.lib()` raises without it)."""
import ctypes as C

import torch

from internnav_amd import _lib, ops


class _Recorder:
    def __init__(self):
        self.calls = []

    def ina_gemm_bf16(self, byref_args, stream):
        a = byref_args._obj
        self.calls.append({k: getattr(a, k) for k, _ in a._fields_})
        return 0


def _linear(monkeypatch, **kw):
    rec = _Recorder()
    monkeypatch.setattr(_lib, "lib", lambda: rec)
    monkeypatch.setattr(ops, "_stream", lambda: 0)
    x = torch.zeros(256, 128, dtype=torch.bfloat16)
    w = torch.zeros(512, 128, dtype=torch.bfloat16)
    out = torch.empty(256, 512, dtype=torch.bfloat16)
    ops.linear(x, w, out=out, **kw)
    return rec.calls[-1]


def test_gemm_args_layout_and_auto_tile_mode(monkeypatch):
    a = _linear(monkeypatch)
    assert (a["M"], a["N"], a["K"], a["lda"], a["ldw"], a["ldc"], a["batch"]) == (256, 512, 128, 128, 128, 512, 1)
    assert a["force_cfg"] == 0 and a["group_m"] == 0 and a["norm_gamma"] is None


def test_shared_tail_marks_auto_gemms_only(monkeypatch):
    """inside ops.shared_tail() (the two-stream System-2 prefill) library-selected tiles are requested with force_cfg = -1 (the cost
    model does not charge the last round the other stream fills); an explicit tile config is left alone; the mode ends with the block,
    also when the block raises."""
    with ops.shared_tail():
        assert _linear(monkeypatch)["force_cfg"] == -1
        assert _linear(monkeypatch, force_cfg=18)["force_cfg"] == 18
        with ops.shared_tail():                      # nests
            assert _linear(monkeypatch)["force_cfg"] == -1
        assert _linear(monkeypatch)["force_cfg"] == -1
    assert _linear(monkeypatch)["force_cfg"] == 0
    try:
        with ops.shared_tail():
            raise RuntimeError("launch failed")
    except RuntimeError:
        pass
    assert _linear(monkeypatch)["force_cfg"] == 0


def test_gemm_struct_mirror_matches_header_field_order():
    """the ctypes mirror lists the fields of ina_gemm_args in the header's order (sizes are checked against the compiled library in test_abi)."""
    import re
    from pathlib import Path

    hdr = (Path(__file__).resolve().parent.parent / "include" / "internnav_amd.h").read_text()
    body = hdr[hdr.index("typedef struct ina_gemm_args {"):hdr.index("} ina_gemm_args;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = []
    for decl in body.split("{", 1)[1].split(";"):
        decl = decl.strip()
        if not decl:
            continue
        for part in decl.split(","):
            names.append(part.replace("*", " ").split()[-1])
    assert names == [k for k, _ in _lib.GemmArgs._fields_]
    assert C.sizeof(_lib.GemmArgs) % 8 == 0
