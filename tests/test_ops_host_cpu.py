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


def test_rowpanel_lds_image_is_bank_conflict_free_and_consistent():
    """csrc/gemm_rowpanel.hip: a W stage is 128 rows x 128 bytes; physical 16-byte chunk cp of row r holds logical chunk cp ^ ((r >> 1) & 7)
    (the DMA applies it on the source address, the fragment reads on the LDS address). Enumerated here: (1) the two maps are inverse to each
    other - lane (row, k half) of k step kk reads the 8 bf16 the MFMA operand layout asks for; (2) every ds_read_b128 of a 32x32x16 fragment
    hits 16 DIFFERENT 16-byte slots of the 256-byte bank row inside each of the four 16-lane service groups of the instruction
    (guides/MI355X_MICROARCH.md, LDS table: {0-3,12-15,20-27}, {4-11,16-19,28-31} and the same + 32)."""
    f = lambda r: (r >> 1) & 7  # noqa: E731
    # (1) DMA: lane (drow, dcp) of the instruction covering rows r0 .. r0+7 lands at row r0 + drow, physical chunk dcp, fetching logical dcp ^ f(r)
    phys = {}
    for r in range(128):
        for dcp in range(8):
            phys[(r, dcp)] = dcp ^ f(r)                      # logical chunk stored at physical position dcp of row r
    for j in range(4):
        for lane in range(64):
            row, khalf = j * 32 + (lane & 31), lane >> 5
            for kk in range(4):
                want = kk * 2 + khalf                        # logical 16-byte chunk (8 bf16) of this lane's operand fragment
                cp = want ^ f(lane & 31)                     # the kernel's foff[]: computed from lrow only
                assert phys[(row, cp)] == want
    # (2) bank slots
    groups = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
    groups += [[x + 32 for x in g] for g in groups]
    for j in range(4):
        for kk in range(4):
            for g in groups:
                slots = set()
                for lane in g:
                    row, khalf = j * 32 + (lane & 31), lane >> 5
                    addr = row * 128 + (((kk * 2 + khalf) ^ f(row)) << 4)
                    slots.add((addr // 16) % 16)
                assert len(slots) == 16, (j, kk, g)


def test_rowpanel_contract_through_gemm_select(built_lib):
    """ina_gemm_select validates force_cfg 34-37 against the row-panel kernels' contract without a GPU."""
    h = _lib.lib()

    def sel(M, N, K, cfg, **kw):
        a = _lib.GemmArgs()
        a.A = a.W = a.C = 0x1000
        a.M, a.N, a.K, a.lda, a.ldw, a.ldc, a.ldr, a.force_cfg = M, N, K, K, K, N, N, cfg
        for k, v in kw.items():
            setattr(a, k, v)
        out = C.c_int(0)
        return out.value if h.ina_gemm_select(C.byref(a), C.byref(out)) == 0 else None
    assert sel(65536, 1536, 384, 34) == 34 and sel(7168, 2048, 384, 35, glu=1, act=4, ldc=1024) == 35
    assert sel(65536, 1536, 512, 34) is None and sel(65536, 1500, 384, 34) is None and sel(100, 1536, 384, 35) is None
    assert sel(65536, 1536, 384, 34, out_dtype=1) is None and sel(65536, 2048, 384, 34, glu=1, act=0) is None and sel(65536, 2048, 384, 34, glu=1, act=4, bias=0x3000) is None
    assert sel(49152, 1152, 384, 34, bias=0x3000) == 34 and sel(49152, 1536, 384, 35, bias=0x3000, act=1) == 35 and sel(49152, 1152, 384, 0, bias=0x3000) == 27
