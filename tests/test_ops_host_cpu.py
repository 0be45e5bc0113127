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


def test_w4p_counted_waits_cover_their_loads():
    """Tile config 40 (csrc/gemm_w4.hip) retires its B-fragment loads and LDS-DMA pieces with counted `s_waitcnt vmcnt(N)` over ONE in-order
    counter. This replays the kernel's fixed instruction stream on the host - prologue, first stage, steady state, the last two stages that
    request nothing - for several K depths and checks every wait: `vmcnt(N)` retires a request iff at least N requests were issued after it.
    The constants are read out of the source (a change there must be made here as well)."""
    import re
    from pathlib import Path

    src = (Path(__file__).resolve().parent.parent / "internnav_amd" / "csrc" / "gemm_w4.hip").read_text()
    assert "wait_vm<(SET == 0) ? 15 + 8 * P : 15 + (7 - j) * P + j * D>();" in src          # B fragment j of slice 0 / slice 1
    assert "wait_vm<9>();" in src and "wait_vm<8 + 16>();" in src                              # DMA share of stage t + 1; prologue: stage 0 of A
    rd, dma_pos = (int(re.search(rf"#define {n} (\d)", src).group(1)) for n in ("W4P_RD_POS", "W4P_DMA_POS"))
    assert 0 <= rd < 7 and 0 <= dma_pos < 7                                                   # both precede the B reload at position 7
    assert "for (; t < 1 && t + 2 < nk; ++t) stage_body(IC<1>{}, IC<0>{}, t);" in src and "for (; t + 2 < nk; ++t) stage_body(IC<1>{}, IC<1>{}, t);" in src
    assert "for (; t < nk; ++t) stage_body(IC<0>{}, IC<0>{}, t);" in src

    def n_b(slice_par, j, D, P):
        return 15 + 8 * P if slice_par == 0 else 15 + (7 - j) * P + j * D

    for nk in (1, 2, 3, 4, 5, 9):
        issued = []                                            # the VMEM requests in issue order: ("A", stage, piece) | ("B", slice, fragment)

        def retired(req, n):                                   # would `s_waitcnt vmcnt(n)` issued NOW have waited for req?
            return len(issued) - 1 - issued.index(req) >= n

        for st in (0, 1):
            issued.extend(("A", st, d) for d in range(8))
        for sl in (0, 1):
            issued.extend(("B", sl, j) for j in range(8))
        assert all(retired(("A", 0, d), 8 + 16) for d in range(8))
        for t in range(nk):
            D, P = (1, 0) if (t == 0 and nk > 2) else (1, 1) if t + 2 < nk else (0, 0)
            real_D = int(t + 2 < nk)
            assert D == real_D                                  # (P may under-state the truth in the last stages: a stricter wait)
            for par in (0, 1):
                sl = 2 * t + par
                for j in range(8):
                    assert retired(("B", sl, j), n_b(par, j, D, P)), (nk, t, par, j)
                    # the 8 MFMAs of fragment j: positions 0 .. 7; the DMA piece (slice 1 of a requesting stage) precedes the B reload
                    if par == 1 and real_D:
                        issued.append(("A", t + 2, j))
                    issued.append(("B", sl + 2, j))            # (clamped to the last slice behind the end: still one request)
                if par == 0 and t + 1 < nk:
                    assert all(retired(("A", t + 1, d), 9) for d in range(8)), (nk, t)
        # exactness in the steady state: one request fewer after the awaited load and the wait would no longer cover it
        if nk >= 5:
            issued2, checks = [], []
            for st in (0, 1):
                issued2.extend(("A", st, d) for d in range(8))
            for sl in (0, 1):
                issued2.extend(("B", sl, j) for j in range(8))
            for t in range(nk):
                for par in (0, 1):
                    for j in range(8):
                        if 2 <= t and t + 2 < nk:
                            checks.append(len(issued2) - 1 - issued2.index(("B", 2 * t + par, j)) - n_b(par, j, 1, 1))
                        if par == 1 and t + 2 < nk:
                            issued2.append(("A", t + 2, j))
                        issued2.append(("B", 2 * t + par + 2, j))
            assert checks and set(checks) == {0}
