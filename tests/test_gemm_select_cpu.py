"""Kernel selection of ina_gemm_bf16, inspected through ina_gemm_select on the CPU (host arithmetic only, nothing is launched).

The table pins what the tile cost model of csrc/gemm.hip chooses for the GEMM shapes of the hot path, in the plain mode (force_cfg = 0)
and in the shared-tail mode of the two-stream System-2 prefill (force_cfg = -1). Every entry is backed by a measurement under profiles/
(r01f / r02m tile sweeps, r03v isolated sweep, r03w / r03x / r03z chain sweeps); every tile shape gives bit-equal results, so a change
here is a performance decision, never a numerical one."""
import ctypes as C

import pytest

H, I, QKV = 3584, 18944, 4608          # Qwen2.5-VL-7B decoder: hidden, MLP width, fused q|k|v rows
RES = dict(R=0x2000, out_dtype=1)      # fp32 output accumulated into the fp32 residual stream
GLU = dict(glu=1, act=4)               # SwiGLU pairing in the epilogue


@pytest.fixture(scope="module")
def select(built_lib):
    from internnav_amd import _lib

    h = _lib.lib()

    def sel(M, N, K, **kw):
        a = _lib.GemmArgs()
        a.A = a.W = a.C = 0x1000          # aligned dummies: selection never dereferences them
        a.M, a.N, a.K = M, N, K
        a.lda = a.ldw = K
        a.ldc = a.ldr = N
        for k, v in kw.items():
            setattr(a, k, v)
        out = C.c_int(0)
        rc = h.ina_gemm_select(C.byref(a), C.byref(out))
        return out.value if rc == 0 else ("error", h.ina_last_error().decode())
    return sel


@pytest.mark.parametrize("rows,plain,shared", [
    # rows of 7 / 6 prompts (joint prefill) and of 4 / 3 prompts (the halves of the two-stream prefill): q|k|v, o + residual, gate|up, down + residual
    (6440, (39, 21, 39, 21), (39, 33, 39, 18)),
    (5520, (39, 21, 39, 21), (39, 33, 39, 18)),
    (3680, (21, 33, 39, 18), (39, 33, 39, 18)),
    (2760, (39, 21, 39, 21), (39, 33, 39, 18)),
])
def test_decoder_layer_gemms(select, rows, plain, shared):
    """plain: 192 x 256 tiles (21) where 256-row tiles quantise badly over the 256 CUs; shared tail: 256 x 256 everywhere (the other half's
    workgroups fill the last round), the o projection (fp32 residual, K = 3584) on the 16-wave tile (33), the K = 18944 down projection not.
    Round 4: wherever the 256 x 256 geometry is selected for q|k|v and gate|up (no residual, K = 3584, N >= 4096) it runs on the four-wave
    kernel (39, gemm_w4.hip; profiles/r04l_native_w4.log, r04m_native_w4_chain.log)."""
    for mode, want in ((0, plain), (-1, shared)):
        got = (select(rows, QKV, H, force_cfg=mode), select(rows, H, H, force_cfg=mode, **RES),
               select(rows, 2 * I, H, force_cfg=mode, **GLU), select(rows, H, I, force_cfg=mode, **RES))
        assert got == want, f"rows {rows}, force_cfg {mode}"


@pytest.mark.parametrize("rows", [6440, 5520, 3680, 2760])
def test_decoder_layer_gemms_with_fragment_ordered_weights(select, rows):
    """Round 5: with the fragment-ordered copy of W at hand (ina_gemm_args.Wp) the four-wave tile takes its B fragments straight from global
    memory (config 40) wherever 39 would run, and instead of the ping-pong tile (18) on the K = 18944 down projection; a choice of the
    192-row tile (21) or of the 16-wave tile (33: o projection) is not touched (profiles/r05u_native_w4p.log). Bit-equal either way."""
    WP = dict(Wp=0x3000)
    for mode in (0, -1):
        base = (select(rows, QKV, H, force_cfg=mode), select(rows, H, H, force_cfg=mode, **RES),
                select(rows, 2 * I, H, force_cfg=mode, **GLU), select(rows, H, I, force_cfg=mode, **RES))
        got = (select(rows, QKV, H, force_cfg=mode, **WP), select(rows, H, H, force_cfg=mode, **RES, **WP),
               select(rows, 2 * I, H, force_cfg=mode, **GLU, **WP), select(rows, H, I, force_cfg=mode, **RES, **WP))
        want = tuple(40 if (b == 39 or (b == 18 and i == 3)) else b for i, b in enumerate(base))
        assert got == want, f"rows {rows}, force_cfg {mode}: {base} -> {got}"
    assert select(7, QKV, H, **WP) == 32 and select(rows, QKV, H, force_cfg=18, **WP) == 18      # single-token passes and forced tiles ignore the copy
    err = select(rows, QKV, H, force_cfg=40)
    assert err[0] == "error" and "fragment-ordered copy" in err[1]


@pytest.mark.parametrize("rows,plain,shared", [
    (21952, (18, 33, 18, 33), (18, 33, 18, 33)),      # 7 prompts x 4 frames x 784 patches
    (18816, (18, 21, 18, 21), (18, 33, 18, 33)),
    (12544, (18, 33, 18, 33), (18, 33, 18, 33)),
    (9408, (21, 21, 18, 21), (18, 33, 18, 33)),
])
def test_vision_block_gemms(select, rows, plain, shared):
    """qkv, proj + residual, gate|up (3420 padded to 3456), down + residual of a Qwen2.5-VL vision block: the two fp32-residual GEMMs take the
    16-wave 256 x 256 tile wherever the 256 x 256 geometry is selected (K = 1280 / 3456 <= 4096)."""
    for mode, want in ((0, plain), (-1, shared)):
        got = (select(rows, 3840, 1280, force_cfg=mode), select(rows, 1280, 1280, force_cfg=mode, **RES),
               select(rows, 6912, 1280, force_cfg=mode, **GLU), select(rows, 1280, 3456, force_cfg=mode, **RES))
        assert got == want, f"rows {rows}, force_cfg {mode}"


def test_d384_heads_and_odd_shapes(select):
    for rows in (65536, 58368, 7168):            # 64 / 57 / 7 envs x 32 samples x 32 tokens
        big = rows >= 16384                        # round 4: the row-panel kernels (34 / 35, gemm_rowpanel.hip) from 16 envs x 1024 rows on
        assert select(rows, 1536, 384) == (34 if big else 26)      # fused q|k|v|q2; below: 128 x 256 single buffer, 3 workgroups / CU
        assert select(rows, 2048, 384, ldc=1024, **GLU) == (35 if big else 26)
        assert select(rows, 1536, 384, bias=0x3000, act=1) == 26   # bias + activation (NavDP decoder FFN): measured slower on the row-panel kernel, stays tiled
        assert select(rows, 1536, 384, colscale=0x3000) == 26      # a LayerScale is outside the row-panel contract: tiled kernel
        assert select(rows, 384, 384, **RES) == 22    # N = 384 with the fp32-residual epilogue: 128 x 128 single buffer, 4 workgroups / CU
        assert select(rows, 384, 1024, **RES) == 22
    assert select(21952, 1280, 1176) == 1         # patch embed: K % 64 != 0 -> register-staged kernel
    assert select(5488, 5120, 5120) == 18 and select(5488, H, 5120) == 21      # patch merger MLP
    assert (select(100, 100, 64), select(300, 48, 64), select(32, 100, 64), select(32, 132, 64)) == (2, 5, 4, 3)


def test_weight_streaming_paths(select):
    assert select(7, QKV, H) == 32                                    # decode: fused weight-streaming kernel
    assert select(7, QKV, H, norm_gamma=0x3000, a_dtype=1) == 30      # with the input RMSNorm inside
    assert select(64, 152064, H) == 32 and select(65, 152064, H) == 11    # lm_head: 64 rows is the boundary
    assert select(7, QKV, H, force_cfg=-1) == 32                      # the shared-tail mode does not touch this path


def test_forced_configs_and_rejections(select):
    assert select(6440, QKV, H, force_cfg=21) == 21 and select(6440, QKV, H, force_cfg=33) == 33 and select(6440, QKV, H, force_cfg=18) == 18
    # four-wave tile: forced anywhere inside its contract (16-byte aligned output rows), selected for wide no-residual K = 2048 .. 4096 GEMMs only
    assert select(300, 264, 64, force_cfg=39) == 39 and select(6440, H, H, force_cfg=39, **RES) == 39
    assert select(2047, QKV, H) != 39 and select(6440, 4096, 2048) == 39 and select(6440, 4096, 4160) == 18 and select(6440, 3584, H) != 39
    bad = select(300, 260, 64, force_cfg=39)               # bf16 rows of 520 bytes
    assert isinstance(bad, tuple) and "39 / 40" in bad[1]
    for bad in (select(100, 100, 63), select(7, QKV, H, force_cfg=30), select(100, 48, 64, glu=1), select(0, 4, 8),
                select(100, 102, 64), select(17, QKV, H, norm_gamma=0x3000, a_dtype=1), select(100, 512, 64, force_cfg=32), select(7, QKV, H, force_cfg=31), select(7, QKV, H, force_cfg=60), select(6440, QKV, H, force_cfg=38), select(6440, QKV, H, force_cfg=23), select(65536, 1536, 384, force_cfg=36),
                select(7, QKV, H, force_cfg=18, norm_gamma=0x3000, a_dtype=1)):
        assert isinstance(bad, tuple) and bad[0] == "error" and bad[1].startswith("gemm")
