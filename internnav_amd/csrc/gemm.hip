// bf16 MFMA GEMM for gfx950:  C[M,N] = epilogue( A[M,K] . W[N,K]^T )
//
// A is the activation matrix (row-major, K contiguous), W is an nn.Linear weight exactly as PyTorch stores
// it ([out_features, in_features], K contiguous) - both operands are therefore K-contiguous and their MFMA
// fragments are plain 16-byte LDS reads.  Replaces every cuBLAS nn.Linear on the reference hot path
// (SURVEY.md 2c "cuBLAS GEMMs (all nn.Linear)").
//
// Structure: 128x128 (or 64x128 / 128x64 / 64x64) output tile per 256-thread workgroup, BK = 64, two LDS
// stages, register-staged global->LDS copies (tile t+1 is in flight in VGPRs while tile t is multiplied),
// LDS rows padded by 16 B so the ds_read_b128 fragment reads are bank-conflict free.
// The MFMA is issued "swapped" (W fragment as the A operand, activation fragment as the B operand) so each
// lane ends up holding 4 consecutive output columns of one output row -> 8-byte bf16 / 16-byte f32 stores.
//
// Epilogue (fused, fp32): +bias[n] -> activation -> *colscale[n] (LayerScale) -> *rowscale[m] -> +residual[m,n]
// -> store bf16 or f32.  Optional GLU mode pairs neighbouring 16-column blocks (gate,up) and writes
// act(gate)*up to N/2 columns (Qwen SwiGLU, LuminaFeedForward).
#include "common.h"
#include "kernels.h"
#include "gemm_epilogue.h"

namespace {

template <int BM, int BN, int BK, int WM, int WN>
struct GemmCfg {
    static constexpr int NT = WM * WN * 64;
    static constexpr int TM = BM / WM;  // wave tile rows
    static constexpr int TN = BN / WN;
    static constexpr int FM = TM / 16;
    static constexpr int FN = TN / 16;
    static constexpr int LDS_K = BK + 8;  // padded row (bf16 elements)
    static constexpr int CPR = BK / 8;    // 16-byte chunks per tile row
    static constexpr int A_CHUNKS = BM * CPR / NT;
    static constexpr int B_CHUNKS = BN * CPR / NT;
    static constexpr size_t LDS_BYTES = size_t(2) * (BM + BN) * LDS_K * sizeof(bf16);
};

template <int BM, int BN, int BK, int WM, int WN>
__global__ __launch_bounds__(WM* WN * 64) void gemm_bf16_nt_kernel(GemmArgs p) {
    using C = GemmCfg<BM, BN, BK, WM, WN>;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    bf16* As = reinterpret_cast<bf16*>(smem_raw);                  // [2][BM][LDS_K]
    bf16* Bs = As + 2 * BM * C::LDS_K;                             // [2][BN][LDS_K]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;

    const int tiles_n = (p.N + BN - 1) / BN;
    const int tiles_m = (p.M + BM - 1) / BM;
    const int nwg = tiles_m * tiles_n;
    const int id = xcd_remap(blockIdx.x, nwg);
    const int tm = id / tiles_n, tn = id % tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;

    const bf16* __restrict__ A = reinterpret_cast<const bf16*>(p.A) + (size_t)blockIdx.y * p.strideA;
    const bf16* __restrict__ W = reinterpret_cast<const bf16*>(p.W) + (size_t)blockIdx.y * p.strideW;

    bf16x8 ra[C::A_CHUNKS], rb[C::B_CHUNKS];
    const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};

    auto load_tiles = [&](int k0) {
#pragma unroll
        for (int i = 0; i < C::A_CHUNKS; ++i) {
            int q = tid + i * C::NT;
            int row = q / C::CPR, c = q % C::CPR;
            int gm = m0 + row, gk = k0 + c * 8;
            ra[i] = (gm < p.M && gk < p.K) ? *reinterpret_cast<const bf16x8*>(A + (size_t)gm * p.lda + gk) : zero8;
        }
#pragma unroll
        for (int i = 0; i < C::B_CHUNKS; ++i) {
            int q = tid + i * C::NT;
            int row = q / C::CPR, c = q % C::CPR;
            int gn = n0 + row, gk = k0 + c * 8;
            rb[i] = (gn < p.N && gk < p.K) ? *reinterpret_cast<const bf16x8*>(W + (size_t)gn * p.ldw + gk) : zero8;
        }
    };
    auto store_tiles = [&](int buf) {
        bf16* as = As + buf * BM * C::LDS_K;
        bf16* bs = Bs + buf * BN * C::LDS_K;
#pragma unroll
        for (int i = 0; i < C::A_CHUNKS; ++i) {
            int q = tid + i * C::NT;
            int row = q / C::CPR, c = q % C::CPR;
            *reinterpret_cast<bf16x8*>(as + row * C::LDS_K + c * 8) = ra[i];
        }
#pragma unroll
        for (int i = 0; i < C::B_CHUNKS; ++i) {
            int q = tid + i * C::NT;
            int row = q / C::CPR, c = q % C::CPR;
            *reinterpret_cast<bf16x8*>(bs + row * C::LDS_K + c * 8) = rb[i];
        }
    };

    f32x4 acc[C::FM][C::FN];
#pragma unroll
    for (int i = 0; i < C::FM; ++i)
#pragma unroll
        for (int j = 0; j < C::FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = (p.K + BK - 1) / BK;
    load_tiles(0);
    store_tiles(0);
    __syncthreads();

    const int frow = lane & 15;        // row inside a 16-row fragment
    const int fk = (lane >> 4) * 8;    // k offset of this lane's 8 contiguous elements

    for (int t = 0; t < nk; ++t) {
        const int buf = t & 1;
        if (t + 1 < nk) load_tiles((t + 1) * BK);
        const bf16* as = As + buf * BM * C::LDS_K + (wm * C::TM + frow) * C::LDS_K + fk;
        const bf16* bs = Bs + buf * BN * C::LDS_K + (wn * C::TN + frow) * C::LDS_K + fk;
#pragma unroll
        for (int kk = 0; kk < BK / 32; ++kk) {
            bf16x8 fa[C::FM], fb[C::FN];
#pragma unroll
            for (int i = 0; i < C::FM; ++i) fa[i] = *reinterpret_cast<const bf16x8*>(as + i * 16 * C::LDS_K + kk * 32);
#pragma unroll
            for (int j = 0; j < C::FN; ++j) fb[j] = *reinterpret_cast<const bf16x8*>(bs + j * 16 * C::LDS_K + kk * 32);
#pragma unroll
            for (int i = 0; i < C::FM; ++i)
#pragma unroll
                for (int j = 0; j < C::FN; ++j)
                    // swapped operands: D[row = n_local][col = m_local]
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
        }
        if (t + 1 < nk) store_tiles(buf ^ 1);
        __syncthreads();
    }

    // LDS-transposed epilogue (gemm_epilogue.h) when the wave tile is 64 columns wide and C / R rows are 16-byte aligned
    bool staged = false;
    if constexpr (C::TN == 64 && C::LDS_BYTES >= size_t(WM * WN) * C::FM * 4096) {
        const size_t oes = p.out_dtype == INA_DT_BF16 ? 2 : 4, res = p.res_dtype == INA_DT_BF16 ? 2 : 4;
        staged = ((uintptr_t)p.C % 16) == 0 && (p.ldc * oes) % 16 == 0 && (p.strideC * oes) % 16 == 0 &&
                 (!p.R || (((uintptr_t)p.R % 16) == 0 && (p.ldr * res) % 16 == 0 && (p.strideR * res) % 16 == 0));
        if (staged) {
            // (the K loop ends with a barrier: every wave is done with the stage buffers, which become the transpose scratch)
            gemm_store_tile_staged<C::FM, C::FN, C::TM, C::TN, true>(p, acc, m0, n0, wm, wn, lane,
                                                                     reinterpret_cast<float*>(smem_raw) + wave * C::FM * 1024);
        }
    }
    if (!staged) gemm_store_tile<C::FM, C::FN, C::TM, C::TN>(p, acc, m0, n0, wm, wn, lane);
}

template <int BM, int BN, int BK, int WM, int WN>
int launch_cfg(const GemmArgs& p, hipStream_t stream) {
    using C = GemmCfg<BM, BN, BK, WM, WN>;
    static bool attr_done = false;
    auto kern = gemm_bf16_nt_kernel<BM, BN, BK, WM, WN>;
    if (!attr_done) {
        INA_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                          (int)C::LDS_BYTES));
        attr_done = true;
    }
    const int tiles = ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
    dim3 grid(tiles, p.batch, 1);
    const double osz = p.out_dtype == INA_DT_BF16 ? 2.0 : 4.0;
    InaProfScope prof(INA_PROF_GEMM, 2.0 * p.M * p.N * p.K * p.batch,
                      (double)p.batch * (2.0 * p.M * p.K + 2.0 * p.N * p.K + osz * p.M * (p.glu ? p.N / 2 : p.N)), stream);
    hipLaunchKernelGGL(kern, grid, dim3(C::NT), C::LDS_BYTES, stream, p);
    INA_HIP_CHECK(hipGetLastError());
    return 0;
}

}  // namespace

// Validation + kernel selection of one GEMM call, without launching anything (host arithmetic only: also reachable as ina_gemm_select so
// that the selection can be inspected / tested without a GPU). `kernel`: 1-5 register-staged tiles (gemm_bf16_nt_kernel), 11-27 / 33
// LDS-DMA tiles (gemm_glds.hip), 39 / 40 the four-wave 256 x 256 tile (gemm_w4.hip), 30 = weight-streaming kernel with the fused input RMSNorm, 32 = weight streaming (gemm_skinny.hip).
int ina_plan_gemm(const GemmArgs& p_in, GemmArgs& p, int& kernel) {
    p = p_in;
    if (p.rowscale_div <= 0) p.rowscale_div = 1;
    if (p.batch <= 0) p.batch = 1;
    INA_REQUIRE(p.M > 0 && p.N > 0 && p.K > 0, "gemm: empty problem M=%d N=%d K=%d", p.M, p.N, p.K);
    INA_REQUIRE(p.K % 8 == 0 && p.lda % 8 == 0 && p.ldw % 8 == 0, "gemm: K/lda/ldw must be multiples of 8 (K=%d lda=%d ldw=%d)",
                p.K, p.lda, p.ldw);
    INA_REQUIRE(p.N % 4 == 0 && p.ldc % 4 == 0, "gemm: N/ldc must be multiples of 4 (N=%d ldc=%d)", p.N, p.ldc);
    INA_REQUIRE(((uintptr_t)p.A % 16) == 0 && ((uintptr_t)p.W % 16) == 0 && ((uintptr_t)p.C % 8) == 0, "gemm: misaligned pointer");
    INA_REQUIRE(!p.R || p.ldr % 4 == 0, "gemm: ldr must be a multiple of 4");
    INA_REQUIRE(!p.glu || (p.N % 32 == 0), "gemm: GLU mode needs N %% 32 == 0");
    // tile selection: big tiles when the grid still fills the 256 CUs, smaller ones otherwise
    // skinny M: HBM-bound weight streaming with split-K (gemm_skinny.hip) instead of an under-filled tile grid
    INA_REQUIRE(!p.seg_stats || (p.M > 64 && !p.norm_gamma), "gemm: seg_stats exist in the row-panel kernels only (M >= 16384 rows; M=%d)", p.M);
    if (p.norm_gamma) {
        INA_REQUIRE(p.M <= 16 && p.batch <= 1 && p.K % 8 == 0 && p.K <= 4096 && p.N >= 256,
                    "gemm: the fused input RMSNorm is built for the decode passes (M <= 16 rows, K <= 4096, one batch): M=%d K=%d N=%d batch=%d", p.M, p.K, p.N, p.batch);
        INA_REQUIRE(p.a_dtype == INA_DT_BF16 || p.a_dtype == INA_DT_F32, "gemm: a_dtype must be bf16 or f32 with norm_gamma");
        INA_REQUIRE(((uintptr_t)p.norm_gamma % 16) == 0 && (p.lda % (p.a_dtype == INA_DT_F32 ? 4 : 8)) == 0, "gemm(prenorm): misaligned gamma / lda");
        INA_REQUIRE(p.force_cfg <= 0, "gemm: the fused input RMSNorm exists in the automatic weight-streaming kernel only (force_cfg %d)", p.force_cfg);
        kernel = 30;
        return 0;
    }
    INA_REQUIRE(p.force_cfg != 30, "gemm: kernel 30 (fused input RMSNorm) is selected by norm_gamma, not by force_cfg");
    if (p.force_cfg <= 0 && p.M <= 64 && p.batch == 1 && p.N >= 256) {
        kernel = 32;
        return 0;
    }
    if (p.force_cfg == 32) {
        INA_REQUIRE(p.M <= 64 && p.batch == 1, "gemm: the weight-streaming kernels need M <= 64, batch 1 (M=%d)", p.M);
        kernel = 32;
        return 0;
    }
    const long t128 = (long)((p.M + 127) / 128) * ((p.N + 127) / 128) * p.batch;
    int cfg = p.force_cfg;
    if (cfg <= 0) {
        if (p.M <= 64) cfg = (p.N >= 128) ? 3 : 4;
        else if (t128 >= 256 || (p.M >= 128 && p.N >= 128 && t128 >= 96)) cfg = 1;
        else if (p.N <= 64) cfg = 5;
        else cfg = 2;
    }
    if (p.force_cfg <= 0 && cfg == 1 && p.K % 64 == 0) {
        // K multiple of 64: LDS-DMA staged kernels. Pick the tile shape by a small cost model: rounds of workgroups over the 256 CUs
        // (tile quantisation) x tile area / measured relative per-CU rate (profiles/r01f_gemm_tile_sweep.log):
        //   11: 128x128, 2 WG/CU, rate 0.85 | 14: 256x128 x 3 stages, 1 WG/CU, 0.92 | 18: 256x256 ping-pong, 1 WG/CU, 1.18 | 21: 192x256 ping-pong, 0.97 (long K only)
        cfg = p.K <= 1024 ? 22 : 11;   // short K: single-buffer 128x128 (32 KiB LDS, 4 workgroups per CU) beats the 2-stage ring by 3-13 %
        // short K and a wide output (the d = 384 heads' fused q|k|v|q2 / SwiGLU projections, K = 384): 48 KiB single-buffer tiles with 85
        // instead of 64 FLOP per operand byte, 3 workgroups per CU: +14-15 % (740-816 vs 632-704 TF/s, profiles/r02m_gemm_s1_tile_sweep.log)
        if (p.K <= 512 && p.N >= 1024 && p.M >= 4096) cfg = p.N >= 1536 ? 26 : 27;
        // K = 384, wide output, many rows, plain or SwiGLU epilogue (the fused q|k|v|q2 and SwiGLU projections of a NextDiT block over >= 16 envs):
        // row-panel kernels - activations as register-resident MFMA fragments, only W streams through LDS (gemm_rowpanel.hip). NOT bit-equal
        // to the tiled kernels (K in ascending order but by 16-wide MFMA steps: last-bit differences in < 20 % of the bf16 outputs,
        // tests/test_ops_gpu.py::test_gemm_rowpanel_k384). Isolated they are within +-10 % of cfg 26 (0.75-0.86 PF/s, profiles/r04b_native_rowpanel.log); inside the
        // System-1 call, where the tiled kernel's operands start cold, the call of 64 envs goes 84.2 -> 75.0 ms (r04f_s1_variants.log).
        // Below ~16 k rows their one-workgroup-per-256-rows grid leaves the chip empty (7168 rows: 72 vs 17 us).
        // The bias + activation epilogue of those kernels exists (force_cfg 34 / 35) but is NOT selected: the biased q|k|v / GELU-FFN
        // projections of the NavDP decoder (49152 x 1152 / 1536) lose on it - NavDPNet call of 64 envs 186.5 vs 146.0 ms (r04g_navdp_variants.log).
        if (p.M >= 16384 && !p.bias && (p.glu || p.act == INA_ACT_NONE) && ina_gemm_rowpanel_contract(p)) cfg = p.glu ? 35 : 34;
        if (p.K >= 768 && p.N > 512) {   // narrow outputs (N = 384 heads): 128x128 measured 6-15 % ahead of 256x128 at K = 1024 / 1536
            // force_cfg = -1 (INA_GEMM_AUTO_SHARED): auto selection for a launch that runs BESIDE another stream's GEMMs (the two half
            // micro-batches of the System-2 prefill): the CUs its last round leaves idle are taken by the other stream's workgroups, so
            // tile quantisation is not charged (fractional rounds) and the tile with the best per-CU rate wins. Measured on the decoder
            // chain of the two halves, two streams: 60.7 vs 62.7 ms (2760 + 2760 rows), 71.7 vs 72.7 ms (3680 + 2760)
            // (profiles/r03w_native_chain_sweep.log). Every tile shape accumulates K in the same order: the choice never changes a bit.
            const bool shared = p.force_cfg == -1;
            auto cost = [&](int bm, int bn, int wg_per_cu, double rate) {
                const long tiles = (long)((p.M + bm - 1) / bm) * ((p.N + bn - 1) / bn) * p.batch;
                const long slots = 256L * wg_per_cu;
                const double rounds = shared ? (double)tiles / slots : (double)((tiles + slots - 1) / slots);
                return rounds * wg_per_cu * (bm / 128.0) * (bn / 128.0) / rate;
            };
            const double c11 = cost(128, 128, 2, 0.85), c14 = cost(256, 128, 1, 0.92), c18 = cost(256, 256, 1, 1.18);
            const double c21 = cost(192, 256, 1, 0.97);   // 192-row ping-pong tiles: the row counts of 6 / 7 prompts (5520 / 6440) round better
            cfg = (c18 < c14 && c18 < c11) ? 18 : (c14 < c11 ? 14 : 11);
            const double best = cfg == 18 ? c18 : cfg == 14 ? c14 : c11;
            if (c21 < best) cfg = 21;
            // 256 x 256 tile, fp32-residual epilogue, short K loop (the vision blocks' proj / down projections; K = 3584 is neutral): the
            // lock-step kernel with SIXTEEN waves (wave tile 64 x 64, cfg 33) instead of the 8-wave ping-pong. Such a tile spends as long
            // reading and writing its 512 KB of fp32 as in its K loop, and twice the waves keep twice the epilogue loads in flight:
            // vision chain of the two prefill halves 26.9 vs 28.1-28.3 ms; decoder chain and the d = 384 heads unchanged
            // (profiles/r03z_native_chain_16wave.log). Same tile geometry and K order: bit-equal (r03z_native_gemm_16wave.log).
            if (cfg == 18 && p.R && p.out_dtype == INA_DT_F32 && p.K <= 4096) cfg = 33;
            // 256 x 256 tile, no residual, wide output, K = 2048 .. 4096 (the decoder's q|k|v and gate|up projections, K = 3584): the FOUR-wave
            // kernel (cfg 39, gemm_w4.hip: wave tile 128 x 128, hand-threaded MFMA / LDS-read / DMA stream). Isolated +2-6 % (1.23-1.26 vs
            // 1.18-1.22 PF/s, profiles/r04l_native_w4.log); decoder chain of the two prefill halves 71.1 -> 70.2 ms, joint 73.2 -> 72.0
            // (r04m_native_w4_chain.log). Bit-equal (same K order). It loses where the epilogue or the prologue weighs more: fp32-residual
            // outputs (o / down: -3 ... -14 %), K = 1280 (vision blocks: -4 ... -6 %) - those stay on cfg 18 / 33.
            if (cfg == 18 && !p.R && p.K >= 2048 && p.K <= 4096 && p.N >= 4096 && p.M >= 2048 && ina_gemm_w4_contract(p)) cfg = 39;
            // ... and cfg 40 where the caller also holds W in MFMA fragment order (Wp, ina_gemm_preshuffle): B fragments straight from global
            // memory into registers, LDS-DMA for A only - half the DMA pieces and half the fragment reads of cfg 39 (gemm_w4.hip). Bit-equal.
            // Also ahead of the ping-pong tile on the long-K fp32-residual down projection (K = 18944: 444 vs 469 us at 3680 rows, 434 vs 442 at 2760;
            // profiles/r05s_native_w4p.log); the short-K residual GEMMs (o, vision blocks) stay on cfg 33 / 18.
            if ((cfg == 39 || (cfg == 18 && p.K >= 8192 && ina_gemm_w4_contract(p))) && p.Wp && p.N % 16 == 0 && p.batch == 1) cfg = 40;
        }
    }
    {
        static const int known[] = {1, 2, 3, 4, 5, 11, 14, 18, 21, 22, 26, 27, 33, 34, 35, 39, 40};     // every one of them is a choice of the cost model above for some shape
        bool ok = false;
        for (int k : known) ok = ok || k == cfg;
        INA_REQUIRE(ok, "gemm: unknown tile config %d", cfg);
    }
    if (cfg == 40) INA_REQUIRE(p.Wp && p.N % 16 == 0 && p.batch == 1, "gemm: tile config 40 needs the fragment-ordered copy of W (Wp), N %% 16 == 0, no batch (N=%d)", p.N);
    if (cfg == 39 || cfg == 40)
        INA_REQUIRE(ina_gemm_w4_contract(p), "gemm: tile configs 39 / 40 (four-wave 256 x 256 tile) need K %% 64 == 0 and 16-byte aligned output / residual rows "
                    "(M=%d N=%d K=%d ldc=%d)", p.M, p.N, p.K, p.ldc);
    INA_REQUIRE(!p.seg_stats || cfg == 34 || cfg == 35, "gemm: seg_stats (LayerNorm statistics of the produced rows) exist in the row-panel kernels only "
                "(K = 384, plain epilogue, M >= 16384): this call would run tile config %d (M=%d N=%d K=%d)", cfg, p.M, p.N, p.K);
    if (cfg == 34 || cfg == 35)
        INA_REQUIRE(ina_gemm_rowpanel_contract(p), "gemm: tile configs 34 / 35 (row-panel kernels) need K = 384, N %% 128 == 0, M %% 32 == 0, bf16 output, "
                    "no scales / residual, bias + activation or SiLU-GLU (M=%d N=%d K=%d)", p.M, p.N, p.K);
    kernel = cfg;
    return 0;
}

int ina_launch_gemm(const GemmArgs& p_in, hipStream_t stream) {
    GemmArgs p;
    int cfg = 0;
    if (int rc = ina_plan_gemm(p_in, p, cfg)) return rc;
    if (cfg == 30) return ina_launch_gemm_skinny_prenorm(p, stream);
    if (cfg == 32) return ina_launch_gemm_skinny_fused(p, stream);
    ina_prof_set_sub(cfg);
    switch (cfg) {
        case 1: return launch_cfg<128, 128, 64, 2, 2>(p, stream);
        case 2: return launch_cfg<64, 128, 64, 2, 2>(p, stream);   // wave tile 32x64
        case 3: return launch_cfg<64, 128, 64, 1, 4>(p, stream);   // wave tile 64x32 (skinny M)
        case 4: return launch_cfg<64, 64, 64, 2, 2>(p, stream);    // wave tile 32x32
        case 5: return launch_cfg<128, 64, 64, 2, 2>(p, stream);   // wave tile 64x32 (narrow N)
        case 11: case 14: case 18: case 21: case 22: case 26: case 27: case 33: case 39: case 40: return ina_launch_gemm_glds(p, stream, cfg);  // LDS-DMA staged kernels (K % 64 == 0)
        case 34: case 35: return ina_launch_gemm_rowpanel(p, stream, cfg);   // K = 384 row-panel kernels (gemm_rowpanel.hip)
        default: ina_set_error("gemm: unknown tile config %d", cfg); return -2;
    }
}
