// Skinny-M GEMM for gfx950 (M <= 64): C[M,N] = epilogue(A[M,K] . W[N,K]^T) as a WEIGHT-STREAMING kernel.
//
// Used by the decode steps of the System-2 LLM (M = number of sequences), the latent-query pass, lm_head on the last position
// and the adaLN modulation GEMMs. At M <= 64 the op is HBM-bound: every weight byte is read once and used for M <= 64 MACs, so
// the design goal is bytes/s, not MFMA utilisation. Common to the kernels in this file:
//   * W is streamed straight from HBM into VGPRs (no LDS round trip: each weight element is used by exactly one wave), 4 x 16-byte
//     loads per lane per 128-wide K step; load s of a wave covers 64 CONTIGUOUS bytes of each of its 16 rows (k = k0 + s*32 + g*8,
//     the natural MFMA k order). The first version gave every lane 64 contiguous bytes instead, i.e. 4 scattered 16-byte pieces
//     per row per instruction: 3.3 -> 4.0 TB/s on the gate/up projection, 3.8 -> 4.9 TB/s on lm_head from this change alone;
//   * the activation rows (M x K bf16, <= 2.4 MB, L2 resident) are read directly as the second MFMA operand;
//   * one summation order per output element: K is cut into 128-wide steps, step j belongs to slice j % NS (NS = sk_group_waves(column
//     tiles, K steps): 1 / 2 / 4 / 8 waves of a group), a slice accumulates its steps in ascending order on the MFMA pipe, and the slices are
//     added in ascending order starting from 0.0f.
// gemm_skinny_fused_kernel / gemm_skinny_prenorm_kernel (column owners; the library's automatic choice at M <= 64): a group of NS waves of
//   one workgroup owns its output columns for all of K, reduces through LDS and applies the epilogue; the prenorm form (M <= 16) also
//   RMS-normalises its input rows itself.
// A split-K form of the single-token passes' GEMMs - the NS slices of a column tile as NS workgroups, write-through fp32 slabs, a ticket per
//   tile, the last arriver adding the slabs in slice order (bit-identical to these kernels), rope + KV append in the q|k|v reducer - was built
//   and measured in round 6 and is NOT here: 4.94-5.24 ms per pass against 4.02 alone, 288 against 295 policy steps/s inside the step
//   (profiles/NEGATIVE.md, r06c_*). The reducer's tail (ticket round trip + slab read-back, 3-12 us per GEMM) costs more than the launch it saves.
// Algorithmic bytes per launch = 2*N*K (weights) + 2*M*K + out; roofline = HBM (measured 3.6 - 4.7 TB/s on the LLM shapes,
// profiles/r01i_skinny_gemm_streaming.log; the 25 - 33 MB projections are latency-bound at ~2 TB/s).
#include "common.h"
#include "kernels.h"

namespace {

constexpr int SK_BK = 128;   // K elements per step

// waves of a group that share one column tile's K steps: enough to put a few thousand waves in
// flight, never more than there are K steps. `tiles` = 16-column tiles (GLU: 32 interleaved gate | up rows of W)
inline int sk_group_waves(int tiles, int ksteps) {
    int nw = tiles >= 4096 ? 1 : tiles >= 2048 ? 2 : tiles >= 1024 ? 4 : 8;   // (16 waves on the K = 18944 down projection: no gain)
    while (nw > 1 && nw > ksteps) nw >>= 1;
    return nw;
}
// ... of the kernels with the fused input norm (every workgroup repeats the normalisation: at least 4 slices per tile)
inline int sk_group_waves_prenorm(int tiles, int ksteps) {
    const int nw = sk_group_waves(tiles, ksteps);
    return nw < 4 ? 4 : nw;
}

// RMSNorm of the <= 16 activation rows into an LDS image of bf16 MFMA operand rows (row stride lds_ld = K + 8 elements: the 16 fragment rows
// of a ds_read_b128 land on 16 different bank quads): wave `wave` of NWAVES takes rows wave, wave + NWAVES, ...; lane l the 8-element chunks
// l, l + 64, ... of the row. ONE definition for every kernel that normalises its own input, so they hand the MFMAs bit-identical operands:
// per lane the squares are accumulated by fma in ascending chunk / element order, the 64 lane sums by the xor butterfly of wave_sum.
template <int NWAVES>
__device__ __forceinline__ void sk_prenorm_rows(const GemmArgs& p, bf16* img, int lds_ld, int wave, int lane) {
    const int nch = p.K >> 3;
    const float invK = 1.0f / (float)p.K;
    const bool a32 = p.a_dtype == INA_DT_F32;
    auto load8 = [&](int m, int c, float (&v)[8]) {
        if (a32) {
            const float* q = reinterpret_cast<const float*>(p.A) + (size_t)m * p.lda + c * 8;
            const f32x4 a = *reinterpret_cast<const f32x4*>(q), b = *reinterpret_cast<const f32x4*>(q + 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) { v[j] = a[j]; v[4 + j] = b[j]; }
        } else {
            const bf16x8 a = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const bf16*>(p.A) + (size_t)m * p.lda + c * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = (float)a[j];
        }
    };
    for (int m = wave; m < p.M; m += NWAVES) {          // wave-uniform
        float sq = 0.f;
        for (int c = lane; c < nch; c += 64) {
            float v[8];
            load8(m, c, v);
#pragma unroll
            for (int j = 0; j < 8; ++j) sq = fmaf(v[j], v[j], sq);
        }
        const float rstd = rsqrtf(fmaf(wave_sum(sq), invK, p.norm_eps));
        for (int c = lane; c < nch; c += 64) {           // second read of the row: L1 / L2 hits, keeps the register budget small
            float v[8];
            load8(m, c, v);
            const f32x4 ga = *reinterpret_cast<const f32x4*>(p.norm_gamma + c * 8), gb = *reinterpret_cast<const f32x4*>(p.norm_gamma + c * 8 + 4);
            bf16x8 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                o[j] = (bf16)(v[j] * rstd * ga[j]);
                o[4 + j] = (bf16)(v[4 + j] * rstd * gb[j]);
            }
            *reinterpret_cast<bf16x8*>(img + (size_t)m * lds_ld + c * 8) = o;
        }
    }
}

// ---- column-owner variant: a group of NW waves owns NT16 x 16 output columns for ALL of K (a workgroup holds NC such groups); the
// group's waves take interleaved 128-wide K steps, keep a DEPTH-deep register ring of loads in flight, reduce their accumulators
// through LDS (NW > 1) and apply the epilogue in the same kernel: no partial workspace, no second launch (a decode pass of the
// 28-layer LLM is 4 such GEMMs per layer: the separate epilogue launches were ~0.8 ms of every pass). NW is chosen per shape so
// the launch has a few thousand waves: 1 for the 152064-row lm_head (long-lived independent waves), 8 for a 3584-row projection.
template <int MF, int NT16, int NW, int NC, int DEPTH>
__global__ __launch_bounds__(NW * NC * 64) void gemm_skinny_fused_kernel(GemmArgs p) {
    __shared__ __attribute__((aligned(16))) float red[NW > 1 ? NC : 1][NW > 1 ? NW : 1][NT16][MF][64 * 4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int grp = wave / NW, w = wave % NW;
    const int n0 = (blockIdx.x * NC + grp) * (16 * NT16);
    const int r16 = lane & 15, g = lane >> 4;
    const bf16* __restrict__ A = reinterpret_cast<const bf16*>(p.A);
    const bf16* __restrict__ W = reinterpret_cast<const bf16*>(p.W);
    const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    const bf16* wrow[NT16];
    bool wok[NT16];
#pragma unroll
    for (int t = 0; t < NT16; ++t) {
        const int wn = n0 + t * 16 + r16;
        wok[t] = wn < p.N;
        wrow[t] = W + (size_t)(wok[t] ? wn : 0) * p.ldw;
    }
    const bf16* arow[MF];
    bool aok[MF];
#pragma unroll
    for (int i = 0; i < MF; ++i) {
        const int m = i * 16 + r16;
        aok[i] = m < p.M;
        arow[i] = A + (size_t)(aok[i] ? m : 0) * p.lda;
    }
    f32x4 acc[NT16][MF];
#pragma unroll
    for (int t = 0; t < NT16; ++t)
#pragma unroll
        for (int i = 0; i < MF; ++i) acc[t][i] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int ksteps = (p.K + SK_BK - 1) / SK_BK;
    const int nmine = w < ksteps ? (ksteps - w + NW - 1) / NW : 0;   // K steps w, w + NW, ...
    bf16x8 wf[DEPTH][NT16][4], af[DEPTH][MF][4];
    auto load = [&](int j, int slot) {
        const int k0 = (w + j * NW) * SK_BK + g * 8;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int k = k0 + s * 32;   // load s: the 4 lane groups of a row cover 64 contiguous bytes
            const bool kok = j < nmine && k < p.K;
#pragma unroll
            for (int t = 0; t < NT16; ++t) wf[slot][t][s] = (kok && wok[t]) ? *reinterpret_cast<const bf16x8*>(wrow[t] + k) : zero8;
#pragma unroll
            for (int i = 0; i < MF; ++i) af[slot][i][s] = (kok && aok[i]) ? *reinterpret_cast<const bf16x8*>(arow[i] + k) : zero8;
        }
    };
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) load(d, d);
    for (int j0 = 0; j0 < nmine; j0 += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {      // static ring slots
#pragma unroll
            for (int t = 0; t < NT16; ++t)
#pragma unroll
                for (int i = 0; i < MF; ++i)
#pragma unroll
                    for (int s = 0; s < 4; ++s) acc[t][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[d][t][s], af[d][i][s], acc[t][i], 0, 0, 0);
            load(j0 + d + DEPTH, d);
        }
    }
    // ---- cross-wave reduction (NW > 1) + epilogue: wave w of a group owns row fragments w, w + NW, ... of every column tile
    if constexpr (NW > 1) {
#pragma unroll
        for (int t = 0; t < NT16; ++t)
#pragma unroll
            for (int i = 0; i < MF; ++i) *reinterpret_cast<f32x4*>(&red[grp][w][t][i][lane * 4]) = acc[t][i];
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < MF; ++i) {
        if (NW > 1 && (i % NW) != w) continue;
        const int m = i * 16 + r16;
        f32x4 sum[NT16];
#pragma unroll
        for (int t = 0; t < NT16; ++t) {
            if constexpr (NW > 1) {
                sum[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ww = 0; ww < NW; ++ww) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(&red[grp][ww][t][i][lane * 4]);
#pragma unroll
                    for (int r = 0; r < 4; ++r) sum[t][r] += v[r];
                }
            } else {
                sum[t] = acc[t][i];
            }
        }
        const int n = n0 + g * 4;
        if (m >= p.M || n >= p.N) continue;
        const float rs = p.rowscale ? p.rowscale[m / p.rowscale_div] : 1.0f;
        float v[4];
        int no = n;
        if constexpr (NT16 == 2) {
            // GLU: tile 0 = gate rows, tile 1 = up rows of the interleaved weight; output column block = pair index
            no = (n0 >> 1) + g * 4;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float gg = sum[0][r], uu = sum[1][r];
                if (p.bias) { gg += p.bias[n + r]; uu += p.bias[n + 16 + r]; }
                v[r] = ina_act(gg, p.act) * uu * rs;
            }
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float x = sum[0][r];
                if (p.bias) x += p.bias[n + r];
                x = ina_act(x, p.act);
                if (p.colscale) x *= p.colscale[n + r];
                v[r] = x * rs;
            }
            if (p.R) {
                const size_t ro = (size_t)m * p.ldr + n;
                if (p.res_dtype == INA_DT_BF16) {
                    const bf16x4 rr = *reinterpret_cast<const bf16x4*>(reinterpret_cast<const bf16*>(p.R) + ro);
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] += (float)rr[r];
                } else {
                    const f32x4 rr = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(p.R) + ro);
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] += rr[r];
                }
            }
        }
        const size_t co = (size_t)m * p.ldc + no;
        if (p.out_dtype == INA_DT_BF16) *reinterpret_cast<bf16x4*>(reinterpret_cast<bf16*>(p.C) + co) = bf16x4{(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
        else *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.C) + co) = f32x4{v[0], v[1], v[2], v[3]};
    }
}

template <int MF, int NT16>
void launch_skinny_fused(const GemmArgs& p, hipStream_t stream) {
    const int ksteps = (p.K + SK_BK - 1) / SK_BK;
    const int tiles = (p.N + 16 * NT16 - 1) / (16 * NT16);
    // waves per column group: enough to put >= ~4096 waves in flight (16 per CU), never more than there are K steps
    constexpr int DEEP = (MF <= 2 && NT16 == 1) ? 4 : 2;
    const int nw = sk_group_waves(tiles, ksteps);
#define INA_SKF(NW_, NC_, D_) hipLaunchKernelGGL((gemm_skinny_fused_kernel<MF, NT16, NW_, NC_, D_>), dim3((tiles + NC_ - 1) / NC_), dim3(NW_ * NC_ * 64), 0, stream, p)
    // (round 3 measured and dropped: twice the register ring for the single-fragment shapes, and 4-wave column groups everywhere - neutral
    //  alone, -10 % inside the step where the chain shares the CUs with System-1: profiles/r03e / r03g_bench_*_experiments.log)
    if (nw == 1) INA_SKF(1, 4, 2);
    else if (nw == 2) INA_SKF(2, 2, 2);
    else if (nw == 4) INA_SKF(4, 1, 2);
    else INA_SKF(8, 1, DEEP);
#undef INA_SKF
}



// ---- column-owner variant with the INPUT RMSNorm fused in front (M <= 16: the single-token decode passes of the LLM, where a separate
// norm launch over 7 x 3584 values costs as much as a 30 MB projection). Every workgroup first requests its weight ring, then - while
// those loads are in flight - normalises the <= 16 activation rows itself: one wave per row reads the row (f32 residual stream or bf16
// embeddings, L2 hits: every workgroup reads the same <= 230 KB), reduces the sum of squares and writes bf16(x * rstd * gamma) into an
// LDS image whose rows sit 16 bytes past a multiple of 256 (the 16 fragment rows of a ds_read_b128 land on 16 different bank quads).
// The main loop is the fused kernel's with the activation fragments read from that image instead of global memory.
template <int NT16, int NW, int NC, int DEPTH>
__global__ __launch_bounds__(NW * NC * 64) void gemm_skinny_prenorm_kernel(GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char sk_smem[];
    constexpr int NWAVES = NW * NC;
    const int lds_ld = p.K + 8;
    bf16* img = reinterpret_cast<bf16*>(sk_smem);                                                     // [M][K + 8]
    float* red = reinterpret_cast<float*>(sk_smem + (((size_t)p.M * lds_ld * sizeof(bf16) + 15) & ~size_t(15)));   // [NC][NW][NT16][256]
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int grp = wave / NW, w = wave % NW;
    const int n0 = (blockIdx.x * NC + grp) * (16 * NT16);
    const int r16 = lane & 15, g = lane >> 4;
    const bf16* __restrict__ W = reinterpret_cast<const bf16*>(p.W);
    const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    const bf16* wrow[NT16];
    bool wok[NT16];
#pragma unroll
    for (int t = 0; t < NT16; ++t) {
        const int wn = n0 + t * 16 + r16;
        wok[t] = wn < p.N;
        wrow[t] = W + (size_t)(wok[t] ? wn : 0) * p.ldw;
    }
    f32x4 acc[NT16];
#pragma unroll
    for (int t = 0; t < NT16; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int ksteps = (p.K + SK_BK - 1) / SK_BK;
    const int nmine = w < ksteps ? (ksteps - w + NW - 1) / NW : 0;
    bf16x8 wf[DEPTH][NT16][4];
    auto load_w = [&](int j, int slot) {
        const int k0 = (w + j * NW) * SK_BK + g * 8;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int k = k0 + s * 32;
            const bool kok = j < nmine && k < p.K;
#pragma unroll
            for (int t = 0; t < NT16; ++t) wf[slot][t][s] = (kok && wok[t]) ? *reinterpret_cast<const bf16x8*>(wrow[t] + k) : zero8;
        }
    };
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) load_w(d, d);

    sk_prenorm_rows<NWAVES>(p, img, lds_ld, wave, lane);   // (the weight ring is in flight meanwhile)
    __syncthreads();

    // rows of the MFMA fragment beyond M read a valid row: their outputs are never stored
    const bf16* arow = img + (size_t)(r16 < p.M ? r16 : p.M - 1) * lds_ld + g * 8;
    for (int j0 = 0; j0 < nmine; j0 += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const int kb = (w + (j0 + d) * NW) * SK_BK;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int k = kb + s * 32;
                const bf16x8 af = (j0 + d < nmine && k + g * 8 < p.K) ? *reinterpret_cast<const bf16x8*>(arow + k) : zero8;
#pragma unroll
                for (int t = 0; t < NT16; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[d][t][s], af, acc[t], 0, 0, 0);
            }
            load_w(j0 + d + DEPTH, d);
        }
    }
    if constexpr (NW > 1) {
#pragma unroll
        for (int t = 0; t < NT16; ++t) *reinterpret_cast<f32x4*>(&red[(((size_t)grp * NW + w) * NT16 + t) * 256 + lane * 4]) = acc[t];
        __syncthreads();
        if (w != 0) return;        // one 16-row fragment: wave 0 of the group reduces and stores
    }
    const int m = r16, n = n0 + g * 4;
    f32x4 sum[NT16];
#pragma unroll
    for (int t = 0; t < NT16; ++t) {
        if constexpr (NW > 1) {
            sum[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ww = 0; ww < NW; ++ww) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(&red[(((size_t)grp * NW + ww) * NT16 + t) * 256 + lane * 4]);
#pragma unroll
                for (int r = 0; r < 4; ++r) sum[t][r] += v[r];
            }
        } else {
            sum[t] = acc[t];
        }
    }
    if (m >= p.M || n >= p.N) return;
    const float rs = p.rowscale ? p.rowscale[m / p.rowscale_div] : 1.0f;
    float v[4];
    int no = n;
    if constexpr (NT16 == 2) {
        no = (n0 >> 1) + g * 4;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float gg = sum[0][r], uu = sum[1][r];
            if (p.bias) { gg += p.bias[n + r]; uu += p.bias[n + 16 + r]; }
            v[r] = ina_act(gg, p.act) * uu * rs;
        }
    } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float x = sum[0][r];
            if (p.bias) x += p.bias[n + r];
            x = ina_act(x, p.act);
            if (p.colscale) x *= p.colscale[n + r];
            v[r] = x * rs;
        }
        if (p.R) {
            const size_t ro = (size_t)m * p.ldr + n;
            if (p.res_dtype == INA_DT_BF16) {
                const bf16x4 rr = *reinterpret_cast<const bf16x4*>(reinterpret_cast<const bf16*>(p.R) + ro);
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] += (float)rr[r];
            } else {
                const f32x4 rr = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(p.R) + ro);
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] += rr[r];
            }
        }
    }
    const size_t co = (size_t)m * p.ldc + no;
    if (p.out_dtype == INA_DT_BF16) *reinterpret_cast<bf16x4*>(reinterpret_cast<bf16*>(p.C) + co) = bf16x4{(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
    else *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.C) + co) = f32x4{v[0], v[1], v[2], v[3]};
}

template <int NT16, int NW, int NC, int DEPTH>
int launch_skinny_prenorm(const GemmArgs& p, hipStream_t stream, int tiles) {
    const size_t lds = (((size_t)p.M * (p.K + 8) * sizeof(bf16) + 15) & ~size_t(15)) + (NW > 1 ? size_t(NC) * NW * NT16 * 256 * sizeof(float) : 0);
    auto kern = gemm_skinny_prenorm_kernel<NT16, NW, NC, DEPTH>;
    static size_t attr = 0;
    if (lds > attr) {
        INA_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr = lds;
    }
    hipLaunchKernelGGL(kern, dim3((tiles + NC - 1) / NC), dim3(NW * NC * 64), lds, stream, p);
    INA_HIP_CHECK(hipGetLastError());
    return 0;
}


}  // namespace

int ina_launch_gemm_skinny_fused(const GemmArgs& p, hipStream_t stream) {
    const double osz = p.out_dtype == INA_DT_BF16 ? 2.0 : 4.0;
    InaProfScope prof(INA_PROF_GEMM_SKINNY, 2.0 * p.M * p.N * p.K, 2.0 * p.M * p.K + 2.0 * p.N * p.K + osz * p.M * (p.glu ? p.N / 2 : p.N), stream);
    const int mf = (p.M + 15) / 16;
    if (p.glu) {
        switch (mf) {
            case 1: launch_skinny_fused<1, 2>(p, stream); break;
            case 2: launch_skinny_fused<2, 2>(p, stream); break;
            case 3: launch_skinny_fused<3, 2>(p, stream); break;
            default: launch_skinny_fused<4, 2>(p, stream); break;
        }
    } else {
        switch (mf) {
            case 1: launch_skinny_fused<1, 1>(p, stream); break;
            case 2: launch_skinny_fused<2, 1>(p, stream); break;
            case 3: launch_skinny_fused<3, 1>(p, stream); break;
            default: launch_skinny_fused<4, 1>(p, stream); break;
        }
    }
    INA_HIP_CHECK(hipGetLastError());
    return 0;
}


int ina_launch_gemm_skinny_prenorm(const GemmArgs& p, hipStream_t stream) {
    const double osz = p.out_dtype == INA_DT_BF16 ? 2.0 : 4.0, asz = p.a_dtype == INA_DT_F32 ? 4.0 : 2.0;
    InaProfScope prof(INA_PROF_GEMM_SKINNY, 2.0 * p.M * p.N * p.K, asz * p.M * p.K + 2.0 * p.N * p.K + osz * p.M * (p.glu ? p.N / 2 : p.N), stream);
    // every workgroup repeats the (cheap, L2-served) normalisation, so the column groups are packed NC per workgroup where one group
    // alone would be a 4-wave workgroup; the group width follows the fused kernel's rule (a few thousand waves per launch)
    const int ksteps = (p.K + SK_BK - 1) / SK_BK;
    if (p.glu) {
        const int tiles = (p.N + 31) / 32;
        if (sk_group_waves_prenorm(tiles, ksteps) == 4) return launch_skinny_prenorm<2, 4, 2, 2>(p, stream, tiles);
        return launch_skinny_prenorm<2, 8, 1, 2>(p, stream, tiles);
    }
    const int tiles = (p.N + 15) / 16;
    if (sk_group_waves_prenorm(tiles, ksteps) == 4) return launch_skinny_prenorm<1, 4, 2, 2>(p, stream, tiles);
    return launch_skinny_prenorm<1, 8, 1, 4>(p, stream, tiles);
}
