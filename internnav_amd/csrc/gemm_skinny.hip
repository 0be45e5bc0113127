// Skinny-M GEMM for gfx950 (M <= 64): C[M,N] = epilogue(A[M,K] . W[N,K]^T) as a WEIGHT-STREAMING kernel.
//
// Used by the decode steps of the System-2 LLM (M = number of sequences), the latent-query pass, lm_head on the last position
// and the adaLN modulation GEMMs. At M <= 64 the op is HBM-bound: every weight byte is read once and used for M <= 64 MACs, so
// the design goal is bytes/s, not MFMA utilisation. Common to both kernels in this file:
//   * W is streamed straight from HBM into VGPRs (no LDS round trip: each weight element is used by exactly one wave), 4 x 16-byte
//     loads per lane per 128-wide K step; load s of a wave covers 64 CONTIGUOUS bytes of each of its 16 rows (k = k0 + s*32 + g*8,
//     the natural MFMA k order). The first version gave every lane 64 contiguous bytes instead, i.e. 4 scattered 16-byte pieces
//     per row per instruction: 3.3 -> 4.0 TB/s on the gate/up projection, 3.8 -> 4.9 TB/s on lm_head from this change alone;
//   * the activation rows (M x K bf16, <= 2.4 MB, L2 resident) are read directly as the second MFMA operand.
// gemm_skinny_fused_kernel (the library's automatic choice at M <= 64): a group of waves owns its output columns for all of K, reduces
//   through LDS and applies the epilogue itself. gemm_skinny_kernel + gemm_skinny_epilogue (cfg 31): grid = (N / 64 column tiles) x SPLITK
//   K-slices, fp32 partial tiles in a workspace [SPLITK][M][N], second kernel sums the slices and applies the epilogue. Round 5: the engine
//   forces cfg 31 for the single-token decoder passes - alone the pair is 1.5 % slower, inside the policy step its many short 256-thread
//   workgroups get into the gaps System-1's long workgroups leave on the CUs (+ 1.5 % policy steps/s) - and its epilogue has a row-owning form
//   (gemm_skinny_epilogue_rows) that also writes the NEXT GEMM's RMS-normed operand (ina_gemm_args.post_gamma).
// Algorithmic bytes per launch = 2*N*K (weights) + 2*M*K + out; roofline = HBM (measured 3.6 - 4.7 TB/s on the LLM shapes,
// profiles/r01i_skinny_gemm_streaming.log; the 25 - 33 MB projections are latency-bound at ~2 TB/s).
#include "common.h"
#include "kernels.h"

namespace {

constexpr int SK_BN = 64;    // output columns (W rows) per workgroup: 4 waves x 16
constexpr int SK_BK = 128;   // K elements per step
// ina_gemm_args.group_m == SK_NT_FLAG (the field orders the tiles of the LDS-DMA kernels and means nothing here): the weight stream of the
// column-owner kernels is loaded NON-TEMPORAL (each weight byte is read exactly once per launch and by one wave: it need not displace the
// activations / KV cache from the L2 and the Infinity Cache on its way)
constexpr int SK_NT_FLAG = 7;

__device__ __forceinline__ bf16x8 sk_load_w(const bf16* ptr, bool nt) {
    const bf16x8* q = reinterpret_cast<const bf16x8*>(ptr);
    return nt ? __builtin_nontemporal_load(q) : *q;
}

template <int MF>  // number of 16-row activation fragments (M <= 16*MF)
__global__ __launch_bounds__(256) void gemm_skinny_kernel(GemmArgs p, float* __restrict__ part, int kslice) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n0 = blockIdx.x * SK_BN + wave * 16;
    const int split = blockIdx.y;
    const int k_begin = split * kslice;
    const int k_end = min(p.K, k_begin + kslice);
    const int r16 = lane & 15, g = lane >> 4;
    const bf16* __restrict__ A = reinterpret_cast<const bf16*>(p.A);
    const bf16* __restrict__ W = reinterpret_cast<const bf16*>(p.W);
    const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};

    f32x4 acc[MF];
#pragma unroll
    for (int i = 0; i < MF; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int wn = n0 + r16;
    const bool wok = wn < p.N;
    const bf16* wrow = W + (size_t)(wok ? wn : 0) * p.ldw;

    // software pipeline: the loads of K step t+1 (4 x 16 B of W + 4*MF x 16 B of A per lane) are issued before the MFMAs of step t
    auto load_w = [&](int k0, bf16x8 (&wf)[4]) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int k = k0 + s * 32 + g * 8;   // load s: the 4 lane groups of a row cover 64 contiguous bytes (natural MFMA k order)
            wf[s] = (wok && k < k_end) ? *reinterpret_cast<const bf16x8*>(wrow + k) : zero8;
        }
    };
    auto load_a = [&](int k0, bf16x8 (&af)[MF][4]) {
#pragma unroll
        for (int i = 0; i < MF; ++i) {
            const int m = i * 16 + r16;
            const bf16* arow = A + (size_t)(m < p.M ? m : 0) * p.lda;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int k = k0 + s * 32 + g * 8;
                af[i][s] = (m < p.M && k < k_end) ? *reinterpret_cast<const bf16x8*>(arow + k) : zero8;
            }
        }
    };
    bf16x8 wf[2][4], af[2][MF][4];
    load_w(k_begin, wf[0]);
    load_a(k_begin, af[0]);
    int cur = 0;
    for (int k0 = k_begin; k0 < k_end; k0 += 2 * SK_BK) {
        // two K steps per trip so the ping-pong register sets are indexed statically
        load_w(k0 + SK_BK, wf[1]);
        load_a(k0 + SK_BK, af[1]);
#pragma unroll
        for (int i = 0; i < MF; ++i)
#pragma unroll
            for (int s = 0; s < 4; ++s) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[0][s], af[0][i][s], acc[i], 0, 0, 0);
        load_w(k0 + 2 * SK_BK, wf[0]);
        load_a(k0 + 2 * SK_BK, af[0]);
#pragma unroll
        for (int i = 0; i < MF; ++i)
#pragma unroll
            for (int s = 0; s < 4; ++s) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[1][s], af[1][i][s], acc[i], 0, 0, 0);
    }
    (void)cur;
    const int n = n0 + g * 4;
    if (n < p.N) {
#pragma unroll
        for (int i = 0; i < MF; ++i) {
            const int m = i * 16 + r16;
            if (m < p.M) *reinterpret_cast<f32x4*>(part + ((size_t)split * p.M + m) * p.N + n) = acc[i];
        }
    }
}

// sum the K-slices and apply the epilogue; one thread per 4 output columns (GLU: per gate/up pair of 4)
__global__ __launch_bounds__(256) void gemm_skinny_epilogue(GemmArgs p, const float* __restrict__ part, int splits) {
    const int n4 = p.N >> 2;
    const long total = (long)p.M * n4;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int m = (int)(i / n4), n = (int)(i % n4) * 4;
        if (p.glu && ((n >> 4) & 1)) continue;  // "up" blocks are consumed together with their gate block
        float v[4] = {0.f, 0.f, 0.f, 0.f}, u[4] = {0.f, 0.f, 0.f, 0.f};
        for (int s = 0; s < splits; ++s) {
            const float* q = part + ((size_t)s * p.M + m) * p.N + n;
            f32x4 a = *reinterpret_cast<const f32x4*>(q);
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] += a[r];
            if (p.glu) {
                f32x4 b = *reinterpret_cast<const f32x4*>(q + 16);
#pragma unroll
                for (int r = 0; r < 4; ++r) u[r] += b[r];
            }
        }
        const float rs = p.rowscale ? p.rowscale[m / p.rowscale_div] : 1.0f;
        int no = n;
        if (p.glu) {
            no = ((n >> 5) << 4) + (n & 15);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float gg = v[r], uu = u[r];
                if (p.bias) { gg += p.bias[n + r]; uu += p.bias[n + 16 + r]; }
                v[r] = ina_act(gg, p.act) * uu * rs;
            }
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float x = v[r];
                if (p.bias) x += p.bias[n + r];
                x = ina_act(x, p.act);
                if (p.colscale) x *= p.colscale[n + r];
                v[r] = x * rs;
            }
            if (p.R) {
                const size_t ro = (size_t)m * p.ldr + n;
                if (p.res_dtype == INA_DT_BF16) {
                    bf16x4 rr = *reinterpret_cast<const bf16x4*>(reinterpret_cast<const bf16*>(p.R) + ro);
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] += (float)rr[r];
                } else {
                    f32x4 rr = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(p.R) + ro);
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

// the same epilogue as ONE workgroup per output row (no GLU, N <= 4096), which then also holds the whole fp32 row: post_gamma -> the RMSNorm of the
// row, i.e. the NEXT GEMM's bf16 operand, leaves with it (ina_gemm_args.post_gamma / post_out) - in a single-token decoder pass the norm launch
// between the o projection and gate|up, and between the down projection and the next layer's q|k|v, disappears
template <int CH>   // 4-column chunks per thread: the row's N / 4 chunks on ceil(N / 4 / CH) threads
__global__ __launch_bounds__(1024) void gemm_skinny_epilogue_rows(GemmArgs p, const float* __restrict__ part, int splits) {
    // every thread's CH x `splits` partial loads are independent and in flight together
    __shared__ float red[16];
    const int m = blockIdx.x, tid = threadIdx.x, n4 = p.N >> 2, nthr = blockDim.x;
    const float rs = p.rowscale ? p.rowscale[m / p.rowscale_div] : 1.0f;
    float v[CH][4];
    float sq = 0.f;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[c][r] = 0.f;
        const int i4 = tid + c * nthr;
        if (i4 < n4) {
            const int n = i4 * 4;
            for (int s = 0; s < splits; ++s) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(part + ((size_t)s * p.M + m) * p.N + n);
#pragma unroll
                for (int r = 0; r < 4; ++r) v[c][r] += a[r];
            }
        }
    }
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const int i4 = tid + c * nthr;
        if (i4 >= n4) continue;
        const int n = i4 * 4;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float x = v[c][r];
            if (p.bias) x += p.bias[n + r];
            x = ina_act(x, p.act);
            if (p.colscale) x *= p.colscale[n + r];
            v[c][r] = x * rs;
        }
        if (p.R) {
            const size_t ro = (size_t)m * p.ldr + n;
            if (p.res_dtype == INA_DT_BF16) {
                const bf16x4 rr = *reinterpret_cast<const bf16x4*>(reinterpret_cast<const bf16*>(p.R) + ro);
#pragma unroll
                for (int r = 0; r < 4; ++r) v[c][r] += (float)rr[r];
            } else {
                const f32x4 rr = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(p.R) + ro);
#pragma unroll
                for (int r = 0; r < 4; ++r) v[c][r] += rr[r];
            }
        }
        const size_t co = (size_t)m * p.ldc + n;
        if (p.out_dtype == INA_DT_BF16) *reinterpret_cast<bf16x4*>(reinterpret_cast<bf16*>(p.C) + co) = bf16x4{(bf16)v[c][0], (bf16)v[c][1], (bf16)v[c][2], (bf16)v[c][3]};
        else *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.C) + co) = f32x4{v[c][0], v[c][1], v[c][2], v[c][3]};
#pragma unroll
        for (int r = 0; r < 4; ++r) sq = fmaf(v[c][r], v[c][r], sq);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
    if ((tid & 63) == 0) red[tid >> 6] = sq;
    __syncthreads();
    float tot = 0.f;
    const int nwv = (nthr + 63) >> 6;
    for (int w = 0; w < nwv; ++w) tot += red[w];
    const float rn = rsqrtf(tot / (float)p.N + p.post_eps);
    bf16* __restrict__ H = reinterpret_cast<bf16*>(p.post_out) + (size_t)m * p.post_ld;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const int i4 = tid + c * nthr;
        if (i4 >= n4) continue;
        const int n = i4 * 4;
        const f32x4 gm = *reinterpret_cast<const f32x4*>(p.post_gamma + n);
        *reinterpret_cast<bf16x4*>(H + n) = bf16x4{(bf16)(v[c][0] * rn * gm[0]), (bf16)(v[c][1] * rn * gm[1]), (bf16)(v[c][2] * rn * gm[2]), (bf16)(v[c][3] * rn * gm[3])};
    }
}

// ---- column-owner variant: a group of NW waves owns NT16 x 16 output columns for ALL of K (a workgroup holds NC such groups); the
// group's waves take interleaved 128-wide K steps, keep a DEPTH-deep register ring of loads in flight, reduce their accumulators
// through LDS (NW > 1) and apply the epilogue in the same kernel: no partial workspace, no second launch (a decode pass of the
// 28-layer LLM is 4 such GEMMs per layer: the separate epilogue launches were ~0.8 ms of every pass). NW is chosen per shape so
// the launch has a few thousand waves: 1 for the 152064-row lm_head (long-lived independent waves), 8 for a 3584-row projection.
// OCC: minimum waves per SIMD the register allocation must allow (1 = unconstrained). The THIN builds (OCC = 5: at most 96 registers, 4-wave
// workgroups) are sized to co-reside with System-1's row-chain workgroups, which leave 96 registers per SIMD and 90 KiB of LDS on a CU.
template <int MF, int NT16, int NW, int NC, int DEPTH, int OCC = 1>
__global__ __launch_bounds__(NW * NC * 64, OCC) void gemm_skinny_fused_kernel(GemmArgs p) {
    __shared__ __attribute__((aligned(16))) float red[NW > 1 ? NC : 1][NW > 1 ? NW : 1][NT16][MF][64 * 4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int grp = wave / NW, w = wave % NW;
    const int n0 = (blockIdx.x * NC + grp) * (16 * NT16);
    const int r16 = lane & 15, g = lane >> 4;
    const bf16* __restrict__ A = reinterpret_cast<const bf16*>(p.A);
    const bf16* __restrict__ W = reinterpret_cast<const bf16*>(p.W);
    const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    const bool nt = p.group_m == SK_NT_FLAG;
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
            for (int t = 0; t < NT16; ++t) wf[slot][t][s] = (kok && wok[t]) ? sk_load_w(wrow[t] + k, nt) : zero8;
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
    int nw = tiles >= 4096 ? 1 : tiles >= 2048 ? 2 : tiles >= 1024 ? 4 : 8;   // (16 waves on the K = 18944 down projection: no gain)
    while (nw > 1 && nw > ksteps) nw >>= 1;
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
template <int NT16, int NW, int NC, int DEPTH, int OCC = 1>
__global__ __launch_bounds__(NW * NC * 64, OCC) void gemm_skinny_prenorm_kernel(GemmArgs p) {
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
    const bool nt = p.group_m == SK_NT_FLAG;
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
            for (int t = 0; t < NT16; ++t) wf[slot][t][s] = (kok && wok[t]) ? sk_load_w(wrow[t] + k, nt) : zero8;
        }
    };
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) load_w(d, d);

    // ---- fused RMSNorm: wave `wave` takes rows wave, wave + NWAVES, ...; lane l the 8-element chunks l, l + 64, ... of the row
    {
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
                for (int j = 0; j < 8; ++j) sq += v[j] * v[j];
            }
            const float rstd = rsqrtf(wave_sum(sq) * invK + p.norm_eps);
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

template <int NT16, int NW, int NC, int DEPTH, int OCC = 1>
int launch_skinny_prenorm(const GemmArgs& p, hipStream_t stream, int tiles) {
    const size_t lds = (((size_t)p.M * (p.K + 8) * sizeof(bf16) + 15) & ~size_t(15)) + (NW > 1 ? size_t(NC) * NW * NT16 * 256 * sizeof(float) : 0);
    auto kern = gemm_skinny_prenorm_kernel<NT16, NW, NC, DEPTH, OCC>;
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

int ina_launch_gemm_skinny(const GemmArgs& p, hipStream_t stream) {
    const int tiles = (p.N + SK_BN - 1) / SK_BN;
    int splits = 1;
    const int ksteps = (p.K + SK_BK - 1) / SK_BK;
    // (a launch aims at >= 1024 workgroups; inside the policy step 512 / 2048 / 4096 measure 314.0 / 316.3 / 316.4 vs 317.3 policy steps/s, profiles/r05F_*)
    while (tiles * splits < 1024 && splits * 2 <= ksteps && splits < 32) splits *= 2;
    const int kslice = ((ksteps + splits - 1) / splits) * SK_BK;
    splits = (p.K + kslice - 1) / kslice;
    float* part = nullptr;
    const int rc = ina_workspace(0, (size_t)splits * p.M * p.N * sizeof(float), stream, &part);
    if (rc) return rc;
    const double osz = p.out_dtype == INA_DT_BF16 ? 2.0 : 4.0;
    InaProfScope prof(INA_PROF_GEMM_SKINNY, 2.0 * p.M * p.N * p.K, 2.0 * p.M * p.K + 2.0 * p.N * p.K + osz * p.M * (p.glu ? p.N / 2 : p.N), stream);
    dim3 grid(tiles, splits);
    const int mf = (p.M + 15) / 16;
    switch (mf) {
        case 1: hipLaunchKernelGGL(gemm_skinny_kernel<1>, grid, dim3(256), 0, stream, p, part, kslice); break;
        case 2: hipLaunchKernelGGL(gemm_skinny_kernel<2>, grid, dim3(256), 0, stream, p, part, kslice); break;
        case 3: hipLaunchKernelGGL(gemm_skinny_kernel<3>, grid, dim3(256), 0, stream, p, part, kslice); break;
        default: hipLaunchKernelGGL(gemm_skinny_kernel<4>, grid, dim3(256), 0, stream, p, part, kslice); break;
    }
    if (p.post_gamma) {                                    // (ina_plan_gemm: no GLU, N <= 4096)
        // one chunk per thread (16-wave workgroups at N = 3584): 2 / 4 chunks on 8- / 4-wave workgroups measure 312.8 / 305.6 vs 315.0 policy steps/s (profiles/r05K_*)
        hipLaunchKernelGGL(gemm_skinny_epilogue_rows<1>, dim3(p.M), dim3(((p.N / 4 + 63) / 64) * 64), 0, stream, p, part, splits);
        INA_HIP_CHECK(hipGetLastError());
        return 0;
    }
    const long total = (long)p.M * (p.N / 4);
    const int eb = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
    hipLaunchKernelGGL(gemm_skinny_epilogue, dim3(eb), dim3(256), 0, stream, p, part, splits);
    INA_HIP_CHECK(hipGetLastError());
    return 0;
}

// THIN decode GEMMs (force_cfg 60, M <= 16): 4-wave workgroups of at most 96 registers - one wave per SIMD - so that a workgroup fits on a CU
// BESIDE System-1's 256-row row-chain workgroup (8 waves x 208 registers, 67 KiB of LDS): the weight-streaming decode chain and the MFMA-bound
// System-1 chain then share the CUs instead of taking turns on them. Same arithmetic and K order per output as the 8-wave builds with NW = 4.
int ina_launch_gemm_skinny_thin(const GemmArgs& p, hipStream_t stream) {
    const double osz = p.out_dtype == INA_DT_BF16 ? 2.0 : 4.0, asz = (p.norm_gamma && p.a_dtype == INA_DT_F32) ? 4.0 : 2.0;
    InaProfScope prof(INA_PROF_GEMM_SKINNY, 2.0 * p.M * p.N * p.K, asz * p.M * p.K + 2.0 * p.N * p.K + osz * p.M * (p.glu ? p.N / 2 : p.N), stream);
    if (p.force_cfg == 61) {      // the same 4-wave workgroups without the register cap (experiment: workgroup size alone)
        if (p.norm_gamma) {
            if (p.glu) return launch_skinny_prenorm<2, 4, 1, 2>(p, stream, (p.N + 31) / 32);
            return launch_skinny_prenorm<1, 4, 1, 2>(p, stream, (p.N + 15) / 16);
        }
        if (p.glu) hipLaunchKernelGGL((gemm_skinny_fused_kernel<1, 2, 4, 1, 2>), dim3((p.N + 31) / 32), dim3(256), 0, stream, p);
        else hipLaunchKernelGGL((gemm_skinny_fused_kernel<1, 1, 4, 1, 2>), dim3((p.N + 15) / 16), dim3(256), 0, stream, p);
        INA_HIP_CHECK(hipGetLastError());
        return 0;
    }
    if (p.norm_gamma) {
        if (p.glu) return launch_skinny_prenorm<2, 4, 1, 2, 5>(p, stream, (p.N + 31) / 32);
        return launch_skinny_prenorm<1, 4, 1, 2, 5>(p, stream, (p.N + 15) / 16);
    }
    if (p.glu) hipLaunchKernelGGL((gemm_skinny_fused_kernel<1, 2, 4, 1, 2, 5>), dim3((p.N + 31) / 32), dim3(256), 0, stream, p);
    else hipLaunchKernelGGL((gemm_skinny_fused_kernel<1, 1, 4, 1, 2, 5>), dim3((p.N + 15) / 16), dim3(256), 0, stream, p);
    INA_HIP_CHECK(hipGetLastError());
    return 0;
}

int ina_launch_gemm_skinny_prenorm(const GemmArgs& p, hipStream_t stream) {
    const double osz = p.out_dtype == INA_DT_BF16 ? 2.0 : 4.0, asz = p.a_dtype == INA_DT_F32 ? 4.0 : 2.0;
    InaProfScope prof(INA_PROF_GEMM_SKINNY, 2.0 * p.M * p.N * p.K, asz * p.M * p.K + 2.0 * p.N * p.K + osz * p.M * (p.glu ? p.N / 2 : p.N), stream);
    // every workgroup repeats the (cheap, L2-served) normalisation, so the column groups are packed NC per workgroup where one group
    // alone would be a 4-wave workgroup; the group width follows the fused kernel's rule (a few thousand waves per launch)
    if (p.glu) {
        const int tiles = (p.N + 31) / 32;
        if (tiles >= 1024) return launch_skinny_prenorm<2, 4, 2, 2>(p, stream, tiles);
        return launch_skinny_prenorm<2, 8, 1, 2>(p, stream, tiles);
    }
    const int tiles = (p.N + 15) / 16;
    if (tiles >= 1024) return launch_skinny_prenorm<1, 4, 2, 2>(p, stream, tiles);
    return launch_skinny_prenorm<1, 8, 1, 4>(p, stream, tiles);
}
