// Row-panel bf16 MFMA GEMM for the d = 384 System-1 heads (gfx950):  C[M,N] = epilogue( A[M,384] . W[N,384]^T ),  N % 128 == 0.
//
// The fused q|k|v|q2 and SwiGLU projections of a NextDiT block (65536 x 1536 / 2048 x 384) have six K tiles per output tile: a
// kernel that stages BOTH operands per output tile re-fetches the activations for every column tile and tops out at what the
// L2 -> LDS path delivers per fetched byte (128 x 256 tiles: 85 FLOP / B, 0.63-0.8 PF/s measured, the vendor library's level too).
// Here a workgroup owns a PANEL of rows and walks all of N:
//   * the panel never touches LDS: each wave keeps its 32 rows x 384 columns of A as MFMA operand fragments in registers (24 fragments
//     of 8 bf16 = 96 VGPRs), loaded once per workgroup;
//   * only W streams: 128 x 64 stages (16 KiB) through an NS-deep LDS ring filled by LDS-DMA (`global_load_lds_dwordx4`, swizzled on
//     the source side like gemm_glds.hip), every stage feeding 16 `mfma_f32_32x32x16_bf16` per wave. 128 (NW = 4) or 256 (NW = 8) FLOP
//     per fetched W byte, and the activations are read from HBM exactly once;
//   * operands are swapped (D = W-fragment x A-fragment), so a lane ends up with one output ROW and runs of 4 consecutive COLUMNS;
//     the two lanes that share a row (l, l + 32) trade halves with `v_permlane32_swap` and every lane stores whole 16-byte pieces
//     straight from registers - no LDS in the epilogue, the stores of column tile n retire under the MFMAs of tile n + 1.
// One `s_barrier` per stage; the ring is retired with COUNTED `s_waitcnt vmcnt` (the epilogue's stores sit in the same in-order
// counter: their number is added while they can still be younger than the stage waited for).
// LDS image of a stage: 128 rows (output columns n) x 128 bytes (64 k); physical 16-byte chunk cp of row r holds logical chunk
// cp ^ ((r >> 1) & 7): a 32x32x16 fragment read (`ds_read_b128`, lane -> row l & 31, k half l >> 5) then hits 16 different 16-byte
// slots of the 256-byte bank row in each of the four 16-lane service groups {0-3,12-15,20-27}, {4-11,16-19,28-31}, + 32
// (guides/MI355X_MICROARCH.md, LDS table) - checked by enumeration in tests/test_ops_host_cpu.py.
// K is accumulated in ascending order but by 16-wide MFMAs: results differ from the 16x16x32 tiles of gemm_glds.hip in the last bits.
#include "common.h"
#include "kernels.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void rp_glds16(const bf16* src, bf16* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)src,
                                     (void __attribute__((address_space(3)))*)lds_wave_base, 16, 0, 0);
}

template <int N>
__device__ __forceinline__ void rp_wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

__device__ __forceinline__ unsigned rp_pack2(float a, float b) {
    const bf16x2 v = {(bf16)a, (bf16)b};
    return __builtin_bit_cast(unsigned, v);
}

// lanes l and l + 32 hold the same output row: lane l columns c .. c+3 (lo pair) and c+8 .. c+11 (hi pair), lane l + 32 columns c+4 .. c+7
// and c+12 .. c+15. After the swap lane l holds c .. c+7 and lane l + 32 holds c+8 .. c+15 (one 16-byte store each).
__device__ __forceinline__ u32x4 rp_merge_rows(unsigned lo0, unsigned lo1, unsigned hi0, unsigned hi1) {
    // v_permlane32_swap x, y: x' = [x.lower | y.lower], y' = [x.upper | y.upper] (lower / upper = lanes 0-31 / 32-63)
    const u32x2 s0 = __builtin_amdgcn_permlane32_swap(lo0, hi0, false, false);
    const u32x2 s1 = __builtin_amdgcn_permlane32_swap(lo1, hi1, false, false);
    return u32x4{s0[0], s1[0], s0[1], s1[1]};
}

// (the ablation builds that priced one ingredient each - no DMA / no stores / no LDS reads / no barriers - and the s_setprio / 1 : 1 interleave
//  schedule variants of round 4 are in profiles/r04*_native_rowpanel*.log, not in the source any more)
// EPI: + bias[n] and an activation in the (non-GLU) epilogue - the biased projections of the nn.Transformer layers (NavDP decoder q|k|v, FFN).
template <int NW, int NS, bool GLU, bool EPI = false>
__global__ __launch_bounds__(NW * 64, 2) void gemm_bf16_rowpanel_kernel(GemmArgs p) {   // two waves per SIMD: one 8-wave or two 4-wave workgroups per CU
    constexpr int K = 384, KC = K / 64, BN = 128, STAGE = BN * 64;      // STAGE: elements of one ring slot (16 KiB)
    constexpr int INST = 16 / NW;                                       // 1 KiB DMA wave-instructions per wave and stage
    constexpr int NST = GLU ? 4 : 8;                                    // store instructions per wave and column tile
    static_assert(NS >= 3 && NS <= 8 && (NW == 4 || NW == 8), "ring / workgroup shape");
    extern __shared__ __attribute__((aligned(1024))) char smem_raw[];
    bf16* Ws = reinterpret_cast<bf16*>(smem_raw);                       // [NS][128][64]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m0 = blockIdx.x * (NW * 32) + wave * 32;
    const bool live = m0 < p.M;                                         // (M % 32 == 0: a wave is either all rows or none)
    const bf16* __restrict__ A = reinterpret_cast<const bf16*>(p.A);
    const bf16* __restrict__ W = reinterpret_cast<const bf16*>(p.W);
    const int ntiles = p.N / BN, total = ntiles * KC;
    const int lrow = lane & 31, khalf = lane >> 5;

    // ---- W stream: per-lane source pointers of this wave's DMA slots for (column tile 0, k chunk 0)
    const bf16* wsrc[INST];
    {
        const int drow = lane >> 3, dcp = lane & 7;
#pragma unroll
        for (int s = 0; s < INST; ++s) {
            const int r = (wave * INST + s) * 8 + drow;
            wsrc[s] = W + (size_t)r * p.ldw + ((dcp ^ ((r >> 1) & 7)) << 3);
        }
    }
    auto issue = [&](int st) {
        // stage st = (column tile st / KC, k chunk st % KC) into ring slot st % NS; past the end the last stage is fetched again into a slot
        // nobody reads any more, so every wave issues the same number of DMA instructions per iteration (constant wait counts)
        const int s2 = st < total ? st : total - 1;
        const int nt = s2 / KC, kc = s2 - nt * KC;
        bf16* dst = Ws + (st % NS) * STAGE + wave * INST * 512;
        const size_t off = (size_t)nt * BN * p.ldw + kc * 64;
#pragma unroll
        for (int s = 0; s < INST; ++s) rp_glds16(wsrc[s] + off, dst + s * 512);
    };
    float* Bs = reinterpret_cast<float*>(Ws + NS * STAGE);             // EPI: bias[N] copied once (read in every column tile's epilogue)
    if constexpr (EPI) {
        if (p.bias) {
            for (int i = tid; i < p.N; i += NW * 64) Bs[i] = p.bias[i];
        }
        __syncthreads();
    }
#pragma unroll
    for (int s = 0; s < NS - 1; ++s) issue(s);

    // ---- the row panel: 24 operand fragments (row lrow of this wave, k = i * 16 + khalf * 8 .. + 7)
    bf16x8 a[KC * 4];
    {
        const int row = min(m0 + lrow, p.M - 1);
        const bf16* ar = A + (size_t)row * p.lda + khalf * 8;
#pragma unroll
        for (int i = 0; i < KC * 4; ++i) a[i] = *reinterpret_cast<const bf16x8*>(ar + i * 16);
    }
    // fragment read offsets inside a stage (elements): column-tile row j * 32 + lrow, logical chunk kk * 2 + khalf
    int foff[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) foff[kk] = lrow * 64 + (((kk * 2 + khalf) ^ ((lrow >> 1) & 7)) << 3);   // (j * 32 keeps (r >> 1) & 7)

    bf16* __restrict__ Cb = reinterpret_cast<bf16*>(p.C);
    const size_t crow = (size_t)(m0 + lrow) * p.ldc;

    // LayerNorm statistics of every 384-wide segment of the rows this launch produces (plain epilogue only; ina_gemm_args.seg_stats): what the
    // attention stage of a NextDiT block needs for q1 / k1 / q2 (dit_attn_stats_kernel reads each projection row once instead of twice). Running
    // sum / sum of squares of this lane's 64 of a tile's 128 columns over the fp32 accumulators, closed every third tile - the row chain's form
    // (dit_rowchain.hip), so both producers hand the attention stage the same numbers.
    float seg_s = 0.f, seg_q = 0.f;
    int seg_t = 0;
    float* __restrict__ seg_out = (!GLU && !EPI && p.seg_stats && live) ? p.seg_stats + (size_t)(m0 + lrow) * (size_t)(p.N / K) * 2 : nullptr;

    int t = 0;
    for (int nt = 0; nt < ntiles; ++nt) {
        f32x16 acc[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
#pragma unroll
        for (int kc = 0; kc < KC; ++kc, ++t) {
            // stage t has landed for this wave's share: the NS - 2 newer stages (and, right behind an epilogue, its NST stores - same
            // in-order counter, issued after the DMA of stage t + NS - 2) may stay outstanding
            // (the epilogue of tile nt - j was issued behind the DMA of stage t iff kc <= NS + 4 - 6 j: j = 1 for every ring depth, j = 2 for NS = 8)
            if (live && nt > 1 && kc <= NS - 8) rp_wait_vm<(NS - 2) * INST + 2 * NST>();
            else if (live && nt > 0 && kc <= NS - 2) rp_wait_vm<(NS - 2) * INST + NST>();
            else rp_wait_vm<(NS - 2) * INST>();
            __builtin_amdgcn_s_barrier();   // stage t visible to every wave; every wave is done reading stage t - 1
            issue(t + NS - 1);              // ... whose slot is refilled
            const bf16* ws = Ws + (t % NS) * STAGE;
            // fragments of k step kk + 1 are requested before the MFMAs of step kk are issued (pinned: the compiler otherwise sinks every
            // read right in front of its MFMA and waits lgkmcnt(0) per pair)
            bf16x8 wf[2][4];
#pragma unroll
            for (int j = 0; j < 4; ++j) wf[0][j] = *reinterpret_cast<const bf16x8*>(ws + j * 32 * 64 + foff[0]);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                if (kk + 1 < 4) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) wf[(kk + 1) & 1][j] = *reinterpret_cast<const bf16x8*>(ws + j * 32 * 64 + foff[kk + 1]);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[kk & 1][j], a[kc * 4 + kk], acc[j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's reads of the slot have returned before it reaches the next barrier
        }
        // ---- epilogue of column tile nt: lane = output row m0 + lrow; acc[j][i * 4 + e] = column nt * 128 + j * 32 + i * 8 + khalf * 4 + e
        if constexpr (!GLU && !EPI) {
            if (seg_out) {
                float ts[4] = {0.f, 0.f, 0.f, 0.f}, tq[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        ts[j] += acc[j][r];
                        tq[j] = fmaf(acc[j][r], acc[j][r], tq[j]);
                    }
                seg_s += (ts[0] + ts[1]) + (ts[2] + ts[3]);
                seg_q += (tq[0] + tq[1]) + (tq[2] + tq[3]);
                asm volatile("" : "+v"(seg_s), "+v"(seg_q));
                if (++seg_t == 3) {
                    seg_s += __shfl_xor(seg_s, 32);
                    seg_q += __shfl_xor(seg_q, 32);
                    const float mean = seg_s * (1.0f / K);
                    const float var = fmaxf(seg_q * (1.0f / K) - mean * mean, 0.f);
                    // both column halves of a row hold the same pair and store it to the same place (no exec-mask branch)
                    *reinterpret_cast<f32x2*>(seg_out + (nt / 3) * 2) = f32x2{mean, rsqrtf(var + p.seg_eps)};
                    seg_s = seg_q = 0.f;
                    seg_t = 0;
                }
            }
        }
        if (live) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if constexpr (GLU) {
                    // 32 W rows = [gate16 | up16]: i = 0, 1 hold the gates of output columns j * 16 + i * 8 + khalf * 4 + e, i = 2, 3 their ups
                    unsigned q[2][2];
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        float v[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = ina_silu(acc[j][i * 4 + e]) * acc[j][(i + 2) * 4 + e];
                        q[i][0] = rp_pack2(v[0], v[1]);
                        q[i][1] = rp_pack2(v[2], v[3]);
                    }
                    const u32x4 o = rp_merge_rows(q[0][0], q[0][1], q[1][0], q[1][1]);
                    *reinterpret_cast<u32x4*>(Cb + crow + nt * 64 + j * 16 + khalf * 8) = o;
                } else {
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        unsigned q[2][2];
#pragma unroll
                        for (int i = 0; i < 2; ++i) {
                            float v[4];
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = acc[j][(2 * h + i) * 4 + e];
                            if constexpr (EPI) {
                                if (p.bias) {     // (from the LDS copy: a global load here would make the compiler drain the DMA ring with vmcnt(0))
                                    const f32x4 bb = *reinterpret_cast<const f32x4*>(Bs + nt * 128 + j * 32 + (2 * h + i) * 8 + khalf * 4);
#pragma unroll
                                    for (int e = 0; e < 4; ++e) v[e] += bb[e];
                                }
#pragma unroll
                                for (int e = 0; e < 4; ++e) v[e] = ina_act(v[e], p.act);
                            }
                            q[i][0] = rp_pack2(v[0], v[1]);
                            q[i][1] = rp_pack2(v[2], v[3]);
                        }
                        const u32x4 o = rp_merge_rows(q[0][0], q[0][1], q[1][0], q[1][1]);
                        *reinterpret_cast<u32x4*>(Cb + crow + nt * 128 + j * 32 + h * 16 + khalf * 8) = o;
                    }
                }
            }
        }
    }
    rp_wait_vm<0>();      // the dummy tail stages land in LDS: nothing of this workgroup is in flight when it retires
}

template <int NW, int NS>
int launch_rowpanel(const GemmArgs& p, hipStream_t stream) {
    const size_t LDS_BYTES = size_t(NS) * 128 * 64 * sizeof(bf16) + (p.bias ? size_t(p.N) * 4 : 0);
    static bool attr_done[3] = {false, false, false};
    const bool epi = !p.glu && (p.bias || p.act != INA_ACT_NONE);
    const int which = p.glu ? 1 : (epi ? 2 : 0);
    auto kern = p.glu ? gemm_bf16_rowpanel_kernel<NW, NS, true> : (epi ? gemm_bf16_rowpanel_kernel<NW, NS, false, true> : gemm_bf16_rowpanel_kernel<NW, NS, false>);
    if (!attr_done[which]) {
        INA_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(size_t(NS) * 128 * 64 * sizeof(bf16) + 4096 * 4)));
        attr_done[which] = true;
    }
    const int wgs = (p.M + NW * 32 - 1) / (NW * 32);
    InaProfScope prof(INA_PROF_GEMM, 2.0 * p.M * p.N * p.K, 2.0 * p.M * p.K + 2.0 * p.N * p.K + 2.0 * p.M * (p.glu ? p.N / 2 : p.N), stream);
    hipLaunchKernelGGL(kern, dim3(wgs), dim3(NW * 64), LDS_BYTES, stream, p);
    INA_HIP_CHECK(hipGetLastError());
    return 0;
}

}  // namespace

bool ina_gemm_rowpanel_contract(const GemmArgs& p) {
    // what the kernel computes: bf16 out = act(A . W^T + bias), or SiLU-GLU over interleaved [gate16 | up16] rows without a bias; no scales, no residual
    if (p.K != 384 || p.N % 128 != 0 || p.M % 32 != 0 || p.M < 32) return false;
    if (p.colscale || p.rowscale || p.R || p.norm_gamma) return false;
    if (p.out_dtype != INA_DT_BF16 || p.batch > 1) return false;
    if (p.glu && (p.act != INA_ACT_SILU || p.bias)) return false;
    if (p.bias && (((uintptr_t)p.bias % 16) || p.N > 4096)) return false;
    if (p.act < INA_ACT_NONE || p.act > INA_ACT_TANH) return false;
    if (p.seg_stats && (p.glu || p.bias || p.act != INA_ACT_NONE || p.N % 384 != 0 || ((uintptr_t)p.seg_stats % 8))) return false;
    if (p.lda % 8 || p.ldw % 8 || p.ldc % 8 || ((uintptr_t)p.C % 16) || ((uintptr_t)p.A % 16) || ((uintptr_t)p.W % 16)) return false;
    return true;
}

// cfg 34: 8 waves = 256-row panels, one workgroup per CU (256 FLOP per fetched W byte); cfg 35: 4 waves = 128-row panels, two per CU
int ina_launch_gemm_rowpanel(const GemmArgs& p, hipStream_t stream, int cfg) {
    INA_REQUIRE(ina_gemm_rowpanel_contract(p), "gemm(row panel): needs K = 384, N %% 128 == 0, M %% 32 == 0, bf16 output, no scales / residual, "
                "bias + activation or SiLU-GLU (M=%d N=%d K=%d)", p.M, p.N, p.K);
    ina_prof_set_sub(cfg);
    switch (cfg) {
        case 34: return launch_rowpanel<8, 4>(p, stream);
        case 35: return launch_rowpanel<4, 4>(p, stream);
        default: ina_set_error("gemm(row panel): unknown config %d", cfg); return -2;
    }
}
