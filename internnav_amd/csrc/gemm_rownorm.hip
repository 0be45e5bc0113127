// Row-block GEMM with the NextDiT "gated norm + residual + next pre-norm" epilogue for gfx950 (N = 384 = NextDiTCrossAttnConfig.dim):
//
//     P = A[M,K] . W[384,K]^T                                (attn2.to_out / feed_forward.linear_2)
//     X += tanh(gate[b]) * rmsnorm(P) * gamma                 (LuminaNextDiTBlock: norm2 / ffn_norm2 + gate + residual, fp32 stream)
//     H  = rmsnorm(X) * gamma2 * (1 + mod_scale2[b])          (ffn_norm1 / next block's norm1 + adaLN scale, bf16 GEMM operand)
//
// with b = row / mod_div. Unfused this is a GEMM writing a bf16 [M,384] projection plus a norm launch reading it back (and the fp32
// residual twice); here a workgroup owns 64 FULL rows (tile 64 x 384, 8 waves as 2 x 4, wave tile 32 x 96), so the row statistics
// are available in the epilogue (lane partials -> 2 shuffles -> 4 wave partials through LDS) and the projection never leaves the
// chip. Main loop = the LDS-DMA pipeline of gemm_glds.hip with 32-wide K stages (swizzled 64-byte LDS rows, 2 stages, swapped MFMA).
// Reference: diffusers LuminaNextDiTBlock.forward (diffusers==0.33.1) as instantiated by nextdit_traj.py:121-188.
#include "common.h"
#include "kernels.h"

namespace {

// 64-row tiles with 32-wide K stages: 2 x 28 KiB of LDS, so two workgroups share a CU and the HBM-bound epilogue of one (fp32 residual
// read + write) overlaps the main loop of the other (the first version, 128 x 384 tiles with 128 KiB of LDS, ran one workgroup per CU
// and was no faster than the unfused pair). 8 waves as 2 x 4, wave tile 32 x 96.
constexpr int RN_BM = 64, RN_BN = 384, RN_BK = 32, RN_WN = 4, RN_NW = 8, RN_TM = 32, RN_TN = 96, RN_FM = 2, RN_FN = 6;
constexpr int RN_SLOTS = (RN_BM + RN_BN) / 16;                                       // 1 KiB DMA wave-instructions (16 rows x 64 B) per stage: 28
constexpr int RN_SPW = (RN_SLOTS + RN_NW - 1) / RN_NW;                               // per wave: 4 (waves 0-3) or 3
constexpr size_t RN_STAGE = size_t(RN_BM + RN_BN) * RN_BK * sizeof(bf16);            // 28 KiB
constexpr size_t RN_LDS = 2 * RN_STAGE;
constexpr int RN_SLD = RN_BN + 4;                                                    // fp32 row of the 32-row epilogue slab
static_assert(size_t(4 * RN_BM + RN_TM * RN_SLD) * sizeof(float) <= RN_LDS, "epilogue scratch must fit in the stage buffers");

__device__ __forceinline__ void rn_glds16(const bf16* src, bf16* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)src,
                                     (void __attribute__((address_space(3)))*)lds_wave_base, 16, 0, 0);
}

// LDS rows are 64 bytes (4 chunks of 16 B). Physical chunk cp of row r holds logical chunk cp ^ sw4(r) with sw4(r) = (-(r >> 2)) & 3:
// under the 4 x 16-lane service groups of ds_read_b128 ({0-3,12-15,20-27}, ...) every group then covers all 16 bank slots.
__device__ __forceinline__ int rn_sw4(int r) { return (0 - (r >> 2)) & 3; }

__global__ __launch_bounds__(RN_NW * 64) void gemm_rownorm_kernel(GemmRownormArgs p) {
    extern __shared__ __attribute__((aligned(1024))) char smem_raw[];
    bf16* As = reinterpret_cast<bf16*>(smem_raw);          // [2][64][32]
    bf16* Bs = As + 2 * RN_BM * RN_BK;                     // [2][384][32]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / RN_WN, wn = wave % RN_WN;
    const int m0 = blockIdx.x * RN_BM;
    const bf16* __restrict__ A = reinterpret_cast<const bf16*>(p.A);
    const bf16* __restrict__ W = reinterpret_cast<const bf16*>(p.W);

    // DMA slots of this wave: slot s covers tile rows [16 s, 16 s + 16) of the stacked [A rows | W rows] stage image
    const bf16* src[RN_SPW];
    int dst[RN_SPW];                                        // element offset inside one stage of As / Bs (negative: no slot)
    const int drow = lane >> 2, dcp = lane & 3;
#pragma unroll
    for (int j = 0; j < RN_SPW; ++j) {
        const int slot = wave + RN_NW * j;
        const int row = slot * 16 + drow;                   // row in the stacked image
        const int lc = (dcp ^ rn_sw4(drow)) << 3;           // (slot * 16 is a multiple of 16: the swizzle only sees drow)
        if (slot >= RN_SLOTS) { src[j] = A; dst[j] = -1; }
        else if (slot < RN_BM / 16) { src[j] = A + (size_t)min(m0 + row, p.M - 1) * p.lda + lc; dst[j] = slot * 512; }
        else { src[j] = W + (size_t)(row - RN_BM) * p.ldw + lc; dst[j] = (slot - RN_BM / 16) * 512; }
    }
    auto stage = [&](int buf, int k0) {
#pragma unroll
        for (int j = 0; j < RN_SPW; ++j) {
            const int slot = wave + RN_NW * j;              // wave-uniform
            if (slot < RN_BM / 16) rn_glds16(src[j] + k0, As + buf * RN_BM * RN_BK + dst[j]);
            else if (slot < RN_SLOTS) rn_glds16(src[j] + k0, Bs + buf * RN_BN * RN_BK + dst[j]);
        }
    };

    f32x4 acc[RN_FM][RN_FN];
#pragma unroll
    for (int i = 0; i < RN_FM; ++i)
#pragma unroll
        for (int j = 0; j < RN_FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = p.K / RN_BK;
    const int frow = lane & 15, g = lane >> 4;
    const int chunk = (g ^ rn_sw4(frow)) << 3;              // this lane's 16-byte piece of a fragment row
    stage(0, 0);
    int buf = 0;
    for (int t = 0; t < nk; ++t) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                        // tile t visible; everyone is done with the other buffer
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (t + 1 < nk) stage(buf ^ 1, (t + 1) * RN_BK);
        const bf16* as = As + buf * RN_BM * RN_BK + (wm * RN_TM + frow) * RN_BK + chunk;
        const bf16* bs = Bs + buf * RN_BN * RN_BK + (wn * RN_TN + frow) * RN_BK + chunk;
        bf16x8 fa[RN_FM], fb[RN_FN];
#pragma unroll
        for (int i = 0; i < RN_FM; ++i) fa[i] = *reinterpret_cast<const bf16x8*>(as + i * 16 * RN_BK);
#pragma unroll
        for (int j = 0; j < RN_FN; ++j) fb[j] = *reinterpret_cast<const bf16x8*>(bs + j * 16 * RN_BK);
#pragma unroll
        for (int i = 0; i < RN_FM; ++i)
#pragma unroll
            for (int j = 0; j < RN_FN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
        buf ^= 1;
    }
    __syncthreads();   // stage buffers are free: LDS becomes [4][64] f32 row partials + one 32-row fp32 slab of the projection

    // lane holds rows m = wm*32 + i*16 + frow, columns n = wn*96 + j*16 + g*4 + {0..3}
    float* ssq = reinterpret_cast<float*>(smem_raw);                 // [4 wave columns][64 rows]
    float* slab = ssq + 4 * RN_BM;                                   // [32][RN_SLD]
    const float invN = 1.0f / RN_BN;
#pragma unroll
    for (int i = 0; i < RN_FM; ++i) {
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < RN_FN; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) s += acc[i][j][r] * acc[i][j][r];
        s += __shfl_xor(s, 16);
        s += __shfl_xor(s, 32);
        if (g == 0) ssq[wn * RN_BM + wm * RN_TM + i * 16 + frow] = s;
    }
    __syncthreads();
    float rstd1[RN_FM];
#pragma unroll
    for (int i = 0; i < RN_FM; ++i) {
        const int row = wm * RN_TM + i * 16 + frow;
        rstd1[i] = rsqrtf((ssq[row] + ssq[RN_BM + row] + ssq[2 * RN_BM + row] + ssq[3 * RN_BM + row]) * invN + p.eps);
    }
    // The residual update is the HBM-bound part: it runs in a ROW layout (16 lanes x 24 columns per row: 256 contiguous bytes per
    // 16 lanes for the fp32 read / write, 128 for the bf16 H) instead of the MFMA fragment layout (64-byte pieces per row). The two
    // wave rows hand their 32-row slab of normalised projections over through LDS one after the other.
    const int rl = tid >> 4, l16 = tid & 15;                        // row stage: thread -> (slab row, 16-lane column group)
#pragma unroll 1
    for (int half = 0; half < 2; ++half) {
        if (wm == half) {
#pragma unroll
            for (int i = 0; i < RN_FM; ++i)
#pragma unroll
                for (int j = 0; j < RN_FN; ++j) {
                    f32x4 t;
#pragma unroll
                    for (int r = 0; r < 4; ++r) t[r] = acc[i][j][r] * rstd1[i];
                    *reinterpret_cast<f32x4*>(&slab[(i * 16 + frow) * RN_SLD + wn * RN_TN + j * 16 + g * 4]) = t;
                }
        }
        __syncthreads();
        const int m = m0 + half * RN_TM + rl;
        const bool live = m < p.M;
        const int mb = (live ? m : 0) / p.mod_div;
        const float* gt = p.gate ? p.gate + (size_t)mb * p.mod_ld : nullptr;
        float* xrow = p.X + (size_t)(live ? m : 0) * p.ldx;
        f32x4 v[RN_BN / 64];
        float s2 = 0.f;
#pragma unroll
        for (int c = 0; c < RN_BN / 64; ++c) {
            const int n = c * 64 + l16 * 4;
            f32x4 t = *reinterpret_cast<const f32x4*>(&slab[rl * RN_SLD + n]);
            const f32x4 gm = *reinterpret_cast<const f32x4*>(p.gamma + n);
#pragma unroll
            for (int r = 0; r < 4; ++r) t[r] *= gm[r];
            if (gt) {
                const f32x4 gv = *reinterpret_cast<const f32x4*>(gt + n);
#pragma unroll
                for (int r = 0; r < 4; ++r) t[r] *= tanhf(gv[r]);
            }
            if (live) {
                const f32x4 xv = *reinterpret_cast<const f32x4*>(xrow + n);
#pragma unroll
                for (int r = 0; r < 4; ++r) t[r] += xv[r];
                *reinterpret_cast<f32x4*>(xrow + n) = t;
            }
            v[c] = t;
#pragma unroll
            for (int r = 0; r < 4; ++r) s2 += t[r] * t[r];
        }
        if (p.H) {
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) s2 += __shfl_xor(s2, o);
            const float rstd2 = rsqrtf(s2 * invN + p.eps);
            const float* ms = p.mod_scale2 ? p.mod_scale2 + (size_t)mb * p.mod_ld : nullptr;
            bf16* hrow = reinterpret_cast<bf16*>(p.H) + (size_t)(live ? m : 0) * p.ldh;
#pragma unroll
            for (int c = 0; c < RN_BN / 64; ++c) {
                const int n = c * 64 + l16 * 4;
                f32x4 t;
#pragma unroll
                for (int r = 0; r < 4; ++r) t[r] = v[c][r] * rstd2;
                if (p.gamma2) {
                    const f32x4 gm = *reinterpret_cast<const f32x4*>(p.gamma2 + n);
#pragma unroll
                    for (int r = 0; r < 4; ++r) t[r] *= gm[r];
                }
                if (ms) {
                    const f32x4 mv = *reinterpret_cast<const f32x4*>(ms + n);
#pragma unroll
                    for (int r = 0; r < 4; ++r) t[r] *= 1.0f + mv[r];
                }
                if (live) *reinterpret_cast<bf16x4*>(hrow + n) = bf16x4{(bf16)t[0], (bf16)t[1], (bf16)t[2], (bf16)t[3]};
            }
        }
        __syncthreads();   // the slab is rewritten by the other wave row
    }
}

}  // namespace

int ina_launch_gemm_rownorm(const GemmRownormArgs& p_in, hipStream_t stream) {
    GemmRownormArgs p = p_in;
    if (p.mod_div <= 0) p.mod_div = 1;
    INA_REQUIRE(p.A && p.W && p.X && p.gamma, "gemm_rownorm: A, W, X and gamma are required");
    INA_REQUIRE(p.N == RN_BN, "gemm_rownorm: N=%d (the row-block kernel is built for N = 384)", p.N);
    INA_REQUIRE(p.M > 0 && p.K > 0 && p.K % 32 == 0, "gemm_rownorm: M=%d, K=%d (K must be a multiple of 32)", p.M, p.K);
    INA_REQUIRE(p.lda % 8 == 0 && p.ldw % 8 == 0 && p.ldx % 4 == 0 && (!p.H || p.ldh % 8 == 0), "gemm_rownorm: row strides must keep 16-byte alignment");
    INA_REQUIRE(((uintptr_t)p.A % 16) == 0 && ((uintptr_t)p.W % 16) == 0 && ((uintptr_t)p.X % 16) == 0 && ((uintptr_t)p.H % 16) == 0, "gemm_rownorm: misaligned pointer");
    INA_REQUIRE((!p.gate && !p.mod_scale2) || (p.mod_ld > 0 && p.mod_ld % 4 == 0), "gemm_rownorm: modulation needs mod_ld (multiple of 4)");
    static bool attr_done = false;
    if (!attr_done) {
        INA_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_rownorm_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)RN_LDS));
        attr_done = true;
    }
    ina_prof_set_sub(41);
    InaProfScope prof(INA_PROF_GEMM, 2.0 * p.M * RN_BN * p.K, 2.0 * p.M * p.K + 2.0 * RN_BN * p.K + p.M * RN_BN * (8.0 + (p.H ? 2.0 : 0.0)), stream);
    hipLaunchKernelGGL(gemm_rownorm_kernel, dim3((p.M + RN_BM - 1) / RN_BM), dim3(RN_NW * 64), RN_LDS, stream, p);
    INA_HIP_CHECK(hipGetLastError());
    return 0;
}
