// Row-block GEMM with the NextDiT "gated norm + residual + next pre-norm" epilogue for gfx950 (N = 384 = NextDiTCrossAttnConfig.dim):
//
//     P = A[M,K] . W[384,K]^T                                (attn2.to_out / feed_forward.linear_2)
//     X += tanh(gate[b]) * rmsnorm(P) * gamma                 (LuminaNextDiTBlock: norm2 / ffn_norm2 + gate + residual, fp32 stream)
//     H  = rmsnorm(X) * gamma2 * (1 + mod_scale2[b])          (ffn_norm1 / next block's norm1 + adaLN scale, bf16 GEMM operand)
//
// with b = row / mod_div. Unfused this is a GEMM writing a bf16 [M,384] projection plus a norm launch reading it back (and the fp32
// residual twice); here a workgroup owns 128 FULL rows (tile 128 x 384, 8 waves as 2 x 4, wave tile 64 x 96), so the row statistics
// are available in the epilogue (lane partials -> 2 shuffles -> 4 wave partials through LDS) and the projection never leaves the
// chip. Main loop = the LDS-DMA pipeline of gemm_glds.hip (swizzled 128-byte LDS rows, 2 stages, swapped-operand MFMA).
// Reference: diffusers LuminaNextDiTBlock.forward (diffusers==0.33.1) as instantiated by nextdit_traj.py:121-188.
#include "common.h"
#include "kernels.h"

namespace {

constexpr int RN_BM = 128, RN_BN = 384, RN_WN = 4, RN_NW = 8, RN_TM = 64, RN_TN = 96, RN_FM = 4, RN_FN = 6;
constexpr int RN_A_INST = RN_BM / 8 / RN_NW, RN_B_INST = RN_BN / 8 / RN_NW;           // 1 KiB DMA wave-instructions per wave per stage
constexpr size_t RN_STAGE = size_t(RN_BM + RN_BN) * 64 * sizeof(bf16);               // 64 KiB
constexpr size_t RN_LDS = 2 * RN_STAGE;
constexpr int RN_HLD = RN_BN + 8;                                                    // bf16 staging row of the H tile

__device__ __forceinline__ void rn_glds16(const bf16* src, bf16* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)src,
                                     (void __attribute__((address_space(3)))*)lds_wave_base, 16, 0, 0);
}

__global__ __launch_bounds__(RN_NW * 64) void gemm_rownorm_kernel(GemmRownormArgs p) {
    extern __shared__ __attribute__((aligned(1024))) char smem_raw[];
    bf16* As = reinterpret_cast<bf16*>(smem_raw);          // [2][128][64]
    bf16* Bs = As + 2 * RN_BM * 64;                        // [2][384][64]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / RN_WN, wn = wave % RN_WN;
    const int m0 = blockIdx.x * RN_BM;
    const bf16* __restrict__ A = reinterpret_cast<const bf16*>(p.A);
    const bf16* __restrict__ W = reinterpret_cast<const bf16*>(p.W);

    const bf16* asrc[RN_A_INST];
    const bf16* bsrc[RN_B_INST];
    const int drow = lane >> 3, dcp = lane & 7;
#pragma unroll
    for (int s = 0; s < RN_A_INST; ++s) {
        const int row = (wave * RN_A_INST + s) * 8 + drow;
        asrc[s] = A + (size_t)min(m0 + row, p.M - 1) * p.lda + ((dcp ^ (row & 7)) << 3);
    }
#pragma unroll
    for (int s = 0; s < RN_B_INST; ++s) {
        const int row = (wave * RN_B_INST + s) * 8 + drow;
        bsrc[s] = W + (size_t)row * p.ldw + ((dcp ^ (row & 7)) << 3);
    }
    auto stage = [&](int buf, int k0) {
        bf16* as = As + buf * RN_BM * 64 + wave * RN_A_INST * 512;
        bf16* bs = Bs + buf * RN_BN * 64 + wave * RN_B_INST * 512;
#pragma unroll
        for (int s = 0; s < RN_A_INST; ++s) rn_glds16(asrc[s] + k0, as + s * 512);
#pragma unroll
        for (int s = 0; s < RN_B_INST; ++s) rn_glds16(bsrc[s] + k0, bs + s * 512);
    };

    f32x4 acc[RN_FM][RN_FN];
#pragma unroll
    for (int i = 0; i < RN_FM; ++i)
#pragma unroll
        for (int j = 0; j < RN_FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = p.K / 64;
    const int frow = lane & 15, g = lane >> 4, sw = lane & 7;
    stage(0, 0);
    int buf = 0;
    for (int t = 0; t < nk; ++t) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                        // tile t visible; everyone is done with the other buffer
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (t + 1 < nk) stage(buf ^ 1, (t + 1) * 64);
        const bf16* as = As + buf * RN_BM * 64 + (wm * RN_TM + frow) * 64;
        const bf16* bs = Bs + buf * RN_BN * 64 + (wn * RN_TN + frow) * 64;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int chunk = ((kk * 4 + g) ^ sw) << 3;
            bf16x8 fa[RN_FM], fb[RN_FN];
#pragma unroll
            for (int i = 0; i < RN_FM; ++i) fa[i] = *reinterpret_cast<const bf16x8*>(as + i * 16 * 64 + chunk);
#pragma unroll
            for (int j = 0; j < RN_FN; ++j) fb[j] = *reinterpret_cast<const bf16x8*>(bs + j * 16 * 64 + chunk);
#pragma unroll
            for (int i = 0; i < RN_FM; ++i)
#pragma unroll
                for (int j = 0; j < RN_FN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
        }
        buf ^= 1;
    }
    __syncthreads();   // stage buffers are free: LDS becomes [2][4][128] f32 row partials + the bf16 H tile

    // lane holds rows m = wm*64 + i*16 + frow, columns n = wn*96 + j*16 + g*4 + {0..3}
    float* ssq = reinterpret_cast<float*>(smem_raw);                 // [2 passes][4 wave columns][128 rows]
    bf16* hs = reinterpret_cast<bf16*>(smem_raw + 4096);             // [128][RN_HLD]
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
#pragma unroll
    for (int i = 0; i < RN_FM; ++i) {
        const int row = wm * RN_TM + i * 16 + frow, m = m0 + row;
        const bool live = m < p.M;
        const float tot = ssq[row] + ssq[RN_BM + row] + ssq[2 * RN_BM + row] + ssq[3 * RN_BM + row];
        const float rstd = rsqrtf(tot * invN + p.eps);
        const float* gt = p.gate ? p.gate + (size_t)((live ? m : 0) / p.mod_div) * p.mod_ld : nullptr;
        float* xrow = p.X + (size_t)(live ? m : 0) * p.ldx;
        float s2 = 0.f;
#pragma unroll
        for (int j = 0; j < RN_FN; ++j) {
            const int n = wn * RN_TN + j * 16 + g * 4;
            const f32x4 gm = *reinterpret_cast<const f32x4*>(p.gamma + n);
            f32x4 t;
#pragma unroll
            for (int r = 0; r < 4; ++r) t[r] = acc[i][j][r] * rstd * gm[r];
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
            acc[i][j] = t;
#pragma unroll
            for (int r = 0; r < 4; ++r) s2 += t[r] * t[r];
        }
        s2 += __shfl_xor(s2, 16);
        s2 += __shfl_xor(s2, 32);
        if (g == 0) ssq[512 + wn * RN_BM + row] = s2;
    }
    if (!p.H) return;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < RN_FM; ++i) {
        const int row = wm * RN_TM + i * 16 + frow, m = m0 + row;
        const float tot = ssq[512 + row] + ssq[512 + RN_BM + row] + ssq[512 + 2 * RN_BM + row] + ssq[512 + 3 * RN_BM + row];
        const float rstd = rsqrtf(tot * invN + p.eps);
        const float* ms = p.mod_scale2 ? p.mod_scale2 + (size_t)((m < p.M ? m : 0) / p.mod_div) * p.mod_ld : nullptr;
#pragma unroll
        for (int j = 0; j < RN_FN; ++j) {
            const int n = wn * RN_TN + j * 16 + g * 4;
            f32x4 t;
#pragma unroll
            for (int r = 0; r < 4; ++r) t[r] = acc[i][j][r] * rstd;
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
            *reinterpret_cast<bf16x4*>(&hs[row * RN_HLD + n]) = bf16x4{(bf16)t[0], (bf16)t[1], (bf16)t[2], (bf16)t[3]};
        }
    }
    __syncthreads();
    // H tile out: whole 768-byte rows, 16 bytes per lane
    bf16* __restrict__ H = reinterpret_cast<bf16*>(p.H);
    constexpr int CPR = RN_BN / 8;   // 48 chunks per row
    for (int c = tid; c < RN_BM * CPR; c += RN_NW * 64) {
        const int row = c / CPR, cc = c % CPR;
        if (m0 + row < p.M) *reinterpret_cast<bf16x8*>(H + (size_t)(m0 + row) * p.ldh + cc * 8) = *reinterpret_cast<const bf16x8*>(&hs[row * RN_HLD + cc * 8]);
    }
}

}  // namespace

int ina_launch_gemm_rownorm(const GemmRownormArgs& p_in, hipStream_t stream) {
    GemmRownormArgs p = p_in;
    if (p.mod_div <= 0) p.mod_div = 1;
    INA_REQUIRE(p.A && p.W && p.X && p.gamma, "gemm_rownorm: A, W, X and gamma are required");
    INA_REQUIRE(p.N == RN_BN, "gemm_rownorm: N=%d (the row-block kernel is built for N = 384)", p.N);
    INA_REQUIRE(p.M > 0 && p.K > 0 && p.K % 64 == 0, "gemm_rownorm: M=%d, K=%d (K must be a multiple of 64)", p.M, p.K);
    INA_REQUIRE(p.lda % 8 == 0 && p.ldw % 8 == 0 && p.ldx % 4 == 0 && (!p.H || p.ldh % 8 == 0), "gemm_rownorm: row strides must keep 16-byte alignment");
    INA_REQUIRE(((uintptr_t)p.A % 16) == 0 && ((uintptr_t)p.W % 16) == 0 && ((uintptr_t)p.X % 16) == 0 && ((uintptr_t)p.H % 16) == 0, "gemm_rownorm: misaligned pointer");
    INA_REQUIRE((!p.gate && !p.mod_scale2) || (p.mod_ld > 0 && p.mod_ld % 4 == 0), "gemm_rownorm: modulation needs mod_ld (multiple of 4)");
    static bool attr_done = false;
    if (!attr_done) {
        INA_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_rownorm_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)RN_LDS));
        attr_done = true;
    }
    InaProfScope prof(INA_PROF_GEMM, 2.0 * p.M * RN_BN * p.K, 2.0 * p.M * p.K + 2.0 * RN_BN * p.K + p.M * RN_BN * (8.0 + (p.H ? 2.0 : 0.0)), stream);
    hipLaunchKernelGGL(gemm_rownorm_kernel, dim3((p.M + RN_BM - 1) / RN_BM), dim3(RN_NW * 64), RN_LDS, stream, p);
    INA_HIP_CHECK(hipGetLastError());
    return 0;
}
