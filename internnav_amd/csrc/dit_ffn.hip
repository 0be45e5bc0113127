// Fused SwiGLU feed-forward of a NextDiT block for gfx950 (dim 384 = NextDiTCrossAttnConfig.dim):
//
//     F  = silu(H . W1^T) * (H . W3^T)                         LuminaFeedForward: linear_1 / linear_3 (stored interleaved, see W13)
//     P  = F . W2^T                                            linear_2
//     X += tanh(gate[b]) * rmsnorm(P) * gamma                  ffn_norm2 + gate_mlp + residual (fp32 stream)
//     Hn = rmsnorm(X) * gamma2 * (1 + mod_scale2[b])           the NEXT block's norm1 + adaLN scale (bf16 GEMM operand), optional
//
// Unfused this is three launches (GLU GEMM writing F, GEMM reading it back, norm kernel) that move 670 MB per block at 65 536 rows and
// run at <50 % of either roofline (short K = 384 GEMMs: prologue / epilogue bound). Here a workgroup owns 64 FULL rows:
//   * H tile [64 x 384] is DMA'd into LDS once (12 k-slices of 64-byte swizzled rows, 48 KiB) and stays for the whole kernel;
//   * the F dimension is walked in chunks of 128 columns: 12 k-steps of GEMM-1 (B = 256 interleaved gate/up rows of W13) produce a
//     64 x 128 chunk of F in registers -> silu(gate)*up -> bf16 -> LDS (16 KiB) -> 4 k-steps of GEMM-2 (B = 384 rows of W2) accumulate
//     P [64 x 384] in registers (never leaves the chip); F never exists in HBM;
//   * ONE LDS-DMA ring (3 stages of 24 KiB, `global_load_lds`, counted vmcnt, one s_barrier per k-step) streams the B operands of
//     both GEMMs as a single sequence of 128 (F = 1024) k-steps, so the prefetch runs across every GEMM-1 / GEMM-2 / chunk boundary;
//   * the fp32 residual rows are prefetched into registers under the last k-steps; the epilogue (row statistics through LDS, row-
//     layout residual update, next pre-norm) is the one of gemm_rownorm.hip.
// 8 waves as 2 x 4: GEMM-1 wave tile 32 x 64 (interleaved columns = 32 F columns = exactly one k-slice of the F chunk), GEMM-2
// wave tile 32 x 96. MFMA: swapped-operand v_mfma_f32_16x16x32_bf16 as everywhere in this tree.
// Reference: diffusers LuminaFeedForward / LuminaNextDiTBlock.forward (diffusers==0.33.1) as wired by nextdit_traj.py:121-188.
#include "common.h"
#include "kernels.h"

namespace {

constexpr int FF_BM = 64, FF_D = 384, FF_BK = 32, FF_NW = 8, FF_WN = 4, FF_TM = 32;
constexpr int FF_CH = 128;                                   // F columns per chunk (256 interleaved W13 rows)
constexpr int FF_K1 = FF_D / FF_BK;                          // 12 k-steps of GEMM-1 per chunk
constexpr int FF_K2 = FF_CH / FF_BK;                         // 4 k-steps of GEMM-2 per chunk
constexpr int FF_NS = 3;                                     // B ring stages
constexpr int FF_STAGE_ROWS = FF_D;                          // a stage holds up to 384 rows x 32 k (GEMM-2); GEMM-1 uses 256 of them
constexpr size_t FF_HS = size_t(FF_K1) * FF_BM * FF_BK * sizeof(bf16);          // 48 KiB
constexpr size_t FF_BS = size_t(FF_NS) * FF_STAGE_ROWS * FF_BK * sizeof(bf16);  // 72 KiB
constexpr size_t FF_FS = size_t(FF_K2) * FF_BM * FF_BK * sizeof(bf16);          // 16 KiB
constexpr size_t FF_LDS = FF_HS + FF_BS + FF_FS;                                // 136 KiB
constexpr int FF_SLD = FF_D + 4;                             // fp32 row of the epilogue slab
static_assert(size_t(4 * FF_BM + FF_TM * FF_SLD) * sizeof(float) <= FF_HS + FF_BS, "epilogue scratch must fit in the tile buffers");

__device__ __forceinline__ void ff_glds16(const bf16* src, bf16* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)src,
                                     (void __attribute__((address_space(3)))*)lds_wave_base, 16, 0, 0);
}
// 64-byte LDS rows (4 chunks of 16 B): physical chunk cp of row r holds logical chunk cp ^ sw4(r) (conflict-free ds_read_b128 under the
// 4 x 16-lane service groups, as gemm_rownorm.hip)
__device__ __forceinline__ int ff_sw4(int r) { return (0 - (r >> 2)) & 3; }

__global__ __launch_bounds__(FF_NW * 64) void dit_ffn_kernel(DitFfnArgs p) {
    extern __shared__ __attribute__((aligned(1024))) char smem_raw[];
    bf16* Hs = reinterpret_cast<bf16*>(smem_raw);                    // [12][64][32]
    bf16* Bs = Hs + FF_K1 * FF_BM * FF_BK;                           // [3][384][32]
    bf16* Fs = Bs + FF_NS * FF_STAGE_ROWS * FF_BK;                   // [4][64][32]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / FF_WN, wn = wave % FF_WN;
    const int m0 = blockIdx.x * FF_BM;
    const bf16* __restrict__ A = reinterpret_cast<const bf16*>(p.A);
    const bf16* __restrict__ W13 = reinterpret_cast<const bf16*>(p.W13);
    const bf16* __restrict__ W2 = reinterpret_cast<const bf16*>(p.W2);
    const int nch = p.F / FF_CH;
    const int nsteps = nch * (FF_K1 + FF_K2);

    const int drow = lane >> 2, dcp = lane & 3;
    const int lc = (dcp ^ ff_sw4(drow)) << 3;                        // logical 16-byte chunk this lane fetches (elements)
    // ---- H tile: 48 DMA slots (k-slice t, 16-row group r4) -> wave w takes slots w, w+8, ...
#pragma unroll
    for (int j = 0; j < (FF_K1 * FF_BM / 16) / FF_NW; ++j) {
        const int slot = wave + FF_NW * j;                            // 0..47
        const int t = slot >> 2, r4 = slot & 3;
        const int row = r4 * 16 + drow;
        ff_glds16(A + (size_t)min(m0 + row, p.M - 1) * p.lda + t * FF_BK + lc, Hs + (t * FF_BM + r4 * 16) * FF_BK);
    }
    // ---- B ring: per-lane row offsets of this wave's slots. GEMM-1 stage = 16 slots (2 per wave), GEMM-2 stage = 24 slots (3 per wave)
    size_t off1[2], off2[3];
#pragma unroll
    for (int j = 0; j < 2; ++j) off1[j] = (size_t)((wave + FF_NW * j) * 16 + drow) * p.ldw13 + lc;
#pragma unroll
    for (int j = 0; j < 3; ++j) off2[j] = (size_t)((wave + FF_NW * j) * 16 + drow) * p.ldw2 + lc;
    // step s -> (chunk c, phase): s % 16 < 12 -> GEMM-1 k-step, else GEMM-2 k-step
    auto issue = [&](int s) {
        const int c = s / (FF_K1 + FF_K2), t = s - c * (FF_K1 + FF_K2);
        bf16* dst = Bs + (s % FF_NS) * FF_STAGE_ROWS * FF_BK;
        if (t < FF_K1) {
            const bf16* src = W13 + (size_t)c * (2 * FF_CH) * p.ldw13 + t * FF_BK;
#pragma unroll
            for (int j = 0; j < 2; ++j) ff_glds16(src + off1[j], dst + (wave + FF_NW * j) * 512);
        } else {
            const bf16* src = W2 + (size_t)c * FF_CH + (t - FF_K1) * FF_BK;
#pragma unroll
            for (int j = 0; j < 3; ++j) ff_glds16(src + off2[j], dst + (wave + FF_NW * j) * 512);
        }
    };
    auto is_g2 = [&](int s) { return (s % (FF_K1 + FF_K2)) >= FF_K1; };
    issue(0);
    issue(1);

    f32x4 acc1[2][4], acc2[2][6];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j) acc2[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int frow = lane & 15, g = lane >> 4;
    const int chunk = (g ^ ff_sw4(frow)) << 3;                       // this lane's 16-byte piece of a fragment row
    // residual prefetch registers (row layout of the epilogue: thread -> (row rl of a 32-row half, 16-lane column group))
    const int rl = tid >> 4, l16 = tid & 15;
    f32x4 xpre[2][FF_D / 64];

    for (int s = 0; s < nsteps; ++s) {
        // stage s has landed once at most the DMA of stage s+1 (2 or 3 wave-instructions, issued later) is outstanding; in the last
        // step the 12 residual loads issued at the end of step nsteps-2 are the only newer requests
        if (s + 1 < nsteps) {
            if (is_g2(s + 1)) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(12)" ::: "memory");                             // only the residual loads may still fly
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");            // this wave's F-chunk stores (GEMM-1 -> GEMM-2 hand-over) are in LDS
        __builtin_amdgcn_s_barrier();
        if (s + 2 < nsteps) issue(s + 2);
        const int t = s % (FF_K1 + FF_K2);
        const bf16* bs = Bs + (s % FF_NS) * FF_STAGE_ROWS * FF_BK;
        if (t < FF_K1) {
            if (t == 0) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc1[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            const bf16* as = Hs + (t * FF_BM + wm * FF_TM + frow) * FF_BK + chunk;
            const bf16* bw = bs + (wn * 64 + frow) * FF_BK + chunk;
            bf16x8 fa[2], fb[4];
#pragma unroll
            for (int i = 0; i < 2; ++i) fa[i] = *reinterpret_cast<const bf16x8*>(as + i * 16 * FF_BK);
#pragma unroll
            for (int j = 0; j < 4; ++j) fb[j] = *reinterpret_cast<const bf16x8*>(bw + j * 16 * FF_BK);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc1[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc1[i][j], 0, 0, 0);
            if (t == FF_K1 - 1) {
                // GLU: W13 rows alternate [gate16 | up16], so fragment 2q holds the gates and 2q+1 the ups of the same 16 F columns;
                // the lane owns F columns wn*32 + q*16 + g*4 + {0..3} of rows wm*32 + i*16 + frow -> k-slice wn of the F chunk
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        bf16x4 v;
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = (bf16)(ina_silu(acc1[i][2 * q][r]) * acc1[i][2 * q + 1][r]);
                        const int row = wm * FF_TM + i * 16 + frow;
                        const int c16 = q * 2 + (g >> 1);
                        *reinterpret_cast<bf16x4*>(Fs + (wn * FF_BM + row) * FF_BK + ((c16 ^ ff_sw4(frow)) << 3) + (g & 1) * 4) = v;
                    }
            }
        } else {
            const int t2 = t - FF_K1;
            const bf16* as = Fs + (t2 * FF_BM + wm * FF_TM + frow) * FF_BK + chunk;
            const bf16* bw = bs + (wn * 96 + frow) * FF_BK + chunk;
            bf16x8 fa[2], fb[6];
#pragma unroll
            for (int i = 0; i < 2; ++i) fa[i] = *reinterpret_cast<const bf16x8*>(as + i * 16 * FF_BK);
#pragma unroll
            for (int j = 0; j < 6; ++j) fb[j] = *reinterpret_cast<const bf16x8*>(bw + j * 16 * FF_BK);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 6; ++j) acc2[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc2[i][j], 0, 0, 0);
        }
        if (s == nsteps - 2) {
            // every B stage has been requested: prefetch this thread's residual rows (both 32-row halves) under the last two k-steps
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int m = min(m0 + half * FF_TM + rl, p.M - 1);
                const float* xrow = p.X + (size_t)m * p.ldx;
#pragma unroll
                for (int c = 0; c < FF_D / 64; ++c) xpre[half][c] = *reinterpret_cast<const f32x4*>(xrow + c * 64 + l16 * 4);
            }
        }
    }
    __syncthreads();   // tile buffers are free: LDS becomes [4][64] f32 row partials + one 32-row fp32 slab of the projection

    // lane holds rows m = wm*32 + i*16 + frow, columns n = wn*96 + j*16 + g*4 + {0..3}
    float* ssq = reinterpret_cast<float*>(smem_raw);                 // [4 wave columns][64 rows]
    float* slab = ssq + 4 * FF_BM;                                   // [32][FF_SLD]
    const float invN = 1.0f / FF_D;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        float sq = 0.f;
#pragma unroll
        for (int j = 0; j < 6; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) sq += acc2[i][j][r] * acc2[i][j][r];
        sq += __shfl_xor(sq, 16);
        sq += __shfl_xor(sq, 32);
        if (g == 0) ssq[wn * FF_BM + wm * FF_TM + i * 16 + frow] = sq;
    }
    __syncthreads();
    float rstd1[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = wm * FF_TM + i * 16 + frow;
        rstd1[i] = rsqrtf((ssq[row] + ssq[FF_BM + row] + ssq[2 * FF_BM + row] + ssq[3 * FF_BM + row]) * invN + p.eps);
    }
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        if (wm == half) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 6; ++j) {
                    f32x4 t;
#pragma unroll
                    for (int r = 0; r < 4; ++r) t[r] = acc2[i][j][r] * rstd1[i];
                    *reinterpret_cast<f32x4*>(&slab[(i * 16 + frow) * FF_SLD + wn * 96 + j * 16 + g * 4]) = t;
                }
        }
        __syncthreads();
        const int m = m0 + half * FF_TM + rl;
        const bool live = m < p.M;
        const int mb = (live ? m : 0) / p.mod_div;
        const float* gt = p.gate ? p.gate + (size_t)mb * p.mod_ld : nullptr;
        float* xrow = p.X + (size_t)(live ? m : 0) * p.ldx;
        f32x4 v[FF_D / 64];
        float s2 = 0.f;
#pragma unroll
        for (int c = 0; c < FF_D / 64; ++c) {
            const int n = c * 64 + l16 * 4;
            f32x4 t = *reinterpret_cast<const f32x4*>(&slab[rl * FF_SLD + n]);
            const f32x4 gm = *reinterpret_cast<const f32x4*>(p.gamma + n);
#pragma unroll
            for (int r = 0; r < 4; ++r) t[r] *= gm[r];
            if (gt) {
                const f32x4 gv = *reinterpret_cast<const f32x4*>(gt + n);
#pragma unroll
                for (int r = 0; r < 4; ++r) t[r] *= tanhf(gv[r]);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) t[r] += xpre[half][c][r];
            if (live) *reinterpret_cast<f32x4*>(xrow + n) = t;
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
            for (int c = 0; c < FF_D / 64; ++c) {
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

int ina_launch_dit_ffn(const DitFfnArgs& p_in, hipStream_t stream) {
    DitFfnArgs p = p_in;
    if (p.mod_div <= 0) p.mod_div = 1;
    INA_REQUIRE(p.A && p.W13 && p.W2 && p.X && p.gamma, "dit_ffn: A, W13, W2, X and gamma are required");
    INA_REQUIRE(p.D == FF_D, "dit_ffn: D=%d (the row-block kernel is built for dim 384)", p.D);
    INA_REQUIRE(p.M > 0 && p.F >= FF_CH && p.F % FF_CH == 0, "dit_ffn: M=%d, F=%d (F must be a multiple of 128)", p.M, p.F);
    // H may alias A: a workgroup reads its own 64 rows of A into LDS before it writes the same 64 rows of H
    INA_REQUIRE(p.lda % 8 == 0 && p.ldw13 % 8 == 0 && p.ldw2 % 8 == 0 && p.ldx % 4 == 0 && (!p.H || p.ldh % 8 == 0), "dit_ffn: row strides must keep 16-byte alignment");
    INA_REQUIRE(((uintptr_t)p.A % 16) == 0 && ((uintptr_t)p.W13 % 16) == 0 && ((uintptr_t)p.W2 % 16) == 0 && ((uintptr_t)p.X % 16) == 0 && ((uintptr_t)p.H % 16) == 0,
                "dit_ffn: misaligned pointer");
    INA_REQUIRE((!p.gate && !p.mod_scale2) || (p.mod_ld > 0 && p.mod_ld % 4 == 0), "dit_ffn: modulation needs mod_ld (multiple of 4)");
    static bool attr_done = false;
    if (!attr_done) {
        INA_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(dit_ffn_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)FF_LDS));
        attr_done = true;
    }
    ina_prof_set_sub(40);
    InaProfScope prof(INA_PROF_GEMM, 2.0 * p.M * FF_D * p.F * 3.0, 2.0 * p.M * FF_D + 6.0 * FF_D * p.F + p.M * FF_D * (8.0 + (p.H ? 2.0 : 0.0)), stream);
    hipLaunchKernelGGL(dit_ffn_kernel, dim3((p.M + FF_BM - 1) / FF_BM), dim3(FF_NW * 64), FF_LDS, stream, p);
    INA_HIP_CHECK(hipGetLastError());
    return 0;
}
