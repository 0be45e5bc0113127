// Fused SwiGLU feed-forward of a NextDiT block for gfx950 (dim 384 = NextDiTCrossAttnConfig.dim):
//
//     F  = silu(H . W1^T) * (H . W3^T)                         LuminaFeedForward: linear_1 / linear_3 (stored interleaved, see W13)
//     P  = F . W2^T                                            linear_2
//     X += tanh(gate[b]) * rmsnorm(P) * gamma                  ffn_norm2 + gate_mlp + residual (fp32 stream)
//     Hn = rmsnorm(X) * gamma2 * (1 + mod_scale2[b])           the NEXT block's norm1 + adaLN scale (bf16 GEMM operand), optional
//
// Unfused this is three launches (GLU GEMM writing F, GEMM reading it back, norm kernel) that move 670 MB per block at 65 536 rows.
// Here a workgroup owns 64 FULL rows: the H tile sits in LDS (48 KiB), F is produced and consumed chunk by chunk (128 columns) through a
// double-buffered 16 KiB LDS tile and never exists in HBM, P [64 x 384] stays in registers until the epilogue (row statistics through
// LDS, row-layout residual update, next pre-norm: the one of gemm_rownorm.hip). See the kernel comment for how the weights are streamed
// (global -> VGPR, never through LDS); the first version streamed them through a 3-stage LDS-DMA ring and measured 366 us against
// 308 us for the three launches (profiles/r02b_bench_ffn.log): 48 KiB in flight per CU bound the L2 -> CU weight stream at ~6.5 TB/s.
// Reference: diffusers LuminaFeedForward / LuminaNextDiTBlock.forward (diffusers==0.33.1) as wired by nextdit_traj.py:121-188.
#include "common.h"
#include "kernels.h"

namespace {

constexpr int FF_BM = 64, FF_D = 384, FF_BK = 32, FF_NW = 8;
constexpr int FF_CH = 128;                                   // F columns per chunk: 16 per wave (a [gate16 | up16] block of W13 rows)
constexpr int FF_K1 = FF_D / FF_BK;                          // 12 k-slices of GEMM-1
constexpr int FF_K2 = FF_CH / FF_BK;                         // 4 k-slices of GEMM-2 per chunk
constexpr size_t FF_HS = size_t(FF_K1) * FF_BM * FF_BK * sizeof(bf16);          // 48 KiB: H tile as 12 k-slices of [64 rows][32] (64-byte swizzled rows)
constexpr size_t FF_FS = size_t(2) * FF_K2 * FF_BM * FF_BK * sizeof(bf16);      // 2 x 16 KiB: double-buffered F chunk
constexpr size_t FF_LDS = FF_HS + FF_FS;                                        // 80 KiB
constexpr int FF_SLD = FF_D + 4;                             // fp32 row of the epilogue slab
static_assert(size_t(FF_NW * FF_BM + 32 * FF_SLD) * sizeof(float) <= FF_LDS, "epilogue scratch must fit in the tile buffers");

__device__ __forceinline__ void ff_glds16(const bf16* src, bf16* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)src,
                                     (void __attribute__((address_space(3)))*)lds_wave_base, 16, 0, 0);
}
// 64-byte LDS rows (4 chunks of 16 B): physical chunk cp of row r holds logical chunk cp ^ sw4(r) (conflict-free ds_read_b128 under the
// 4 x 16-lane service groups, as gemm_rownorm.hip)
__device__ __forceinline__ int ff_sw4(int r) { return (0 - (r >> 2)) & 3; }

// v2: the weights never touch LDS. 8 waves side by side over the output columns (1 x 8), every wave owns ALL 64 rows:
//   GEMM-1: wave w computes F columns [16 w, 16 w + 16) of the chunk = W13 rows [32 w, 32 w + 32) (a gate16 | up16 block): 2 B fragments
//   GEMM-2: wave w computes P columns [48 w, 48 w + 48): 3 B fragments per k-slice
// B fragments are loaded global -> VGPR in exactly the MFMA operand layout (lane (n, g) reads 16 bytes of row n at k = 8 g; 4 lanes cover
// 64 contiguous bytes of a row, the next k-slice completes the 128-byte line), 12 fragments (a "phase": half of a chunk's GEMM-1, or
// its GEMM-2) ahead of the MFMAs that consume them: 12 KiB in flight per wave, 96 KiB per CU - twice what the 3-stage LDS ring of the
// first version held, without the ring's barriers (the H tile is read-only, the F chunk is double buffered: ONE barrier per chunk).
// A fragments (H rows, F rows) come from LDS (ds_read_b128, swizzled 64-byte rows). LDS traffic per k-slice: 8 waves x 4 KiB of A.
__global__ __launch_bounds__(FF_NW * 64) void dit_ffn_kernel(DitFfnArgs p) {
    extern __shared__ __attribute__((aligned(1024))) char smem_raw[];
    bf16* Hs = reinterpret_cast<bf16*>(smem_raw);                    // [12][64][32]
    bf16* Fs = Hs + FF_K1 * FF_BM * FF_BK;                           // [2][4][64][32]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m0 = blockIdx.x * FF_BM;
    const bf16* __restrict__ A = reinterpret_cast<const bf16*>(p.A);
    const int nch = p.F / FF_CH;
    const int frow = lane & 15, g = lane >> 4;

    // ---- H tile -> LDS by DMA: 48 slots (k-slice t, 16-row group r4) of 1 KiB, 6 per wave
    {
        const int drow = lane >> 2, dcp = lane & 3;
        const int lc = (dcp ^ ff_sw4(drow)) << 3;
#pragma unroll
        for (int j = 0; j < (FF_K1 * FF_BM / 16) / FF_NW; ++j) {
            const int slot = wave + FF_NW * j;
            const int t = slot >> 2, r4 = slot & 3;
            ff_glds16(A + (size_t)min(m0 + r4 * 16 + drow, p.M - 1) * p.lda + t * FF_BK + lc, Hs + (t * FF_BM + r4 * 16) * FF_BK);
        }
    }
    // ---- per-lane weight pointers: W13 rows 32 w + {frow, 16 + frow} of a chunk, W2 rows 48 w + 16 j + frow; k offset 8 g
    const bf16* w13p = reinterpret_cast<const bf16*>(p.W13) + (size_t)(wave * 32 + frow) * p.ldw13 + g * 8;
    const bf16* w2p = reinterpret_cast<const bf16*>(p.W2) + (size_t)(wave * 48 + frow) * p.ldw2 + g * 8;
    const size_t w13_up = (size_t)16 * p.ldw13, w13_chunk = (size_t)(2 * FF_CH) * p.ldw13, w2_frag = (size_t)16 * p.ldw2;

    bf16x8 ba[6][2], bb[6][2], bc[4][3];           // GEMM-1 slices 0-5 / 6-11 (gate, up), GEMM-2 slices 0-3 (3 column fragments)
    auto load_a = [&](int c) {
        const bf16* q = w13p + (size_t)c * w13_chunk;
#pragma unroll
        for (int t = 0; t < 6; ++t) {
            ba[t][0] = *reinterpret_cast<const bf16x8*>(q + t * FF_BK);
            ba[t][1] = *reinterpret_cast<const bf16x8*>(q + w13_up + t * FF_BK);
        }
    };
    auto load_b = [&](int c) {
        const bf16* q = w13p + (size_t)c * w13_chunk + 6 * FF_BK;
#pragma unroll
        for (int t = 0; t < 6; ++t) {
            bb[t][0] = *reinterpret_cast<const bf16x8*>(q + t * FF_BK);
            bb[t][1] = *reinterpret_cast<const bf16x8*>(q + w13_up + t * FF_BK);
        }
    };
    auto load_c = [&](int c) {
        const bf16* q = w2p + (size_t)c * FF_CH;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int j = 0; j < 3; ++j) bc[t][j] = *reinterpret_cast<const bf16x8*>(q + j * w2_frag + t * FF_BK);
    };

    f32x4 acc1[4][2], acc2[4][3];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) acc2[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int chunk = (g ^ ff_sw4(frow)) << 3;                       // this lane's 16-byte piece of an A fragment row
    auto gemm1 = [&](bf16x8 (&bw)[6][2], int t0) {
#pragma unroll
        for (int t = 0; t < 6; ++t) {
            const bf16* as = Hs + ((t0 + t) * FF_BM + frow) * FF_BK + chunk;
            bf16x8 fa[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) fa[i] = *reinterpret_cast<const bf16x8*>(as + i * 16 * FF_BK);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc1[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bw[t][j], fa[i], acc1[i][j], 0, 0, 0);
        }
    };

    // chunk order is rotated per workgroup (the sum over F chunks commutes): the ~256 resident workgroups otherwise walk the SAME weight
    // lines in lock step and queue on the same L2 channels (both weight-streaming variants plateaued at ~6.5 TB/s of L2 -> CU traffic)
    const int c_rot = (p.rotate & 1) ? (int)(blockIdx.x % (unsigned)nch) : 0;
    const bool dbg_noload = p.rotate & 2, dbg_noio = p.rotate & 4;   // timing experiments only (tools/bench_ffn.py)
    auto chunk_of = [&](int cc) { const int c = cc + c_rot; return c >= nch ? c - nch : c; };
    load_a(chunk_of(0));
    asm volatile("s_waitcnt vmcnt(12)" ::: "memory");                // the H tile DMA (issued first) has landed; the 12 weight loads may fly
    __syncthreads();
    for (int cc = 0; cc < nch; ++cc) {
        const int c = chunk_of(cc);
        bf16* fs = Fs + (cc & 1) * FF_K2 * FF_BM * FF_BK;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc1[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        // every batch of 12 weight loads is ISSUED before the MFMAs of the phase it hides under (sched_barrier: the machine scheduler
        // otherwise sinks the loads next to their first use and the prefetch distance collapses)
        if (!dbg_noload || cc == 0) load_b(c);                      // phase 2 of this chunk in flight under phase 1
        __builtin_amdgcn_sched_barrier(0);
        gemm1(ba, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (!dbg_noload || cc == 0) load_c(c);
        __builtin_amdgcn_sched_barrier(0);
        gemm1(bb, 6);
        __builtin_amdgcn_sched_barrier(0);
        if (cc + 1 < nch && !dbg_noload) load_a(chunk_of(cc + 1));
        __builtin_amdgcn_sched_barrier(0);
        // GLU: fragment 0 holds the gates, fragment 1 the ups of F columns 16 w + 4 g + {0..3}, rows 16 i + frow -> k-slice w / 2 of the F chunk
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            bf16x4 v;
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = (bf16)(ina_silu(acc1[i][0][r]) * acc1[i][1][r]);
            const int c16 = (wave & 1) * 2 + (g >> 1);
            *reinterpret_cast<bf16x4*>(fs + ((wave >> 1) * FF_BM + i * 16 + frow) * FF_BK + ((c16 ^ ff_sw4(frow)) << 3) + (g & 1) * 4) = v;
        }
        __syncthreads();                                             // F chunk c complete (its buffer was last read two chunks ago)
#pragma unroll
        for (int t = 0; t < FF_K2; ++t) {
            const bf16* as = fs + (t * FF_BM + frow) * FF_BK + chunk;
            bf16x8 fa[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) fa[i] = *reinterpret_cast<const bf16x8*>(as + i * 16 * FF_BK);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) acc2[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bc[t][j], fa[i], acc2[i][j], 0, 0, 0);
        }
    }
    __syncthreads();   // tile buffers are free: LDS becomes [8][64] f32 row partials + one 32-row fp32 slab of the projection

    // lane holds rows m = 16 i + frow, columns n = 48 w + 16 j + 4 g + {0..3}
    float* ssq = reinterpret_cast<float*>(smem_raw);                 // [8 waves][64 rows]
    float* slab = ssq + FF_NW * FF_BM;                               // [32][FF_SLD]
    const float invN = 1.0f / FF_D;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float sq = 0.f;
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) sq += acc2[i][j][r] * acc2[i][j][r];
        sq += __shfl_xor(sq, 16);
        sq += __shfl_xor(sq, 32);
        if (g == 0) ssq[wave * FF_BM + i * 16 + frow] = sq;
    }
    __syncthreads();
    float rstd1[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float tot = 0.f;
#pragma unroll
        for (int w = 0; w < FF_NW; ++w) tot += ssq[w * FF_BM + i * 16 + frow];
        rstd1[i] = rsqrtf(tot * invN + p.eps);
    }
    const int rl = tid >> 4, l16 = tid & 15;                         // row stage: thread -> (row of a 32-row half, 16-lane column group)
#pragma unroll
    for (int half = 0; half < 2; ++half) {
#pragma unroll
        for (int ii = 0; ii < 2; ++ii) {
            const int i = half * 2 + ii;
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                f32x4 t;
#pragma unroll
                for (int r = 0; r < 4; ++r) t[r] = acc2[i][j][r] * rstd1[i];
                *reinterpret_cast<f32x4*>(&slab[(ii * 16 + frow) * FF_SLD + wave * 48 + j * 16 + g * 4]) = t;
            }
        }
        __syncthreads();
        const int m = m0 + half * 32 + rl;
        const bool live = m < p.M && !dbg_noio;
        const int mb = (m < p.M ? m : 0) / p.mod_div;
        const float* gt = p.gate ? p.gate + (size_t)mb * p.mod_ld : nullptr;
        float* xrow = p.X + (size_t)(m < p.M ? m : 0) * p.ldx;
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
        __syncthreads();   // the slab is rewritten for the other half
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
