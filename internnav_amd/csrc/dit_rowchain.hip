// Row-chain kernel of a NextDiT block (gfx950): everything between two attention stages that is LOCAL TO A ROW, in one launch.
//
//     P  = A[M,K1] . W1[384,K1]^T                                  attn2.to_out (K1 = 384) / feed_forward.linear_2 at FFN width 1024 (K1 = 1024), rounded to bf16
//                                                                  (FFN 1536 - the reference's block under its pinned diffusers 0.33.1 - runs the FIRST launch form only,
//                                                                  with N2 = 3072: a K1 = 1536 build of the second form was measured slower inside the policy step,
//                                                                  profiles/r06i_rowchain_halves_in_step.txt, and is not built)
//     X += tanh(gate[b]) * rmsnorm(P) * gamma1                     norm2 / ffn_norm2 + tanh gate + residual (fp32 stream, in place)
//     H  = rmsnorm(X) * gamma2 * (1 + mod_scale2[b])               ffn_norm1 / the next block's norm1 + adaLN scale (bf16 GEMM operand)
//     C2 = H . W2[N2,384]^T   or   silu(H . Wg^T) * (H . Wu^T)      feed_forward.linear_1/3 + SiLU gate (GLU)  /  the next block's q1|k1|v1|q2
//
// with b = row / mod_div  (reference: diffusers LuminaNextDiTBlock.forward as wired by nextdit_traj.py:121-178). Unfused these are a tiled
// GEMM writing the bf16 projection, a norm launch reading it back together with the fp32 residual and writing H, and a row-panel GEMM reading
// H: per row 768 + 768 (P) + 768 + 768 (H) bytes of HBM traffic and two launches that this kernel does not have - the norm launches were the
// largest System-1 kernel by time (15.8 of 74.7 ms per 64-env call).
//
// Structure = the row-panel GEMM of gemm_rowpanel.hip (a wave owns 32 rows, W streams through an LDS ring filled by LDS-DMA, operands swapped so
// a lane ends with ONE output row and runs of 4 columns) run twice around an in-register epilogue:
//   GEMM 1  column-tile major (3 tiles of 128 columns), the A fragments of a 64-wide k chunk loaded global -> VGPR one stage ahead (the
//           same chunks are re-read per column tile: L1 / L2 hits); each finished tile is packed to bf16 (the rounding the unfused
//           projection buffer had) - 96 registers hold the wave's whole 32 x 384 projection;
//   epilogue the two lanes of a row (l, l + 32) hold 192 columns each: sum of squares in the lane + one cross-half add; the residual is read
//           and written in 16-byte pieces straight from / to the fragment layout; the new residual row stays in registers (192 fp32) until its
//           own statistic is known, then becomes the 24 bf16 operand fragments of GEMM 2 through the same `v_permlane32_swap` merge the GEMM
//           epilogue uses for its 16-byte stores (the merged 8 columns ARE a fragment) - H never exists in memory;
//   GEMM 2  the row-panel main loop on those fragments; its first W stages are already in the ring (the DMA ring runs through both GEMMs).
// gamma1 / tanh(gate) / gamma2 / (1 + mod_scale2) of the workgroup's environment are tabulated in LDS once (a workgroup's rows share b).
// seg_stats (plain second GEMM): the epilogue of GEMM 2 also sums every 384-wide segment of the row it stores and writes (mean, rstd) per
// segment - the LayerNorm statistics the attention stage of the NEXT block needs for q1 / k1 / q2 (dit_attention.hip: dit_attn_stats_kernel),
// which then reads every projection row once instead of twice.
// Numerics: the same roundings as the unfused chain (bf16 projection, fp32 statistics and residual, bf16 H); the K summation runs in
// 16-wide MFMA steps (as gemm_rowpanel.hip), so results differ from the tiled kernels in the last bits.
#include <type_traits>
#include <utility>

#pragma clang diagnostic ignored "-Winline-asm"   // (the DMA asm names m0 as clobbered, as gemm_w4.hip does)

#include "common.h"
#include "kernels.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int RC_NS = 4, RC_BN = 128, RC_STAGE = RC_BN * 64, RC_D = 384, RC_NT1 = RC_D / RC_BN, RC_KC2 = RC_D / 64;
constexpr size_t RC_LDS = size_t(RC_NS) * RC_STAGE * sizeof(bf16) + 2 * RC_D * sizeof(float);   // 64 KiB ring + 3 KiB tables

// one 1 KiB LDS-DMA piece with a SCALAR base + 32-bit lane byte offset (no 64-bit per-lane source pointers: eight of them did not fit beside
// the accumulators); M0 = LDS byte address of the piece
__device__ __forceinline__ void rc_glds16(const void* base, uint32_t voff, uint32_t lds_byte_addr) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_byte_addr), "v"(voff), "s"(base) : "memory", "m0");
}

template <int N>
__device__ __forceinline__ void rc_wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// A fragment straight from global memory into its operand registers. An asm statement, not a plain load: the compiler's own wait-count pass
// answers a plain VGPR load inside the DMA ring with `s_waitcnt vmcnt(0)` (it does not count LDS-DMA instructions it cannot match to a use),
// which would drain the ring every stage. The kernel retires these loads with its counted waits instead.
template <int OFF_BYTES>
__device__ __forceinline__ void rc_gload16(bf16x8& dst, const bf16* src) {
    asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(dst) : "v"(src), "n"(OFF_BYTES) : "memory");
}

template <int N>
using IC = std::integral_constant<int, N>;
// f(IC<0>{}), ..., f(IC<N - 1>{}): a fully unrolled loop whose index is a constant expression inside f (asm immediates, register-array slots)
template <class F, int... I>
__device__ __forceinline__ void rc_for_seq_impl(F&& f, std::integer_sequence<int, I...>) {
    (f(IC<I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void rc_for_seq(F&& f) {
    rc_for_seq_impl(f, std::make_integer_sequence<int, N>{});
}

__device__ __forceinline__ unsigned rc_pack2(float a, float b) {
    const bf16x2 v = {(bf16)a, (bf16)b};
    return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ float rc_lo(unsigned u) { return __builtin_bit_cast(float, u << 16); }
__device__ __forceinline__ float rc_hi(unsigned u) { return __builtin_bit_cast(float, u & 0xffff0000u); }

// lanes l and l + 32 hold the same row: lane l columns c .. c+3 (lo pair) and c+8 .. c+11 (hi pair), lane l + 32 columns c+4 .. c+7 and
// c+12 .. c+15. After the swap lane l holds c .. c+7 and lane l + 32 holds c+8 .. c+15: one 16-byte store each - and exactly the
// operand fragment of a 32x32x16 MFMA for k = c .. c+15 (row l & 31, k half l >> 5).
__device__ __forceinline__ u32x4 rc_merge_rows(unsigned lo0, unsigned lo1, unsigned hi0, unsigned hi1) {
    const u32x2 s0 = __builtin_amdgcn_permlane32_swap(lo0, hi0, false, false);
    const u32x2 s1 = __builtin_amdgcn_permlane32_swap(lo1, hi1, false, false);
    return u32x4{s0[0], s1[0], s0[1], s1[1]};
}

// 16 MFMAs of one W stage (128 output columns x 64 k) against the four 16-wide k fragments a0 .. a3 of this wave's 32 rows
__device__ __forceinline__ void rc_stage_mfma(const bf16* ws, const int (&foff)[4], const bf16x8& a0, const bf16x8& a1, const bf16x8& a2,
                                              const bf16x8& a3, f32x16 (&acc)[4]) {
    // fragments of k step kk + 1 are requested before the MFMAs of step kk are issued (pinned: the compiler otherwise sinks every read right
    // in front of its MFMA and waits lgkmcnt(0) per pair)
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
        const bf16x8& a = kk == 0 ? a0 : kk == 1 ? a1 : kk == 2 ? a2 : a3;
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[kk & 1][j], a, acc[j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's reads of the slot have returned before it reaches the next barrier
}

template <int NW, int KC1, bool HAS2, bool GLU2>
__global__ __launch_bounds__(NW * 64, 2) void dit_rowchain_kernel(ina_dit_rowchain_args p) {
    constexpr int NS = RC_NS, BN = RC_BN, STAGE = RC_STAGE, D = RC_D, NT1 = RC_NT1, KC2 = RC_KC2;
    constexpr int INST = 16 / NW;                         // 1 KiB DMA wave-instructions per wave and stage
    constexpr int NST = GLU2 ? 4 : 8;                     // store instructions per wave and column tile of GEMM 2
    static_assert(KC1 % 2 == 0 && NW == 4, "the A double buffer alternates per k chunk across column tiles; a workgroup is four 32-row waves");
    extern __shared__ __attribute__((aligned(1024))) char smem_raw[];
    bf16* Ws = reinterpret_cast<bf16*>(smem_raw);         // [NS][128][64]
    float* tab = reinterpret_cast<float*>(Ws + NS * STAGE);   // [2][384]: gamma1 * tanh(gate[b]) | gamma2 * (1 + mod_scale2[b])
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m0 = blockIdx.x * (NW * 32) + wave * 32;    // (M % (NW * 32) == 0: every wave owns 32 real rows)
    const int lrow = lane & 31, khalf = lane >> 5;
    const bf16* __restrict__ A = reinterpret_cast<const bf16*>(p.A);
    const bf16* __restrict__ W1 = reinterpret_cast<const bf16*>(p.W1);
    const bf16* __restrict__ W2 = reinterpret_cast<const bf16*>(p.W2);
    constexpr int total1 = NT1 * KC1;
    const int ntiles2 = HAS2 ? p.N2 / BN : 0;
    const int total = total1 + ntiles2 * KC2;

    // ---- per-environment tables (read in the epilogue, many barriers from here)
    {
        const int mb = (blockIdx.x * (NW * 32)) / p.mod_div;
        const float* gt = p.gate ? p.gate + (size_t)mb * p.mod_ld : nullptr;
        const float* ms = p.mod_scale2 ? p.mod_scale2 + (size_t)mb * p.mod_ld : nullptr;
        for (int i = tid; i < D; i += NW * 64) {
            tab[i] = gt ? p.gamma1[i] * tanhf(gt[i]) : p.gamma1[i];
            const float g2 = p.gamma2 ? p.gamma2[i] : 1.0f;
            tab[D + i] = ms ? g2 * (1.0f + ms[i]) : g2;
        }
    }

    // ---- W stream: per-lane BYTE offsets of this wave's DMA slots inside a (column tile, k chunk) stage of W1 / W2
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem_raw;
    uint32_t off1[INST];
    {
        const int drow = lane >> 3, dcp = lane & 7;
#pragma unroll
        for (int s = 0; s < INST; ++s) {
            const int r = (wave * INST + s) * 8 + drow;
            off1[s] = (uint32_t)(r * p.ldw1 + ((dcp ^ ((r >> 1) & 7)) << 3)) * 2u;
        }
    }
    uint32_t off2[INST];                                    // (filled behind the epilogue: nothing of GEMM 2 occupies registers before it)
    auto issue = [&](int st) {
        // stage st -> ring slot st % NS; stages [0, total1) walk W1 (column tile st / KC1, k chunk st % KC1), the rest W2. Past the end the
        // last stage is fetched again into a slot nobody reads any more: every iteration issues the same number of DMA instructions
        const int s2 = st < total ? st : total - 1;
        const uint32_t dst = lds0 + ((st & (NS - 1)) * STAGE + wave * INST * 512) * 2u;
        if (!HAS2 || s2 < total1) {
            const int nt = s2 / KC1, kc = s2 - nt * KC1;
            const bf16* b = W1 + (size_t)nt * BN * p.ldw1 + kc * 64;
#pragma unroll
            for (int s = 0; s < INST; ++s) rc_glds16(b, off1[s], __builtin_amdgcn_readfirstlane(dst + s * 1024u));
        } else {
            const int u = s2 - total1;
            const int nt = u / KC2, kc = u - nt * KC2;
            const bf16* b = W2 + (size_t)nt * BN * p.ldw2 + kc * 64;
#pragma unroll
            for (int s = 0; s < INST; ++s) rc_glds16(b, off2[s], __builtin_amdgcn_readfirstlane(dst + s * 1024u));
        }
    };
    auto lane_offsets2 = [&](int ln) {
        const int drow = ln >> 3, dcp = ln & 7;
#pragma unroll
        for (int s = 0; s < INST; ++s) {
            const int r = (wave * INST + s) * 8 + drow;
            off2[s] = HAS2 ? (uint32_t)(r * p.ldw2 + ((dcp ^ ((r >> 1) & 7)) << 3)) * 2u : 0u;
        }
    };
    lane_offsets2(lane);                                    // (GEMM 1's last iterations already request the first W2 stages)
    // fragment read offsets inside a stage (elements): column-tile row j * 32 + lrow, logical chunk kk * 2 + khalf
    int foff[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) foff[kk] = lrow * 64 + (((kk * 2 + khalf) ^ ((lrow >> 1) & 7)) << 3);

    // ---- GEMM 1: P = A . W1^T, column-tile major; A fragments of the NEXT stage are requested first in every iteration, then the ring refill
    const int row = m0 + lrow;
    const bf16* arow = A + (size_t)row * p.lda + khalf * 8;
    bf16x8 ab[2][4];
    issue(0);
    rc_for_seq<4>([&](auto KK) { rc_gload16<KK.value * 32>(ab[0][KK.value], arow); });
    issue(1);
    issue(2);
    unsigned P[NT1][4][8];                                  // the wave's 32 x 384 projection as packed bf16 pairs (fragment layout)
    rc_for_seq<NT1>([&](auto NT) {
        constexpr int nt = NT.value;
        f32x16 acc[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
        rc_for_seq<KC1>([&](auto KC) {
            constexpr int kc = KC.value, t = nt * KC1 + kc;
            // stage t and the A chunk of stage t have landed: behind the A loads only the refill of the previous iteration was issued
            // (prologue: DMA 0, A 0, DMA 1, DMA 2)
            if constexpr (t == 0) rc_wait_vm<2 * INST>();
            else rc_wait_vm<INST>();
            __builtin_amdgcn_s_barrier();                  // stage t visible to every wave; every wave is done reading stage t - 1
            __builtin_amdgcn_sched_barrier(0);             // (nothing that consumes the A fragments moves above the wait)
            if constexpr (t + 1 < total1) {
                constexpr int kn = kc + 1 == KC1 ? 0 : kc + 1;
                rc_for_seq<4>([&](auto KK) { rc_gload16<kn * 128 + KK.value * 32>(ab[(kc + 1) & 1][KK.value], arow); });
            }
            issue(t + NS - 1);                             // ... whose slot is refilled
            __builtin_amdgcn_sched_barrier(0);
            rc_stage_mfma(Ws + (t & (NS - 1)) * STAGE, foff, ab[kc & 1][0], ab[kc & 1][1], ab[kc & 1][2], ab[kc & 1][3], acc);
        });
        // acc[j][i * 4 + e] = column nt * 128 + j * 32 + i * 8 + khalf * 4 + e of row `row`
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                P[nt][j][2 * i] = rc_pack2(acc[j][i * 4], acc[j][i * 4 + 1]);
                P[nt][j][2 * i + 1] = rc_pack2(acc[j][i * 4 + 2], acc[j][i * 4 + 3]);
                // opaque to the optimiser: it otherwise sees through pack -> unpack and keeps every rounded element in its own register
                // (192 instead of 96 live registers across the remaining column tiles)
                asm("" : "+v"(P[nt][j][2 * i]), "+v"(P[nt][j][2 * i + 1]));
            }
        __builtin_amdgcn_sched_barrier(0);
    });

    // ---- epilogue: X += tanh(gate) * rmsnorm(P) * gamma1 ; H = rmsnorm(X) * gamma2 * (1 + mod_scale2)
    float s1 = 0.f;
#pragma unroll
    for (int nt = 0; nt < NT1; ++nt) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const float a = rc_lo(P[nt][j][q]), b = rc_hi(P[nt][j][q]);
                s1 += a * a;
                s1 += b * b;
            }
        asm volatile("" : "+v"(s1));
        __builtin_amdgcn_sched_barrier(0);
    }
    // opaque again: the residual pass below unpacks the same registers, and as common subexpressions of the loop above all 192 unpacked
    // values would stay alive through it
#pragma unroll
    for (int nt = 0; nt < NT1; ++nt)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int q = 0; q < 8; q += 2) asm("" : "+v"(P[nt][j][q]), "+v"(P[nt][j][q + 1]));
    s1 += __shfl_xor(s1, 32);
    const float rstd1 = rsqrtf(s1 * (1.0f / D) + p.eps);
    float* xrow = p.X + (size_t)row * p.ldx + khalf * 4;
    float s2 = 0.f;
    // one 32-column block (four 16-byte pieces per lane) at a time, the next block's pieces requested before this one's arithmetic; the
    // projection registers of a block die as the block is consumed
    if constexpr (!HAS2) {
        // no second GEMM: the new residual row stays in registers (192 fp32) until its statistic is known, then H is written (if asked for)
        float xn[NT1 * 4][16];                              // block nt * 4 + j, element i * 4 + e
#pragma unroll
        for (int blk = 0; blk < NT1 * 4; ++blk) {
            f32x4 xv[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) xv[i] = *reinterpret_cast<const f32x4*>(xrow + blk * 32 + i * 8);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int c = blk * 32 + i * 8 + khalf * 4;
                const f32x4 gg = *reinterpret_cast<const f32x4*>(tab + c);
                const unsigned u0 = P[blk >> 2][blk & 3][2 * i], u1 = P[blk >> 2][blk & 3][2 * i + 1];
                const float v[4] = {rc_lo(u0), rc_hi(u0), rc_lo(u1), rc_hi(u1)};
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float w = v[e] * rstd1;
                    w *= gg[e];
                    w += xv[i][e];
                    o[e] = w;
                    xn[blk][i * 4 + e] = w;
                    s2 += w * w;
                }
                *reinterpret_cast<f32x4*>(xrow + blk * 32 + i * 8) = o;
            }
            asm volatile("" : "+v"(s2));   // the running sum is MATERIALISED here (the compiler otherwise keeps all 192 squares and sinks the adds behind the loop)
            __builtin_amdgcn_sched_barrier(0);
        }
        if (p.H) {
            s2 += __shfl_xor(s2, 32);
            const float rstd2 = rsqrtf(s2 * (1.0f / D) + p.eps);
            bf16* hrow = reinterpret_cast<bf16*>(p.H) + (size_t)row * p.ldh + khalf * 8;
#pragma unroll
            for (int blk = 0; blk < NT1 * 4; ++blk) {
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    unsigned q[2][2];
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const int i = 2 * m + h, c = blk * 32 + i * 8 + khalf * 4;
                        const f32x4 hh = *reinterpret_cast<const f32x4*>(tab + D + c);
                        float w[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            w[e] = xn[blk][i * 4 + e] * rstd2;
                            w[e] *= hh[e];
                        }
                        q[h][0] = rc_pack2(w[0], w[1]);
                        q[h][1] = rc_pack2(w[2], w[3]);
                    }
                    *reinterpret_cast<u32x4*>(hrow + blk * 32 + m * 16) = rc_merge_rows(q[0][0], q[0][1], q[1][0], q[1][1]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        rc_wait_vm<0>();                                    // (the dummy tail stages land in LDS before the workgroup retires)
    } else {
        // second GEMM follows: the operand fragments of GEMM 2 are built block by block from U = X_new * gamma2 * (1 + mod_scale2), WITHOUT the
        // row's 1 / rms - that statistic needs the whole row, and keeping 192 fp32 values per lane for it does not fit beside anything else.
        // It is a per-row scalar and this lane owns exactly one row, so it is applied to GEMM 2's fp32 accumulators instead:
        // (diag(r) U) W^T = diag(r) (U W^T). Only the place of the bf16 rounding moves (bf16(U) instead of bf16(r U): the same relative step).
        bf16x8 a2[NT1 * 4 * 2];                             // U as the 24 operand fragments of GEMM 2 (k = q * 16 + khalf * 8 .. + 7)
        f32x4 xv[2][4];
#pragma unroll
        for (int i = 0; i < 4; ++i) xv[0][i] = *reinterpret_cast<const f32x4*>(xrow + i * 8);
#pragma unroll
        for (int blk = 0; blk < NT1 * 4; ++blk) {
            if (blk + 1 < NT1 * 4) {
#pragma unroll
                for (int i = 0; i < 4; ++i) xv[(blk + 1) & 1][i] = *reinterpret_cast<const f32x4*>(xrow + (blk + 1) * 32 + i * 8);
            }
            __builtin_amdgcn_sched_barrier(0);
            unsigned q[4][2];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int c = blk * 32 + i * 8 + khalf * 4;
                const f32x4 gg = *reinterpret_cast<const f32x4*>(tab + c), hh = *reinterpret_cast<const f32x4*>(tab + D + c);
                const unsigned u0 = P[blk >> 2][blk & 3][2 * i], u1 = P[blk >> 2][blk & 3][2 * i + 1];
                const float v[4] = {rc_lo(u0), rc_hi(u0), rc_lo(u1), rc_hi(u1)};
                f32x4 o;
                float u[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float w = v[e] * rstd1;
                    w *= gg[e];
                    w += xv[blk & 1][i][e];
                    o[e] = w;
                    s2 += w * w;
                    u[e] = w * hh[e];
                }
                *reinterpret_cast<f32x4*>(xrow + blk * 32 + i * 8) = o;
                q[i][0] = rc_pack2(u[0], u[1]);
                q[i][1] = rc_pack2(u[2], u[3]);
            }
#pragma unroll
            for (int m = 0; m < 2; ++m)
                a2[blk * 2 + m] = __builtin_bit_cast(bf16x8, rc_merge_rows(q[2 * m][0], q[2 * m][1], q[2 * m + 1][0], q[2 * m + 1][1]));
            asm volatile("" : "+v"(s2));   // the running sum is MATERIALISED here (the compiler otherwise keeps all 192 squares and sinks the adds behind the loop)
            __builtin_amdgcn_sched_barrier(0);
        }
        s2 += __shfl_xor(s2, 32);
        const float rstd2 = rsqrtf(s2 * (1.0f / D) + p.eps);
        rc_wait_vm<0>();   // the epilogue's loads / stores are retired; the first W2 stages (requested during GEMM 1) have long landed
        // lane constants of GEMM 2 from a lane id the compiler cannot tie to the one above (they would otherwise be computed at kernel start
        // and held through the epilogue)
        const int lane2 = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
        const int lrow2 = lane2 & 31, khalf2 = lane2 >> 5;
        lane_offsets2(lane2);
        int foff[4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) foff[kk] = lrow2 * 64 + (((kk * 2 + khalf2) ^ ((lrow2 >> 1) & 7)) << 3);
        int t = total1;
        // ---- GEMM 2: the row-panel main loop of gemm_rowpanel.hip on the fragments a2
        bf16* __restrict__ Cb = reinterpret_cast<bf16*>(p.C2);
        const size_t crow = (size_t)(m0 + lrow2) * p.ldc2;
        const int khalf = khalf2;
        // LayerNorm statistics of the 384-wide segments of the C2 row (plain second GEMM only): running sum / sum of squares of this lane's
        // 64 of a tile's 128 columns, closed every third tile
        float seg_s = 0.f, seg_q = 0.f;
        int seg_t = 0;
        float* __restrict__ seg_out = p.seg_stats ? p.seg_stats + (size_t)(m0 + lrow2) * (size_t)(p.N2 / D) * 2 : nullptr;
        for (int nt = 0; nt < ntiles2; ++nt) {
            f32x16 acc[4];
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
#pragma unroll
            for (int kc = 0; kc < KC2; ++kc, ++t) {
                // stage t has landed for this wave's share: the NS - 2 newer stages (and, right behind a tile epilogue, its NST stores - same
                // in-order counter, issued after the DMA of stage t + NS - 2 iff kc <= NS - 2) may stay outstanding
                if (nt > 0 && kc <= NS - 2) rc_wait_vm<(NS - 2) * INST + NST>();
                else rc_wait_vm<(NS - 2) * INST>();
                __builtin_amdgcn_s_barrier();
                issue(t + NS - 1);
                rc_stage_mfma(Ws + (t & (NS - 1)) * STAGE, foff, a2[kc * 4], a2[kc * 4 + 1], a2[kc * 4 + 2], a2[kc * 4 + 3], acc);
            }
            // epilogue of column tile nt: lane = row; acc[j][i * 4 + e] = column nt * 128 + j * 32 + i * 8 + khalf * 4 + e.
            // First the deferred 1 / rms of this lane's row (see above): H . W2^T = rstd2 * (U . W2^T)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[j][r] *= rstd2;
            if constexpr (!GLU2) {
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
                        const float mean = seg_s * (1.0f / D);
                        const float var = fmaxf(seg_q * (1.0f / D) - mean * mean, 0.f);
                        // both column halves of a row hold the same pair and store it to the same place (no exec-mask branch)
                        *reinterpret_cast<f32x2*>(seg_out + (nt / 3) * 2) = f32x2{mean, rsqrtf(var + p.seg_eps)};
                        seg_s = seg_q = 0.f;
                        seg_t = 0;
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if constexpr (GLU2) {
                    // 32 W rows = [gate16 | up16]: i = 0, 1 hold the gates of output columns j * 16 + i * 8 + khalf * 4 + e, i = 2, 3 their ups
                    unsigned q[2][2];
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        float v[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = ina_silu(acc[j][i * 4 + e]) * acc[j][(i + 2) * 4 + e];
                        q[i][0] = rc_pack2(v[0], v[1]);
                        q[i][1] = rc_pack2(v[2], v[3]);
                    }
                    const u32x4 o = rc_merge_rows(q[0][0], q[0][1], q[1][0], q[1][1]);
                    *reinterpret_cast<u32x4*>(Cb + crow + nt * 64 + j * 16 + khalf * 8) = o;
                } else {
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        unsigned q[2][2];
#pragma unroll
                        for (int i = 0; i < 2; ++i) {
                            q[i][0] = rc_pack2(acc[j][(2 * h + i) * 4], acc[j][(2 * h + i) * 4 + 1]);
                            q[i][1] = rc_pack2(acc[j][(2 * h + i) * 4 + 2], acc[j][(2 * h + i) * 4 + 3]);
                        }
                        const u32x4 o = rc_merge_rows(q[0][0], q[0][1], q[1][0], q[1][1]);
                        *reinterpret_cast<u32x4*>(Cb + crow + nt * 128 + j * 32 + h * 16 + khalf * 8) = o;
                    }
                }
            }
        }
        rc_wait_vm<0>();      // the dummy tail stages land in LDS: nothing of this workgroup is in flight when it retires
    }
}

template <int NW, int KC1, bool HAS2, bool GLU2>
int launch_rowchain(const ina_dit_rowchain_args& p, hipStream_t stream) {
    auto kern = dit_rowchain_kernel<NW, KC1, HAS2, GLU2>;
    static bool attr_done = false;
    if (!attr_done) {
        INA_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)RC_LDS));
        attr_done = true;
    }
    const double n2 = HAS2 ? (double)p.N2 : 0.0;
    InaProfScope prof(INA_PROF_GEMM, 2.0 * p.M * RC_D * (p.K1 + n2),
                      2.0 * p.M * p.K1 + 2.0 * RC_D * (p.K1 + n2) + 8.0 * p.M * RC_D + (p.H ? 2.0 * p.M * RC_D : 0.0) + 2.0 * p.M * (GLU2 ? n2 / 2 : n2), stream);
    hipLaunchKernelGGL(kern, dim3(p.M / (NW * 32)), dim3(NW * 64), RC_LDS, stream, p);
    INA_HIP_CHECK(hipGetLastError());
    return 0;
}

}  // namespace

int ina_launch_dit_rowchain(const ina_dit_rowchain_args& p_in, hipStream_t stream) {
    ina_dit_rowchain_args p = p_in;
    if (p.mod_div <= 0) p.mod_div = p.M;
    constexpr int nw = 4;
    INA_REQUIRE(p.A && p.W1 && p.X && p.gamma1, "dit_rowchain: A, W1, X and gamma1 are required");
    INA_REQUIRE(p.K1 == 384 || p.K1 == 1024, "dit_rowchain: K1=%d (built for 384 = attn2.to_out and 1024 = feed_forward.linear_2)", p.K1);
    INA_REQUIRE(p.M > 0 && p.M % (nw * 32) == 0 && p.mod_div % (nw * 32) == 0, "dit_rowchain: M=%d and mod_div=%d must be multiples of the %d-row panel", p.M,
                p.mod_div, nw * 32);
    INA_REQUIRE(p.lda % 8 == 0 && p.ldw1 % 8 == 0 && p.ldx % 4 == 0 && (!p.H || p.ldh % 8 == 0), "dit_rowchain: row strides must keep 16-byte alignment");
    INA_REQUIRE(((uintptr_t)p.A % 16) == 0 && ((uintptr_t)p.W1 % 16) == 0 && ((uintptr_t)p.X % 16) == 0 && ((uintptr_t)p.H % 16) == 0, "dit_rowchain: misaligned pointer");
    INA_REQUIRE((!p.gate && !p.mod_scale2) || p.mod_ld >= RC_D, "dit_rowchain: modulation needs mod_ld >= 384");
    INA_REQUIRE((size_t)384 * p.ldw1 < (1u << 30) && (!p.W2 || (size_t)p.N2 * p.ldw2 < (1u << 30)), "dit_rowchain: weight too large for 32-bit element offsets");
    INA_REQUIRE(p.W2 || !p.seg_stats, "dit_rowchain: seg_stats are statistics of the second GEMM's output rows");
    if (p.W2) {
        INA_REQUIRE(p.C2 && p.N2 > 0 && p.N2 % 128 == 0 && p.ldw2 % 8 == 0 && p.ldc2 % 8 == 0 && ((uintptr_t)p.W2 % 16) == 0 && ((uintptr_t)p.C2 % 16) == 0,
                    "dit_rowchain: second GEMM needs C2, N2 %% 128 == 0 and 16-byte aligned rows (N2=%d)", p.N2);
        INA_REQUIRE(!p.seg_stats || (!p.glu2 && p.N2 % 384 == 0 && ((uintptr_t)p.seg_stats % 8) == 0), "dit_rowchain: seg_stats needs a plain second GEMM with N2 %% 384 == 0 (N2=%d)", p.N2);
        INA_REQUIRE(p.glu2 ? p.K1 == 384 : p.K1 != 384, "dit_rowchain: built pairs are (K1 = 384, SwiGLU second GEMM) and (K1 = 1024, plain second GEMM)");
    }
    ina_prof_set_sub(42);
    if (p.K1 == 384) return !p.W2 ? launch_rowchain<4, 6, false, false>(p, stream) : launch_rowchain<4, 6, true, true>(p, stream);
    return !p.W2 ? launch_rowchain<4, 16, false, false>(p, stream) : launch_rowchain<4, 16, true, false>(p, stream);
}
