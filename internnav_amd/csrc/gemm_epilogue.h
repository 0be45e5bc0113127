// Fused GEMM epilogue shared by the tiled kernels (gemm.hip, gemm_glds.hip):
//   +bias[n] -> activation -> *colscale[n] (LayerScale) -> *rowscale[m] -> +residual[m,n] -> store bf16 or f32,
//   or GLU mode: neighbouring 16-column blocks (gate, up) -> act(gate) * up written to N/2 columns.
// Accumulators come from the swapped-operand MFMA (D[row = n_local][col = m_local]): lane holds output row m = .. + (lane & 15) and
// the 4 consecutive columns n = .. + (lane >> 4) * 4 + {0..3} of every 16x16 fragment.
#pragma once
#include "common.h"
#include "kernels.h"

template <int FM, int FN, int TM, int TN>
__device__ __forceinline__ void gemm_store_tile(const GemmArgs& p, f32x4 (&acc)[FM][FN], int m0, int n0, int wm, int wn, int lane) {
    // ---- epilogue: lane holds rows m = ..+ (lane&15), columns n = ..+ (lane>>4)*4 + {0..3}
    const int out_bf16 = (p.out_dtype == INA_DT_BF16);
    const float* __restrict__ bias = p.bias;
    const float* __restrict__ colscale = p.colscale;
    const float* __restrict__ rowscale = p.rowscale;
    const size_t cbatch = (size_t)blockIdx.y * p.strideC;
#pragma unroll
    for (int i = 0; i < FM; ++i) {
        const int m = m0 + wm * TM + i * 16 + (lane & 15);
        if (m >= p.M) continue;
        const float rs = rowscale ? rowscale[m / p.rowscale_div] : 1.0f;
        if (!p.glu) {
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                const int n = n0 + wn * TN + j * 16 + (lane >> 4) * 4;
                if (n >= p.N) continue;
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float x = acc[i][j][r];
                    if (bias) x += bias[n + r];
                    x = ina_act(x, p.act);
                    if (colscale) x *= colscale[n + r];
                    v[r] = x * rs;
                }
                if (p.R) {
                    const size_t ro = (size_t)blockIdx.y * p.strideR + (size_t)m * p.ldr + n;
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
                const size_t co = cbatch + (size_t)m * p.ldc + n;
                if (out_bf16) {
                    bf16x4 o = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
                    *reinterpret_cast<bf16x4*>(reinterpret_cast<bf16*>(p.C) + co) = o;
                } else {
                    f32x4 o = {v[0], v[1], v[2], v[3]};
                    *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.C) + co) = o;
                }
            }
        } else {
            // GLU: W rows are interleaved in 16-row blocks [gate16 | up16]; output column block = pair index
#pragma unroll
            for (int j = 0; j < FN; j += 2) {
                const int n = n0 + wn * TN + j * 16 + (lane >> 4) * 4;  // gate column in interleaved space
                if (n >= p.N) continue;
                const int no = ((n0 + wn * TN + j * 16) >> 1) + (lane >> 4) * 4;  // output column
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float g = acc[i][j][r], u = acc[i][j + 1][r];
                    if (bias) { g += bias[n + r]; u += bias[n + 16 + r]; }
                    v[r] = ina_act(g, p.act) * u * rs;
                }
                const size_t co = cbatch + (size_t)m * p.ldc + no;
                if (out_bf16) {
                    bf16x4 o = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
                    *reinterpret_cast<bf16x4*>(reinterpret_cast<bf16*>(p.C) + co) = o;
                } else {
                    f32x4 o = {v[0], v[1], v[2], v[3]};
                    *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.C) + co) = o;
                }
            }
        }
    }
}


// ---- LDS-transposed variant -------------------------------------------------------------------------------------------------
// The direct epilogue above does its per-element math (runtime activation switch, bias / scale / residual tests) inside fully
// unrolled register loops and stores 8-16 bytes per lane at a ROW stride: tens of thousands of instructions (far beyond the
// instruction cache) and quarter-line stores - it dominated small-K GEMMs (373 -> 1046 TFLOP/s at M=65536, N=1536, K=384 with the
// epilogue removed). Here every wave drops one raw 16-row accumulator slab at a time into a private 4 KiB LDS scratch
// (XOR-swizzled 16-byte chunks, conflict free) and re-reads it row-major: ALL epilogue math then runs in a small rolled loop on 8
// (bf16 out) or 4 (f32 out) consecutive columns per lane, with vector bias / scale / residual loads and whole-line 16-byte stores.
// Requires 16-byte aligned C (and R) rows and 64-column wave tiles; launchers fall back to the direct epilogue otherwise.
template <int N>
__device__ __forceinline__ void ina_act_vec(float (&v)[8], int act) {
    switch (act) {
        case INA_ACT_GELU_ERF:
#pragma unroll
            for (int q = 0; q < N; ++q) v[q] = ina_gelu_erf(v[q]);
            break;
        case INA_ACT_GELU_TANH:
#pragma unroll
            for (int q = 0; q < N; ++q) v[q] = ina_gelu_tanh(v[q]);
            break;
        case INA_ACT_RELU:
#pragma unroll
            for (int q = 0; q < N; ++q) v[q] = v[q] > 0.f ? v[q] : 0.f;
            break;
        case INA_ACT_SILU:
#pragma unroll
            for (int q = 0; q < N; ++q) v[q] = ina_silu(v[q]);
            break;
        default: break;
    }
}

// ALL = true: the wave's scratch holds all FM slabs at once ([FM][16][64] f32) and ONE rolled loop walks every slab and row group, so
// the epilogue math exists once in the instruction stream; ALL = false: 4 KiB scratch, the (rolled) row-group loop is emitted per slab.
template <int FM, int FN, int TM, int TN, bool ALL, bool OUT_BF16>
__device__ __forceinline__ void gemm_store_tile_staged_t(const GemmArgs& p, f32x4 (&acc)[FM][FN], int m0, int n0, int wm, int wn, int lane,
                                                       float* __restrict__ scratch /* this wave's [FM or 1][16][64] f32 */) {
    static_assert(TN == 64, "staged epilogue is written for 64-column wave tiles");
    constexpr bool out_bf16 = OUT_BF16;
    const size_t cbatch = (size_t)blockIdx.y * p.strideC, rbatch = (size_t)blockIdx.y * p.strideR;
    const int r16 = lane & 15, g = lane >> 4;
    constexpr int W = OUT_BF16 ? 8 : 4;                          // output columns per lane per pass (16 bytes)
    const int ncols = p.glu ? TN / 2 : TN;                       // output columns of this wave's tile
    const int nbase = p.glu ? ((n0 + wn * TN) >> 1) : (n0 + wn * TN);
    const int nout = p.glu ? p.N / 2 : p.N;
    const int cpr = ncols / W, rpp = 64 / cpr;                   // lanes per row, rows per pass
    const int rr0 = lane / cpr, cc = lane % cpr;
    if constexpr (ALL) {
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j) *reinterpret_cast<f32x4*>(scratch + i * 1024 + r16 * 64 + (((j * 4 + g) ^ r16) << 2)) = acc[i][j];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
    }
#pragma unroll
    for (int i = 0; i < (ALL ? 1 : FM); ++i) {
        if constexpr (!ALL) {
            // 1) raw accumulators -> scratch: logical [16][64] f32, 16-byte chunk c of row r stored at chunk position c ^ r
#pragma unroll
            for (int j = 0; j < FN; ++j) *reinterpret_cast<f32x4*>(scratch + r16 * 64 + (((j * 4 + g) ^ r16) << 2)) = acc[i][j];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
        }
        // 2) rolled loop over (slabs and) row groups: each lane owns W consecutive output columns of one row
#pragma unroll 1
        for (int r0 = 0; r0 < (ALL ? FM * 16 : 16); r0 += rpp) {
            const int slab = ALL ? (r0 >> 4) : i;
            const int rr = (r0 & 15) + rr0;
            const int m = m0 + wm * TM + slab * 16 + rr, n = nbase + cc * W;   // output column (GLU: in the halved space)
            float v[8];
            const float* srow = scratch + (ALL ? slab * 1024 : 0) + rr * 64;
            if (!p.glu) {
                const int c0 = (cc * W) >> 2;                     // first 16-byte chunk of this lane
                f32x4 a = *reinterpret_cast<const f32x4*>(srow + ((c0 ^ rr) << 2));
                v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3];
                if (out_bf16) {
                    f32x4 b = *reinterpret_cast<const f32x4*>(srow + (((c0 + 1) ^ rr) << 2));
                    v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
                }
                if (m < p.M && n < nout) {
                    const bool full = n + W <= nout;              // else: ragged last chunk (N % 8 == 4 with bf16 out), upper half unused
                    if (p.bias) {
#pragma unroll
                        for (int q = 0; q < W; ++q) if (q < 4 || full) v[q] += p.bias[n + q];
                    }
                    ina_act_vec<W>(v, p.act);
                    if (p.colscale) {
#pragma unroll
                        for (int q = 0; q < W; ++q) if (q < 4 || full) v[q] *= p.colscale[n + q];
                    }
                }
            } else {
                // GLU: output block jb (16 columns) pairs gate fragment 2*jb with up fragment 2*jb + 1 of the interleaved tile
                const int oc = cc * W, jb = oc >> 4, q0 = (oc & 15) >> 2;     // first 4-column chunk inside the block
                const int gc = (2 * jb) * 4 + q0, uc = (2 * jb + 1) * 4 + q0;
                f32x4 ga = *reinterpret_cast<const f32x4*>(srow + ((gc ^ rr) << 2)), ua = *reinterpret_cast<const f32x4*>(srow + ((uc ^ rr) << 2));
                float u[8];
                v[0] = ga[0]; v[1] = ga[1]; v[2] = ga[2]; v[3] = ga[3];
                u[0] = ua[0]; u[1] = ua[1]; u[2] = ua[2]; u[3] = ua[3];
                if (out_bf16) {
                    f32x4 gb = *reinterpret_cast<const f32x4*>(srow + (((gc + 1) ^ rr) << 2)), ub = *reinterpret_cast<const f32x4*>(srow + (((uc + 1) ^ rr) << 2));
                    v[4] = gb[0]; v[5] = gb[1]; v[6] = gb[2]; v[7] = gb[3];
                    u[4] = ub[0]; u[5] = ub[1]; u[6] = ub[2]; u[7] = ub[3];
                }
                if (m < p.M && n < nout) {
                    const int ni = n0 + wn * TN + jb * 32 + (oc & 15);        // gate column in the interleaved space
                    if (p.bias) {
#pragma unroll
                        for (int q = 0; q < W; ++q) { v[q] += p.bias[ni + q]; u[q] += p.bias[ni + 16 + q]; }
                    }
                    ina_act_vec<W>(v, p.act);
#pragma unroll
                    for (int q = 0; q < W; ++q) v[q] *= u[q];
                }
            }
            if (m < p.M && n < nout) {
                const int wv = (n + W <= nout) ? W : 4;
                if (p.rowscale) {
                    const float rs = p.rowscale[m / p.rowscale_div];
#pragma unroll
                    for (int q = 0; q < W; ++q) v[q] *= rs;
                }
                if (p.R) {
                    const size_t ro = rbatch + (size_t)m * p.ldr + n;
                    if (p.res_dtype == INA_DT_BF16) {
                        const bf16* R = reinterpret_cast<const bf16*>(p.R) + ro;
                        if (wv == 8) {
                            bf16x8 rv = *reinterpret_cast<const bf16x8*>(R);
                            for (int q = 0; q < 8; ++q) v[q] += (float)rv[q];
                        } else {
                            bf16x4 rv = *reinterpret_cast<const bf16x4*>(R);
                            for (int q = 0; q < 4; ++q) v[q] += (float)rv[q];
                        }
                    } else {
                        const float* R = reinterpret_cast<const float*>(p.R) + ro;
                        f32x4 ra = *reinterpret_cast<const f32x4*>(R);
                        for (int q = 0; q < 4; ++q) v[q] += ra[q];
                        if (wv == 8) {
                            f32x4 rb = *reinterpret_cast<const f32x4*>(R + 4);
                            for (int q = 0; q < 4; ++q) v[4 + q] += rb[q];
                        }
                    }
                }
                const size_t co = cbatch + (size_t)m * p.ldc + n;
                if (out_bf16) {
                    if (wv == 8) {
                        bf16x8 o;
                        for (int q = 0; q < 8; ++q) o[q] = (bf16)v[q];
                        *reinterpret_cast<bf16x8*>(reinterpret_cast<bf16*>(p.C) + co) = o;
                    } else {
                        *reinterpret_cast<bf16x4*>(reinterpret_cast<bf16*>(p.C) + co) = bf16x4{(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
                    }
                } else {
                    *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.C) + co) = f32x4{v[0], v[1], v[2], v[3]};
                }
            }
        }
        if constexpr (!ALL) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
        }
    }
}

template <int FM, int FN, int TM, int TN, bool ALL>
__device__ __forceinline__ void gemm_store_tile_staged(const GemmArgs& p, f32x4 (&acc)[FM][FN], int m0, int n0, int wm, int wn, int lane,
                                                       float* __restrict__ scratch) {
    if (p.out_dtype == INA_DT_BF16) gemm_store_tile_staged_t<FM, FN, TM, TN, ALL, true>(p, acc, m0, n0, wm, wn, lane, scratch);
    else gemm_store_tile_staged_t<FM, FN, TM, TN, ALL, false>(p, acc, m0, n0, wm, wn, lane, scratch);
}
