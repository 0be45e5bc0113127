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
