// Backward / optimiser kernels of the SFT step (reference: the autograd of internvla_n1.py:222-286 under
// HF Trainer + adamw_torch, train_dual_system.sh:72-77). The GEMMs of the backward pass reuse the forward MFMA kernels
// (dX = dY . W and dW = dY^T . X are "x @ w.T" products of transposed copies, see ina_transpose); everything here is the
// HBM-bound rest: element-wise derivatives, normalisation backward, column reductions (bias / gain / modulation gradients),
// the flat fused AdamW with global-norm clipping, the flow-matching loss, and the weight-streaming dX of the frozen LLM rows.
// All tensors carry a run-time dtype flag (bf16 | f32): these kernels move bytes, the branch is uniform.
#include "common.h"
#include "kernels.h"

namespace {

__device__ __forceinline__ float ldf(const void* p, int dt, size_t i) {
    return dt == INA_DT_BF16 ? (float)reinterpret_cast<const bf16*>(p)[i] : reinterpret_cast<const float*>(p)[i];
}
__device__ __forceinline__ void stf(void* p, int dt, size_t i, float v) {
    if (dt == INA_DT_BF16) reinterpret_cast<bf16*>(p)[i] = (bf16)v;
    else reinterpret_cast<float*>(p)[i] = v;
}

__device__ __forceinline__ float act_grad(float x, int act) {
    switch (act) {
        case INA_ACT_GELU_ERF: {
            const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
            return cdf + x * 0.3989422804014327f * __expf(-0.5f * x * x);
        }
        case INA_ACT_GELU_TANH: {
            const float k0 = 0.7978845608028654f, k1 = 0.044715f;
            const float t = tanhf(k0 * (x + k1 * x * x * x));
            return 0.5f * (1.0f + t) + 0.5f * x * (1.0f - t * t) * k0 * (1.0f + 3.0f * k1 * x * x);
        }
        case INA_ACT_RELU: return x > 0.f ? 1.f : 0.f;
        case INA_ACT_SILU: {
            const float s = 1.0f / (1.0f + __expf(-x));
            return s * (1.0f + x * (1.0f - s));
        }
        case INA_ACT_TANH: {
            const float t = tanhf(x);
            return 1.0f - t * t;
        }
        default: return 1.f;
    }
}

__device__ __forceinline__ float scale_fn(float s, int f) { return f == 1 ? 1.0f + s : (f == 2 ? tanhf(s) : s); }

// ---------------------------------------------------------------------------------------------------------------- element-wise
__global__ __launch_bounds__(256) void ew_kernel(EwArgs p) {
    const size_t n = (size_t)p.rows * p.C;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const int r = (int)(i / p.C), c = (int)(i % p.C);
        float y = 0.f, y2 = 0.f;
        switch (p.op) {
            case INA_EW_AFFINE: {
                y = ldf(p.A, p.a_dt, (size_t)r * p.lda + c);
                if (p.S) y *= scale_fn(ldf(p.S, p.s_dt, (size_t)(r / p.s_div) * p.lds + c), p.s_f);
                if (p.B) y += ldf(p.B, p.b_dt, (size_t)r * p.ldb + c);
                if (p.tab) y += p.tab[(size_t)(r % p.tab_mod) * p.C + c];
                break;
            }
            case INA_EW_ACT_FWD: y = ina_act(ldf(p.A, p.a_dt, (size_t)r * p.lda + c), p.act); break;
            case INA_EW_ACT_BWD: y = ldf(p.B, p.b_dt, (size_t)r * p.ldb + c) * act_grad(ldf(p.A, p.a_dt, (size_t)r * p.lda + c), p.act); break;
            case INA_EW_GLU_FWD: y = ina_silu(ldf(p.A, p.a_dt, (size_t)r * p.lda + c)) * ldf(p.B, p.b_dt, (size_t)r * p.ldb + c); break;
            case INA_EW_GLU_BWD: {
                const float av = ldf(p.A, p.a_dt, (size_t)r * p.lda + c), bv = ldf(p.B, p.b_dt, (size_t)r * p.ldb + c);
                const float dy = ldf(p.D, p.d_dt, (size_t)r * p.ldd + c);
                y = dy * bv * act_grad(av, INA_ACT_SILU);
                y2 = dy * ina_silu(av);
                break;
            }
            case INA_EW_DROPOUT:
                y = ina_hash(p.drop_seed + (p.drop_salt ? *p.drop_salt : 0u), (uint64_t)r * p.C + c) >= p.drop_thresh ? ldf(p.A, p.a_dt, (size_t)r * p.lda + c) * p.drop_scale : 0.f;
                break;
            default: break;
        }
        if (p.accumulate) y += ldf(p.Y, p.y_dt, (size_t)r * p.ldy + c);
        stf(p.Y, p.y_dt, (size_t)r * p.ldy + c, y);
        if (p.op == INA_EW_GLU_BWD) stf(p.Y2, p.y2_dt, (size_t)r * p.ldy2 + c, y2);
    }
}

__device__ __forceinline__ f32x4 ld4(const void* p, int dt, size_t i) {
    if (dt == INA_DT_BF16) {
        const bf16x4 t = *reinterpret_cast<const bf16x4*>(reinterpret_cast<const bf16*>(p) + i);
        return f32x4{(float)t[0], (float)t[1], (float)t[2], (float)t[3]};
    }
    return *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(p) + i);
}
__device__ __forceinline__ void st4(void* p, int dt, size_t i, f32x4 v) {
    if (dt == INA_DT_BF16) *reinterpret_cast<bf16x4*>(reinterpret_cast<bf16*>(p) + i) = bf16x4{(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
    else *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p) + i) = v;
}
// the same operations on 4 consecutive columns per thread (8- / 16-byte accesses) when every operand allows it
__global__ __launch_bounds__(256) void ew_vec_kernel(EwArgs p) {
    const int C4 = p.C / 4;
    const size_t n = (size_t)p.rows * C4;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const int r = (int)(i / C4), c = (int)(i % C4) * 4;
        f32x4 y = f32x4{0.f, 0.f, 0.f, 0.f}, y2 = y;
        const f32x4 av = ld4(p.A, p.a_dt, (size_t)r * p.lda + c);
        switch (p.op) {
            case INA_EW_AFFINE: {
                y = av;
                if (p.S) {
                    const f32x4 sv = ld4(p.S, p.s_dt, (size_t)(r / p.s_div) * p.lds + c);
#pragma unroll
                    for (int j = 0; j < 4; ++j) y[j] *= scale_fn(sv[j], p.s_f);
                }
                if (p.B) y += ld4(p.B, p.b_dt, (size_t)r * p.ldb + c);
                if (p.tab) y += *reinterpret_cast<const f32x4*>(p.tab + (size_t)(r % p.tab_mod) * p.C + c);
                break;
            }
            case INA_EW_ACT_FWD:
#pragma unroll
                for (int j = 0; j < 4; ++j) y[j] = ina_act(av[j], p.act);
                break;
            case INA_EW_ACT_BWD: {
                const f32x4 bv = ld4(p.B, p.b_dt, (size_t)r * p.ldb + c);
#pragma unroll
                for (int j = 0; j < 4; ++j) y[j] = bv[j] * act_grad(av[j], p.act);
                break;
            }
            case INA_EW_GLU_FWD: {
                const f32x4 bv = ld4(p.B, p.b_dt, (size_t)r * p.ldb + c);
#pragma unroll
                for (int j = 0; j < 4; ++j) y[j] = ina_silu(av[j]) * bv[j];
                break;
            }
            case INA_EW_GLU_BWD: {
                const f32x4 bv = ld4(p.B, p.b_dt, (size_t)r * p.ldb + c), dy = ld4(p.D, p.d_dt, (size_t)r * p.ldd + c);
#pragma unroll
                for (int j = 0; j < 4; ++j) { y[j] = dy[j] * bv[j] * act_grad(av[j], INA_ACT_SILU); y2[j] = dy[j] * ina_silu(av[j]); }
                break;
            }
            case INA_EW_DROPOUT:
#pragma unroll
                for (int j = 0; j < 4; ++j) y[j] = ina_hash(p.drop_seed + (p.drop_salt ? *p.drop_salt : 0u), (uint64_t)r * p.C + c + j) >= p.drop_thresh ? av[j] * p.drop_scale : 0.f;
                break;
            default: break;
        }
        if (p.accumulate) y += ld4(p.Y, p.y_dt, (size_t)r * p.ldy + c);
        st4(p.Y, p.y_dt, (size_t)r * p.ldy + c, y);
        if (p.op == INA_EW_GLU_BWD) st4(p.Y2, p.y2_dt, (size_t)r * p.ldy2 + c, y2);
    }
}

// ---------------------------------------------------------------------------------------------------------------- column sums
// out[g, c] (+)= scale * sum over the rows of group g of X[r, c] * X2[r, c]; chunks of `chunk_rows` rows per workgroup, deterministic two-stage.
// Chunk = 32 rows up to 2048 rows per group, 64 up to 4096, 128 up to 8192, 256 beyond (at most 64 chunks below 8192 rows): the gradients of the SFT
// step are sums over 24 ... 768 rows, where 256-row chunks meant 2-6 workgroups per launch, each a chain of 16 dependent memory round trips (33 us for
// 1 MB of input); the big reductions (the squared gradient norm: 95 k rows) keep their 256-row chunks.
inline int colsum_chunk_rows(int group_rows) {
    return group_rows <= 32 ? group_rows : group_rows <= 2048 ? 32 : group_rows <= 4096 ? 64 : group_rows <= 8192 ? 128 : 256;
}
__global__ __launch_bounds__(256) void colsum_kernel(ColsumArgs p, int nchunk, int group_rows, int chunk_rows) {
    __shared__ float red[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane, chunk = blockIdx.y, grp = blockIdx.z;
    const int r0 = grp * group_rows + chunk * chunk_rows, r1 = min(grp * group_rows + group_rows, r0 + chunk_rows);
    float acc = 0.f;
    if (c < p.C) {
        // 8 independent loads in flight per lane: the loop is latency-bound otherwise (64 dependent round trips per workgroup)
        constexpr int U = 8;
        for (int r = r0 + wave; r < r1; r += 4 * U) {
            float v[U], w[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int rr = r + 4 * u;
                const bool ok = rr < r1;
                v[u] = ok ? ldf(p.X, p.x_dt, (size_t)rr * p.ldx + (size_t)c * p.x_cs) : 0.f;
                w[u] = (ok && p.X2) ? ldf(p.X2, p.x2_dt, (size_t)rr * p.ldx2 + (size_t)c * p.x2_cs) : 1.f;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) acc += v[u] * w[u];
        }
    }
    red[wave][lane] = acc;
    __syncthreads();
    if (wave == 0 && c < p.C) {
        const float s = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
        if (nchunk == 1) {
            float* o = p.out + (size_t)grp * p.ldo + (size_t)c * p.out_cs;
            *o = (p.accumulate ? *o : 0.f) + s * p.scale;
        } else {
            p.partial[((size_t)grp * nchunk + chunk) * p.C + c] = s;
        }
    }
}
// 4 columns per lane (8- / 16-byte loads): the dense case of every bias / gain / modulation gradient. XB: X is bf16; X2M: 0 none, 1 bf16, 2 f32
template <bool XB, int X2M>
__global__ __launch_bounds__(256) void colsum_vec_kernel(ColsumArgs p, int nchunk, int group_rows, int chunk_rows) {
    __shared__ float red[4][64][4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = (blockIdx.x * 64 + lane) * 4, chunk = blockIdx.y, grp = blockIdx.z;
    const int r0 = grp * group_rows + chunk * chunk_rows, r1 = min(grp * group_rows + group_rows, r0 + chunk_rows);
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    if (c < p.C) {
        constexpr int U = 8;     // a 32-row chunk = one round trip per wave (8 rows each, all 8 loads in flight)
        for (int r = r0 + wave; r < r1; r += 4 * U) {
            f32x4 v[U], w[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int rr = r + 4 * u;
                v[u] = f32x4{0.f, 0.f, 0.f, 0.f};
                w[u] = f32x4{1.f, 1.f, 1.f, 1.f};
                if (rr < r1) {
                    if constexpr (XB) {
                        const bf16x4 t = *reinterpret_cast<const bf16x4*>(reinterpret_cast<const bf16*>(p.X) + (size_t)rr * p.ldx + c);
                        v[u] = f32x4{(float)t[0], (float)t[1], (float)t[2], (float)t[3]};
                    } else {
                        v[u] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(p.X) + (size_t)rr * p.ldx + c);
                    }
                    if constexpr (X2M == 1) {
                        const bf16x4 t = *reinterpret_cast<const bf16x4*>(reinterpret_cast<const bf16*>(p.X2) + (size_t)rr * p.ldx2 + c);
                        w[u] = f32x4{(float)t[0], (float)t[1], (float)t[2], (float)t[3]};
                    } else if constexpr (X2M == 2) {
                        w[u] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(p.X2) + (size_t)rr * p.ldx2 + c);
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] += v[u][j] * w[u][j];
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) red[wave][lane][j] = acc[j];
    __syncthreads();
    if (wave == 0 && c < p.C) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float sm = (red[0][lane][j] + red[1][lane][j]) + (red[2][lane][j] + red[3][lane][j]);
            if (nchunk == 1) {
                float* o = p.out + (size_t)grp * p.ldo + c + j;
                *o = (p.accumulate ? *o : 0.f) + sm * p.scale;
            } else {
                p.partial[((size_t)grp * nchunk + chunk) * p.C + c + j] = sm;
            }
        }
    }
}
__global__ __launch_bounds__(256) void colsum_final_kernel(ColsumArgs p, int nchunk) {
    const int c = blockIdx.x * 256 + threadIdx.x, grp = blockIdx.y;
    if (c >= p.C) return;
    float s = 0.f;
    const float* part = p.partial + (size_t)grp * nchunk * p.C + c;
    int k = 0;
    for (; k + 8 <= nchunk; k += 8) {      // 8 loads in flight, added in ascending chunk order
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = part[(size_t)(k + u) * p.C];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; k < nchunk; ++k) s += part[(size_t)k * p.C];
    float* o = p.out + (size_t)grp * p.ldo + (size_t)c * p.out_cs;
    *o = (p.accumulate ? *o : 0.f) + s * p.scale;
}

// ---------------------------------------------------------------------------------------------------------------- norm backward
// LayerNorm / RMSNorm backward of y = xhat * gamma (+ beta): one wave per row, statistics recomputed from x.
//   g = dy * gamma;  LN: dx = rstd * (g - mean(g) - xhat * mean(g * xhat));  RMS: dx = rstd * (g - xhat * mean(g * xhat))
// xhat (bf16) is written when asked for: dgamma = colsum(dy * xhat), dbeta = colsum(dy).
__global__ __launch_bounds__(256) void norm_bwd_kernel(NormBwdArgs p) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= p.rows) return;
    const size_t xo = (size_t)row * p.ldx, go = (size_t)row * p.lddy;
    float mean = 0.f;
    if (!p.rms) {
        float s = 0.f;
        for (int c = lane; c < p.C; c += 64) s += ldf(p.X, p.x_dt, xo + c);
        mean = wave_sum(s) / p.C;
    }
    float v = 0.f;
    for (int c = lane; c < p.C; c += 64) {
        const float d = ldf(p.X, p.x_dt, xo + c) - mean;
        v += d * d;
    }
    const float rstd = rsqrtf(wave_sum(v) / p.C + p.eps);
    float s1 = 0.f, s2 = 0.f;
    for (int c = lane; c < p.C; c += 64) {
        const float xh = (ldf(p.X, p.x_dt, xo + c) - mean) * rstd;
        const float g = ldf(p.DY, p.dy_dt, go + c) * (p.gamma ? p.gamma[c] : 1.f);
        s1 += g;
        s2 += g * xh;
    }
    s1 = p.rms ? 0.f : wave_sum(s1) / p.C;
    s2 = wave_sum(s2) / p.C;
    for (int c = lane; c < p.C; c += 64) {
        const float xh = (ldf(p.X, p.x_dt, xo + c) - mean) * rstd;
        const float g = ldf(p.DY, p.dy_dt, go + c) * (p.gamma ? p.gamma[c] : 1.f);
        float dx = rstd * (g - s1 - xh * s2);
        const size_t o = (size_t)row * p.lddx + c;
        if (p.accumulate) dx += ldf(p.DX, p.dx_dt, o);
        stf(p.DX, p.dx_dt, o, dx);
        if (p.XHAT) reinterpret_cast<bf16*>(p.XHAT)[(size_t)row * p.ldxh + c] = (bf16)xh;
    }
}

// 4 columns per lane, the row kept in registers (C <= 4096): one pass over x / dy instead of four
template <bool XB, bool DYB>
__global__ __launch_bounds__(256) void norm_bwd_vec_kernel(NormBwdArgs p) {
    constexpr int MAXV = 16;                                   // 16 x 4 x 64 = 4096 columns
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= p.rows) return;
    f32x4 xv[MAXV], gv[MAXV];
    const int nv = (p.C / 4 + 63) / 64;
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = (i * 64 + lane) * 4;
        xv[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        gv[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (i < nv && c < p.C) {
            if constexpr (XB) {
                const bf16x4 t = *reinterpret_cast<const bf16x4*>(reinterpret_cast<const bf16*>(p.X) + (size_t)row * p.ldx + c);
                xv[i] = f32x4{(float)t[0], (float)t[1], (float)t[2], (float)t[3]};
            } else {
                xv[i] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(p.X) + (size_t)row * p.ldx + c);
            }
            if constexpr (DYB) {
                const bf16x4 t = *reinterpret_cast<const bf16x4*>(reinterpret_cast<const bf16*>(p.DY) + (size_t)row * p.lddy + c);
                gv[i] = f32x4{(float)t[0], (float)t[1], (float)t[2], (float)t[3]};
            } else {
                gv[i] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(p.DY) + (size_t)row * p.lddy + c);
            }
            if (p.gamma) {
                const f32x4 gm = *reinterpret_cast<const f32x4*>(p.gamma + c);
                gv[i] *= gm;
            }
            s += (xv[i][0] + xv[i][1]) + (xv[i][2] + xv[i][3]);
        }
    }
    const float mean = p.rms ? 0.f : wave_sum(s) / p.C;
    float v = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = (i * 64 + lane) * 4;
        if (i < nv && c < p.C) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { xv[i][j] -= mean; v += xv[i][j] * xv[i][j]; }
        }
    }
    const float rstd = rsqrtf(wave_sum(v) / p.C + p.eps);
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = (i * 64 + lane) * 4;
        if (i < nv && c < p.C) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { xv[i][j] *= rstd; s1 += gv[i][j]; s2 += gv[i][j] * xv[i][j]; }
        }
    }
    s1 = p.rms ? 0.f : wave_sum(s1) / p.C;
    s2 = wave_sum(s2) / p.C;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = (i * 64 + lane) * 4;
        if (i < nv && c < p.C) {
            f32x4 dx;
#pragma unroll
            for (int j = 0; j < 4; ++j) dx[j] = rstd * (gv[i][j] - s1 - xv[i][j] * s2);
            const size_t o = (size_t)row * p.lddx + c;
            if (p.dx_dt == INA_DT_BF16) {
                bf16* d = reinterpret_cast<bf16*>(p.DX) + o;
                if (p.accumulate) { const bf16x4 t = *reinterpret_cast<const bf16x4*>(d); for (int j = 0; j < 4; ++j) dx[j] += (float)t[j]; }
                *reinterpret_cast<bf16x4*>(d) = bf16x4{(bf16)dx[0], (bf16)dx[1], (bf16)dx[2], (bf16)dx[3]};
            } else {
                float* d = reinterpret_cast<float*>(p.DX) + o;
                if (p.accumulate) dx += *reinterpret_cast<const f32x4*>(d);
                *reinterpret_cast<f32x4*>(d) = dx;
            }
            if (p.XHAT) *reinterpret_cast<bf16x4*>(reinterpret_cast<bf16*>(p.XHAT) + (size_t)row * p.ldxh + c) =
                            bf16x4{(bf16)xv[i][0], (bf16)xv[i][1], (bf16)xv[i][2], (bf16)xv[i][3]};
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------- transpose
// Y[c, r] = bf16(X[r, c]) for r < rows, 0 for rows <= r < ldy (the padded K dimension of the GEMM that consumes it)
__global__ __launch_bounds__(256) void transpose_kernel(TransposeArgs p) {
    __shared__ float tile[64][65];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int r0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int r = r0 + ty + 4 * i, c = c0 + tx;
        tile[ty + 4 * i][tx] = (r < p.rows && c < p.cols) ? ldf(p.X, p.x_dt, (size_t)r * p.ldx + c) : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int c = c0 + ty + 4 * i, r = r0 + tx;
        if (c < p.cols && r < p.ldy) reinterpret_cast<bf16*>(p.Y)[(size_t)c * p.ldy + r] = (bf16)tile[tx][ty + 4 * i];
    }
}

// ---------------------------------------------------------------------------------------------------------------- sparse row mix
// out[t, :] (+)= sum_j coef[t, j] * in[idx[t, j], :]   (the bicubic position-embedding interpolation of DINOv2 and its transpose)
__global__ __launch_bounds__(256) void sparse_rows_kernel(SparseRowsArgs p) {
    const int t = blockIdx.x;
    for (int c = threadIdx.x; c < p.C; c += 256) {
        float s = 0.f;
        for (int j = 0; j < p.taps; ++j) {
            const int src = p.idx[(size_t)t * p.taps + j];
            if (src >= 0) s += p.coef[(size_t)t * p.taps + j] * p.in[(size_t)src * p.C + c];
        }
        float* o = p.out + (size_t)t * p.C + c;
        *o = (p.accumulate ? *o : 0.f) + s;
    }
}

// ---------------------------------------------------------------------------------------------------------------- tiny linear
// Y[r, n] = sum_k X[r, k] * W[n * w_ns + k * w_ks] + bias[n] + tab[r % tab_mod, n]  (nn.Linear(3, 384) / (384, 3) and their dX)
__global__ __launch_bounds__(256) void small_linear_kernel(SmallLinearArgs p) {
    const size_t n_out = (size_t)p.rows * p.N;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n_out; i += (size_t)gridDim.x * 256) {
        const int r = (int)(i / p.N), n = (int)(i % p.N);
        float s = p.bias ? p.bias[n] : 0.f;
        for (int k = 0; k < p.K; ++k) s += ldf(p.X, p.x_dt, (size_t)r * p.ldx + k) * p.W[(size_t)n * p.w_ns + (size_t)k * p.w_ks];
        if (p.tab) s += p.tab[(size_t)(r % p.tab_mod) * p.N + n];
        stf(p.Y, p.y_dt, (size_t)r * p.ldy + n, s);
    }
}

// ---------------------------------------------------------------------------------------------------------------- masked MSE
// loss = sum_s mask[s] * sum_{t,d} (pred - target)^2 / (sum_s mask[s] * T * D)   (internvla_n1.py:283-286); dpred = d loss / d pred
__global__ __launch_bounds__(256) void mse_kernel(MseArgs p) {
    __shared__ float red[256];
    const int tid = threadIdx.x;
    float ms = 0.f;
    for (int s = tid; s < p.nseq; s += 256) ms += p.mask[s];
    red[tid] = ms;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (tid < o) red[tid] += red[tid + o]; __syncthreads(); }
    const float denom = red[0] * p.T * p.D;
    __syncthreads();
    const float inv = denom > 0.f ? 1.0f / denom : 0.f;
    float acc = 0.f;
    const int n = p.nseq * p.T * p.D;
    for (int i = tid; i < n; i += 256) {
        const int r = i / p.D, d = i % p.D;
        const float m = p.mask[r / p.T];
        const float e = ldf(p.pred, p.pred_dt, (size_t)r * p.ldp + d) - p.target[(size_t)r * p.D + d];
        acc += m * e * e;
        if (p.dpred) stf(p.dpred, p.dpred_dt, (size_t)r * p.lddp + d, 2.0f * m * e * inv * p.loss_scale);
    }
    red[tid] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (tid < o) red[tid] += red[tid + o]; __syncthreads(); }
    if (tid == 0) p.loss[0] = red[0] * inv;
}

// ---------------------------------------------------------------------------------------------------------------- AdamW
// torch.optim.AdamW (adamw_torch) on one flat fp32 buffer, fused with clip_grad_norm_ (coefficient from the sum-of-squares partials
// of the gradient, no host round trip), the gradient averaging over data-parallel ranks and the bf16 working copy of the weights.
__global__ __launch_bounds__(256) void adamw_kernel(AdamwArgs p) {
    __shared__ float red[256];
    float s = 0.f;
    for (int i = threadIdx.x; i < p.n_parts; i += 256) s += p.sumsq_parts[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
    const float total = p.n_parts > 0 ? sqrtf(red[0]) * p.grad_scale : 0.f;
    float clip = 1.0f;
    if (p.max_norm > 0.f) clip = fminf(1.0f, p.max_norm / (total + 1e-6f));
    if (p.norm_out && blockIdx.x == 0 && threadIdx.x == 0) p.norm_out[0] = total;
    const float gs = p.grad_scale * clip;
    const float step = p.lr / p.bc1, rbc2 = rsqrtf(p.bc2);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < (size_t)p.n; i += (size_t)gridDim.x * 256) {
        const float g = p.g[i] * gs;
        float w = p.p[i] * (1.0f - p.lr * p.wd);
        const float m = p.beta1 * p.m[i] + (1.0f - p.beta1) * g;
        const float v = p.beta2 * p.v[i] + (1.0f - p.beta2) * g * g;
        w -= step * m / (sqrtf(v) * rbc2 + p.eps);
        p.p[i] = w; p.m[i] = m; p.v[i] = v;
        if (p.p_bf16) reinterpret_cast<bf16*>(p.p_bf16)[i] = (bf16)w;
        if (p.zero_grad) p.g[i] = 0.f;
    }
}

// ---------------------------------------------------------------------------------------------------------------- skinny dX
// part[split, m, k] = sum over the split's n of X[m, n] * W[n, k]: the input gradient of a frozen nn.Linear for a handful of rows
// (the latent-query rows of the LLM), streaming W [N, K] once in its stored layout at HBM rate. M <= 16.
constexpr int NN_COLS = 512;   // 64 lanes x 8 columns
template <int MR>
__global__ __launch_bounds__(256) void gemm_nn_kernel(GemmNnArgs p, int rows_per_split) {
    extern __shared__ float xs[];   // [rows_per_split][MR]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int k0 = blockIdx.x * NN_COLS + lane * 8, split = blockIdx.y;
    const int n0 = split * rows_per_split, n1 = min(p.N, n0 + rows_per_split);
    for (int i = threadIdx.x; i < rows_per_split * MR; i += 256) {
        const int n = n0 + i / MR, m = i % MR;
        xs[i] = (n < n1 && m < p.M) ? (float)reinterpret_cast<const bf16*>(p.X)[(size_t)m * p.ldx + n] : 0.f;
    }
    __syncthreads();
    float acc[MR][8];
#pragma unroll
    for (int m = 0; m < MR; ++m)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[m][j] = 0.f;
    if (k0 < p.K) {
        const bf16* W = reinterpret_cast<const bf16*>(p.W);
        for (int n = n0 + wave; n < n1; n += 4) {
            const bf16x8 w = *reinterpret_cast<const bf16x8*>(W + (size_t)n * p.ldw + k0);
            const float* xr = xs + (size_t)(n - n0) * MR;
#pragma unroll
            for (int m = 0; m < MR; ++m) {
                const float x = xr[m];
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[m][j] += x * (float)w[j];
            }
        }
    }
    // each wave owns its own partial slot (4 per split); the caller reduces the slots with ina_colsum (deterministic order)
    if (k0 < p.K) {
        float* o = p.partial + ((size_t)(split * 4 + wave) * MR) * p.K + k0;
#pragma unroll
        for (int m = 0; m < MR; ++m)
#pragma unroll
            for (int j = 0; j < 8; ++j) o[(size_t)m * p.K + j] = acc[m][j];
    }
}

}  // namespace

int ina_launch_ew(const EwArgs& p, hipStream_t stream) {
    INA_REQUIRE(p.rows > 0 && p.C > 0 && p.A && p.Y, "ew: empty problem or null tensor");
    INA_REQUIRE(p.op >= INA_EW_AFFINE && p.op <= INA_EW_DROPOUT, "ew: unknown op %d", p.op);
    INA_REQUIRE(p.op == INA_EW_AFFINE || p.op == INA_EW_ACT_FWD || p.op == INA_EW_DROPOUT || p.B, "ew: op %d needs the second input", p.op);
    INA_REQUIRE(p.op != INA_EW_GLU_BWD || (p.D && p.Y2), "ew: glu backward needs dy and the second output");
    INA_REQUIRE(!p.S || p.s_div > 0, "ew: s_div must be positive");
    INA_REQUIRE(!p.tab || p.tab_mod > 0, "ew: tab_mod must be positive");
    auto ok4 = [](const void* ptr, int dt, int ld) { return !ptr || (ld % 4 == 0 && ((uintptr_t)ptr % (dt == INA_DT_BF16 ? 8 : 16)) == 0); };
    const bool vec = p.C % 4 == 0 && ok4(p.A, p.a_dt, p.lda) && ok4(p.B, p.b_dt, p.ldb) && ok4(p.D, p.d_dt, p.ldd) && ok4(p.S, p.s_dt, p.lds) &&
                     ok4(p.Y, p.y_dt, p.ldy) && ok4(p.Y2, p.y2_dt, p.ldy2) && ok4(p.tab, INA_DT_F32, 4);
    const size_t n = vec ? (size_t)p.rows * (p.C / 4) : (size_t)p.rows * p.C;
    const int blocks = (int)((n + 255) / 256 < 8192 ? (n + 255) / 256 : 8192);
    InaProfScope prof(INA_PROF_ELEMENTWISE, 0.0, 0.0, stream);
    if (vec) hipLaunchKernelGGL(ew_vec_kernel, dim3(blocks), dim3(256), 0, stream, p);
    else hipLaunchKernelGGL(ew_kernel, dim3(blocks), dim3(256), 0, stream, p);
    INA_HIP_CHECK(hipGetLastError());
    return 0;
}

int ina_launch_colsum(const ColsumArgs& p_in, hipStream_t stream) {
    ColsumArgs p = p_in;
    INA_REQUIRE(p.rows > 0 && p.C > 0 && p.X && p.out, "colsum: empty problem or null tensor");
    const int group_rows = p.group_rows > 0 ? p.group_rows : p.rows;
    INA_REQUIRE(p.rows % group_rows == 0, "colsum: rows %d not a multiple of group_rows %d", p.rows, group_rows);
    if (p.x_cs == 0 && p.X2 == nullptr) p.x_cs = 1;
    if (p.out_cs <= 0) p.out_cs = 1;
    if (p.ldo <= 0) p.ldo = p.C * p.out_cs;
    if (p.scale == 0.f) p.scale = 1.f;
    const int chunk_rows = colsum_chunk_rows(group_rows);
    const int groups = p.rows / group_rows, nchunk = (group_rows + chunk_rows - 1) / chunk_rows;
    INA_REQUIRE(nchunk == 1 || (p.partial && p.partial_elems >= (int64_t)groups * nchunk * p.C),
                "colsum: needs a partial buffer of %lld floats", (long long)groups * nchunk * p.C);
    InaProfScope prof(INA_PROF_ELEMENTWISE, 0.0, 0.0, stream);
    const size_t xes = p.x_dt == INA_DT_BF16 ? 2 : 4, x2es = p.x2_dt == INA_DT_BF16 ? 2 : 4;
    const bool vec = p.x_cs == 1 && p.out_cs == 1 && p.C % 4 == 0 && p.ldx % 4 == 0 && ((uintptr_t)p.X % (4 * xes)) == 0 &&
                     (!p.X2 || (p.x2_cs == 1 && p.ldx2 % 4 == 0 && ((uintptr_t)p.X2 % (4 * x2es)) == 0));
    if (vec) {
        const dim3 grid((p.C + 255) / 256, nchunk, groups);
        const int x2m = !p.X2 ? 0 : (p.x2_dt == INA_DT_BF16 ? 1 : 2);
        const bool xb = p.x_dt == INA_DT_BF16;
#define INA_CS(XB, M) hipLaunchKernelGGL((colsum_vec_kernel<XB, M>), grid, dim3(256), 0, stream, p, nchunk, group_rows, chunk_rows)
        if (xb) { if (x2m == 0) INA_CS(true, 0); else if (x2m == 1) INA_CS(true, 1); else INA_CS(true, 2); }
        else { if (x2m == 0) INA_CS(false, 0); else if (x2m == 1) INA_CS(false, 1); else INA_CS(false, 2); }
#undef INA_CS
    } else {
        hipLaunchKernelGGL(colsum_kernel, dim3((p.C + 63) / 64, nchunk, groups), dim3(256), 0, stream, p, nchunk, group_rows, chunk_rows);
    }
    if (nchunk > 1) hipLaunchKernelGGL(colsum_final_kernel, dim3((p.C + 255) / 256, groups), dim3(256), 0, stream, p, nchunk);
    INA_HIP_CHECK(hipGetLastError());
    return 0;
}

int ina_launch_norm_bwd(const NormBwdArgs& p, hipStream_t stream) {
    INA_REQUIRE(p.rows > 0 && p.C > 0 && p.X && p.DY && p.DX, "norm_bwd: empty problem or null tensor");
    InaProfScope prof(INA_PROF_NORM, 0.0, 0.0, stream);
    const size_t xes = p.x_dt == INA_DT_BF16 ? 2 : 4, des = p.dy_dt == INA_DT_BF16 ? 2 : 4, oes = p.dx_dt == INA_DT_BF16 ? 2 : 4;
    const bool vec = p.C % 4 == 0 && p.C <= 4096 && p.ldx % 4 == 0 && p.lddy % 4 == 0 && p.lddx % 4 == 0 && (!p.XHAT || p.ldxh % 4 == 0) &&
                     ((uintptr_t)p.X % (4 * xes)) == 0 && ((uintptr_t)p.DY % (4 * des)) == 0 && ((uintptr_t)p.DX % (4 * oes)) == 0 &&
                     (!p.XHAT || ((uintptr_t)p.XHAT % 8) == 0) && (!p.gamma || ((uintptr_t)p.gamma % 16) == 0);
    const dim3 grid((p.rows + 3) / 4);
    if (vec) {
        const bool xb = p.x_dt == INA_DT_BF16, db = p.dy_dt == INA_DT_BF16;
        if (xb && db) hipLaunchKernelGGL((norm_bwd_vec_kernel<true, true>), grid, dim3(256), 0, stream, p);
        else if (xb) hipLaunchKernelGGL((norm_bwd_vec_kernel<true, false>), grid, dim3(256), 0, stream, p);
        else if (db) hipLaunchKernelGGL((norm_bwd_vec_kernel<false, true>), grid, dim3(256), 0, stream, p);
        else hipLaunchKernelGGL((norm_bwd_vec_kernel<false, false>), grid, dim3(256), 0, stream, p);
    } else {
        hipLaunchKernelGGL(norm_bwd_kernel, grid, dim3(256), 0, stream, p);
    }
    INA_HIP_CHECK(hipGetLastError());
    return 0;
}

int ina_launch_transpose(const TransposeArgs& p, hipStream_t stream) {
    INA_REQUIRE(p.rows > 0 && p.cols > 0 && p.X && p.Y && p.ldy >= p.rows, "transpose: bad problem rows=%d cols=%d ldy=%d", p.rows, p.cols, p.ldy);
    InaProfScope prof(INA_PROF_ELEMENTWISE, 0.0, 0.0, stream);
    hipLaunchKernelGGL(transpose_kernel, dim3((p.ldy + 63) / 64, (p.cols + 63) / 64), dim3(256), 0, stream, p);
    INA_HIP_CHECK(hipGetLastError());
    return 0;
}

int ina_launch_sparse_rows(const SparseRowsArgs& p, hipStream_t stream) {
    INA_REQUIRE(p.n_out > 0 && p.C > 0 && p.taps > 0 && p.in && p.out && p.idx && p.coef, "sparse_rows: empty problem or null tensor");
    hipLaunchKernelGGL(sparse_rows_kernel, dim3(p.n_out), dim3(256), 0, stream, p);
    INA_HIP_CHECK(hipGetLastError());
    return 0;
}

int ina_launch_small_linear(const SmallLinearArgs& p, hipStream_t stream) {
    INA_REQUIRE(p.rows > 0 && p.N > 0 && p.K > 0 && p.X && p.W && p.Y, "small_linear: empty problem or null tensor");
    INA_REQUIRE(!p.tab || p.tab_mod > 0, "small_linear: tab_mod must be positive");
    const size_t n = (size_t)p.rows * p.N;
    hipLaunchKernelGGL(small_linear_kernel, dim3((int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096)), dim3(256), 0, stream, p);
    INA_HIP_CHECK(hipGetLastError());
    return 0;
}

int ina_launch_mse(const MseArgs& p, hipStream_t stream) {
    INA_REQUIRE(p.nseq > 0 && p.T > 0 && p.D > 0 && p.pred && p.target && p.mask && p.loss, "mse: empty problem or null tensor");
    hipLaunchKernelGGL(mse_kernel, dim3(1), dim3(256), 0, stream, p);
    INA_HIP_CHECK(hipGetLastError());
    return 0;
}

int ina_launch_adamw(const AdamwArgs& p, hipStream_t stream) {
    INA_REQUIRE(p.n > 0 && p.p && p.g && p.m && p.v, "adamw: empty problem or null tensor");
    INA_REQUIRE(p.bc1 > 0.f && p.bc2 > 0.f, "adamw: bias corrections must be positive (step >= 1)");
    INA_REQUIRE(p.n_parts == 0 || p.sumsq_parts, "adamw: n_parts without sumsq_parts");
    const int64_t blocks = (p.n + 255) / 256;
    InaProfScope prof(INA_PROF_ELEMENTWISE, 0.0, 0.0, stream);
    hipLaunchKernelGGL(adamw_kernel, dim3((int)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, stream, p);
    INA_HIP_CHECK(hipGetLastError());
    return 0;
}

int ina_launch_gemm_nn(const GemmNnArgs& p, hipStream_t stream) {
    INA_REQUIRE(p.M > 0 && p.M <= 16 && p.N > 0 && p.K > 0 && p.X && p.W && p.partial, "gemm_nn: bad problem M=%d N=%d K=%d", p.M, p.N, p.K);
    INA_REQUIRE(p.K % 8 == 0 && p.ldw % 8 == 0 && ((uintptr_t)p.W % 16) == 0, "gemm_nn: K / ldw must be multiples of 8 and W 16-byte aligned");
    INA_REQUIRE(p.splits > 0, "gemm_nn: splits must be positive");
    const int rps = (p.N + p.splits - 1) / p.splits;
    const int mr = p.M <= 8 ? 8 : 16;
    INA_REQUIRE(p.partial_elems >= (int64_t)p.splits * 4 * mr * p.K, "gemm_nn: partial buffer too small (%lld floats needed)",
                (long long)p.splits * 4 * mr * p.K);
    dim3 grid((p.K + NN_COLS - 1) / NN_COLS, p.splits);
    const size_t lds = (size_t)rps * mr * sizeof(float);
    INA_REQUIRE(lds <= 48 * 1024, "gemm_nn: %d rows per split do not fit LDS, raise splits", rps);
    InaProfScope prof(INA_PROF_GEMM_SKINNY, 2.0 * p.M * p.N * p.K, 2.0 * p.N * p.K, stream);
    if (mr == 8) hipLaunchKernelGGL(gemm_nn_kernel<8>, grid, dim3(256), lds, stream, p, rps);
    else hipLaunchKernelGGL(gemm_nn_kernel<16>, grid, dim3(256), lds, stream, p, rps);
    INA_HIP_CHECK(hipGetLastError());
    return 0;
}
