// LayerNorm / RMSNorm family for gfx950: one 64-lane wave per row, 16/32-byte loads, fp32 statistics.
// Covers nn.LayerNorm (DINOv2 eps 1e-6, nn.Transformer layers eps 1e-5), Qwen2_5_VLRMSNorm, diffusers RMSNorm and the
// NextDiT modulated forms (reference nextdit_traj.py:146,172-176):
//   t = norm(x) * gamma + beta ;  t *= (1 + mod_scale[b]) ;  t = tanh(gate[b]) * t ;  t += G[row] ;  t += P[row % p_mod]
// with b = row / mod_div.  The input may be the fp32 residual stream or a bf16 activation; the result is written as bf16
// (next GEMM operand) and/or fp32 (next residual).  Logical rows can be gathered / scattered through a one-level row
// map (drop the cls token, concatenate token groups) so no separate copy kernels are needed.
// HBM-bound: algorithmic bytes per row = C * (sizeof(in) + sizeof(out)) (+ optional operands).
#include "common.h"
#include "kernels.h"

namespace {

__device__ __forceinline__ int map_row(const ina_rowmap& m, int r) {
    return m.seg_len > 0 ? (r / m.seg_len) * m.seg_stride + m.off + (r % m.seg_len) : r;
}

template <int NCH, bool XF32, int RPW>  // 8-element chunks per lane, input type, rows per wave (sub-wave groups of 64/RPW lanes)
__global__ __launch_bounds__(256) void norm_kernel(NormArgs p) {
    constexpr int GL = 64 / RPW;  // lanes per row
    const int lane = threadIdx.x & 63;
    const int sub = lane / GL, gl = lane % GL;
    const int row = (blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW + sub;
    const bool live = row < p.rows;
    const int nchunks = p.C >> 3;
    const int in_row = live ? map_row(p.in_map, row) : 0;
    const int out_row = live ? map_row(p.out_map, row) : 0;

    float v[NCH][8];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c = gl + i * GL;
        if (live && c < nchunks) {
            if (XF32) {
                const float* x = reinterpret_cast<const float*>(p.X) + (size_t)in_row * p.ldx + c * 8;
                f32x4 a = *reinterpret_cast<const f32x4*>(x), b = *reinterpret_cast<const f32x4*>(x + 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) { v[i][j] = a[j]; v[i][4 + j] = b[j]; }
            } else {
                bf16x8 xv = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const bf16*>(p.X) + (size_t)in_row * p.ldx + c * 8);
#pragma unroll
                for (int j = 0; j < 8; ++j) v[i][j] = (float)xv[j];
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) sum += v[i][j];
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[i][j] = 0.f;
        }
    }
    auto group_sum = [&](float s) {
#pragma unroll
        for (int o = GL / 2; o > 0; o >>= 1) s += __shfl_xor(s, o);
        return s;
    };
    const float invC = 1.0f / (float)p.C;
    float mean = 0.f;
    if (!p.rms) mean = group_sum(sum) * invC;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c = gl + i * GL;
        if (c < nchunks) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float d = v[i][j] - mean;
                sq += d * d;
            }
        }
    }
    const float rstd = rsqrtf(group_sum(sq) * invC + p.eps);
    if (!live) return;   // the lanes of a row group leave together: later group reductions stay well defined

    const int mrow = row / p.mod_div;
    const float* ms = p.mod_scale ? p.mod_scale + (size_t)mrow * p.mod_ld : nullptr;
    const float* gt = p.gate ? p.gate + (size_t)mrow * p.mod_ld : nullptr;
    const float* pt = p.P ? p.P + (size_t)(row % p.p_mod) * p.C : nullptr;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c = gl + i * GL;
        if (c >= nchunks) continue;
        const int col = c * 8;
        float t[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) t[j] = (v[i][j] - mean) * rstd;
        auto mul8 = [&](const float* w, bool plus1) {
            f32x4 a = *reinterpret_cast<const f32x4*>(w + col), b = *reinterpret_cast<const f32x4*>(w + col + 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                t[j] *= plus1 ? 1.0f + a[j] : a[j];
                t[4 + j] *= plus1 ? 1.0f + b[j] : b[j];
            }
        };
        auto add8 = [&](const float* w) {
            f32x4 a = *reinterpret_cast<const f32x4*>(w + col), b = *reinterpret_cast<const f32x4*>(w + col + 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) { t[j] += a[j]; t[4 + j] += b[j]; }
        };
        if (p.gamma) mul8(p.gamma, false);
        if (p.beta) add8(p.beta);
        if (ms) mul8(ms, true);
        if (gt) {
            f32x4 a = *reinterpret_cast<const f32x4*>(gt + col), b = *reinterpret_cast<const f32x4*>(gt + col + 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) { t[j] *= tanhf(a[j]); t[4 + j] *= tanhf(b[j]); }
        }
        if (p.G) {
            if (p.g_dtype == INA_DT_F32) {
                add8(reinterpret_cast<const float*>(p.G) + (size_t)row * p.ldg);
            } else {
                bf16x8 gv = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const bf16*>(p.G) + (size_t)row * p.ldg + col);
#pragma unroll
                for (int j = 0; j < 8; ++j) t[j] += (float)gv[j];
            }
        }
        if (pt) add8(pt);
        if (p.Y) {
            bf16x8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = (bf16)t[j];
            *reinterpret_cast<bf16x8*>(reinterpret_cast<bf16*>(p.Y) + (size_t)out_row * p.ldy + col) = o;
        }
        if (p.Y32) {
            float* y = reinterpret_cast<float*>(p.Y32) + (size_t)out_row * p.ldy32 + col;
            *reinterpret_cast<f32x4*>(y) = f32x4{t[0], t[1], t[2], t[3]};
            *reinterpret_cast<f32x4*>(y + 4) = f32x4{t[4], t[5], t[6], t[7]};
        }
        if (p.Y2) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[i][j] = t[j];
        }
    }
    if (!p.Y2) return;
    // ---- chained second norm on the row just produced (the next sub-block's pre-norm of the new residual): saves re-reading it
    float mean2 = 0.f;
    if (!p.rms) {
        float s2 = 0.f;
#pragma unroll
        for (int i = 0; i < NCH; ++i)
            if (gl + i * GL < nchunks)
#pragma unroll
                for (int j = 0; j < 8; ++j) s2 += v[i][j];
        mean2 = group_sum(s2) * invC;
    }
    float sq2 = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i)
        if (gl + i * GL < nchunks)
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float d = v[i][j] - mean2; sq2 += d * d; }
    const float rstd2 = rsqrtf(group_sum(sq2) * invC + p.eps);
    const float* ms2 = p.mod_scale2 ? p.mod_scale2 + (size_t)mrow * p.mod_ld : nullptr;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c = gl + i * GL;
        if (c >= nchunks) continue;
        const int col = c * 8;
        float t[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) t[j] = (v[i][j] - mean2) * rstd2;
        if (p.gamma2) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(p.gamma2 + col), b = *reinterpret_cast<const f32x4*>(p.gamma2 + col + 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) { t[j] *= a[j]; t[4 + j] *= b[j]; }
        }
        if (ms2) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(ms2 + col), b = *reinterpret_cast<const f32x4*>(ms2 + col + 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) { t[j] *= 1.0f + a[j]; t[4 + j] *= 1.0f + b[j]; }
        }
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (bf16)t[j];
        *reinterpret_cast<bf16x8*>(reinterpret_cast<bf16*>(p.Y2) + (size_t)out_row * p.ldy2 + col) = o;
    }
}

template <int NCH, int RPW>
void launch_norm(const NormArgs& p, hipStream_t stream) {
    dim3 grid((p.rows + 4 * RPW - 1) / (4 * RPW)), block(256);
    if (p.x_dtype == INA_DT_F32) hipLaunchKernelGGL((norm_kernel<NCH, true, RPW>), grid, block, 0, stream, p);
    else hipLaunchKernelGGL((norm_kernel<NCH, false, RPW>), grid, block, 0, stream, p);
}

}  // namespace

int ina_launch_norm(const NormArgs& p_in, hipStream_t stream) {
    NormArgs p = p_in;
    if (p.mod_div <= 0) p.mod_div = 1;
    if (p.p_mod <= 0) p.p_mod = 1;
    INA_REQUIRE(p.rows > 0 && p.C > 0, "norm: empty problem rows=%d C=%d", p.rows, p.C);
    INA_REQUIRE(p.X && (p.Y || p.Y32), "norm: X and at least one of Y / Y32 are required");
    INA_REQUIRE(p.C % 8 == 0 && p.ldx % 8 == 0, "norm: C/ldx must be multiples of 8 (C=%d ldx=%d)", p.C, p.ldx);
    INA_REQUIRE(!p.Y || p.ldy % 8 == 0, "norm: ldy must be a multiple of 8");
    INA_REQUIRE(!p.Y32 || p.ldy32 % 4 == 0, "norm: ldy32 must be a multiple of 4");
    INA_REQUIRE(!p.G || p.ldg % 8 == 0, "norm: ldg must be a multiple of 8");
    INA_REQUIRE(!p.Y2 || p.ldy2 % 8 == 0, "norm: ldy2 must be a multiple of 8");
    INA_REQUIRE((!p.mod_scale && !p.gate && !p.mod_scale2) || (p.mod_ld > 0 && p.mod_ld % 4 == 0), "norm: modulation needs mod_ld (multiple of 4)");
    const int nchunks = p.C / 8;
    InaProfScope prof(INA_PROF_NORM, 8.0 * p.rows * p.C,
                      (double)p.rows * p.C * ((p.x_dtype == INA_DT_F32 ? 4.0 : 2.0) + (p.Y ? 2.0 : 0.0) + (p.Y32 ? 4.0 : 0.0) + (p.Y2 ? 2.0 : 0.0) +
                                              (p.G ? (p.g_dtype == INA_DT_F32 ? 4.0 : 2.0) : 0.0)), stream);
    if (nchunks <= 16) launch_norm<1, 4>(p, stream);        // C <= 128: 4 rows per wave
    else if (nchunks <= 32) launch_norm<1, 2>(p, stream);   // C <= 256
    else if (nchunks <= 64) launch_norm<1, 1>(p, stream);
    else if (nchunks <= 128) launch_norm<2, 1>(p, stream);
    else if (nchunks <= 256) launch_norm<4, 1>(p, stream);
    else if (nchunks <= 512) launch_norm<8, 1>(p, stream);
    else if (nchunks <= 1024) launch_norm<16, 1>(p, stream);
    else { ina_set_error("norm: C=%d too wide (max 8192)", p.C); return -2; }
    INA_HIP_CHECK(hipGetLastError());
    return 0;
}
