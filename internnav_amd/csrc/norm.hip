// LayerNorm / RMSNorm family for gfx950: one 64-lane wave per row, 16-byte bf16 loads, fp32 statistics.
// Covers nn.LayerNorm (DINOv2 eps 1e-6, nn.Transformer layers eps 1e-5), Qwen2_5_VLRMSNorm, and the NextDiT
// modulated forms (reference nextdit_traj.py:146,172-176):
//   y = norm(x [+ r]) * gamma + beta ;  y *= (1 + mod_scale[row / mod_div]) ;  y = G + tanh(gate[row / mod_div]) * y
// HBM-bound: algorithmic bytes = 2*C (read) + 2*C (write) per row (+2*C per optional operand).
#include "common.h"
#include "kernels.h"

namespace {

template <int NCH>  // 16-byte chunks per lane
__global__ __launch_bounds__(256) void norm_kernel(NormArgs p) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= p.rows) return;
    const int nchunks = p.C >> 3;
    const bf16* __restrict__ x = reinterpret_cast<const bf16*>(p.X) + (size_t)row * p.ldx;
    const bf16* __restrict__ r = p.R ? reinterpret_cast<const bf16*>(p.R) + (size_t)row * p.ldr : nullptr;

    float v[NCH][8];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        int c = lane + i * 64;
        if (c < nchunks) {
            bf16x8 xv = *reinterpret_cast<const bf16x8*>(x + c * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[i][j] = (float)xv[j];
            if (r) {
                bf16x8 rv = *reinterpret_cast<const bf16x8*>(r + c * 8);
#pragma unroll
                for (int j = 0; j < 8; ++j) v[i][j] += (float)rv[j];
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) sum += v[i][j];
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[i][j] = 0.f;
        }
    }
    if (p.S) {
        bf16* s = reinterpret_cast<bf16*>(p.S) + (size_t)row * p.ldy;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            int c = lane + i * 64;
            if (c < nchunks) {
                bf16x8 o;
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = (bf16)v[i][j];
                *reinterpret_cast<bf16x8*>(s + c * 8) = o;
            }
        }
    }
    const float invC = 1.0f / (float)p.C;
    float mean = 0.f;
    if (!p.rms) mean = wave_sum(sum) * invC;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        int c = lane + i * 64;
        if (c < nchunks) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float d = v[i][j] - mean;
                sq += d * d;
            }
        }
    }
    const float rstd = rsqrtf(wave_sum(sq) * invC + p.eps);

    const float* ms = p.mod_scale ? p.mod_scale + (size_t)(row / p.mod_div) * p.mod_ld : nullptr;
    const float* gt = p.gate ? p.gate + (size_t)(row / p.mod_div) * p.mod_ld : nullptr;
    const bf16* gb = p.G ? reinterpret_cast<const bf16*>(p.G) + (size_t)row * p.ldg : nullptr;
    bf16* y = reinterpret_cast<bf16*>(p.Y) + (size_t)row * p.ldy;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        int c = lane + i * 64;
        if (c < nchunks) {
            bf16x8 o;
            bf16x8 gv;
            if (gb) gv = *reinterpret_cast<const bf16x8*>(gb + c * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                int col = c * 8 + j;
                float t = (v[i][j] - mean) * rstd;
                if (p.gamma) t *= p.gamma[col];
                if (p.beta) t += p.beta[col];
                if (ms) t *= (1.0f + ms[col]);
                if (gt) t = tanhf(gt[col]) * t;
                if (gb) t += (float)gv[j];
                o[j] = (bf16)t;
            }
            *reinterpret_cast<bf16x8*>(y + c * 8) = o;
        }
    }
}

}  // namespace

int ina_launch_norm(const NormArgs& p_in, hipStream_t stream) {
    NormArgs p = p_in;
    if (p.mod_div <= 0) p.mod_div = 1;
    INA_REQUIRE(p.rows > 0 && p.C > 0, "norm: empty problem rows=%d C=%d", p.rows, p.C);
    INA_REQUIRE(p.C % 8 == 0 && p.ldx % 8 == 0 && p.ldy % 8 == 0, "norm: C/ldx/ldy must be multiples of 8 (C=%d)", p.C);
    INA_REQUIRE(!p.R || p.ldr % 8 == 0, "norm: ldr must be a multiple of 8");
    INA_REQUIRE(!p.G || p.ldg % 8 == 0, "norm: ldg must be a multiple of 8");
    INA_REQUIRE((!p.mod_scale && !p.gate) || (p.mod_ld > 0), "norm: modulation needs mod_div/mod_ld");
    const int nch = (p.C / 8 + 63) / 64;
    dim3 grid((p.rows + 3) / 4), block(256);
    if (nch <= 1) hipLaunchKernelGGL(norm_kernel<1>, grid, block, 0, stream, p);
    else if (nch <= 2) hipLaunchKernelGGL(norm_kernel<2>, grid, block, 0, stream, p);
    else if (nch <= 4) hipLaunchKernelGGL(norm_kernel<4>, grid, block, 0, stream, p);
    else if (nch <= 8) hipLaunchKernelGGL(norm_kernel<8>, grid, block, 0, stream, p);
    else if (nch <= 16) hipLaunchKernelGGL(norm_kernel<16>, grid, block, 0, stream, p);
    else { ina_set_error("norm: C=%d too wide (max 8192)", p.C); return -2; }
    INA_HIP_CHECK(hipGetLastError());
    return 0;
}
