// Elementwise / normalisation kernels of the diffusion-policy ConditionalUnet1D head (conditional_unet1d.py:14-241) for gfx950.
// The convolutions themselves are implicit GEMMs on the MFMA kernels of gemm*.hip (internnav_amd/unet1d.py builds the overlapping
// row windows); what remains per Conv1dBlock is  GroupNorm(8) -> Mish [-> FiLM scale/bias] [-> + residual]  on channels-last,
// sequence-padded activations, plus the pad-row clean-up after the strided (down / transposed) convolutions and the DDIM update.
//
// Layout: a level holds Bs sequences of T valid rows between `pad` zero rows on both sides: row (b, t) of a padded buffer is
// b * (T + 2 pad) + pad + t, C contiguous bf16 channels per row. A k-tap convolution then reads, for output row (b, t), the k * C
// CONTIGUOUS elements starting at padded row t + pad - k/2: the im2col matrix is an overlapping strided view, never materialised.
// All three kernels are HBM-bound (bytes touched / launch).
#include "common.h"
#include "kernels.h"

namespace {

__device__ __forceinline__ float mish_f(float x) {
    // x * tanh(softplus(x)); softplus with the torch threshold (20)
    const float sp = x > 20.f ? x : log1pf(__expf(x));
    return x * tanhf(sp);
}

// one workgroup per sequence: T x C elements (8192 at every level of the 256/512/1024 network), 8 channels (16 B) per thread and row.
// in: conv output rows b * in_seq_stride + t (t < T); out: padded rows b * (T + 2 pad) + pad + t, pad rows zeroed.
__global__ __launch_bounds__(256) void gn_mish_kernel(GnMishArgs p) {
    __shared__ float s_sum[32], s_sq[32];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int C = p.C, T = p.T, G = p.groups;
    const int cpr = C >> 3;                        // 16-byte chunks per row
    const int chunk = tid % cpr, r0 = tid / cpr, rstep = 256 / cpr;
    const int grp = chunk / (cpr / G);
    if (tid < 32) { s_sum[tid] = 0.f; s_sq[tid] = 0.f; }
    __syncthreads();
    const size_t in0 = (size_t)b * p.in_seq_stride * p.ldx + chunk * 8;
    auto load8 = [&](int t, float* f) {      // conv output row t of this sequence: fp32 (x_f32) or bf16
        if (p.x_f32) {
            const float* q = reinterpret_cast<const float*>(p.X) + in0 + (size_t)t * p.ldx;
            const f32x4 a = *reinterpret_cast<const f32x4*>(q), c = *reinterpret_cast<const f32x4*>(q + 4);
#pragma unroll
            for (int i = 0; i < 4; ++i) { f[i] = a[i]; f[4 + i] = c[i]; }
        } else {
            const bf16x8 v = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const bf16*>(p.X) + in0 + (size_t)t * p.ldx);
#pragma unroll
            for (int i = 0; i < 8; ++i) f[i] = (float)v[i];
        }
    };
    float sum = 0.f, sq = 0.f;
    for (int t = r0; t < T; t += rstep) {
        float f[8];
        load8(t, f);
#pragma unroll
        for (int i = 0; i < 8; ++i) { sum += f[i]; sq += f[i] * f[i]; }
    }
    atomicAdd(&s_sum[grp], sum);
    atomicAdd(&s_sq[grp], sq);
    __syncthreads();
    const float n = (float)(T * (C / G));
    const float mean = s_sum[grp] / n;
    const float rstd = rsqrtf(fmaxf(s_sq[grp] / n - mean * mean, 0.f) + p.eps);
    float ga[8], be[8], fs[8], fb[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int c = chunk * 8 + i;
        ga[i] = p.gamma[c] * rstd;
        be[i] = p.beta[c] - mean * ga[i];
        fs[i] = 1.f; fb[i] = 0.f;
    }
    if (p.film_env) {    // FiLM: [scale | bias] = film_env[env] + film_step, env = sequence / seq_per_env
        const float* fe = p.film_env + (size_t)(b / p.seq_per_env) * p.film_ld + p.film_off;
        const float* fst = p.film_step + p.film_off;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int c = chunk * 8 + i;
            fs[i] = fe[c] + fst[c];
            fb[i] = fe[C + c] + fst[C + c];
        }
    }
    const int Tp = T + 2 * p.pad;
    bf16* out = reinterpret_cast<bf16*>(p.Y) + (size_t)b * Tp * p.ldy + chunk * 8;
    const bf16* res = p.R ? reinterpret_cast<const bf16*>(p.R) + (size_t)b * Tp * p.ldr + chunk * 8 : nullptr;
    for (int t = r0; t < Tp; t += rstep) {
        bf16x8 o;
        const int tv = t - p.pad;
        if (tv < 0 || tv >= T) {
#pragma unroll
            for (int i = 0; i < 8; ++i) o[i] = (bf16)0.f;
        } else {
            float v[8];
            load8(tv, v);
            bf16x8 rv;
            if (res) rv = *reinterpret_cast<const bf16x8*>(res + (size_t)t * p.ldr);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                float y = mish_f(v[i] * ga[i] + be[i]);
                y = y * fs[i] + fb[i];
                if (res) y += (float)rv[i];
                o[i] = (bf16)y;
            }
        }
        *reinterpret_cast<bf16x8*>(out + (size_t)t * p.ldy) = o;
    }
}

// zero the pad rows of a padded buffer (the strided convolutions write junk windows into them); optionally add a per-channel bias to
// the valid rows first (ConvTranspose1d bias is added once, after its two phase GEMMs)
__global__ __launch_bounds__(256) void pad_rows_kernel(PadRowsArgs p) {
    const int cpr = p.C >> 3;
    const size_t total = (size_t)p.seqs * (p.T + 2 * p.pad) * cpr;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int chunk = (int)(i % cpr);
        const size_t row = i / cpr;
        const int t = (int)(row % (p.T + 2 * p.pad)) - p.pad;
        bf16* q = reinterpret_cast<bf16*>(p.X) + row * p.ldx + chunk * 8;
        if (t < 0 || t >= p.T) {
            bf16x8 z;
#pragma unroll
            for (int k = 0; k < 8; ++k) z[k] = (bf16)0.f;
            *reinterpret_cast<bf16x8*>(q) = z;
        } else if (p.bias) {
            bf16x8 v = *reinterpret_cast<bf16x8*>(q);
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = (bf16)((float)v[k] + p.bias[chunk * 8 + k]);
            *reinterpret_cast<bf16x8*>(q) = v;
        }
    }
}

// DDIM step (eta = 0, epsilon prediction, clip_sample): x0 = clip((x - sqrt(1-a_t) eps) / sqrt(a_t)); the direction term uses the network's
// eps as diffusers' DDIMScheduler.step does by default, or eps' = (x - sqrt(a_t) x0) / sqrt(1-a_t) re-derived from the clipped x0 when
// use_clipped_model_output is set; x <- sqrt(a_prev) x0 + sqrt(1-a_prev) eps. Updates the fp32 sample in place and refreshes the bf16 network input.
__global__ __launch_bounds__(256) void ddim_step_kernel(DdimStepArgs p) {
    const int i = blockIdx.x * 256 + threadIdx.x;          // (sequence, t)
    if (i >= p.seqs * p.T) return;
    const int b = i / p.T, t = i % p.T;
    const int Tp = p.T + 2 * p.pad;
    const float* e = p.eps + ((size_t)b * Tp + p.pad + t) * p.lde;
    float* x = p.sample + (size_t)i * p.D;
    bf16* xin = reinterpret_cast<bf16*>(p.Xin) + ((size_t)b * Tp + p.pad + t) * p.ldx;
    for (int d = 0; d < p.D; ++d) {
        const float xv = x[d];
        float ev = e[d];
        float x0 = (xv - p.sqrt_b * ev) * p.inv_sqrt_a;
        if (p.clip > 0.f) x0 = fminf(fmaxf(x0, -p.clip), p.clip);
        if (p.use_clipped_model_output) ev = (xv - x0 / p.inv_sqrt_a) / p.sqrt_b;
        const float nx = p.sqrt_ap * x0 + p.sqrt_bp * ev;
        x[d] = nx;
        xin[d] = (bf16)nx;
    }
}

}  // namespace

int ina_launch_gn_mish(const GnMishArgs& p, hipStream_t stream) {
    INA_REQUIRE(p.X && p.Y && p.gamma && p.beta, "gn_mish: X, Y, gamma, beta are required");
    INA_REQUIRE(p.seqs > 0 && p.T > 0 && p.C > 0 && p.groups > 0 && p.groups <= 32, "gn_mish: bad shape seqs=%d T=%d C=%d groups=%d", p.seqs, p.T, p.C, p.groups);
    INA_REQUIRE(p.C % 8 == 0 && (p.C / 8) % p.groups == 0 && 256 % (p.C / 8) == 0, "gn_mish: C=%d must give 16-byte chunks that tile 256 threads and %d groups", p.C, p.groups);
    INA_REQUIRE(p.ldx % 8 == 0 && p.ldy % 8 == 0 && (!p.R || p.ldr % 8 == 0), "gn_mish: row strides must keep 16-byte alignment");
    INA_REQUIRE(!p.film_env || (p.film_step && p.seq_per_env > 0), "gn_mish: FiLM needs film_step and seq_per_env");
    InaProfScope prof(INA_PROF_NORM, 0.0, (double)p.seqs * p.T * p.C * ((p.R ? 4.0 : 2.0) + (p.x_f32 ? 8.0 : 4.0)), stream);
    hipLaunchKernelGGL(gn_mish_kernel, dim3(p.seqs), dim3(256), 0, stream, p);
    INA_HIP_CHECK(hipGetLastError());
    return 0;
}

int ina_launch_pad_rows(const PadRowsArgs& p, hipStream_t stream) {
    INA_REQUIRE(p.X && p.seqs > 0 && p.T > 0 && p.pad >= 0 && p.C % 8 == 0 && p.ldx % 8 == 0, "pad_rows: bad arguments");
    const size_t total = (size_t)p.seqs * (p.T + 2 * p.pad) * (p.C / 8);
    InaProfScope prof(INA_PROF_ELEMENTWISE, 0.0, (double)total * 16.0, stream);
    hipLaunchKernelGGL(pad_rows_kernel, dim3((unsigned)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096)), dim3(256), 0, stream, p);
    INA_HIP_CHECK(hipGetLastError());
    return 0;
}

int ina_launch_ddim_step(const DdimStepArgs& p, hipStream_t stream) {
    INA_REQUIRE(p.eps && p.sample && p.Xin && p.seqs > 0 && p.T > 0 && p.D > 0 && p.D <= p.lde && p.D <= p.ldx, "ddim_step: bad arguments");
    InaProfScope prof(INA_PROF_ELEMENTWISE, 0.0, (double)p.seqs * p.T * p.D * 14.0, stream);
    hipLaunchKernelGGL(ddim_step_kernel, dim3((p.seqs * p.T + 255) / 256), dim3(256), 0, stream, p);
    INA_HIP_CHECK(hipGetLastError());
    return 0;
}
