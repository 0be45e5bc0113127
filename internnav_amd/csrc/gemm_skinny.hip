// Skinny-M GEMM for gfx950 (M <= 64): C[M,N] = epilogue(A[M,K] . W[N,K]^T) as a WEIGHT-STREAMING kernel.
//
// Used by the decode steps of the System-2 LLM (M = number of sequences), the latent-query pass, lm_head on the last position
// and the adaLN modulation GEMMs. At M <= 64 the op is HBM-bound: every weight byte is read once and used for M <= 64 MACs, so
// the design goal is bytes/s, not MFMA utilisation:
//   * grid = (N / 64 column tiles) x SPLITK K-slices, chosen so the launch has >= ~1024 workgroups (a 28-tile N = 3584 GEMM would
//     otherwise leave 228 of the 256 CUs idle);
//   * W is streamed straight from HBM into VGPRs (no LDS round trip: each weight element is used by exactly one wave), 4 x 16-byte
//     loads per lane per 128-wide K step, arranged so a wave covers whole 256-byte row segments (K is permuted identically for
//     both MFMA operands, which leaves the dot product unchanged);
//   * the activation rows (M x K bf16, <= 2.4 MB, L2 resident) are read directly as the second MFMA operand;
//   * fp32 partial tiles go to a workspace [SPLITK][M][N]; a second small kernel sums the slices and applies the fused epilogue
//     (bias, activation, GLU pairing, column / row scale, residual, bf16|f32 store) - the same epilogue as the tiled GEMM.
// Algorithmic bytes per launch = 2*N*K (weights) + 2*M*K + out; roofline = HBM.
#include "common.h"
#include "kernels.h"

namespace {

constexpr int SK_BN = 64;    // output columns (W rows) per workgroup: 4 waves x 16
constexpr int SK_BK = 128;   // K elements per step

template <int MF>  // number of 16-row activation fragments (M <= 16*MF)
__global__ __launch_bounds__(256) void gemm_skinny_kernel(GemmArgs p, float* __restrict__ part, int kslice) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n0 = blockIdx.x * SK_BN + wave * 16;
    const int split = blockIdx.y;
    const int k_begin = split * kslice;
    const int k_end = min(p.K, k_begin + kslice);
    const int r16 = lane & 15, g = lane >> 4;
    const bf16* __restrict__ A = reinterpret_cast<const bf16*>(p.A);
    const bf16* __restrict__ W = reinterpret_cast<const bf16*>(p.W);
    const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};

    f32x4 acc[MF];
#pragma unroll
    for (int i = 0; i < MF; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int wn = n0 + r16;
    const bool wok = wn < p.N;
    const bf16* wrow = W + (size_t)(wok ? wn : 0) * p.ldw;

    // software pipeline: the loads of K step t+1 (4 x 16 B of W + 4*MF x 16 B of A per lane) are issued before the MFMAs of step t
    auto load_w = [&](int k0, bf16x8 (&wf)[4]) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int k = k0 + g * 32 + s * 8;   // lane group g owns k in [k0 + g*32, k0 + g*32 + 32): 4 MFMA k-steps of 8 elements
            wf[s] = (wok && k < k_end) ? *reinterpret_cast<const bf16x8*>(wrow + k) : zero8;
        }
    };
    auto load_a = [&](int k0, bf16x8 (&af)[MF][4]) {
#pragma unroll
        for (int i = 0; i < MF; ++i) {
            const int m = i * 16 + r16;
            const bf16* arow = A + (size_t)(m < p.M ? m : 0) * p.lda;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int k = k0 + g * 32 + s * 8;
                af[i][s] = (m < p.M && k < k_end) ? *reinterpret_cast<const bf16x8*>(arow + k) : zero8;
            }
        }
    };
    bf16x8 wf[2][4], af[2][MF][4];
    load_w(k_begin, wf[0]);
    load_a(k_begin, af[0]);
    int cur = 0;
    for (int k0 = k_begin; k0 < k_end; k0 += 2 * SK_BK) {
        // two K steps per trip so the ping-pong register sets are indexed statically
        load_w(k0 + SK_BK, wf[1]);
        load_a(k0 + SK_BK, af[1]);
#pragma unroll
        for (int i = 0; i < MF; ++i)
#pragma unroll
            for (int s = 0; s < 4; ++s) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[0][s], af[0][i][s], acc[i], 0, 0, 0);
        load_w(k0 + 2 * SK_BK, wf[0]);
        load_a(k0 + 2 * SK_BK, af[0]);
#pragma unroll
        for (int i = 0; i < MF; ++i)
#pragma unroll
            for (int s = 0; s < 4; ++s) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[1][s], af[1][i][s], acc[i], 0, 0, 0);
    }
    (void)cur;
    const int n = n0 + g * 4;
    if (n < p.N) {
#pragma unroll
        for (int i = 0; i < MF; ++i) {
            const int m = i * 16 + r16;
            if (m < p.M) *reinterpret_cast<f32x4*>(part + ((size_t)split * p.M + m) * p.N + n) = acc[i];
        }
    }
}

// sum the K-slices and apply the epilogue; one thread per 4 output columns (GLU: per gate/up pair of 4)
__global__ __launch_bounds__(256) void gemm_skinny_epilogue(GemmArgs p, const float* __restrict__ part, int splits) {
    const int n4 = p.N >> 2;
    const long total = (long)p.M * n4;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int m = (int)(i / n4), n = (int)(i % n4) * 4;
        if (p.glu && ((n >> 4) & 1)) continue;  // "up" blocks are consumed together with their gate block
        float v[4] = {0.f, 0.f, 0.f, 0.f}, u[4] = {0.f, 0.f, 0.f, 0.f};
        for (int s = 0; s < splits; ++s) {
            const float* q = part + ((size_t)s * p.M + m) * p.N + n;
            f32x4 a = *reinterpret_cast<const f32x4*>(q);
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] += a[r];
            if (p.glu) {
                f32x4 b = *reinterpret_cast<const f32x4*>(q + 16);
#pragma unroll
                for (int r = 0; r < 4; ++r) u[r] += b[r];
            }
        }
        const float rs = p.rowscale ? p.rowscale[m / p.rowscale_div] : 1.0f;
        int no = n;
        if (p.glu) {
            no = ((n >> 5) << 4) + (n & 15);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float gg = v[r], uu = u[r];
                if (p.bias) { gg += p.bias[n + r]; uu += p.bias[n + 16 + r]; }
                v[r] = ina_act(gg, p.act) * uu * rs;
            }
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float x = v[r];
                if (p.bias) x += p.bias[n + r];
                x = ina_act(x, p.act);
                if (p.colscale) x *= p.colscale[n + r];
                v[r] = x * rs;
            }
            if (p.R) {
                const size_t ro = (size_t)m * p.ldr + n;
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
        }
        const size_t co = (size_t)m * p.ldc + no;
        if (p.out_dtype == INA_DT_BF16) *reinterpret_cast<bf16x4*>(reinterpret_cast<bf16*>(p.C) + co) = bf16x4{(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
        else *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.C) + co) = f32x4{v[0], v[1], v[2], v[3]};
    }
}

}  // namespace

int ina_launch_gemm_skinny(const GemmArgs& p, hipStream_t stream) {
    const int tiles = (p.N + SK_BN - 1) / SK_BN;
    int splits = 1;
    const int ksteps = (p.K + SK_BK - 1) / SK_BK;
    while (tiles * splits < 1024 && splits * 2 <= ksteps && splits < 32) splits *= 2;
    const int kslice = ((ksteps + splits - 1) / splits) * SK_BK;
    splits = (p.K + kslice - 1) / kslice;
    float* part = nullptr;
    const int rc = ina_workspace(0, (size_t)splits * p.M * p.N * sizeof(float), stream, &part);
    if (rc) return rc;
    const double osz = p.out_dtype == INA_DT_BF16 ? 2.0 : 4.0;
    InaProfScope prof(INA_PROF_GEMM_SKINNY, 2.0 * p.M * p.N * p.K, 2.0 * p.M * p.K + 2.0 * p.N * p.K + osz * p.M * (p.glu ? p.N / 2 : p.N), stream);
    dim3 grid(tiles, splits);
    const int mf = (p.M + 15) / 16;
    switch (mf) {
        case 1: hipLaunchKernelGGL(gemm_skinny_kernel<1>, grid, dim3(256), 0, stream, p, part, kslice); break;
        case 2: hipLaunchKernelGGL(gemm_skinny_kernel<2>, grid, dim3(256), 0, stream, p, part, kslice); break;
        case 3: hipLaunchKernelGGL(gemm_skinny_kernel<3>, grid, dim3(256), 0, stream, p, part, kslice); break;
        default: hipLaunchKernelGGL(gemm_skinny_kernel<4>, grid, dim3(256), 0, stream, p, part, kslice); break;
    }
    const long total = (long)p.M * (p.N / 4);
    const int eb = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
    hipLaunchKernelGGL(gemm_skinny_epilogue, dim3(eb), dim3(256), 0, stream, p, part, splits);
    INA_HIP_CHECK(hipGetLastError());
    return 0;
}
