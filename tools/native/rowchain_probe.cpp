// Torch-free probe of the row-local part of a NextDiT block over the C-ABI (round 5): per block-step the launches between two attention stages,
//   unfused: GEMM(attn2.to_out) -> norm (gated rmsnorm + residual + next pre-norm) -> GEMM(linear_1/3, SwiGLU) -> GEMM(linear_2) -> norm -> GEMM(next q|k|v|q2)
//   chained: ina_dit_rowchain x 2 (csrc/dit_rowchain.hip), 128-row panels
// timed back to back on one stream (12 block-steps per repetition, distinct weights per block so the W stream is not an L2 replay), plus the max
// difference of the residual stream and of the last q|k|v|q2 projection between the two forms.
// Build: tools/native/build.sh; run from the repo root: tools/native/rowchain_probe [lib] [ffn width]
#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/internnav_amd.h"

#define HIP_OK(x)                                                                                      \
    do {                                                                                               \
        hipError_t e_ = (x);                                                                           \
        if (e_ != hipSuccess) {                                                                        \
            fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_));          \
            exit(2);                                                                                   \
        }                                                                                              \
    } while (0)

__global__ void fill_bf16(uint16_t* p, size_t n, uint32_t seed, float scale) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        uint32_t h = (uint32_t)i * 0x9E3779B9u ^ seed ^ (uint32_t)(i >> 32) * 0x85EBCA6Bu;
        h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
        const float v = ((h >> 8) * (1.0f / 8388608.0f) - 1.0f) * scale;
        uint32_t b = __float_as_uint(v);
        b += 0x7FFFu + ((b >> 16) & 1u);
        p[i] = (uint16_t)(b >> 16);
    }
}
__global__ void fill_f32(float* p, size_t n, uint32_t seed, float scale, float offset) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        uint32_t h = (uint32_t)i * 0x9E3779B9u ^ seed;
        h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
        p[i] = ((h >> 8) * (1.0f / 8388608.0f) - 1.0f) * scale + offset;
    }
}
__global__ void maxdiff_f32(const float* a, const float* b, size_t n, float* out) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    float m = 0.f;
    for (; i < n; i += stride) m = fmaxf(m, fabsf(a[i] - b[i]));
    atomicMax(reinterpret_cast<int*>(out), __float_as_int(m));
}
__global__ void maxdiff_bf16(const uint16_t* a, const uint16_t* b, size_t n, float* out) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    float m = 0.f;
    for (; i < n; i += stride) m = fmaxf(m, fabsf(__uint_as_float((uint32_t)a[i] << 16) - __uint_as_float((uint32_t)b[i] << 16)));
    atomicMax(reinterpret_cast<int*>(out), __float_as_int(m));
}
static void* bf16_buf(size_t n, uint32_t seed, float scale) {
    void* p;
    HIP_OK(hipMalloc(&p, n * 2));
    hipLaunchKernelGGL(fill_bf16, dim3(2048), dim3(256), 0, 0, (uint16_t*)p, n, seed, scale);
    return p;
}
static float* f32_buf(size_t n, uint32_t seed, float scale, float offset) {
    float* p;
    HIP_OK(hipMalloc(&p, n * 4));
    hipLaunchKernelGGL(fill_f32, dim3(2048), dim3(256), 0, 0, p, n, seed, scale, offset);
    return p;
}

typedef int (*gemm_t)(const ina_gemm_args*, void*);
typedef int (*norm_t)(const ina_norm_args*, void*);
typedef int (*chain_t)(const ina_dit_rowchain_args*, void*);
typedef const char* (*err_t)(void);

int main(int argc, char** argv) {
    const char* lib = argc > 1 ? argv[1] : "internnav_amd/libinternnav_amd.so";
    void* h = dlopen(lib, RTLD_NOW | RTLD_LOCAL);
    if (!h) { fprintf(stderr, "dlopen %s: %s\n", lib, dlerror()); return 1; }
    gemm_t gemm = (gemm_t)dlsym(h, "ina_gemm_bf16");
    norm_t norm = (norm_t)dlsym(h, "ina_norm_bf16");
    chain_t chain = (chain_t)dlsym(h, "ina_dit_rowchain");
    err_t err = (err_t)dlsym(h, "ina_last_error");
    if (!gemm || !norm || !chain || !err) { fprintf(stderr, "missing symbols\n"); return 1; }
    printf("# %s\n", lib);
    const int F = argc > 2 ? atoi(argv[2]) : 1536;          // FFN width (1536: diffusers 0.33.1 as pinned; 1024: <= 0.32)
    const int abl = argc > 3 ? atoi(argv[3]) : 0;           // experimental builds only: ablation bits handed to the kernel in the reserved field
    const int D = 384, NL = 12, ST = 1024, MODLD = NL * 4 * D + D;
    printf("# FFN %d, ablation %d\n", F, abl);
    const int ENVS_MAX = 64;
    const size_t rows_max = (size_t)ENVS_MAX * ST;
    void *att = bf16_buf(rows_max * D, 1, 1.0f), *ff, *proj, *hbuf, *qkvq, *qkvq_ref;
    HIP_OK(hipMalloc(&ff, rows_max * F * 2));
    HIP_OK(hipMalloc(&proj, rows_max * D * 2));
    HIP_OK(hipMalloc(&hbuf, rows_max * D * 2));
    HIP_OK(hipMalloc(&qkvq, rows_max * 4 * D * 2));
    HIP_OK(hipMalloc(&qkvq_ref, rows_max * 4 * D * 2));
    float *x0 = f32_buf(rows_max * D, 2, 1.5f, 0.0f), *x, *x_ref;
    HIP_OK(hipMalloc(&x, rows_max * D * 4));
    HIP_OK(hipMalloc(&x_ref, rows_max * D * 4));
    float* mod = f32_buf((size_t)ENVS_MAX * MODLD, 3, 0.5f, 0.0f);
    std::vector<void*> wo(NL), w13(NL), w2(NL), wq(NL);
    std::vector<float*> n2(NL), fn1(NL), fn2(NL), n1(NL);
    for (int l = 0; l < NL; ++l) {
        wo[l] = bf16_buf((size_t)D * D, 10 + l, 0.09f);
        w13[l] = bf16_buf((size_t)2 * F * D, 30 + l, 0.09f);
        w2[l] = bf16_buf((size_t)D * F, 50 + l, 0.055f);
        wq[l] = bf16_buf((size_t)4 * D * D, 70 + l, 0.09f);
        n2[l] = f32_buf(D, 90 + l, 0.2f, 1.0f); fn1[l] = f32_buf(D, 110 + l, 0.2f, 1.0f);
        fn2[l] = f32_buf(D, 130 + l, 0.2f, 1.0f); n1[l] = f32_buf(D, 150 + l, 0.2f, 1.0f);
    }
    float* dmax;
    HIP_OK(hipMalloc(&dmax, 8));
    HIP_OK(hipDeviceSynchronize());

    auto run_gemm = [&](const void* A, int lda, const void* W, int K, int N, void* C, int glu) {
        ina_gemm_args a;
        memset(&a, 0, sizeof a);
        a.A = A; a.W = W; a.C = C; a.K = K; a.N = N; a.lda = lda; a.ldw = K; a.ldc = glu ? N / 2 : N; a.out_dtype = INA_BF16; a.batch = 1;
        a.glu = glu; a.act = glu ? INA_ACT_SILU_C : INA_ACT_NONE_C;
        return a;
    };
    hipEvent_t e0, e1;
    HIP_OK(hipEventCreate(&e0)); HIP_OK(hipEventCreate(&e1));
    const int env_cases[] = {64, 57, 16};
    for (int envs : env_cases) {
        const int M = envs * ST;
        auto unfused = [&](float* X) {
            for (int l = 0; l < NL; ++l) {
                const float* m = mod + (size_t)l * 4 * D;
                ina_gemm_args g = run_gemm(att, D, wo[l], D, D, proj, 0);
                g.M = M;
                if (gemm(&g, nullptr)) return 1;
                ina_norm_args n;
                memset(&n, 0, sizeof n);
                n.X = proj; n.x_dtype = INA_BF16; n.ldx = D; n.rows = M; n.C = D; n.gamma = n2[l]; n.gate = m + D; n.G = X; n.g_dtype = INA_F32; n.ldg = D;
                n.Y32 = X; n.ldy32 = D; n.Y2 = hbuf; n.ldy2 = D; n.gamma2 = fn1[l]; n.mod_scale2 = m + 2 * D; n.mod_div = ST; n.mod_ld = MODLD; n.rms = 1; n.eps = 1e-5f;
                if (norm(&n, nullptr)) return 1;
                g = run_gemm(hbuf, D, w13[l], D, 2 * F, ff, 1);
                g.M = M;
                if (gemm(&g, nullptr)) return 1;
                g = run_gemm(ff, F, w2[l], F, D, proj, 0);
                g.M = M;
                if (gemm(&g, nullptr)) return 1;
                const int ln = (l + 1) % NL;
                n.gamma = fn2[l]; n.gate = m + 3 * D; n.gamma2 = n1[ln]; n.mod_scale2 = mod + (size_t)ln * 4 * D;
                if (norm(&n, nullptr)) return 1;
                g = run_gemm(hbuf, D, wq[ln], D, 4 * D, qkvq, 0);
                g.M = M;
                if (gemm(&g, nullptr)) return 1;
            }
            return 0;
        };
        auto chained = [&](float* X, int waves) {
            for (int l = 0; l < NL; ++l) {
                const float* m = mod + (size_t)l * 4 * D;
                ina_dit_rowchain_args c;
                memset(&c, 0, sizeof c);
                c.A = att; c.W1 = wo[l]; c.gamma1 = n2[l]; c.gate = m + D; c.X = X; c.gamma2 = fn1[l]; c.mod_scale2 = m + 2 * D; c.W2 = w13[l]; c.C2 = ff;
                c.M = M; c.K1 = D; c.N2 = 2 * F; c.lda = D; c.ldw1 = D; c.ldx = D; c.ldw2 = D; c.ldc2 = F; c.glu2 = 1; c.mod_div = ST; c.mod_ld = MODLD; c.eps = 1e-5f;
                c._reserved = abl;
                (void)waves;
                if (chain(&c, nullptr)) return 1;
                const int ln = (l + 1) % NL;
                if (F > 1024) {      // the second half as the engine runs it at FFN 1536: GEMM, norm (+ next pre-norm), GEMM
                    ina_gemm_args g = run_gemm(ff, F, w2[l], F, D, proj, 0);
                    g.M = M;
                    if (gemm(&g, nullptr)) return 1;
                    ina_norm_args n;
                    memset(&n, 0, sizeof n);
                    n.X = proj; n.x_dtype = INA_BF16; n.ldx = D; n.rows = M; n.C = D; n.gamma = fn2[l]; n.gate = m + 3 * D; n.G = X; n.g_dtype = INA_F32; n.ldg = D;
                    n.Y32 = X; n.ldy32 = D; n.Y2 = hbuf; n.ldy2 = D; n.gamma2 = n1[ln]; n.mod_scale2 = mod + (size_t)ln * 4 * D; n.mod_div = ST; n.mod_ld = MODLD; n.rms = 1; n.eps = 1e-5f;
                    if (norm(&n, nullptr)) return 1;
                    g = run_gemm(hbuf, D, wq[ln], D, 4 * D, qkvq, 0);
                    g.M = M;
                    if (gemm(&g, nullptr)) return 1;
                    continue;
                }
                c.A = ff; c.W1 = w2[l]; c.gamma1 = fn2[l]; c.gate = m + 3 * D; c.gamma2 = n1[ln]; c.mod_scale2 = mod + (size_t)ln * 4 * D; c.W2 = wq[ln]; c.C2 = qkvq;
                c.K1 = F; c.N2 = 4 * D; c.lda = F; c.ldw1 = F; c.ldc2 = 4 * D; c.glu2 = 0;
                if (chain(&c, nullptr)) return 1;
            }
            return 0;
        };
        // reference pass (unfused) for the difference check
        HIP_OK(hipMemcpy(x_ref, x0, (size_t)M * D * 4, hipMemcpyDeviceToDevice));
        if (unfused(x_ref)) { fprintf(stderr, "unfused: %s\n", err()); return 3; }
        HIP_OK(hipMemcpy(qkvq_ref, qkvq, (size_t)M * 4 * D * 2, hipMemcpyDeviceToDevice));
        HIP_OK(hipDeviceSynchronize());
        const int reps = 5;
        for (int mode = 0; mode < 2; ++mode) {
            const int waves = 4;
            HIP_OK(hipMemcpy(x, x0, (size_t)M * D * 4, hipMemcpyDeviceToDevice));
            if (mode == 0 ? unfused(x) : chained(x, waves)) { fprintf(stderr, "mode %d: %s\n", mode, err()); return 3; }
            float d[2] = {0.f, 0.f};
            HIP_OK(hipMemset(dmax, 0, 8));
            hipLaunchKernelGGL(maxdiff_f32, dim3(1024), dim3(256), 0, 0, x, x_ref, (size_t)M * D, dmax);
            hipLaunchKernelGGL(maxdiff_bf16, dim3(1024), dim3(256), 0, 0, (const uint16_t*)qkvq, (const uint16_t*)qkvq_ref, (size_t)M * 4 * D, dmax + 1);
            HIP_OK(hipMemcpy(d, dmax, 8, hipMemcpyDeviceToHost));
            HIP_OK(hipEventRecord(e0, 0));
            for (int i = 0; i < reps; ++i) {
                if (mode == 0 ? unfused(x) : chained(x, waves)) { fprintf(stderr, "mode %d: %s\n", mode, err()); return 3; }
            }
            HIP_OK(hipEventRecord(e1, 0));
            HIP_OK(hipEventSynchronize(e1));
            float ms = 0;
            HIP_OK(hipEventElapsedTime(&ms, e0, e1));
            const double us = ms * 1e3 / (reps * NL);
            const double flop = 2.0 * M * D * (D + 2.0 * F + F + 4.0 * D);
            printf("%2d envs (%6d rows)  %-28s %8.1f us per block-step  %6.1f TF/s   max|dx| vs unfused %.3e (after 12 blocks)  max|dqkvq| %.3e\n", envs, M,
                   mode == 0 ? "unfused (6 launches)" : (F > 1024 ? "chain launch a + 3 launches" : "row chain, 128-row panels"), us, flop / us * 1e-6, d[0], d[1]);
        }
    }
    return 0;
}
