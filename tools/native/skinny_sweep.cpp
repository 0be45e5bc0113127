// Torch-free probe of the weight-streaming GEMMs of a System-2 decode pass over the C-ABI: 28 decoder layers (fused-RMSNorm q|k|v, o + f32
// residual, fused-RMSNorm gate|up SwiGLU, down + f32 residual; M = 7 sequences) + lm_head, with 8 DISTINCT layers of weights cycled (3.7 GB,
// so nothing is served from the 256 MB Infinity Cache - as in the real pass, whose 15.2 GB are read once). Prints the time of a pass, the
// per-GEMM average inside it with its HBM rate, and a checksum of the outputs (two builds of the library must print the same one).
// Build: tools/native/build.sh; run from the repo root: tools/native/skinny_sweep [lib] [passes]
#include <hip/hip_runtime.h>

#include <dlfcn.h>

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

typedef int (*gemm_fn)(const ina_gemm_args*, void*);
typedef const char* (*err_fn)(void);
static gemm_fn g_gemm;
static err_fn g_err;

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
__global__ void checksum(const uint32_t* p, size_t n, unsigned long long* out) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    unsigned long long s = 0;
    for (; i < n; i += stride) s += (unsigned long long)p[i] * (2 * i + 1);
    atomicAdd(out, s);
}

static void* bf16_buf(size_t elems, uint32_t seed, float scale) {
    void* p;
    HIP_OK(hipMalloc(&p, elems * 2));
    hipLaunchKernelGGL(fill_bf16, dim3(2048), dim3(256), 0, 0, (uint16_t*)p, elems, seed, scale);
    return p;
}
static float* f32_buf(size_t elems, uint32_t seed, float scale, float offset) {
    float* p;
    HIP_OK(hipMalloc(&p, elems * 4));
    hipLaunchKernelGGL(fill_f32, dim3(256), dim3(256), 0, 0, p, elems, seed, scale, offset);
    return p;
}

int main(int argc, char** argv) {
    const char* lib = argc > 1 ? argv[1] : "internnav_amd/libinternnav_amd.so";
    const int passes = argc > 2 ? atoi(argv[2]) : 4;
    const int group_m = argc > 3 ? atoi(argv[3]) : 0;      // 7 = non-temporal weight loads (SK_NT_FLAG)
    const int force_cfg = argc > 4 ? atoi(argv[4]) : 0;    // 60 / 61: four-wave builds
    void* h = dlopen(lib, RTLD_NOW | RTLD_LOCAL);
    if (!h) { fprintf(stderr, "dlopen %s: %s\n", lib, dlerror()); return 1; }
    g_gemm = (gemm_fn)dlsym(h, "ina_gemm_bf16");
    g_err = (err_fn)dlsym(h, "ina_last_error");
    if (!g_gemm || !g_err) { fprintf(stderr, "missing symbols\n"); return 1; }
    printf("# %s  group_m %d  force_cfg %d\n", lib, group_m, force_cfg);

    const int M = 7, H = 3584, I = 18944, QKV = 4608, V = 152064, NL = 8, LAYERS = 28;
    void *Wq[NL], *Wo[NL], *Wg[NL], *Wd[NL];
    float *g1[NL], *g2[NL];
    for (int l = 0; l < NL; ++l) {
        Wq[l] = bf16_buf((size_t)QKV * H, 10 + l, 0.02f); Wo[l] = bf16_buf((size_t)H * H, 20 + l, 0.02f);
        Wg[l] = bf16_buf((size_t)2 * I * H, 30 + l, 0.02f); Wd[l] = bf16_buf((size_t)H * I, 40 + l, 0.01f);
        g1[l] = f32_buf(H, 50 + l, 0.1f, 1.0f); g2[l] = f32_buf(H, 60 + l, 0.1f, 1.0f);
    }
    void* Wlm = bf16_buf((size_t)V * H, 99, 0.02f);
    float* x = f32_buf((size_t)M * H, 1, 1.0f, 0.0f);            // residual stream
    void* qkv = bf16_buf((size_t)M * QKV, 2, 1.0f);
    void* att = bf16_buf((size_t)M * H, 3, 1.0f);                // stands for the attention output (the probe runs no attention)
    void* ff = bf16_buf((size_t)M * I, 4, 1.0f);
    void* xn = bf16_buf((size_t)M * H, 5, 1.0f);                 // normed last hidden state for lm_head
    float* logits;
    HIP_OK(hipMalloc(&logits, (size_t)M * V * 4));
    unsigned long long* cs;
    HIP_OK(hipMalloc(&cs, 8));
    HIP_OK(hipDeviceSynchronize());

    auto base = [&](const void* A, const void* W, void* C, int N, int K) {
        ina_gemm_args a;
        memset(&a, 0, sizeof a);
        a.A = A; a.W = W; a.C = C; a.M = M; a.N = N; a.K = K; a.lda = K; a.ldw = K; a.ldc = N; a.ldr = N;
        a.out_dtype = INA_BF16; a.res_dtype = INA_F32; a.rowscale_div = 1; a.batch = 1; a.group_m = group_m; a.force_cfg = force_cfg;
        return a;
    };
    auto go = [&](const ina_gemm_args& a, hipStream_t st) {
        if (g_gemm(&a, (void*)st) != 0) { fprintf(stderr, "ina_gemm_bf16: %s\n", g_err()); exit(3); }
    };
    hipStream_t st;
    HIP_OK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    const char* names[5] = {"q|k|v (fused RMSNorm)", "o + residual", "gate|up SwiGLU (fused RMSNorm)", "down + residual", "lm_head"};
    const double bytes[5] = {2.0 * QKV * H, 2.0 * H * H, 4.0 * I * H, 2.0 * H * I, 2.0 * V * H};
    std::vector<hipEvent_t> ev((LAYERS * 4 + 1) * 2);
    for (auto& e : ev) HIP_OK(hipEventCreate(&e));
    auto pass = [&](bool timed) {
        int k = 0;
        auto rec = [&]() { if (timed) HIP_OK(hipEventRecord(ev[k], st)); ++k; };
        for (int l = 0; l < LAYERS; ++l) {
            const int w = l % NL;
            ina_gemm_args a = base(x, Wq[w], qkv, QKV, H);
            a.norm_gamma = g1[w]; a.norm_eps = 1e-6f; a.a_dtype = INA_F32;
            rec(); go(a, st); rec();
            a = base(att, Wo[w], x, H, H);
            a.R = x; a.out_dtype = INA_F32;
            rec(); go(a, st); rec();
            a = base(x, Wg[w], ff, 2 * I, H);
            a.norm_gamma = g2[w]; a.norm_eps = 1e-6f; a.a_dtype = INA_F32; a.glu = 1; a.act = INA_ACT_SILU_C; a.ldc = I;
            rec(); go(a, st); rec();
            a = base(ff, Wd[w], x, H, I);
            a.R = x; a.out_dtype = INA_F32;
            rec(); go(a, st); rec();
        }
        ina_gemm_args a = base(xn, Wlm, logits, V, H);
        a.out_dtype = INA_F32;
        rec(); go(a, st); rec();
    };
    hipEvent_t e0, e1;
    HIP_OK(hipEventCreate(&e0)); HIP_OK(hipEventCreate(&e1));
    pass(false);                                       // warm-up
    HIP_OK(hipStreamSynchronize(st));
    const double pass_bytes = LAYERS * (bytes[0] + bytes[1] + bytes[2] + bytes[3]) + bytes[4];
    double best = 1e30;
    for (int r = 0; r < passes; ++r) {
        HIP_OK(hipEventRecord(e0, st));
        pass(false);
        HIP_OK(hipEventRecord(e1, st));
        HIP_OK(hipEventSynchronize(e1));
        float ms = 0;
        HIP_OK(hipEventElapsedTime(&ms, e0, e1));
        printf("pass %d: %7.3f ms  %6.2f TB/s over the %.2f GB of weights\n", r, ms, pass_bytes / (ms * 1e-3) * 1e-12, pass_bytes * 1e-9);
        if (ms < best) best = ms;
    }
    pass(true);
    HIP_OK(hipStreamSynchronize(st));
    for (int g = 0; g < 5; ++g) {
        double us = 0;
        const int n = g < 4 ? LAYERS : 1;
        for (int l = 0; l < n; ++l) {
            const int k = g < 4 ? (l * 4 + g) * 2 : LAYERS * 4 * 2;
            float ms = 0;
            HIP_OK(hipEventElapsedTime(&ms, ev[k], ev[k + 1]));
            us += ms * 1e3;
        }
        us /= n;
        printf("  %-32s %8.1f us  %6.2f TB/s  (%.1f MB)\n", names[g], us, bytes[g] / us * 1e-6, bytes[g] * 1e-6);
    }
    HIP_OK(hipMemset(cs, 0, 8));
    hipLaunchKernelGGL(checksum, dim3(256), dim3(256), 0, 0, (const uint32_t*)x, (size_t)M * H, cs);
    hipLaunchKernelGGL(checksum, dim3(256), dim3(256), 0, 0, (const uint32_t*)logits, (size_t)M * V, cs);
    hipLaunchKernelGGL(checksum, dim3(256), dim3(256), 0, 0, (const uint32_t*)ff, (size_t)M * I / 2, cs);
    unsigned long long c = 0;
    HIP_OK(hipMemcpy(&c, cs, 8, hipMemcpyDeviceToHost));
    printf("best pass %7.3f ms  %6.2f TB/s   checksum %016llx\n", best, pass_bytes / (best * 1e-3) * 1e-12, c);
    return 0;
}
