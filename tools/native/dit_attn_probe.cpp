// Torch-free probe of ina_dit_attention (the fused attention stage of a NextDiT block) over the C-ABI: time per launch, achieved HBM rate
// against the kernel's algorithmic bytes (5 x rows x D x 2 B) and a checksum of the output. Two builds of the library must print the same
// checksums when a change is meant to be bit-identical (restructured loads, occupancy, barriers).
// Build: tools/native/build.sh; run from the repo root: tools/native/dit_attn_probe [lib]
#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

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
__global__ void checksum(const uint32_t* p, size_t n, unsigned long long* out) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    unsigned long long s = 0;
    for (; i < n; i += stride) s += (unsigned long long)p[i] * (2 * i + 1);
    atomicAdd(out, s);
}
// (mean, rstd) of every D-wide segment of the X rows: what ina_dit_rowchain's seg_stats epilogue hands to the attention stage
__global__ void row_stats(const uint16_t* x, float* st, size_t nseg, int D, float eps) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= nseg) return;
    float s = 0.f, q = 0.f;
    for (int c = 0; c < D; ++c) {
        const float v = __uint_as_float((uint32_t)x[i * D + c] << 16);
        s += v; q += v * v;
    }
    const float m = s / D, var = fmaxf(q / D - m * m, 0.f);
    st[2 * i] = m; st[2 * i + 1] = rsqrtf(var + eps);
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
    hipLaunchKernelGGL(fill_f32, dim3(64), dim3(256), 0, 0, p, n, seed, scale, offset);
    return p;
}

int main(int argc, char** argv) {
    const char* lib = argc > 1 ? argv[1] : "internnav_amd/libinternnav_amd.so";
    void* h = dlopen(lib, RTLD_NOW | RTLD_LOCAL);
    if (!h) { fprintf(stderr, "dlopen %s: %s\n", lib, dlerror()); return 1; }
    typedef int (*fn_t)(const ina_dit_attn_args*, void*);
    typedef const char* (*err_t)(void);
    fn_t dit = (fn_t)dlsym(h, "ina_dit_attention");
    err_t err = (err_t)dlsym(h, "ina_last_error");
    if (!dit || !err) { fprintf(stderr, "missing symbols\n"); return 1; }
    printf("# %s\n", lib);
    const int NH = 6, D = NH * 64, ENVS = 64, SPE = 32, TMAX = 32, LZMAX = 64;
    const size_t rows_max = (size_t)ENVS * SPE * TMAX;
    void* X = bf16_buf(rows_max * 4 * D, 1, 2.0f);
    void* O;
    HIP_OK(hipMalloc(&O, rows_max * D * 2));
    void* KV = bf16_buf((size_t)ENVS * LZMAX * 2 * D, 2, 1.5f);          // [env][row][k | v][heads x 64]
    void* V2T;
    HIP_OK(hipMalloc(&V2T, (size_t)ENVS * NH * 64 * 64 * 2));
    float *g1 = f32_buf(D, 3, 0.2f, 1.0f), *b1 = f32_buf(D, 4, 0.1f, 0.0f), *g2 = f32_buf(D, 5, 0.2f, 1.0f), *b2 = f32_buf(D, 6, 0.1f, 0.0f),
          *g3 = f32_buf(D, 7, 0.2f, 1.0f), *b3 = f32_buf(D, 8, 0.1f, 0.0f), *gate = f32_buf(NH, 9, 1.0f, 0.0f);
    float* ST;
    HIP_OK(hipMalloc(&ST, rows_max * 4 * 2 * sizeof(float)));
    hipLaunchKernelGGL(row_stats, dim3((unsigned)((rows_max * 4 + 255) / 256)), dim3(256), 0, 0, (const uint16_t*)X, ST, rows_max * 4, D, 1e-5f);
    unsigned long long* cs;
    HIP_OK(hipMalloc(&cs, 8));
    HIP_OK(hipDeviceSynchronize());

    struct Case { const char* name; int envs, T, Lz, gated; };
    const Case cases[] = {{"64 envs x 32 samples, T 32, Lz 64, gated", 64, 32, 64, 1},
                          {"57 envs x 32 samples, T 32, Lz 36, gated", 57, 32, 36, 1},
                          {" 7 envs x 32 samples, T 32, Lz 36, gated", 7, 32, 36, 1},
                          {"64 envs x 32 samples, T 24, Lz 17, no gate", 64, 24, 17, 0}};
    hipEvent_t e0, e1;
    HIP_OK(hipEventCreate(&e0)); HIP_OK(hipEventCreate(&e1));
    for (const Case& c : cases) {
        ina_dit_attn_args a;
        memset(&a, 0, sizeof a);
        a.O = O;
        a.g_q1 = g1; a.b_q1 = b1; a.g_k1 = g2; a.b_k1 = b2; a.g_q2 = g3; a.b_q2 = b3;
        a.K2 = KV; a.V2T = V2T;
        a.V2T_src = (const uint16_t*)KV + D;                      // the v half of every condition row
        a.head_gate = c.gated ? gate : nullptr;
        a.k2_bs = (int64_t)c.Lz * 2 * D; a.k2_rs = 2 * D; a.v2_bs = a.k2_bs; a.v2_rs = a.k2_rs;
        a.nseq = c.envs * SPE; a.T = c.T; a.heads = NH; a.seq_per_env = SPE; a.Lz = c.Lz; a.ldx = 4 * D; a.ldo = D;
        a.scale = 0.125f; a.eps = 1e-5f;
        a.X = nullptr;                                             // first call: build the transposed condition-V image only
        if (dit(&a, nullptr) != 0) { fprintf(stderr, "ina_dit_attention (V2T): %s\n", err()); return 3; }
        a.V2T_src = nullptr; a.X = X;
        for (int with_stats = 0; with_stats < 2; ++with_stats) {
            a.stats = with_stats ? ST : nullptr;
            a.stats_ld = 8;
            HIP_OK(hipMemset(O, 0, rows_max * D * 2));
            if (dit(&a, nullptr) != 0) { fprintf(stderr, "ina_dit_attention: %s\n", err()); return 3; }
            HIP_OK(hipMemset(cs, 0, 8));
            hipLaunchKernelGGL(checksum, dim3(256), dim3(256), 0, 0, (const uint32_t*)O, rows_max * D / 2, cs);
            unsigned long long v = 0;
            HIP_OK(hipMemcpy(&v, cs, 8, hipMemcpyDeviceToHost));
            const int reps = 20;
            HIP_OK(hipEventRecord(e0, 0));
            for (int i = 0; i < reps; ++i) dit(&a, nullptr);
            HIP_OK(hipEventRecord(e1, 0));
            HIP_OK(hipEventSynchronize(e1));
            float ms = 0;
            HIP_OK(hipEventElapsedTime(&ms, e0, e1));
            const double us = ms * 1e3 / reps, bytes = 5.0 * a.nseq * c.T * D * 2.0;
            printf("%-46s %-14s %8.1f us  %5.2f TB/s  checksum %016llx\n", c.name, with_stats ? "stats given" : "own stats", us, bytes / us * 1e-6, v);
        }
    }
    return 0;
}
