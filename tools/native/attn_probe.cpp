// Torch-free probe of ina_attention_bf16 over the C-ABI at the attention shapes of the hot path: LLM prefill (GQA 28 / 4 x d128, causal, 920
// tokens), Qwen ViT full attention (784 tokens x 16 heads x d80) and window attention (varlen, 64-token windows), DINOv2 (257 x 6 x d64), and
// the decode / latent-query passes against the KV cache (packed q|k|v rows as the engine lays them out are not needed: plain strided tensors).
// Prints time, algorithmic TF/s (unmasked keys), the HBM rate of the q / k / v / o bytes and a checksum of the output.
// Build: tools/native/build.sh; run from the repo root: tools/native/attn_probe [lib]
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
__global__ void checksum(const uint32_t* p, size_t n, unsigned long long* out) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    unsigned long long s = 0;
    for (; i < n; i += stride) s += (unsigned long long)p[i] * (2 * i + 1);
    atomicAdd(out, s);
}

int main(int argc, char** argv) {
    const char* lib = argc > 1 ? argv[1] : "internnav_amd/libinternnav_amd.so";
    void* h = dlopen(lib, RTLD_NOW | RTLD_LOCAL);
    if (!h) { fprintf(stderr, "dlopen %s: %s\n", lib, dlerror()); return 1; }
    typedef int (*fn_t)(const ina_attn_args*, void*);
    typedef const char* (*err_t)(void);
    fn_t attn = (fn_t)dlsym(h, "ina_attention_bf16");
    err_t err = (err_t)dlsym(h, "ina_last_error");
    if (!attn || !err) { fprintf(stderr, "missing symbols\n"); return 1; }
    printf("# %s\n", lib);
    // one pool of q / k / v / o, large enough for every case; tensors are [B, L, H, D] contiguous
    const size_t POOL = (size_t)32 * 1024 * 1024;      // elements
    uint16_t *Q, *K, *V, *O;
    for (uint16_t** p : {&Q, &K, &V, &O}) HIP_OK(hipMalloc(p, POOL * 2));
    hipLaunchKernelGGL(fill_bf16, dim3(2048), dim3(256), 0, 0, Q, POOL, 1u, 1.0f);
    hipLaunchKernelGGL(fill_bf16, dim3(2048), dim3(256), 0, 0, K, POOL, 2u, 1.0f);
    hipLaunchKernelGGL(fill_bf16, dim3(2048), dim3(256), 0, 0, V, POOL, 3u, 1.0f);
    unsigned long long* cs;
    HIP_OK(hipMalloc(&cs, 8));
    // window attention: 28 frames x 784 tokens cut into 64-token windows (the real plan has a few shorter edge windows; equal windows here)
    const int WIN = 64, NTOK = 28 * 784;
    std::vector<int32_t> cu(NTOK / WIN + 1);
    for (size_t i = 0; i < cu.size(); ++i) cu[i] = (int32_t)(i * WIN);
    int32_t* d_cu;
    HIP_OK(hipMalloc(&d_cu, cu.size() * 4));
    HIP_OK(hipMemcpy(d_cu, cu.data(), cu.size() * 4, hipMemcpyHostToDevice));
    HIP_OK(hipDeviceSynchronize());

    struct Case { const char* name; int B, Lq, Lk, H, Hkv, D, causal, varlen, kernel; };
    const Case cases[] = {{"LLM prefill      7 x 920, 28 / 4 heads x d128, causal", 7, 920, 920, 28, 4, 128, 1, 0},
                          {"LLM prefill half 4 x 920", 4, 920, 920, 28, 4, 128, 1, 0},
                          {"ViT full         28 x 784, 16 heads x d80", 28, 784, 784, 16, 16, 80, 0, 0},
                          {"ViT windows      343 x 64 (varlen), 16 heads x d80, 16-row kernel", 343, 64, 64, 16, 16, 80, 0, 1, 1},
                          {"ViT windows      343 x 64 (varlen), 32-row kernel (2 waves, 1 buffer)", 343, 64, 64, 16, 16, 80, 0, 1, 2},
                          {"ViT windows half 196 x 64 (varlen), 16-row kernel", 196, 64, 64, 16, 16, 80, 0, 1, 1},
                          {"ViT windows half 196 x 64 (varlen), 32-row kernel", 196, 64, 64, 16, 16, 80, 0, 1, 2},
                          {"DINOv2           128 x 257, 6 heads x d64", 128, 257, 257, 6, 6, 64, 0, 0},
                          {"decode           7 x (1 query, 927 keys), 28 / 4 x d128", 7, 1, 927, 28, 4, 128, 1, 0},
                          {"latent queries   7 x (5 queries, 933 keys)", 7, 5, 933, 28, 4, 128, 1, 0},
                          {"decode           7 x (1, 927), 4-wave one-launch kernel (kernel 3)", 7, 1, 927, 28, 4, 128, 1, 0, 3},
                          {"latent queries   7 x (5, 933), 4-wave one-launch kernel (kernel 3)", 7, 5, 933, 28, 4, 128, 1, 0, 3},
                          {"decode           7 x (1, 927), split + combine pair (kernel 1)", 7, 1, 927, 28, 4, 128, 1, 0, 1},
                          {"decode          13 x (1, 927), 28 / 4 x d128", 13, 1, 927, 28, 4, 128, 1, 0},
                          {"decode          13 x (1, 927), 4-wave one-launch kernel (kernel 3)", 13, 1, 927, 28, 4, 128, 1, 0, 3}};
    hipEvent_t e0, e1;
    HIP_OK(hipEventCreate(&e0)); HIP_OK(hipEventCreate(&e1));
    for (const Case& c : cases) {
        ina_attn_args a;
        memset(&a, 0, sizeof a);
        a.Q = Q; a.K = K; a.V = V; a.O = O;
        a.H = c.H; a.Hkv = c.Hkv; a.D = c.D; a.causal = c.causal; a.kv_bdiv = 1; a.kernel = c.kernel;
        a.scale = 1.0f / sqrtf((float)c.D);
        size_t q_elems, kv_elems;
        if (c.varlen) {
            a.B = c.B; a.Lq = c.Lq; a.Lk = c.Lk; a.cu_q = d_cu; a.cu_k = d_cu;
            a.q_rs = (int64_t)c.H * c.D; a.q_hs = c.D; a.k_rs = (int64_t)c.Hkv * c.D; a.k_hs = c.D; a.v_rs = a.k_rs; a.v_hs = c.D; a.o_rs = a.q_rs; a.o_hs = c.D;
            q_elems = (size_t)c.B * WIN * c.H * c.D; kv_elems = (size_t)c.B * WIN * c.Hkv * c.D;
        } else {
            a.B = c.B; a.Lq = c.Lq; a.Lk = c.Lk;
            a.q_hs = c.D; a.q_rs = (int64_t)c.H * c.D; a.q_bs = a.q_rs * c.Lq;
            a.k_hs = c.D; a.k_rs = (int64_t)c.Hkv * c.D; a.k_bs = a.k_rs * c.Lk;
            a.v_hs = a.k_hs; a.v_rs = a.k_rs; a.v_bs = a.k_bs;
            a.o_hs = a.q_hs; a.o_rs = a.q_rs; a.o_bs = a.q_bs;
            q_elems = (size_t)c.B * c.Lq * c.H * c.D; kv_elems = (size_t)c.B * c.Lk * c.Hkv * c.D;
        }
        if (q_elems > POOL || kv_elems > POOL) { printf("%-62s too large for the pool\n", c.name); continue; }
        HIP_OK(hipMemset(O, 0, q_elems * 2));
        if (attn(&a, nullptr) != 0) { fprintf(stderr, "ina_attention_bf16 (%s): %s\n", c.name, err()); return 3; }
        HIP_OK(hipMemset(cs, 0, 8));
        hipLaunchKernelGGL(checksum, dim3(256), dim3(256), 0, 0, (const uint32_t*)O, q_elems / 2, cs);
        unsigned long long v = 0;
        HIP_OK(hipMemcpy(&v, cs, 8, hipMemcpyDeviceToHost));
        const int reps = 20;
        HIP_OK(hipEventRecord(e0, 0));
        for (int i = 0; i < reps; ++i) attn(&a, nullptr);
        HIP_OK(hipEventRecord(e1, 0));
        HIP_OK(hipEventSynchronize(e1));
        float ms = 0;
        HIP_OK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms * 1e3 / reps;
        const double keys = c.causal ? 0.5 * ((double)c.Lk + (double)(c.Lk - c.Lq) + 1.0) : (double)c.Lk;
        const double flop = 4.0 * (double)c.B * c.Lq * c.H * keys * c.D;
        const double bytes = 2.0 * (2.0 * q_elems + 2.0 * kv_elems);
        printf("%-62s %8.1f us  %6.1f TF/s  %5.2f TB/s  checksum %016llx\n", c.name, us, flop / us * 1e-6, bytes / us * 1e-6, v);
    }
    return 0;
}
