// Torch-free probe of ina_embed3 (nn.Linear(3, C) + bias + positional table, scattered through a row map) over the C-ABI: time per launch
// and a checksum of the output for the call shapes of the engines; two builds of the library must print the same checksums.
// Build: tools/native/build.sh; run from the repo root: tools/native/embed3_probe [lib]
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

__global__ void fill_f32(float* p, size_t n, uint32_t seed, float scale) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        uint32_t h = (uint32_t)i * 0x9E3779B9u ^ seed;
        h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
        p[i] = ((h >> 8) * (1.0f / 8388608.0f) - 1.0f) * scale;
    }
}
__global__ void checksum(const uint32_t* p, size_t n, unsigned long long* out) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    unsigned long long s = 0;
    for (; i < n; i += stride) s += (unsigned long long)p[i] * (2 * i + 1);
    atomicAdd(out, s);
}
static float* f32_buf(size_t n, uint32_t seed, float scale) {
    float* p;
    HIP_OK(hipMalloc(&p, n * 4));
    hipLaunchKernelGGL(fill_f32, dim3(1024), dim3(256), 0, 0, p, n, seed, scale);
    return p;
}

int main(int argc, char** argv) {
    const char* lib = argc > 1 ? argv[1] : "internnav_amd/libinternnav_amd.so";
    void* h = dlopen(lib, RTLD_NOW | RTLD_LOCAL);
    if (!h) { fprintf(stderr, "dlopen %s: %s\n", lib, dlerror()); return 1; }
    typedef int (*fn_t)(const ina_embed3_args*, void*);
    typedef const char* (*err_t)(void);
    fn_t embed3 = (fn_t)dlsym(h, "ina_embed3");
    err_t err = (err_t)dlsym(h, "ina_last_error");
    if (!embed3 || !err) { fprintf(stderr, "missing symbols\n"); return 1; }
    printf("# %s\n", lib);
    const int C = 384, ROWS = 65536;
    float* X = f32_buf((size_t)ROWS * 3, 1, 1.0f), *W = f32_buf((size_t)C * 3, 2, 0.5f), *b = f32_buf(C, 3, 0.1f), *P = f32_buf((size_t)64 * C, 4, 1.0f);
    const size_t ybytes = (size_t)ROWS * 2 * C * 4;                   // room for the scattered layouts
    void* Y;
    HIP_OK(hipMalloc(&Y, ybytes));
    unsigned long long* cs;
    HIP_OK(hipMalloc(&cs, 8));
    struct Case { const char* name; int rows, C, out_f32, has_x, has_b, has_p, p_mod, x_div, seg_len, seg_stride, off; };
    const Case cases[] = {{"action encoder  65536 x 384 f32 out, bias + 32-row table", ROWS, 384, 1, 1, 1, 1, 32, 1, 0, 0, 0},
                          {"same, bf16 out", ROWS, 384, 0, 1, 1, 1, 32, 1, 0, 0, 0},
                          {"table fill (no input vector), f32 out", ROWS, 384, 1, 0, 0, 1, 64, 1, 0, 0, 0},
                          {"scattered: rows of 32 into slots of 40 at offset 8, bf16", ROWS, 384, 0, 1, 1, 1, 32, 1, 32, 40, 8},
                          {"one vector per 32 rows (x_div 32), no table", ROWS, 384, 1, 1, 1, 0, 1, 32, 0, 0, 0},
                          {"small: 100 x 256 f32", 100, 256, 1, 1, 1, 1, 7, 1, 0, 0, 0}};
    hipEvent_t e0, e1;
    HIP_OK(hipEventCreate(&e0)); HIP_OK(hipEventCreate(&e1));
    for (const Case& c : cases) {
        HIP_OK(hipMemset(Y, 0, ybytes));
        ina_embed3_args a;
        memset(&a, 0, sizeof a);
        a.X = c.has_x ? X : nullptr; a.W = W; a.b = c.has_b ? b : nullptr; a.P = c.has_p ? P : nullptr; a.Y = Y;
        a.out_map.seg_len = c.seg_len; a.out_map.seg_stride = c.seg_stride; a.out_map.off = c.off;
        a.rows = c.rows; a.C = c.C; a.ldy = c.C; a.p_mod = c.p_mod; a.out_dtype = c.out_f32 ? INA_F32 : INA_BF16; a.x_div = c.x_div;
        if (embed3(&a, nullptr) != 0) { fprintf(stderr, "ina_embed3: %s\n", err()); return 3; }
        HIP_OK(hipMemset(cs, 0, 8));
        hipLaunchKernelGGL(checksum, dim3(256), dim3(256), 0, 0, (const uint32_t*)Y, ybytes / 4, cs);
        unsigned long long v = 0;
        HIP_OK(hipMemcpy(&v, cs, 8, hipMemcpyDeviceToHost));
        const int reps = 20;
        HIP_OK(hipEventRecord(e0, 0));
        for (int i = 0; i < reps; ++i) embed3(&a, nullptr);
        HIP_OK(hipEventRecord(e1, 0));
        HIP_OK(hipEventSynchronize(e1));
        float ms = 0;
        HIP_OK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms * 1e3 / reps, bytes = (double)c.rows * c.C * (c.out_f32 ? 4 : 2);
        printf("%-62s %8.1f us  %5.2f TB/s  checksum %016llx\n", c.name, us, bytes / us * 1e-6, v);
    }
    return 0;
}
