// Torch-free probe of ina_head3 (final norm + Linear(C, 3) + sampler update) over the C-ABI: the NextDiT Euler call (65536 x 384 fp32 rows,
// adaLN modulation, no affine), the NavDP DDPM call (affine LayerNorm, noise), the prediction-only call, a bf16 input and a small call.
// Prints the time per launch and a checksum of the outputs after three launches: two builds of the library must print the same checksums.
// Build: tools/native/build.sh; run from the repo root: tools/native/head3_probe [lib]
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

__global__ void fill_f32(float* p, size_t n, uint32_t seed, float scale, float offset) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        uint32_t h = (uint32_t)i * 0x9E3779B9u ^ seed;
        h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
        p[i] = ((h >> 8) * (1.0f / 8388608.0f) - 1.0f) * scale + offset;
    }
}
__global__ void to_bf16(const float* x, uint16_t* y, size_t n) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        uint32_t b = __float_as_uint(x[i]);
        b += 0x7FFFu + ((b >> 16) & 1u);
        y[i] = (uint16_t)(b >> 16);
    }
}
__global__ void checksum(const uint32_t* p, size_t n, unsigned long long* out) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    unsigned long long s = 0;
    for (; i < n; i += stride) s += (unsigned long long)p[i] * (2 * i + 1);
    atomicAdd(out, s);
}
static float* f32_buf(size_t n, uint32_t seed, float scale, float offset) {
    float* p;
    HIP_OK(hipMalloc(&p, n * 4));
    hipLaunchKernelGGL(fill_f32, dim3(1024), dim3(256), 0, 0, p, n, seed, scale, offset);
    return p;
}

int main(int argc, char** argv) {
    const char* lib = argc > 1 ? argv[1] : "internnav_amd/libinternnav_amd.so";
    void* h = dlopen(lib, RTLD_NOW | RTLD_LOCAL);
    if (!h) { fprintf(stderr, "dlopen %s: %s\n", lib, dlerror()); return 1; }
    typedef int (*fn_t)(const ina_head3_args*, void*);
    typedef const char* (*err_t)(void);
    fn_t head3 = (fn_t)dlsym(h, "ina_head3");
    err_t err = (err_t)dlsym(h, "ina_last_error");
    if (!head3 || !err) { fprintf(stderr, "missing symbols\n"); return 1; }
    printf("# %s\n", lib);
    const int C = 384, ROWS = 65536, ENVS = 64, MOD_LD = 6 * C;
    float* X = f32_buf((size_t)ROWS * C, 1, 2.0f, 0.1f);
    uint16_t* Xb;
    HIP_OK(hipMalloc(&Xb, (size_t)ROWS * C * 2));
    hipLaunchKernelGGL(to_bf16, dim3(1024), dim3(256), 0, 0, X, Xb, (size_t)ROWS * C);
    float* gamma = f32_buf(C, 2, 0.2f, 1.0f), *beta = f32_buf(C, 3, 0.1f, 0.0f), *W = f32_buf(3 * C, 4, 0.1f, 0.0f), *b = f32_buf(3, 5, 0.1f, 0.0f);
    float* mod = f32_buf((size_t)ENVS * MOD_LD, 6, 0.3f, 0.0f);
    float* sample = f32_buf((size_t)ROWS * 3, 7, 1.0f, 0.0f), *noise = f32_buf((size_t)ROWS * 3, 8, 1.0f, 0.0f), *eps_out = f32_buf((size_t)ROWS * 3, 9, 0.0f, 0.0f);
    unsigned long long* cs;
    HIP_OK(hipMalloc(&cs, 8));
    HIP_OK(hipDeviceSynchronize());

    struct Case { const char* name; int rows, mode, affine, modulated, bf16, noisy; };
    const Case cases[] = {{"NextDiT Euler step  65536 x 384 f32, adaLN", ROWS, 2, 0, 1, 0, 0},
                          {"DDPM step           65536 x 384 f32, affine LN + noise", ROWS, 1, 1, 0, 0, 1},
                          {"prediction only     65536 x 384 f32, adaLN", ROWS, 0, 0, 1, 0, 0},
                          {"Euler step          58368 x 384 bf16, adaLN", 57 * 1024, 2, 0, 1, 1, 0},
                          {"DDPM step            1024 x 384 f32 (one env)", 1024, 1, 1, 0, 0, 1},
                          {"Euler step          65535 x 384 f32 (ragged tail)", ROWS - 1, 2, 0, 1, 0, 0}};
    hipEvent_t e0, e1;
    HIP_OK(hipEventCreate(&e0)); HIP_OK(hipEventCreate(&e1));
    for (const Case& c : cases) {
        hipLaunchKernelGGL(fill_f32, dim3(1024), dim3(256), 0, 0, sample, (size_t)ROWS * 3, 7u, 1.0f, 0.0f);   // same start for every case / library
        hipLaunchKernelGGL(fill_f32, dim3(1024), dim3(256), 0, 0, eps_out, (size_t)ROWS * 3, 9u, 0.0f, 0.0f);
        ina_head3_args a;
        memset(&a, 0, sizeof a);
        a.X = c.bf16 ? (const void*)Xb : (const void*)X;
        a.x_dtype = c.bf16 ? INA_BF16 : INA_F32;
        a.gamma = c.affine ? gamma : nullptr; a.beta = c.affine ? beta : nullptr;
        a.mod_scale = c.modulated ? mod + 3 * C : nullptr;        // a column slice of the modulation matrix, as the engine passes it
        a.mod_ld = MOD_LD; a.mod_div = 1024;
        a.W = W; a.b = b;
        a.sample = c.mode ? sample : nullptr; a.noise = c.noisy ? noise : nullptr; a.eps_out = c.mode == 0 ? eps_out : nullptr;
        a.coef[0] = c.mode == 2 ? -0.1f : 1.02f; a.coef[1] = 0.2f; a.coef[2] = 0.3f; a.coef[3] = 0.69f; a.coef[4] = 0.05f;
        a.clip = 1.0f; a.eps = c.affine ? 1e-5f : 1e-6f;
        a.rows = c.rows; a.C = C; a.ldx = C; a.mode = c.mode;
        for (int i = 0; i < 3; ++i)
            if (head3(&a, nullptr) != 0) { fprintf(stderr, "ina_head3: %s\n", err()); return 3; }
        HIP_OK(hipMemset(cs, 0, 8));
        hipLaunchKernelGGL(checksum, dim3(256), dim3(256), 0, 0, (const uint32_t*)sample, (size_t)ROWS * 3, cs);
        hipLaunchKernelGGL(checksum, dim3(256), dim3(256), 0, 0, (const uint32_t*)eps_out, (size_t)ROWS * 3, cs);
        unsigned long long v = 0;
        HIP_OK(hipMemcpy(&v, cs, 8, hipMemcpyDeviceToHost));
        const int reps = 20;
        HIP_OK(hipEventRecord(e0, 0));
        for (int i = 0; i < reps; ++i) head3(&a, nullptr);
        HIP_OK(hipEventRecord(e1, 0));
        HIP_OK(hipEventSynchronize(e1));
        float ms = 0;
        HIP_OK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms * 1e3 / reps, bytes = (double)c.rows * C * (c.bf16 ? 2 : 4);
        printf("%-58s %8.1f us  %5.2f TB/s  checksum %016llx\n", c.name, us, bytes / us * 1e-6, v);
    }
    return 0;
}
