// Torch-free GEMM sweep over the C-ABI (libinternnav_amd.so) for the System-2 prefill shapes: starts in seconds on a fresh GPU box
// (no `import torch`), so a short gpurun call is enough. Build: tools/native/build.sh (hipcc, gfx950); run from the repo root.
//
//   1. isolated launches: every prefill GEMM shape (LLM 28 / 4 heads x d128, H 3584, I 18944; Qwen ViT width 1280, I 3456) at the row counts
//      of the joint (7 / 6 envs) and the split (4 / 3 / 3 envs) micro-batches x tile configs (force_cfg) x tile orders (group_m);
//      outputs compared BIT FOR BIT against the auto-selected config on the device (every tile shape accumulates K in the same order);
//   2. the decoder-layer GEMM chain (qkv -> o -> gate|up -> down, 28 layers back to back = the sustained in-situ state) for the joint
//      batch on one stream and for the two halves on two streams, under different tile policies.
// Output: one line per measurement on stdout (the caller redirects it under gpurun_out/).
#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
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
typedef int (*shuf_fn)(const void*, void*, int32_t, int32_t, int64_t, void*);
static gemm_fn g_gemm;
static err_fn g_err;
static shuf_fn g_shuffle;
static const void* g_Wp;   // fragment-ordered copy of the current W (tile config 40)

__global__ void fill_bf16(uint16_t* p, size_t n, uint32_t seed, float scale) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        uint32_t h = (uint32_t)i * 0x9E3779B9u ^ seed ^ (uint32_t)(i >> 32) * 0x85EBCA6Bu;
        h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
        const float v = ((h >> 8) * (1.0f / 8388608.0f) - 1.0f) * scale;      // uniform [-scale, scale)
        uint32_t b = __float_as_uint(v);
        b += 0x7FFFu + ((b >> 16) & 1u);                                         // round to nearest even
        p[i] = (uint16_t)(b >> 16);
    }
}

__global__ void count_diff(const uint32_t* a, const uint32_t* b, size_t n32, unsigned long long* out) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    unsigned long long d = 0;
    for (; i < n32; i += stride) d += (a[i] != b[i]);
    if (d) atomicAdd(out, d);
}

// bf16 outputs that differ in bits (a kernel with another K summation order): max |a - b| and max |a| over the elements
__global__ void max_diff_bf16(const uint16_t* a, const uint16_t* b, size_t n, unsigned int* out) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    float d = 0.f, m = 0.f;
    for (; i < n; i += stride) {
        const float x = __uint_as_float((uint32_t)a[i] << 16), y = __uint_as_float((uint32_t)b[i] << 16);
        d = fmaxf(d, fabsf(x - y));
        m = fmaxf(m, fabsf(x));
    }
    atomicMax(out, __float_as_uint(d));          // non-negative floats order like their bit patterns
    atomicMax(out + 1, __float_as_uint(m));
}

struct Buf {
    void* p = nullptr;
    size_t bytes = 0;
    void alloc(size_t b) { bytes = b; HIP_OK(hipMalloc(&p, b)); }
};

static void fill(Buf& b, uint32_t seed, float scale) {
    hipLaunchKernelGGL(fill_bf16, dim3(2048), dim3(256), 0, 0, (uint16_t*)b.p, b.bytes / 2, seed, scale);
    HIP_OK(hipGetLastError());
}

struct Shape {
    const char* name;
    int M, N, K;
    int glu, res_f32, out_f32, act;
};

static ina_gemm_args make(const Shape& s, const void* A, const void* W, void* C, const void* R, int cfg, int group_m) {
    ina_gemm_args a;
    memset(&a, 0, sizeof a);
    a.A = A; a.W = W; a.C = C; a.R = s.res_f32 ? R : nullptr;
    a.M = s.M; a.N = s.N; a.K = s.K;
    a.lda = s.K; a.ldw = s.K;
    a.ldc = s.glu ? s.N / 2 : s.N;
    a.ldr = a.ldc;
    a.act = s.act;
    a.out_dtype = s.out_f32 ? INA_F32 : INA_BF16;
    a.res_dtype = INA_F32;
    a.glu = s.glu;
    a.rowscale_div = 1; a.batch = 1;
    a.force_cfg = cfg; a.group_m = group_m;
    a.Wp = cfg == 40 ? g_Wp : nullptr;
    return a;
}

static void launch(const ina_gemm_args& a, hipStream_t st) {
    if (g_gemm(&a, (void*)st) != 0) {
        fprintf(stderr, "ina_gemm_bf16 failed: %s\n", g_err());
        exit(3);
    }
}

static double time_us(const ina_gemm_args& a, int reps) {
    hipEvent_t e0, e1;
    HIP_OK(hipEventCreate(&e0)); HIP_OK(hipEventCreate(&e1));
    for (int i = 0; i < 2; ++i) launch(a, 0);
    HIP_OK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; ++i) launch(a, 0);
    HIP_OK(hipEventRecord(e1, 0));
    HIP_OK(hipEventSynchronize(e1));
    float ms = 0;
    HIP_OK(hipEventElapsedTime(&ms, e0, e1));
    HIP_OK(hipEventDestroy(e0)); HIP_OK(hipEventDestroy(e1));
    return ms * 1e3 / reps;
}

int main(int argc, char** argv) {
    const char* lib = argc > 1 ? argv[1] : "internnav_amd/libinternnav_amd.so";
    const bool quick = argc > 2 && !strcmp(argv[2], "quick");   // otherwise argv[2] names a spec file (section 0 below)
    void* h = dlopen(lib, RTLD_NOW | RTLD_LOCAL);
    if (!h) { fprintf(stderr, "dlopen %s: %s\n", lib, dlerror()); return 1; }
    g_gemm = (gemm_fn)dlsym(h, "ina_gemm_bf16");
    g_err = (err_fn)dlsym(h, "ina_last_error");
    g_shuffle = (shuf_fn)dlsym(h, "ina_gemm_preshuffle");
    typedef int (*chk_fn)(char*, int);
    chk_fn chk = (chk_fn)dlsym(h, "ina_device_check");
    if (!g_gemm || !g_err || !chk) { fprintf(stderr, "missing symbols\n"); return 1; }
    char arch[128] = {0};
    if (chk(arch, sizeof arch) != 0) { fprintf(stderr, "device check: %s\n", g_err()); return 1; }
    printf("# device %s\n", arch);

    const int H = 3584, I = 18944, QKV = 4608, VH = 1280, VI = 3456;
    const int rows_llm[] = {6440, 5520, 3680, 2760};         // 7 / 6 envs joint; 4 / 3 envs (the halves of the two-stream prefill)
    const int rows_vit[] = {21952, 18816, 12544, 9408};
    Buf A, W, Wp, C0, C1, R, cnt;
    Wp.alloc((size_t)2 * I * H * 2);
    A.alloc((size_t)21952 * I);                               // >= every A (bf16): 6440 x 18944 x 2 B = 244 MB, 21952 x 3456 x 2 B
    W.alloc((size_t)2 * I * H * 2);                           // gate|up weight 37888 x 3584 bf16
    C0.alloc((size_t)6440 * I * 2 + (size_t)21952 * 3840 * 4);
    C1.alloc(C0.bytes);
    R.alloc((size_t)21952 * 3840 * 4);
    cnt.alloc(8);
    fill(A, 1, 1.0f); fill(W, 2, 0.03f); fill(R, 3, 1.0f);
    HIP_OK(hipDeviceSynchronize());

    auto tf = [](const Shape& s, double us) { return 2.0 * s.M * s.N * s.K / us * 1e-6; };
    const int reps = quick ? 4 : 10;

    // ---------------------------------------------------------------------------------------------- 0. spec file: free-form sweeps
    // tools/native/gemm_sweep <lib> <spec file>: one line per shape, `name M N K glu res cfg[,group_m] cfg[,group_m] ...`; the first config
    // is the reference of the bit comparison (0 = auto); res = 1: f32 output with an f32 residual, else bf16 output; `#` starts a comment
    if (argc > 2 && !quick) {
        FILE* f = fopen(argv[2], "r");
        if (!f) { fprintf(stderr, "cannot open %s\n", argv[2]); return 1; }
        printf("# spec %s: us, TF/s; '=' bit-equal to the first config of the line\n", argv[2]);
        char line[1024];
        while (fgets(line, sizeof line, f)) {
            if (line[0] == '#' || line[0] == '\n') { if (line[0] == '#') fputs(line, stdout); continue; }
            char name[64];
            int M, N, K, glu, res, off = 0;
            if (sscanf(line, "%63s %d %d %d %d %d%n", name, &M, &N, &K, &glu, &res, &off) != 6) continue;
            const Shape s{name, M, N, K, glu, res, res, glu ? INA_ACT_SILU_C : INA_ACT_NONE_C};
            const size_t out_bytes = (size_t)M * (glu ? N / 2 : N) * (res ? 4 : 2);
            if ((size_t)M * K * 2 > A.bytes || (size_t)N * K * 2 > W.bytes || out_bytes > C0.bytes || (res && (size_t)M * N * 4 > R.bytes)) {
                printf("%-13s too large for the probe's buffers\n", name);
                continue;
            }
            printf("%-13s M %5d N %5d K %5d :", name, M, N, K);
            g_Wp = nullptr;
            if (g_shuffle && N % 16 == 0 && K % 32 == 0) {       // the fragment-ordered copy of this line's W for config 40
                if (g_shuffle(W.p, Wp.p, N, K, K, nullptr) != 0) { fprintf(stderr, "ina_gemm_preshuffle: %s\n", g_err()); return 3; }
                g_Wp = Wp.p;
            }
            const char* q = line + off;
            int cfg, gm, adv, first = 1;
            while (sscanf(q, " %d%n", &cfg, &adv) == 1) {
                q += adv;
                gm = 0;
                if (*q == ',') { ++q; sscanf(q, "%d%n", &gm, &adv); q += adv; }
                void* out = first ? C0.p : C1.p;
                ina_gemm_args a = make(s, A.p, W.p, out, R.p, cfg, gm);
                const double us = time_us(a, reps);
                unsigned long long d = 0;
                if (!first) {
                    HIP_OK(hipMemsetAsync(cnt.p, 0, 8, 0));
                    hipLaunchKernelGGL(count_diff, dim3(2048), dim3(256), 0, 0, (const uint32_t*)C0.p, (const uint32_t*)C1.p, out_bytes / 4,
                                       (unsigned long long*)cnt.p);
                    HIP_OK(hipMemcpy(&d, cnt.p, 8, hipMemcpyDeviceToHost));
                }
                printf("  cfg%d/g%d %7.1f us %6.1f TF%s", cfg, gm, us, tf(s, us), first ? "" : (d ? " DIFF" : " ="));
                if (d) printf("(%llu)", d);
                if (d && !res) {     // bf16 outputs: how far apart (kernels with a different K summation order differ in the last bits)
                    unsigned int mm[2] = {0, 0};
                    HIP_OK(hipMemsetAsync(cnt.p, 0, 8, 0));
                    hipLaunchKernelGGL(max_diff_bf16, dim3(2048), dim3(256), 0, 0, (const uint16_t*)C0.p, (const uint16_t*)C1.p, out_bytes / 2, (unsigned int*)cnt.p);
                    HIP_OK(hipMemcpy(mm, cnt.p, 8, hipMemcpyDeviceToHost));
                    float fd, fm;
                    memcpy(&fd, &mm[0], 4); memcpy(&fm, &mm[1], 4);
                    printf("[max|d| %.3g of max %.3g]", fd, fm);
                }
                first = 0;
            }
            printf("\n");
            fflush(stdout);
        }
        fclose(f);
        printf("# done\n");
        return 0;
    }

    // ---------------------------------------------------------------------------------------------- 1. isolated launches
    printf("# isolated launches: us, TF/s (algorithmic), bit-equal to the auto config (differing 32-bit words)\n");
    struct Var { int cfg, gm; };
    const Var vars[] = {{0, 0}, {18, 0}, {21, 0}, {14, 0}, {18, 4}, {18, 16}, {21, 4}};
    std::vector<Shape> shapes;
    for (int m : rows_llm) {
        shapes.push_back({"llm.qkv", m, QKV, H, 0, 0, 0, INA_ACT_NONE_C});
        shapes.push_back({"llm.o+res", m, H, H, 0, 1, 1, INA_ACT_NONE_C});
        shapes.push_back({"llm.gate|up", m, 2 * I, H, 1, 0, 0, INA_ACT_SILU_C});
        shapes.push_back({"llm.down+res", m, H, I, 0, 1, 1, INA_ACT_NONE_C});
    }
    for (int m : rows_vit) {
        shapes.push_back({"vit.qkv", m, 3 * VH, VH, 0, 0, 0, INA_ACT_NONE_C});
        shapes.push_back({"vit.proj+res", m, VH, VH, 0, 1, 1, INA_ACT_NONE_C});
        shapes.push_back({"vit.gate|up", m, 2 * VI, VH, 1, 0, 0, INA_ACT_SILU_C});
        shapes.push_back({"vit.down+res", m, VH, VI, 0, 1, 1, INA_ACT_NONE_C});
    }
    for (const Shape& s : shapes) {
        const size_t out_bytes = (size_t)s.M * (s.glu ? s.N / 2 : s.N) * (s.out_f32 ? 4 : 2);
        printf("%-13s M %5d N %5d K %5d :", s.name, s.M, s.N, s.K);
        for (const Var& v : vars) {
            void* out = (v.cfg == 0) ? C0.p : C1.p;
            ina_gemm_args a = make(s, A.p, W.p, out, R.p, v.cfg, v.gm);
            const double us = time_us(a, reps);
            unsigned long long d = 0;
            if (v.cfg != 0) {
                HIP_OK(hipMemsetAsync(cnt.p, 0, 8, 0));
                hipLaunchKernelGGL(count_diff, dim3(2048), dim3(256), 0, 0, (const uint32_t*)C0.p, (const uint32_t*)C1.p, out_bytes / 4,
                                   (unsigned long long*)cnt.p);
                HIP_OK(hipMemcpy(&d, cnt.p, 8, hipMemcpyDeviceToHost));
            }
            char tag[32];
            if (v.cfg == 0) snprintf(tag, sizeof tag, "auto");
            else if (v.gm) snprintf(tag, sizeof tag, "cfg%d/g%d", v.cfg, v.gm);
            else snprintf(tag, sizeof tag, "cfg%d", v.cfg);
            printf("  %s %7.1f us %6.1f TF%s", tag, us, tf(s, us), v.cfg == 0 ? "" : (d ? " DIFF" : " ="));
            if (d) printf("(%llu)", d);
        }
        printf("\n");
        fflush(stdout);
    }

    // ---------------------------------------------------------------------------------------------- 2. decoder-layer chains
    // buffers of one chain instance: x (bf16 [M,H]) -> qkv (bf16 [M,4608]) ; att = first H columns view is not contiguous, so the o
    // projection reads a separate bf16 [M,H] buffer; res f32 [M,H]; ff bf16 [M,I]
    printf("# decoder-layer GEMM chain x 28 layers (qkv, o + f32 residual, gate|up SwiGLU, down + f32 residual): ms per chain, TF/s\n");
    struct Chain {
        int M;
        Buf x, qkv, att, res, ff;
        void init(int m, int H, int I, int QKV) {
            M = m;
            x.alloc((size_t)m * H * 2); qkv.alloc((size_t)m * QKV * 2); att.alloc((size_t)m * H * 2); res.alloc((size_t)m * H * 4); ff.alloc((size_t)m * I * 2);
            fill(x, 11, 1.0f); fill(att, 12, 1.0f); fill(res, 13, 1.0f); fill(ff, 14, 1.0f);
        }
    };
    Buf Wq, Wo, Wg, Wd;
    Wq.alloc((size_t)QKV * H * 2); Wo.alloc((size_t)H * H * 2); Wg.alloc((size_t)2 * I * H * 2); Wd.alloc((size_t)H * I * 2);
    fill(Wq, 21, 0.03f); fill(Wo, 22, 0.03f); fill(Wg, 23, 0.03f); fill(Wd, 24, 0.01f);
    auto run_chain = [&](Chain& c, hipStream_t st, const int cfgs[4], int layers) {
        const Shape sq{"", c.M, QKV, H, 0, 0, 0, INA_ACT_NONE_C}, so{"", c.M, H, H, 0, 1, 1, INA_ACT_NONE_C};
        const Shape sg{"", c.M, 2 * I, H, 1, 0, 0, INA_ACT_SILU_C}, sd{"", c.M, H, I, 0, 1, 1, INA_ACT_NONE_C};
        for (int l = 0; l < layers; ++l) {
            launch(make(sq, c.x.p, Wq.p, c.qkv.p, nullptr, cfgs[0], 0), st);
            launch(make(so, c.att.p, Wo.p, c.res.p, c.res.p, cfgs[1], 0), st);
            launch(make(sg, c.x.p, Wg.p, c.ff.p, nullptr, cfgs[2], 0), st);
            launch(make(sd, c.ff.p, Wd.p, c.res.p, c.res.p, cfgs[3], 0), st);
        }
    };
    const double chain_flop = 2.0 * ((double)QKV * H + (double)H * H + 2.0 * I * H + (double)H * I) * 28;   // per row
    hipStream_t s1, s2;
    HIP_OK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    HIP_OK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    hipEvent_t e0, e1, ej;
    HIP_OK(hipEventCreate(&e0)); HIP_OK(hipEventCreate(&e1)); HIP_OK(hipEventCreate(&ej));
    struct Policy { const char* name; int cfgs[4]; };
    const Policy pol[] = {{"auto", {0, 0, 0, 0}}, {"all cfg18", {18, 18, 18, 18}}, {"all cfg21", {21, 21, 21, 21}},
                          {"qkv/o 21, gate|up/down 18", {21, 21, 18, 18}}, {"qkv/o 18, gate|up/down 21", {18, 18, 21, 21}}};
    const int joint_rows[] = {6440, 5520};
    const int split_rows[][2] = {{3680, 2760}, {2760, 2760}};
    for (int k = 0; k < 2; ++k) {
        Chain cj, ca, cb;
        cj.init(joint_rows[k], H, I, QKV); ca.init(split_rows[k][0], H, I, QKV); cb.init(split_rows[k][1], H, I, QKV);
        HIP_OK(hipDeviceSynchronize());
        for (const Policy& p : pol) {
            for (int mode = 0; mode < 3; ++mode) {      // 0 joint on one stream, 1 halves back to back on one stream, 2 halves on two streams
                double best = 1e30;
                for (int rep = 0; rep < (quick ? 2 : 3); ++rep) {
                    HIP_OK(hipDeviceSynchronize());
                    HIP_OK(hipEventRecord(e0, s1));
                    if (mode == 0) run_chain(cj, s1, p.cfgs, 28);
                    else if (mode == 1) { run_chain(ca, s1, p.cfgs, 28); run_chain(cb, s1, p.cfgs, 28); }
                    else {
                        HIP_OK(hipStreamWaitEvent(s2, e0, 0));
                        // interleave the issue order layer by layer, as the captured launch sequence replays both branches concurrently
                        for (int l = 0; l < 28; ++l) { run_chain(ca, s1, p.cfgs, 1); run_chain(cb, s2, p.cfgs, 1); }
                        HIP_OK(hipEventRecord(ej, s2));
                        HIP_OK(hipStreamWaitEvent(s1, ej, 0));
                    }
                    HIP_OK(hipEventRecord(e1, s1));
                    HIP_OK(hipEventSynchronize(e1));
                    float ms = 0;
                    HIP_OK(hipEventElapsedTime(&ms, e0, e1));
                    if (ms < best) best = ms;
                }
                const int rows = mode == 0 ? joint_rows[k] : split_rows[k][0] + split_rows[k][1];
                printf("rows %5d  %-28s %-34s %8.2f ms  %7.1f TF/s\n", rows, p.name,
                       mode == 0 ? "joint, one stream" : mode == 1 ? "halves back to back, one stream" : "halves on two streams", best,
                       chain_flop * rows / (best * 1e-3) * 1e-12);
                fflush(stdout);
            }
        }
        {   // per-GEMM time INSIDE the joint chain (auto tiles): an event pair around every launch of a second pass
            const int cf[4] = {0, 0, 0, 0};
            std::vector<hipEvent_t> ev(28 * 8);
            for (auto& e : ev) HIP_OK(hipEventCreate(&e));
            const Shape sh[4] = {{"qkv", cj.M, QKV, H, 0, 0, 0, INA_ACT_NONE_C}, {"o+res", cj.M, H, H, 0, 1, 1, INA_ACT_NONE_C},
                                 {"gate|up", cj.M, 2 * I, H, 1, 0, 0, INA_ACT_SILU_C}, {"down+res", cj.M, H, I, 0, 1, 1, INA_ACT_NONE_C}};
            run_chain(cj, s1, cf, 4);                    // warm: the sustained state
            for (int l = 0; l < 28; ++l) {
                const void* Ain[4] = {cj.x.p, cj.att.p, cj.x.p, cj.ff.p};
                const void* Win[4] = {Wq.p, Wo.p, Wg.p, Wd.p};
                void* Cout[4] = {cj.qkv.p, cj.res.p, cj.ff.p, cj.res.p};
                for (int g = 0; g < 4; ++g) {
                    HIP_OK(hipEventRecord(ev[(l * 4 + g) * 2], s1));
                    launch(make(sh[g], Ain[g], Win[g], Cout[g], cj.res.p, 0, 0), s1);
                    HIP_OK(hipEventRecord(ev[(l * 4 + g) * 2 + 1], s1));
                }
            }
            HIP_OK(hipStreamSynchronize(s1));
            for (int g = 0; g < 4; ++g) {
                double us = 0;
                for (int l = 0; l < 28; ++l) {
                    float ms = 0;
                    HIP_OK(hipEventElapsedTime(&ms, ev[(l * 4 + g) * 2], ev[(l * 4 + g) * 2 + 1]));
                    us += ms * 1e3;
                }
                us /= 28;
                printf("rows %5d  inside the chain (auto)  %-10s %8.1f us  %7.1f TF/s\n", cj.M, sh[g].name, us, tf(sh[g], us));
            }
            for (auto& e : ev) HIP_OK(hipEventDestroy(e));
            fflush(stdout);
        }
        for (Buf* b : {&cj.x, &cj.qkv, &cj.att, &cj.res, &cj.ff, &ca.x, &ca.qkv, &ca.att, &ca.res, &ca.ff, &cb.x, &cb.qkv, &cb.att, &cb.res, &cb.ff})
            HIP_OK(hipFree(b->p));
    }
    printf("# done\n");
    return 0;
}
