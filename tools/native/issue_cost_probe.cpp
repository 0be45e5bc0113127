// What does an operand fetch cost the wave that issues it, between MFMAs?  (DESIGN 8.0 / review item "pre-shuffled weights")
//
// One wave per SIMD (256-thread workgroups, one per CU: the geometry of gemm_w4.hip), a loop of "stages" of 64 v_mfma_f32_32x32x16_bf16
// (= the 2048 MFMA cycles a 256 x 256 x 64 stage costs a SIMD) with the memory instructions of a staging scheme threaded evenly between
// them as asm statements:
//     NDMA  x  global_load_lds_dwordx4   (1 KiB LDS-DMA piece: global -> LDS, scalar base + lane offset, M0 = LDS address)
//     NVEC  x  global_load_dwordx4       (1 KiB per wave: global -> VGPR; what a pre-shuffled B operand would use)
//     NDS   x  ds_read_b128              (fragment reads out of LDS)
// The cfg-39 tile issues per wave and stage 16 pieces + 32 reads; a tile that takes pre-shuffled B fragments straight from global memory
// issues 8 pieces (A only) + 16 vector loads (its own 128 columns x 64 k = 16 KiB: the two waves that share a column half cannot share
// registers) + 16 reads. Loads stream through an L2-resident window (operands of a GEMM tile are L2 / MALL hits); nothing waits for data
// except one counted s_waitcnt per stage that keeps ONE stage of requests in flight - the number printed is the issue-side cost.
// Build: tools/native/build.sh; run: tools/native/issue_cost_probe
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <utility>

#define HIP_OK(x)                                                                                      \
    do {                                                                                               \
        hipError_t e_ = (x);                                                                           \
        if (e_ != hipSuccess) {                                                                        \
            fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_));          \
            exit(2);                                                                                   \
        }                                                                                              \
    } while (0)

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int N>
using IC = std::integral_constant<int, N>;
template <class F, int... I>
__device__ __forceinline__ void for_seq_impl(F&& f, std::integer_sequence<int, I...>) {
    (f(IC<I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void for_seq(F&& f) {
    for_seq_impl(f, std::make_integer_sequence<int, N>{});
}

constexpr int NMFMA = 64;

// memory instruction m of a stage -> (class, index inside its class): the three classes are interleaved by largest deficit, so each is spread
// over the whole stage
struct Slot { int cls, idx; };
constexpr Slot slot_of(int ndma, int nvec, int nds, int m) {
    const int n[3] = {ndma, nvec, nds};
    int done[3] = {0, 0, 0};
    const int total = ndma + nvec + nds;
    Slot r{0, 0};
    for (int k = 0; k <= m; ++k) {
        int best = -1;
        long best_def = -(1L << 60);
        for (int c = 0; c < 3; ++c) {
            if (done[c] >= n[c]) continue;
            const long def = (long)n[c] * (k + 1) - (long)done[c] * total;   // target share minus issued, scaled by total
            if (def > best_def) { best_def = def; best = c; }
        }
        r = Slot{best, done[best]};
        ++done[best];
    }
    return r;
}


// the memory instructions that sit behind MFMA number I of a stage
template <int NDMA, int NVEC, int NDS, int I, int M>
__device__ __forceinline__ void mem_op(u32x4* ring, u32x4* frag, const char* src, unsigned goff, unsigned lds_wave, unsigned lds_lane) {
    constexpr int NMEM = NDMA + NVEC + NDS;
    if constexpr ((M + 1) * NMFMA / (NMEM + 1) == I) {
        constexpr Slot sl = slot_of(NDMA, NVEC, NDS, M);
        if constexpr (sl.cls == 0) {
            const unsigned voff = goff + sl.idx * 1024u;
            const unsigned la = lds_wave + sl.idx * 1024u;
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(la), "v"(voff), "s"(src) : "memory", "m0");
        } else if constexpr (sl.cls == 1) {
            const unsigned voff = goff + 8192u + (sl.idx & 7) * 1024u + (sl.idx >> 3) * 65536u;
            asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(ring[sl.idx]) : "v"(voff), "s"(src) : "memory");
        } else {
            const unsigned la = (lds_lane + sl.idx * 1024u) & 65535u;
            asm volatile("ds_read_b128 %0, %1" : "=v"(frag[sl.idx & 3]) : "v"(la) : "memory");
        }
    }
}
template <int NDMA, int NVEC, int NDS, int I, int... M>
__device__ __forceinline__ void mem_ops(std::integer_sequence<int, M...>, u32x4* ring, u32x4* frag, const char* src, unsigned goff, unsigned lds_wave, unsigned lds_lane) {
    (mem_op<NDMA, NVEC, NDS, I, M>(ring, frag, src, goff, lds_wave, lds_lane), ...);
}

template <int NDMA, int NVEC, int NDS, int OCC = 1>
__global__ __launch_bounds__(256, OCC) void probe_kernel(const char* __restrict__ src, unsigned window_mask, int iters, float* __restrict__ out, long long* __restrict__ cyc) {
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr int NACC = OCC == 1 ? 8 : 4;   // (two workgroups per CU: 256 registers per wave)
    f32x16 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    bf16x8 a, b;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        a[i] = (__bf16)(0.001f * (lane + i));
        b[i] = (__bf16)(0.002f * (lane - i));
    }
    u32x4 ring[NVEC > 0 ? NVEC : 1];
    u32x4 frag[NDS > 0 ? 4 : 1];
    // every workgroup walks its own 64 KiB-per-stage stream through the window
    unsigned pos = (blockIdx.x * 2654435761u) & window_mask & ~0xffffu;
    const unsigned lane_off = wave * 16384u + lane * 16u;
    const unsigned lds_wave = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)lds + wave * 16384u;   // LDS byte address of this wave's 16 KiB region
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        constexpr int NMEM = NDMA + NVEC + NDS;
        // memory instruction m goes behind MFMA number (m + 1) * NMFMA / (NMEM + 1): an even spread, the interleave of the hand-written streams
        for_seq<NMFMA>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            acc[i & (NACC - 1)] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i & (NACC - 1)], 0, 0, 0);
            mem_ops<NDMA, NVEC, NDS, i>(std::make_integer_sequence<int, NMEM>{}, ring, frag, src, pos + lane_off, lds_wave, wave * 16384u + lane * 16u);
        });
        // one stage of requests stays in flight (counted wait, as the tiles do); LDS reads are retired
        if constexpr (NDMA + NVEC > 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA + NVEC) : "memory");
        if constexpr (NDS > 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        pos = (pos + 65536u) & window_mask;
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    const long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][lane & 15];
    if constexpr (NVEC > 0) {
#pragma unroll
        for (int i = 0; i < NVEC; ++i) s += (float)(ring[i][0] & 1u);
    }
    if constexpr (NDS > 0) s += (float)(frag[0][0] & 1u) + (float)(frag[3][1] & 1u);
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int NDMA, int NVEC, int NDS, int OCC = 1>
static void run(const char* label, const char* src, unsigned window_mask, float* out, long long* cyc, int ncu_in) {
    auto kern = probe_kernel<NDMA, NVEC, NDS, OCC>;
    const int ncu = ncu_in * OCC;
    const int LDSB = OCC == 1 ? 96 * 1024 : 64 * 1024;
    HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDSB));
    const int iters = 2000;
    hipEvent_t e0, e1;
    HIP_OK(hipEventCreate(&e0));
    HIP_OK(hipEventCreate(&e1));
    hipLaunchKernelGGL(kern, dim3(ncu), dim3(256), LDSB, 0, src, window_mask, 200, out, cyc);
    HIP_OK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(kern, dim3(ncu), dim3(256), LDSB, 0, src, window_mask, iters, out, cyc);
    HIP_OK(hipEventRecord(e1, 0));
    HIP_OK(hipEventSynchronize(e1));
    float ms = 0;
    HIP_OK(hipEventElapsedTime(&ms, e0, e1));
    const double us_stage = ms * 1e3 / iters;
    // MFMA cycles of a stage at the clock the run sustained are unknown from the host: report time and the ratio to the bare-MFMA stage
    static double bare_[3] = {0, 0, 0};
    double& bare = bare_[OCC];
    if (NDMA + NVEC + NDS == 0) bare = us_stage;
    const double flops = 64.0 * 32768.0 * 4 * ncu;   // (OCC workgroups per CU share its SIMDs: per-workgroup stage time doubles at the same rate)   // per stage, all SIMDs
    printf("%-58s %7.3f us / stage   x%5.3f of bare MFMAs   %7.1f TF/s   +%6.0f MFMA-cycles-equivalent\n", label, us_stage, bare > 0 ? us_stage / bare : 1.0,
           flops / us_stage * 1e-6, bare > 0 ? (us_stage / bare - 1.0) * 2048.0 : 0.0);
    fflush(stdout);
}

int main() {
    hipDeviceProp_t prop;
    HIP_OK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount;
    const unsigned window = 16u << 20;   // 16 MiB: L2 / MALL resident
    char* src;
    HIP_OK(hipMalloc(&src, (size_t)window + (1u << 20)));
    HIP_OK(hipMemset(src, 1, (size_t)window + (1u << 20)));
    float* out;
    HIP_OK(hipMalloc(&out, (size_t)ncu * 2 * 256 * 4));
    long long* cyc;
    HIP_OK(hipMalloc(&cyc, (size_t)ncu * 2 * 8));
    printf("# %s, %d CUs; stage = 64 x v_mfma_f32_32x32x16_bf16 per wave, one wave per SIMD, memory instructions spread evenly between them\n", prop.name, ncu);
    const unsigned mask = window - 1;
    run<0, 0, 0>("bare MFMAs", src, mask, out, cyc, ncu);
    run<0, 0, 32>("32 ds_read_b128", src, mask, out, cyc, ncu);
    run<0, 0, 16>("16 ds_read_b128", src, mask, out, cyc, ncu);
    run<16, 0, 0>("16 LDS-DMA pieces", src, mask, out, cyc, ncu);
    run<8, 0, 0>(" 8 LDS-DMA pieces", src, mask, out, cyc, ncu);
    run<0, 16, 0>("16 global_load_dwordx4 -> VGPR", src, mask, out, cyc, ncu);
    run<0, 24, 0>("24 global_load_dwordx4 -> VGPR", src, mask, out, cyc, ncu);
    run<16, 0, 32>("cfg 39 mix: 16 pieces + 32 reads", src, mask, out, cyc, ncu);
    run<8, 16, 16>("pre-shuffled B: 8 pieces + 16 vector loads + 16 reads", src, mask, out, cyc, ncu);
    run<8, 8, 16>("(if the column pair could share B: 8 + 8 + 16)", src, mask, out, cyc, ncu);
    run<0, 24, 0>("(no LDS at all: 24 vector loads)", src, mask, out, cyc, ncu);
    // the row-chain kernel's stage mix (dit_rowchain.hip: 32 rows per wave, W streamed through an LDS ring): per 64 MFMAs a wave issues 16 LDS-DMA
    // pieces and 64 fragment reads, two workgroups per CU; against W fragments fetched straight from global memory (64 vector loads, no LDS)
    printf("# two workgroups per CU (two waves per SIMD), the row-chain geometry:\n");
    run<0, 0, 0, 2>("bare MFMAs, 2 waves / SIMD", src, mask, out, cyc, ncu);
    run<0, 0, 32, 2>("32 fragment reads", src, mask, out, cyc, ncu);
    run<0, 0, 48, 2>("48 fragment reads (the row chain issues 64 + 16 LDS-DMA pieces per 64 MFMAs)", src, mask, out, cyc, ncu);
    run<0, 32, 0, 2>("W fragments from global: 32 vector loads (half of the 64 the kernel would need; register budget of the probe)", src, mask, out, cyc, ncu);
    run<0, 16, 0, 2>("16 vector loads", src, mask, out, cyc, ncu);
    return 0;
}
