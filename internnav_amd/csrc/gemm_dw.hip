// Weight gradient of nn.Linear straight from the row-major operands (gfx950):
//     dW[n, k] += sum_r DY[r, n] * X[r, k]          db[n] += sum_r DY[r, n]   (optional)
// DY [rows, N] (f32 or bf16) and X [rows, K] (bf16) are what the backward tape holds; the reduction runs over their SLOW dimension, so the tiled NT
// GEMM needed K-contiguous copies of both - two transpose launches, the GEMM, and two column-sum launches for the bias: five launches per trained
// layer in a ~3 000-launch backward chain that is bound by launch latency, not by work (rows = 24 ... 768 in the SFT step). This kernel stages
// 32-row slabs of both operands in LDS TRANSPOSED (16-byte global loads along n / k, 2-byte LDS stores, 16-byte fragment reads along r), runs
// MFMA 16x16x32 over them and adds the 64 x 64 tile into the fp32 gradient; workgroups of the first k tile also sum their DY slab columns (from the
// UNROUNDED values) for the bias gradient. One launch, no scratch, deterministic (r ascending inside a tile, one workgroup per output element).
// Long reductions over few tiles (the memory-encoder layers: 12 288 rows into a 384 x 384 gradient = 36 tiles of 192 slabs) are cut into `splits`
// row ranges (grid.z): every range writes its fp32 tile (and bias sums) into the caller's partial buffer and a second launch adds the ranges in
// ascending order into dW / db - two launches instead of five, 8x the workgroups, still one summation order.
// Numerics: DY rounded to bf16 for the MFMAs (as the transposed copy was), fp32 accumulation over r in 32-wide steps.
#include "common.h"
#include "kernels.h"

namespace {

constexpr int DW_T = 64, DW_R = 32, DW_PITCH = 36;   // tile edge, rows per slab, LDS row pitch in bf16 (72 B = 18 dwords: the eight 8-row groups of a
                                                     // write instruction start 16 banks apart, 8-byte accesses are 2-way = the minimum for 128 dwords)
constexpr int DW_WAVE_LDS = 2 * DW_T * DW_PITCH;     // bf16 elements of one wave's operand images (n-major and k-major)

// The four waves of a workgroup work on the SAME 64 x 64 tile over DIFFERENT slabs (wave w: slabs w, w + 4, ...): no workgroup barrier in the
// main loop, four slabs' loads in flight per CU - the launch is latency-bound (36-144 workgroups), so the waves' memory round trips have to overlap.
// A lane loads FOUR CONSECUTIVE rows 4 (l / 8) + h x 8 columns of each operand per slab (two slabs ahead in registers), so the transposed LDS image is
// written 8 bytes at a time ([column][4 rows]) instead of 2 - the 2-byte form (32 conflicted writes per operand) was what bound the first versions. The waves' tiles are added at the
// end in a fixed order ((w0 + w2) + (w1 + w3)) through LDS.
template <bool DY32>
__global__ __launch_bounds__(256) void gemm_dw_kernel(ina_gemm_dw_args p) {
    __shared__ __attribute__((aligned(16))) bf16 lds[4 * DW_WAVE_LDS];                  // 36 KiB; reused as 32 KiB of fp32 for the final adds
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    bf16* As = lds + wave * DW_WAVE_LDS;                                                // [n][r]
    bf16* Bs = As + DW_T * DW_PITCH;                                                    // [k][r]
    const int n0 = blockIdx.x * DW_T, k0 = blockIdx.y * DW_T;
    const int rq = lane >> 3, c8 = (lane & 7) * 8;
    const bool n_ok = n0 + c8 < p.N, k_ok = k0 + c8 < p.K;                             // (N, K multiples of 8: a run of 8 is inside or outside)
    const bool bias = p.db != nullptr && blockIdx.y == 0;
    const int fi = lane & 15, fg = lane >> 4;

    f32x4 acc[4][4];                                                                    // [n sub-tile][k sub-tile]: lane holds n = fi, k = 4 fg .. 4 fg + 3
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    float bs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    const int total = (p.rows + DW_R - 1) / DW_R;
    const int per = (total + (int)gridDim.z - 1) / (int)gridDim.z;                      // slabs per row range (splits > 1: grid.z ranges, partial outputs)
    const int s0 = blockIdx.z * per;
    const int nslab = max(0, min(total - s0, per));                                     // slabs of this workgroup; wave w takes w, w + 4, ...

    float a[2][4][8];
    bf16x8 b[2][4];
    auto load = [&](int set, int slab) {
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            const int r = (s0 + slab) * DW_R + 4 * rq + h;
            const bool ok = slab < nslab && r < p.rows;
#pragma unroll
            for (int j = 0; j < 8; ++j) a[set][h][j] = 0.f;
            b[set][h] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
            if (ok && n_ok) {
                if constexpr (DY32) {
                    const float* q = reinterpret_cast<const float*>(p.DY) + (size_t)r * p.lddy + n0 + c8;
                    const f32x4 u = *reinterpret_cast<const f32x4*>(q), v = *reinterpret_cast<const f32x4*>(q + 4);
#pragma unroll
                    for (int j = 0; j < 4; ++j) { a[set][h][j] = u[j]; a[set][h][4 + j] = v[j]; }
                } else {
                    const bf16x8 u = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const bf16*>(p.DY) + (size_t)r * p.lddy + n0 + c8);
#pragma unroll
                    for (int j = 0; j < 8; ++j) a[set][h][j] = (float)u[j];
                }
            }
            if (ok && k_ok) b[set][h] = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const bf16*>(p.X) + (size_t)r * p.ldx + k0 + c8);
        }
    };
    auto step = [&](int set) {      // one slab: registers -> this wave's LDS images (transposed), 16 MFMAs; the bias sums take the unrounded values
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const bf16x4 va = {(bf16)a[set][0][j], (bf16)a[set][1][j], (bf16)a[set][2][j], (bf16)a[set][3][j]};
            const bf16x4 vb = {b[set][0][j], b[set][1][j], b[set][2][j], b[set][3][j]};
            *reinterpret_cast<bf16x4*>(As + (c8 + j) * DW_PITCH + 4 * rq) = va;
            *reinterpret_cast<bf16x4*>(Bs + (c8 + j) * DW_PITCH + 4 * rq) = vb;
        }
        if (bias) {
#pragma unroll
            for (int j = 0; j < 8; ++j) bs[j] += (a[set][0][j] + a[set][1][j]) + (a[set][2][j] + a[set][3][j]);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                              // the wave's own writes have landed (no other wave touches this image)
        bf16x8 fa[4], fb[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {                                                   // (8-byte halves: the 72-byte pitch is not 16-byte aligned)
            const bf16x4 a0 = *reinterpret_cast<const bf16x4*>(As + (i * 16 + fi) * DW_PITCH + fg * 8);
            const bf16x4 a1 = *reinterpret_cast<const bf16x4*>(As + (i * 16 + fi) * DW_PITCH + fg * 8 + 4);
            const bf16x4 b0 = *reinterpret_cast<const bf16x4*>(Bs + (i * 16 + fi) * DW_PITCH + fg * 8);
            const bf16x4 b1 = *reinterpret_cast<const bf16x4*>(Bs + (i * 16 + fi) * DW_PITCH + fg * 8 + 4);
            fa[i] = bf16x8{a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
            fb[i] = bf16x8{b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                              // fragments are in registers before the image is overwritten
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
    };

    // two register sets: slab t is multiplied while slabs t + 4 and t + 8 (this wave's next two) are in flight
    load(0, wave);
    load(1, wave + 4);
    for (int t = wave; t < nslab; t += 8) {
        step(0);
        load(0, t + 8);
        if (t + 4 < nslab) {
            step(1);
            load(1, t + 12);
        }
    }

    // ---- add the four waves' tiles: w2, w3 -> LDS; w0 += w2, w1 += w3; w1 -> LDS; w0 += w1 (fixed order)
    __syncthreads();                                                                    // every wave is done with its operand images
    float* red = reinterpret_cast<float*>(lds);                                         // [2][16 sub-tiles][64 lanes][4] f32 = 32 KiB
    if (wave >= 2) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x4*>(red + (((wave - 2) * 16 + i * 4 + j) * 64 + lane) * 4) = acc[i][j];
    }
    __syncthreads();
    if (wave < 2) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] += *reinterpret_cast<const f32x4*>(red + ((wave * 16 + i * 4 + j) * 64 + lane) * 4);
    }
    __syncthreads();
    if (wave == 1) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x4*>(red + ((i * 4 + j) * 64 + lane) * 4) = acc[i][j];
    }
    __syncthreads();
    if (wave == 0) {
        // D[4 fg + t][fi] of mfma(first = X^T fragment, second = DY^T fragment): k = 4 fg + t, n = fi
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int n = n0 + i * 16 + fi;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k = k0 + j * 16 + fg * 4;
                const f32x4 v = acc[i][j] + *reinterpret_cast<const f32x4*>(red + ((i * 4 + j) * 64 + lane) * 4);
                if (n < p.N && k < p.K) {                                               // (K multiple of 4: a run of 4 is inside or outside)
                    if (gridDim.z > 1) {
                        *reinterpret_cast<f32x4*>(p.partial + ((size_t)blockIdx.z * p.N + n) * p.K + k) = v;
                    } else {
                        f32x4* o = reinterpret_cast<f32x4*>(p.dW + (size_t)n * p.ldw + k);
                        *o = *o + v;
                    }
                }
            }
        }
    }
    if (bias) {
        __syncthreads();                                                                // wave 0 has read the tile scratch
        float* rb = reinterpret_cast<float*>(lds);                                      // [4 waves x 8 row lanes][64 n]
#pragma unroll
        for (int j = 0; j < 8; ++j) rb[(wave * 8 + rq) * DW_T + c8 + j] = bs[j];
        __syncthreads();
        if (tid < DW_T && n0 + tid < p.N) {
            float s = 0.f;
#pragma unroll
            for (int r = 0; r < 32; ++r) s += rb[r * DW_T + tid];
            if (gridDim.z > 1) p.partial[(size_t)gridDim.z * p.N * p.K + (size_t)blockIdx.z * p.N + n0 + tid] = s;
            else p.db[n0 + tid] += s;
        }
    }
}

// dW[n, k] += sum_z partial[z][n][k] (z ascending), db[n] += sum_z partial_b[z][n]: 4 columns per thread
__global__ __launch_bounds__(256) void gemm_dw_reduce_kernel(ina_gemm_dw_args p) {
    const size_t quads = (size_t)p.N * (p.K / 4);
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < quads) {
        const int n = (int)(i / (p.K / 4)), k = (int)(i % (p.K / 4)) * 4;
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        for (int z = 0; z < p.splits; ++z) s += *reinterpret_cast<const f32x4*>(p.partial + ((size_t)z * p.N + n) * p.K + k);
        f32x4* o = reinterpret_cast<f32x4*>(p.dW + (size_t)n * p.ldw + k);
        *o = *o + s;
    } else if (p.db && i - quads < (size_t)p.N) {
        const int n = (int)(i - quads);
        float s = 0.f;
        for (int z = 0; z < p.splits; ++z) s += p.partial[(size_t)p.splits * p.N * p.K + (size_t)z * p.N + n];
        p.db[n] += s;
    }
}

}  // namespace

int ina_launch_gemm_dw(const ina_gemm_dw_args& p, hipStream_t stream) {
    INA_REQUIRE(p.rows > 0 && p.N > 0 && p.K > 0 && p.DY && p.X && p.dW, "gemm_dw: empty problem or null tensor");
    INA_REQUIRE(p.dy_dt == INA_DT_F32 || p.dy_dt == INA_DT_BF16, "gemm_dw: DY must be f32 or bf16");
    const size_t des = p.dy_dt == INA_DT_F32 ? 4 : 2;
    INA_REQUIRE(p.N % 8 == 0 && p.K % 8 == 0 && p.lddy % 8 == 0 && p.ldx % 8 == 0 && p.ldw % 4 == 0 && ((uintptr_t)p.DY % (8 * des)) == 0 &&
                    ((uintptr_t)p.X % 16) == 0 && ((uintptr_t)p.dW % 16) == 0,
                "gemm_dw: N, K, lddy, ldx must be multiples of 8 (ldw of 4) and the tensors 16-byte aligned (N=%d K=%d lddy=%d ldx=%d ldw=%d)", p.N, p.K,
                p.lddy, p.ldx, p.ldw);
    ina_prof_set_sub(43);
    InaProfScope prof(INA_PROF_GEMM, 2.0 * p.rows * p.N * p.K, (double)p.rows * (des * p.N + 2.0 * p.K) + 8.0 * p.N * p.K, stream);
    const int splits = p.splits > 1 ? p.splits : 1;
    INA_REQUIRE(splits <= 64 && (splits == 1 || (p.partial && ((uintptr_t)p.partial % 16) == 0 && p.partial_elems >= (int64_t)splits * p.N * (p.K + 1))),
                "gemm_dw: %d row ranges need a 16-byte aligned partial buffer of splits * N * (K + 1) = %lld floats (got %lld)", splits,
                (long long)splits * p.N * (p.K + 1), (long long)p.partial_elems);
    const dim3 grid((p.N + DW_T - 1) / DW_T, (p.K + DW_T - 1) / DW_T, splits);
    ina_gemm_dw_args q = p;
    q.splits = splits;
    if (p.dy_dt == INA_DT_F32) hipLaunchKernelGGL(gemm_dw_kernel<true>, grid, dim3(256), 0, stream, q);
    else hipLaunchKernelGGL(gemm_dw_kernel<false>, grid, dim3(256), 0, stream, q);
    if (splits > 1) {
        const size_t work = (size_t)p.N * (p.K / 4) + (p.db ? (size_t)p.N : 0);
        hipLaunchKernelGGL(gemm_dw_reduce_kernel, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, stream, q);
    }
    INA_HIP_CHECK(hipGetLastError());
    return 0;
}
