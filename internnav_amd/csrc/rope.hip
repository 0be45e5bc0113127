// Token-level kernels of the System-2 VLM (gfx950), all HBM-bound:
//   gather_rows   16-byte-vector row gather / scatter (embedding lookup, image-embed scatter, ViT window permutation)
//   rope          in-place rotary embedding from per-row cos/sin tables (vision 2-D rope and text m-RoPE share it)
//   mrope_table   cos/sin tables of the 3-D (t,h,w) multimodal rope from int32 position ids
//   argmax_rows   greedy token selection over the vocabulary
#include "common.h"
#include "kernels.h"

namespace {

__device__ __forceinline__ int map_row(const ina_rowmap& m, int r) {
    return m.seg_len > 0 ? (r / m.seg_len) * m.seg_stride + m.off + (r % m.seg_len) : r;
}

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void gather_kernel(GatherArgs p) {
    const int vec_per_row = p.row_bytes >> 4;
    const long total = (long)p.rows * vec_per_row;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int r = (int)(i / vec_per_row), v = (int)(i % vec_per_row);
        const long sr = p.src ? p.src[r] : r;
        const long dr = p.dst ? p.dst[r] : r;
        const u32x4 val = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(p.X) + sr * p.ldx_bytes + (long)v * 16);
        *reinterpret_cast<u32x4*>(reinterpret_cast<char*>(p.Y) + dr * p.ldy_bytes + (long)v * 16) = val;
    }
}

// one thread handles 8 consecutive columns j..j+7 of the first half AND their partners j + D/2 of HC consecutive heads of one row: the 32
// cos / sin values of its columns are loaded once and reused for those heads (round 5: one head per thread re-read the row's tables for every
// head - 4x the bytes of the activations themselves, all through the L2: 40 us per launch at 3680 rows where the data alone are 14 us).
// With KV set (decoder layer, X rows = q | k | v): query heads are rotated in place, key heads are rotated INTO cache row kv_dst[r],
// value heads (the v_heads heads behind the key heads) are copied there unrotated - rope + KV-cache append in one launch.
template <int HC>
__global__ __launch_bounds__(256) void rope_kernel(RopeArgs p) {
    const int half = p.D >> 1;
    const int groups = half >> 3;  // 8-wide groups per half
    const int hv = p.heads + (p.KV ? p.v_heads : 0);
    const int hchunks = (hv + HC - 1) / HC;
    const long total = (long)p.rows * hchunks * groups;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int g = (int)(i % groups);
        const int hc = (int)((i / groups) % hchunks);
        const int r = (int)(i / ((long)groups * hchunks));
        const int h0 = hc * HC;
        bf16* xr = reinterpret_cast<bf16*>(p.X) + (size_t)map_row(p.map, r) * p.ldx + p.col0 + g * 8;
        float c0[8], s0[8], c1[8], s1[8];
        if (h0 < p.heads) {
            const int tr = p.tab ? p.tab[r] : r;
            const float* c = p.cos + (size_t)tr * p.D + g * 8;
            const float* s = p.sin + (size_t)tr * p.D + g * 8;
#pragma unroll
            for (int j = 0; j < 8; ++j) { c0[j] = c[j]; s0[j] = s[j]; c1[j] = c[half + j]; s1[j] = s[half + j]; }
        }
#pragma unroll
        for (int k = 0; k < HC; ++k) {
            const int h = h0 + k;
            if (h >= hv) break;
            bf16* x = xr + h * p.D;
            bf16x8 lo = *reinterpret_cast<const bf16x8*>(x), hi = *reinterpret_cast<const bf16x8*>(x + half);
            bf16* out = x;
            if (p.KV && h >= p.kv_head0) out = reinterpret_cast<bf16*>(p.KV) + (size_t)p.kv_dst[r] * p.ldkv + (h - p.kv_head0) * p.D + g * 8;
            if (h < p.heads) {
                bf16x8 olo, ohi;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float a = (float)lo[j], b = (float)hi[j];
                    olo[j] = (bf16)ina_rope_lo(a, b, c0[j], s0[j]);        // x*cos + (-x2)*sin
                    ohi[j] = (bf16)ina_rope_hi(a, b, c1[j], s1[j]);        // x2*cos + x1*sin
                }
                lo = olo;
                hi = ohi;
            }
            *reinterpret_cast<bf16x8*>(out) = lo;
            *reinterpret_cast<bf16x8*>(out + half) = hi;
        }
    }
}

__global__ __launch_bounds__(256) void mrope_table_kernel(MropeTableArgs p) {
    const int half = p.D >> 1;
    const long total = (long)p.n * half;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int t = (int)(i / half), f = (int)(i % half);
        const float ang = (float)p.pos[(size_t)p.axis_of[f] * p.n + t] * p.inv_freq[f];
        const float c = cosf(ang), s = sinf(ang);
        const size_t o = (size_t)t * p.D + f;
        p.cos[o] = c; p.cos[o + half] = c;
        p.sin[o] = s; p.sin[o + half] = s;
    }
}

// 1024 threads per row, 16-byte loads: the row (152064 logits = 594 KiB) is a single latency-bound stream for one workgroup
__global__ __launch_bounds__(1024) void argmax_kernel(ArgmaxArgs p) {
    __shared__ float bv[16];
    __shared__ int bi[16];
    const int r = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* x = p.X + (size_t)r * p.ldx;
    float best = -INFINITY;
    int idx = 0x7fffffff;
    const int n4 = ((reinterpret_cast<uintptr_t>(x) & 15) == 0) ? (p.n >> 2) : 0;
    for (int j = threadIdx.x; j < n4; j += 1024) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(x + 4 * j);
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (v[q] > best) { best = v[q]; idx = 4 * j + q; }   // ascending index per thread: the first maximum is kept
    }
    for (int j = n4 * 4 + threadIdx.x; j < p.n; j += 1024) {
        const float v = x[j];
        if (v > best || (v == best && j < idx)) { best = v; idx = j; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o);
        const int oi = __shfl_xor(idx, o);
        if (ov > best || (ov == best && oi < idx)) { best = ov; idx = oi; }
    }
    if (lane == 0) { bv[wave] = best; bi[wave] = idx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 16; ++w)
            if (bv[w] > best || (bv[w] == best && bi[w] < idx)) { best = bv[w]; idx = bi[w]; }
        p.out[r] = idx;
    }
}

}  // namespace

int ina_launch_gather(const GatherArgs& p, hipStream_t stream) {
    INA_REQUIRE(p.rows > 0 && p.row_bytes > 0 && p.row_bytes % 16 == 0, "gather_rows: rows=%d row_bytes=%d (multiple of 16 required)", p.rows, p.row_bytes);
    INA_REQUIRE(p.X && p.Y && p.ldx_bytes % 16 == 0 && p.ldy_bytes % 16 == 0, "gather_rows: X/Y and 16-byte aligned row strides required");
    InaProfScope prof(INA_PROF_ELEMENTWISE, 0.0, 2.0 * p.rows * p.row_bytes, stream);
    const long total = (long)p.rows * (p.row_bytes / 16);
    const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(gather_kernel, dim3(blocks), dim3(256), 0, stream, p);
    INA_HIP_CHECK(hipGetLastError());
    return 0;
}

int ina_launch_rope(const RopeArgs& p, hipStream_t stream) {
    INA_REQUIRE(p.rows > 0 && p.heads > 0 && p.D % 16 == 0 && p.ldx % 8 == 0 && p.col0 % 8 == 0, "rope: bad shape rows=%d heads=%d D=%d ldx=%d col0=%d",
                p.rows, p.heads, p.D, p.ldx, p.col0);
    INA_REQUIRE(p.X && p.cos && p.sin, "rope: X, cos, sin required");
    INA_REQUIRE(!p.KV || (p.kv_dst && p.kv_head0 >= 0 && p.kv_head0 <= p.heads && p.v_heads >= 0 && p.ldkv % 8 == 0), "rope: bad KV-append arguments");
    const int hv = p.heads + (p.KV ? p.v_heads : 0);
    InaProfScope prof(INA_PROF_ELEMENTWISE, 6.0 * p.rows * p.heads * p.D, 4.0 * p.rows * hv * p.D, stream);
    // heads per thread: 4 where that still leaves a few hundred thousand threads (the prefill / vision launches), 1 for the small decode launches
    const bool chunked = (long)p.rows * hv * (p.D / 16) >= (1L << 19);
    const int hc = chunked ? 4 : 1;
    const long total = (long)p.rows * ((hv + hc - 1) / hc) * (p.D / 16);
    const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    if (chunked) hipLaunchKernelGGL(rope_kernel<4>, dim3(blocks), dim3(256), 0, stream, p);
    else hipLaunchKernelGGL(rope_kernel<1>, dim3(blocks), dim3(256), 0, stream, p);
    INA_HIP_CHECK(hipGetLastError());
    return 0;
}

int ina_launch_mrope_table(const MropeTableArgs& p, hipStream_t stream) {
    INA_REQUIRE(p.n > 0 && p.D > 0 && p.D % 2 == 0 && p.pos && p.inv_freq && p.axis_of && p.cos && p.sin, "mrope_table: bad arguments n=%d D=%d", p.n, p.D);
    InaProfScope prof(INA_PROF_ELEMENTWISE, 0.0, 8.0 * p.n * p.D, stream);
    const long total = (long)p.n * (p.D / 2);
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(mrope_table_kernel, dim3(blocks), dim3(256), 0, stream, p);
    INA_HIP_CHECK(hipGetLastError());
    return 0;
}

int ina_launch_argmax(const ArgmaxArgs& p, hipStream_t stream) {
    INA_REQUIRE(p.rows > 0 && p.n > 0 && p.X && p.out, "argmax_rows: bad arguments rows=%d n=%d", p.rows, p.n);
    InaProfScope prof(INA_PROF_ELEMENTWISE, 0.0, 4.0 * p.rows * p.n, stream);
    hipLaunchKernelGGL(argmax_kernel, dim3(p.rows), dim3(1024), 0, stream, p);
    INA_HIP_CHECK(hipGetLastError());
    return 0;
}
