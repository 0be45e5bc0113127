// Shared device/host helpers for the gfx950 (CDNA4) kernels of internnav_amd.
// Wave = 64 lanes everywhere in this tree; MFMA tiles are 16x16x32 bf16 -> f32.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16;
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define INA_WAVE 64

// activation codes shared by the GEMM epilogue and the C-ABI (include/internnav_amd.h)
enum : int {
    INA_ACT_NONE = 0,
    INA_ACT_GELU_ERF = 1,   // nn.GELU() default (DINOv2 Mlp, NavDP decoder 'gelu')
    INA_ACT_GELU_TANH = 2,  // nn.GELU(approximate="tanh") (cond_projector, caption_projection)
    INA_ACT_RELU = 3,       // nn.TransformerDecoderLayer default FFN, vlm_embed_mlp
    INA_ACT_SILU = 4,       // timestep embedder, adaLN SiLU
    INA_ACT_MISH = 5,       // diffusion-policy ConditionalUnet1D (cond_encoder, Conv1dBlock)
    INA_ACT_TANH = 6,       // NextDiT gates (training path: ina_ew)
};

enum : int { INA_DT_BF16 = 0, INA_DT_F32 = 1 };

__device__ __forceinline__ float ina_gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float ina_gelu_tanh(float x) {
    const float k0 = 0.7978845608028654f, k1 = 0.044715f;
    float u = k0 * (x + k1 * x * x * x);
    return 0.5f * x * (1.0f + tanhf(u));
}
__device__ __forceinline__ float ina_silu(float x) { return x / (1.0f + __expf(-x)); }

__device__ __forceinline__ float ina_act(float v, int act) {
    switch (act) {
        case INA_ACT_GELU_ERF: return ina_gelu_erf(v);
        case INA_ACT_GELU_TANH: return ina_gelu_tanh(v);
        case INA_ACT_RELU: return v > 0.f ? v : 0.f;
        case INA_ACT_SILU: return ina_silu(v);
        case INA_ACT_MISH: return v * tanhf(v > 20.f ? v : log1pf(__expf(v)));
        case INA_ACT_TANH: return tanhf(v);
        default: return v;
    }
}

// counter-based dropout mask (training kernels): murmur3 finaliser over (seed, 64-bit element index); keep iff hash >= p * 2^32
__device__ __forceinline__ uint32_t ina_fmix32(uint32_t h) {
    h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
    return h;
}
__device__ __forceinline__ uint32_t ina_hash(uint32_t seed, uint64_t idx) {
    // the seed goes through its own finaliser round before it meets the index: with a plain seed ^ idx two seeds give the same field
    // under an index permutation (hash(s1, i) == hash(s2, i ^ s1 ^ s2)), i.e. masks of different sites / steps would not be independent
    const uint32_t h = ina_fmix32(ina_fmix32(seed) + 0x9E3779B9u * (uint32_t)idx);
    return ina_fmix32(h + 0x9E3779B9u * (uint32_t)(idx >> 32) + 0x7F4A7C15u);
}

// rotary embedding of a (first half, second half) pair, x * cos + rotate_half(x) * sin: ONE definition (explicit fma) for rope.hip's kernel and
// the decode attention kernel's prologue (attention.hip), so the fused and the unfused path round identically
__device__ __forceinline__ float ina_rope_lo(float lo, float hi, float c, float s) { return fmaf(lo, c, -(hi * s)); }
__device__ __forceinline__ float ina_rope_hi(float lo, float hi, float c, float s) { return fmaf(hi, c, lo * s); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

// XCD-aware remap of a 1-D workgroup id: MI355X dispatches block b to XCD b % 8, each XCD has a
// private 4 MiB L2. Give every XCD one contiguous chunk of the logical tile order so neighbouring
// tiles (which share operand panels) hit the same L2. Bijective for any nwg (guide §5, T1).
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int NX = 8;
    if (nwg < NX * 2) return bid;
    int xcd = bid % NX, idx = bid / NX;
    int q = nwg / NX, r = nwg % NX;
    int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

#define INA_HIP_CHECK(expr)                                                          \
    do {                                                                             \
        hipError_t _e = (expr);                                                      \
        if (_e != hipSuccess) {                                                      \
            ina_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
            return -1;                                                               \
        }                                                                            \
    } while (0)

#define INA_REQUIRE(cond, ...)                 \
    do {                                       \
        if (!(cond)) {                         \
            ina_set_error(__VA_ARGS__);        \
            return -2;                         \
        }                                      \
    } while (0)

void ina_set_error(const char* fmt, ...);

// Library-owned scratch (split-K partials, flash-decoding partials) lives in numbered workspace slots. Launches use the slot that
// is current when they are issued (ina_set_workspace_slot, default 0); the pointer is baked into a captured graph, so graphs that
// may be replayed CONCURRENTLY on different streams must be captured under different slots. Buffers grow on demand outside capture.
constexpr int INA_WS_SLOTS = 8;
constexpr int INA_WS_KINDS = 2;   // 0 (unused since round 6), 1 decode-attention partials
int ina_workspace(int kind, size_t bytes, hipStream_t stream, float** out);

// Optional per-launch timing (ina_prof_enable): every launch function opens a scope that records a hipEvent pair on the
// launch stream around its kernel and tallies the algorithmic FLOPs / bytes of that launch. bench.py reads the totals per
// kernel class for the live roofline line; disabled (the default) it costs one branch. Not usable under graph capture.
enum : int { INA_PROF_GEMM = 0, INA_PROF_ATTN = 1, INA_PROF_NORM = 2, INA_PROF_ELEMENTWISE = 3, INA_PROF_GEMM_SKINNY = 4, INA_PROF_KINDS = 5 };
// sub-tag of the NEXT profiled launch of this thread (the GEMM tile config id; 42 = dit_rowchain): lets bench.py tally one
// named kernel on its own (ina_prof_read_sub). Consumed (reset to 0) by the next InaProfScope.
void ina_prof_set_sub(int sub);
struct InaProfScope {
    InaProfScope(int kind, double flops, double bytes, hipStream_t stream);
    ~InaProfScope();
    int idx;
    hipStream_t stream;
};
