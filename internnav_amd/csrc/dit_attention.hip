// Fused attention stage of one NextDiT block for gfx950 (reference nextdit_traj.py:121-188 on top of diffusers'
// LuminaNextDiTBlock / LuminaAttnProcessor2_0 with qk_norm = "layer_norm_across_heads"):
//
//     att = SDPA(LN(q1), LN(k1), v1)  +  tanh(gate[h]) * SDPA(LN(q2), K2, V2)
//
// where q1 | k1 | v1 | q2 are the four D-wide segments of one fused projection row (D = heads * 64), LN is a LayerNorm over the whole
// D-wide segment (i.e. across heads), the self-attention runs inside one short action sequence (T <= 32 tokens) and the gated
// cross-attention reads the per-env condition K/V (Lz <= 64 rows, shared by the env's samples, input independent of the sampler step).
// Unfused this is 3 LayerNorm launches (each a read + write of a [rows, D] segment) + 2 attention launches (the second a
// read-modify-write of the output): 650 MB of HBM traffic per block at 65536 rows. Here the projection row is read once and the
// output written once (250 MB), in one launch:
//   * one workgroup = one sequence, one wave per head;
//   * phase A: per-(token, segment) LayerNorm statistics -> LDS (16 lanes per row, fp32, two-pass variance); the wave's V1 head
//     slice is transposed into its private LDS tile meanwhile;
//   * phase B: Q / K MFMA fragments are loaded straight from global (L2 hits: phase A just read the rows) and normalised in
//     registers; S^T = K . Q^T and O^T = V^T . P^T in the swapped-operand form of attention.hip (P never leaves its lane);
//   * the condition V is consumed from a pre-transposed, key-permuted image V2T[env][head][64][64] built once per call.
#include "common.h"
#include "kernels.h"

namespace {

__device__ __forceinline__ int dit_vt_pos(int kv_local) {
    // key permutation inside a 32-key sub-block (same as attention.hip): MFMA k index g*8 + j <-> key (j<4 ? g*4+j : 16+g*4+(j-4))
    int sub = kv_local >> 5, w = kv_local & 31;
    int t = w >> 4, x = w & 15;
    return (sub << 5) + ((x >> 2) << 3) + (x & 3) + (t << 2);
}

// 8 consecutive bf16 of a row, LayerNorm-ed with the row's (mean, rstd) and the affine parameters of those 8 columns
__device__ __forceinline__ bf16x8 ln_load8(const bf16* ptr, float mean, float rstd, const float* __restrict__ gam, const float* __restrict__ bet) {
    const bf16x8 x = *reinterpret_cast<const bf16x8*>(ptr);
    const f32x4 g0 = *reinterpret_cast<const f32x4*>(gam), g1 = *reinterpret_cast<const f32x4*>(gam + 4);
    const f32x4 b0 = *reinterpret_cast<const f32x4*>(bet), b1 = *reinterpret_cast<const f32x4*>(bet + 4);
    bf16x8 y;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        y[i] = (bf16)(((float)x[i] - mean) * rstd * g0[i] + b0[i]);
        y[4 + i] = (bf16)(((float)x[4 + i] - mean) * rstd * g1[i] + b1[i]);
    }
    return y;
}

template <int NH>
__global__ __launch_bounds__(NH * 64) void dit_attn_kernel(DitAttnArgs p) {
    constexpr int HD = 64, D = NH * HD, VT_LD = 40, CPL = NH / 2;   // CPL: 16-byte chunks per lane in the statistics pass
    static_assert(NH % 2 == 0, "statistics pass covers a D-wide row with 16 lanes x NH/2 chunks");
    __shared__ float stats[3][32][2];
    __shared__ __attribute__((aligned(16))) bf16 Vt[NH][HD * VT_LD];
    const int tid = threadIdx.x, lane = tid & 63, h = tid >> 6;
    const int g = lane >> 4, lq = lane & 15;
    const int seq = blockIdx.x, env = seq / p.seq_per_env;
    const int T = p.T;
    const bf16* __restrict__ base = reinterpret_cast<const bf16*>(p.X) + (size_t)seq * T * p.ldx;
    const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};

    // ---- phase A: LayerNorm statistics of the q1 / k1 / q2 segment of every token (row r = s*32 + token, 4 rows per wave pass)
    for (int r = h * 4 + g; r < 96; r += NH * 4) {
        const int s = r >> 5, tok = r & 31, seg = s == 2 ? 3 : s;
        float x[CPL * 8];
        float sum = 0.f;
        if (tok < T) {
            const bf16* row = base + (size_t)tok * p.ldx + seg * D;
#pragma unroll
            for (int c = 0; c < CPL; ++c) {
                const bf16x8 v = *reinterpret_cast<const bf16x8*>(row + (c * 16 + lq) * 8);
#pragma unroll
                for (int i = 0; i < 8; ++i) { x[c * 8 + i] = (float)v[i]; sum += x[c * 8 + i]; }
            }
        } else {
#pragma unroll
            for (int i = 0; i < CPL * 8; ++i) x[i] = 0.f;
        }
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
        const float mean = sum * (1.0f / D);
        float var = 0.f;
#pragma unroll
        for (int i = 0; i < CPL * 8; ++i) { const float d = x[i] - mean; var += d * d; }
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) var += __shfl_xor(var, o);
        if (lq == 0) { stats[s][tok][0] = mean; stats[s][tok][1] = rsqrtf(var * (1.0f / D) + p.eps); }
    }
    // ---- this wave's V1 head slice [32 keys x 64] -> transposed, key-permuted LDS tile
    {
        bf16* vt = Vt[h];
        const int c = lane & 7;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = i * 8 + (lane >> 3);
            const bf16x8 v = row < T ? *reinterpret_cast<const bf16x8*>(base + (size_t)row * p.ldx + 2 * D + h * HD + c * 8) : zero8;
            const int pos = dit_vt_pos(row);
#pragma unroll
            for (int e = 0; e < 8; ++e) vt[(c * 8 + e) * VT_LD + pos] = v[e];
        }
    }
    __syncthreads();

    const float sc = p.scale * 1.4426950408889634f;   // exp2 domain
    const float gate = p.head_gate ? tanhf(p.head_gate[h]) : 1.0f;
    const bf16* __restrict__ K2 = reinterpret_cast<const bf16*>(p.K2) + (size_t)env * p.k2_bs + h * HD;
    const bf16* __restrict__ V2T = reinterpret_cast<const bf16*>(p.V2T) + ((size_t)env * NH + h) * HD * 64;
    bf16* __restrict__ O = reinterpret_cast<bf16*>(p.O) + (size_t)seq * T * p.ldo + h * HD;
    const int dcol = h * HD + g * 8;                     // first of this lane's 8 columns inside a 32-wide k step (+ kk*32)

#pragma unroll 1
    for (int qt = 0; qt < 2; ++qt) {
        const int tok = qt * 16 + lq;
        if (qt * 16 >= T) break;
        const bool qok = tok < T;
        const bf16* qrow = base + (size_t)(qok ? tok : 0) * p.ldx;
        // ================= self-attention over the sequence's own tokens
        f32x4 o1[4];
        {
            bf16x8 qf[2];
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
                qf[kk] = qok ? ln_load8(qrow + dcol + kk * 32, stats[0][tok][0], stats[0][tok][1], p.g_q1 + dcol + kk * 32, p.b_q1 + dcol + kk * 32) : zero8;
            f32x4 s[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int key = t * 16 + lq;
                s[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    const bf16x8 kf = key < T ? ln_load8(base + (size_t)key * p.ldx + D + dcol + kk * 32, stats[1][key][0], stats[1][key][1],
                                                         p.g_k1 + dcol + kk * 32, p.b_k1 + dcol + kk * 32) : zero8;
                    s[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[kk], s[t], 0, 0, 0);
                }
            }
            float mx = -INFINITY;
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int kv = t * 16 + g * 4 + r;
                    const float v = kv < T ? s[t][r] * sc : -INFINITY;
                    s[t][r] = v;
                    mx = fmaxf(mx, v);
                }
            mx = fmaxf(mx, __shfl_xor(mx, 16));
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            float l = 0.f;
            bf16x8 pf;
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float e = exp2f(s[t][r] - mx);
                    l += e;
                    pf[t * 4 + r] = (bf16)e;
                }
            l += __shfl_xor(l, 16);
            l += __shfl_xor(l, 32);
            const float inv_l = 1.0f / l;
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const bf16x8 vf = *reinterpret_cast<const bf16x8*>(&Vt[h][(nt * 16 + lq) * VT_LD + g * 8]);
                o1[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 4; ++r) o1[nt][r] = (float)(bf16)(o1[nt][r] * inv_l);   // the unfused path stores bf16 here
            }
        }
        // ================= gated cross-attention against the env's condition rows
        {
            bf16x8 qf[2];
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
                qf[kk] = qok ? ln_load8(qrow + 3 * D + dcol + kk * 32, stats[2][tok][0], stats[2][tok][1], p.g_q2 + dcol + kk * 32, p.b_q2 + dcol + kk * 32) : zero8;
            f32x4 s[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int key = t * 16 + lq;
                s[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    const bf16x8 kf = key < p.Lz ? *reinterpret_cast<const bf16x8*>(K2 + (size_t)key * p.k2_rs + g * 8 + kk * 32) : zero8;
                    s[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[kk], s[t], 0, 0, 0);
                }
            }
            float mx = -INFINITY;
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int kv = t * 16 + g * 4 + r;
                    const float v = kv < p.Lz ? s[t][r] * sc : -INFINITY;
                    s[t][r] = v;
                    mx = fmaxf(mx, v);
                }
            mx = fmaxf(mx, __shfl_xor(mx, 16));
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            float l = 0.f;
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float e = exp2f(s[t][r] - mx);
                    s[t][r] = e;
                    l += e;
                }
            l += __shfl_xor(l, 16);
            l += __shfl_xor(l, 32);
            const float w = gate / l;
            f32x4 o2[4];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) o2[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int sb = 0; sb < 2; ++sb) {
                bf16x8 pf;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    pf[r] = (bf16)s[2 * sb][r];
                    pf[4 + r] = (bf16)s[2 * sb + 1][r];
                }
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    const bf16x8 vf = *reinterpret_cast<const bf16x8*>(V2T + (size_t)(nt * 16 + lq) * 64 + sb * 32 + g * 8);
                    o2[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf, o2[nt], 0, 0, 0);
                }
            }
            if (qok) {
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    const bf16x4 o = {(bf16)(o1[nt][0] + o2[nt][0] * w), (bf16)(o1[nt][1] + o2[nt][1] * w),
                                      (bf16)(o1[nt][2] + o2[nt][2] * w), (bf16)(o1[nt][3] + o2[nt][3] * w)};
                    *reinterpret_cast<bf16x4*>(O + (size_t)tok * p.ldo + nt * 16 + g * 4) = o;
                }
            }
        }
    }
}

// condition V [env][Lz rows][heads x 64] -> V2T[env][head][64 dims][64 key slots], slot = dit_vt_pos(key), zero beyond Lz
__global__ __launch_bounds__(256) void dit_v2t_kernel(const bf16* __restrict__ V, bf16* __restrict__ V2T, long v_bs, long v_rs, int nh, int Lz, long total) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int slot = (int)(i & 63), d = (int)((i >> 6) & 63);
        const long eh = i >> 12;
        const int hh = (int)(eh % nh);
        const long env = eh / nh;
        // invert the permutation: slot -> key
        const int sub = slot >> 5, w = slot & 31, gg = w >> 3, j = w & 7;
        const int key = sub * 32 + (j < 4 ? gg * 4 + j : 16 + gg * 4 + (j - 4));
        V2T[i] = key < Lz ? V[env * v_bs + (long)key * v_rs + hh * 64 + d] : (bf16)0.f;
    }
}

}  // namespace

int ina_launch_dit_attention(const DitAttnArgs& p, hipStream_t stream) {
    INA_REQUIRE(p.heads == 6, "dit_attention: heads=%d (only the 6 x 64 geometry of NextDiTCrossAttnConfig is built)", p.heads);
    INA_REQUIRE(p.T >= 1 && p.T <= 32 && p.Lz >= 1 && p.Lz <= 64, "dit_attention: T=%d (<= 32), Lz=%d (<= 64)", p.T, p.Lz);
    INA_REQUIRE(p.ldx % 8 == 0 && p.ldo % 4 == 0 && p.k2_rs % 8 == 0 && p.k2_bs % 8 == 0 && p.seq_per_env >= 1, "dit_attention: strides must keep 16-byte alignment");
    if (p.V2T_src) {
        const long total = (long)((p.nseq + p.seq_per_env - 1) / p.seq_per_env) * p.heads * 64 * 64;
        InaProfScope prof(INA_PROF_ELEMENTWISE, 0.0, 4.0 * total, stream);
        const int nb = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
        hipLaunchKernelGGL(dit_v2t_kernel, dim3(nb), dim3(256), 0, stream, reinterpret_cast<const bf16*>(p.V2T_src), reinterpret_cast<bf16*>(p.V2T),
                           (long)p.v2_bs, (long)p.v2_rs, p.heads, p.Lz, total);
        INA_HIP_CHECK(hipGetLastError());
        if (!p.X) return 0;
    }
    if (p.nseq == 0) return 0;
    const double D = p.heads * 64.0, rows = (double)p.nseq * p.T;
    InaProfScope prof(INA_PROF_ATTN, 4.0 * rows * D * (p.T + p.Lz), 2.0 * rows * D * 5.0, stream);
    hipLaunchKernelGGL(dit_attn_kernel<6>, dim3(p.nseq), dim3(6 * 64), 0, stream, p);
    INA_HIP_CHECK(hipGetLastError());
    return 0;
}
