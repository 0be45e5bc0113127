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
//   * phase A: per-(token, segment) LayerNorm statistics -> LDS, computed with MFMAs (row sums = X . 1^T, sums of squares =
//     diag(X . X^T); the kernel was VALU-issue bound: PMC 1981 VALU instructions per wave); the wave's V1 head slice is
//     transposed into its private LDS tile meanwhile;
//   * phase B: Q / K MFMA fragments are loaded straight from global (L2 hits: phase A just read the rows) and normalised in
//     registers; S^T = K . Q^T and O^T = V^T . P^T in the swapped-operand form of attention.hip (P never leaves its lane);
//   * the condition V is consumed from a pre-transposed, key-permuted image V2T[env][head][64][64] built once per call;
//   * the kernel is latency-bound (few MFMAs per byte), so every load is hoisted: condition K / V fragments first, then all
//     statistics-pass rows and the V1 slice in one batch, then all phase-B q / k slices in one batch.
#include "common.h"
#include "kernels.h"

namespace {

__device__ __forceinline__ int dit_vt_pos(int kv_local) {
    // key permutation inside a 32-key sub-block (same as attention.hip): MFMA k index g*8 + j <-> key (j<4 ? g*4+j : 16+g*4+(j-4))
    int sub = kv_local >> 5, w = kv_local & 31;
    int t = w >> 4, x = w & 15;
    return (sub << 5) + ((x >> 2) << 3) + (x & 3) + (t << 2);
}

// raw 8-element slice + LayerNorm with the row's (mean, rstd) and the 8 affine parameters of those columns
// (scalar fp32 ops on purpose: the packed v_pk_* form measured 15 % slower here, next to the MFMAs)
__device__ __forceinline__ bf16x8 ln_apply8(bf16x8 x, float mean, float rstd, const f32x4& g0, const f32x4& g1, const f32x4& b0, const f32x4& b1) {
    bf16x8 y;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        y[i] = (bf16)(((float)x[i] - mean) * rstd * g0[i] + b0[i]);
        y[4 + i] = (bf16)(((float)x[4 + i] - mean) * rstd * g1[i] + b1[i]);
    }
    return y;
}

template <int NH>
__global__ __launch_bounds__(NH * 64, 3) void dit_attn_kernel(DitAttnArgs p) {
    constexpr int HD = 64, D = NH * HD, VT_LD = 40;
    __shared__ float stats[3][32][2];
    __shared__ __attribute__((aligned(16))) bf16 Vt[NH][HD * VT_LD];
    const int tid = threadIdx.x, lane = tid & 63, h = tid >> 6;
    const int g = lane >> 4, lq = lane & 15;
    const int seq = blockIdx.x, env = seq / p.seq_per_env;
    const int T = p.T;
    const bf16* __restrict__ base = reinterpret_cast<const bf16*>(p.X) + (size_t)seq * T * p.ldx;
    // Rows / keys beyond T or Lz are read from a clamped (valid) row instead of being predicated: their scores are masked to -inf and
    // their outputs never stored, and no load needs an exec-mask branch.
    const bf16* __restrict__ K2 = reinterpret_cast<const bf16*>(p.K2) + (size_t)env * p.k2_bs + h * HD;
    const bf16* __restrict__ V2T = reinterpret_cast<const bf16*>(p.V2T) + ((size_t)env * NH + h) * HD * 64;

    // ---- condition K / V fragments of this head: independent of everything else, requested first (L2 hits: shared by the env's
    //      samples) so their latency hides under the statistics pass
    bf16x8 k2f[4][2], v2f[2][4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int key = t * 16 + lq;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
            k2f[t][kk] = *reinterpret_cast<const bf16x8*>(K2 + (size_t)min(key, p.Lz - 1) * p.k2_rs + g * 8 + kk * 32);
    }

    // ---- phase A: LayerNorm statistics of the q1 / k1 / q2 segment of every token ON THE MATRIX CORES: wave h owns one
    //      (segment, 16-token tile) unit; with X = its [16 tokens x D] slice, X . 1^T gives the row sums and diag(X . X^T) the row sums
    //      of squares (bf16 products are exact in fp32), 2 x D/32 MFMAs instead of ~500 VALU instructions per wave
    {
        static_assert(NH == 6, "one (segment, token tile) unit per wave: 3 segments x 2 tiles");
        constexpr int NCH = D / 32;
        const int us = h >> 1, utt = h & 1, useg = us == 2 ? 3 : us;
        const int utok = utt * 16 + lq;
        bf16x8 raw[NCH];
        {
            const bf16* row = base + (size_t)(utok < T ? utok : 0) * p.ldx + useg * D + g * 8;
#pragma unroll
            for (int c = 0; c < NCH; ++c) raw[c] = *reinterpret_cast<const bf16x8*>(row + c * 32);
        }
        // this wave's V1 head slice [32 keys x 64] -> transposed, key-permuted LDS tile
        // (lane -> 8-column chunk c = lane & 7 of key rows {2m, 2m+1, 16+2m, 16+2m+1}, m = lane >> 3: neighbouring keys land on
        //  neighbouring slots of the permuted image, so two keys go out per 32-bit LDS store)
        bf16x8 vraw[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = (i >> 1) * 16 + (lane >> 3) * 2 + (i & 1);
            vraw[i] = *reinterpret_cast<const bf16x8*>(base + (size_t)min(row, T - 1) * p.ldx + 2 * D + h * HD + (lane & 7) * 8);
        }
        {
            const bf16 one = (bf16)1.0f;
            const bf16x8 ones = {one, one, one, one, one, one, one, one};
            f32x4 asum = {0.f, 0.f, 0.f, 0.f}, asq = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                asum = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, raw[c], asum, 0, 0, 0);     // D[i][j] = sum_k X[j][k]
                asq = __builtin_amdgcn_mfma_f32_16x16x32_bf16(raw[c], raw[c], asq, 0, 0, 0);      // D[i][j] = X[i] . X[j]
            }
            // lane (lq, g) holds D[i = g*4 + r][j = lq]: the diagonal entry of token lq sits in lane group g = lq >> 2, element lq & 3
            if (g == (lq >> 2)) {
                const int r = lq & 3;
                const float ss = r == 0 ? asq[0] : r == 1 ? asq[1] : r == 2 ? asq[2] : asq[3];
                const float mean = asum[0] * (1.0f / D);
                const float var = fmaxf(ss * (1.0f / D) - mean * mean, 0.f);
                stats[us][utok][0] = mean;
                stats[us][utok][1] = rsqrtf(var + p.eps);
            }
        }
        // image row d keeps its four 8-slot groups rotated by d >> 3 (the chunk index c): the 8 chunk lanes of a store then hit 8
        // different bank groups instead of one
        bf16* vt = Vt[h];
        const int c = lane & 7;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int pos = (dit_vt_pos(i * 16 + (lane >> 3) * 2) + 8 * c) & 31;
#pragma unroll
            for (int e = 0; e < 8; ++e)
                *reinterpret_cast<bf16x2*>(&vt[(c * 8 + e) * VT_LD + pos]) = bf16x2{vraw[2 * i][e], vraw[2 * i + 1][e]};
        }
    }
    __syncthreads();

    const float sc = p.scale * 1.4426950408889634f;   // exp2 domain
    const float gate = p.head_gate ? tanhf(p.head_gate[h]) : 1.0f;
    bf16* __restrict__ O = reinterpret_cast<bf16*>(p.O) + (size_t)seq * T * p.ldo + h * HD;
    const int dcol = h * HD + g * 8;                     // first of this lane's 8 columns inside a 32-wide k step (+ kk*32)

    // ---- phase B loads (L2 hits: phase A just read these rows): raw q1 / q2 of both query tiles, raw k1 of both key tiles
    bf16x8 k1r[2][2];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        const int tok = qt * 16 + lq;
        const bf16* row = base + (size_t)(tok < T ? tok : 0) * p.ldx + dcol;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) k1r[qt][kk] = *reinterpret_cast<const bf16x8*>(row + D + kk * 32);
    }
    // condition V fragments (requested here rather than with K2: 32 fewer live registers across the statistics pass, no spills)
#pragma unroll
    for (int sb = 0; sb < 2; ++sb)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) v2f[sb][nt] = *reinterpret_cast<const bf16x8*>(V2T + (size_t)(nt * 16 + lq) * 64 + sb * 32 + g * 8);
    // normalised K1 fragments (keys t*16 + lq)
    bf16x8 k1f[2][2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        const f32x4 g0 = *reinterpret_cast<const f32x4*>(p.g_k1 + dcol + kk * 32), g1 = *reinterpret_cast<const f32x4*>(p.g_k1 + dcol + kk * 32 + 4);
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(p.b_k1 + dcol + kk * 32), b1 = *reinterpret_cast<const f32x4*>(p.b_k1 + dcol + kk * 32 + 4);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int key = t * 16 + lq;
            k1f[t][kk] = ln_apply8(k1r[t][kk], stats[1][key][0], stats[1][key][1], g0, g1, b0, b1);
        }
    }

#pragma unroll 1
    for (int qt = 0; qt < 2; ++qt) {
        const int tok = qt * 16 + lq;
        if (qt * 16 >= T) break;
        const bool qok = tok < T;
        bf16x8 q1r[2], q2r[2];
        {
            const bf16* row = base + (size_t)(qok ? tok : 0) * p.ldx + dcol;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                q1r[kk] = *reinterpret_cast<const bf16x8*>(row + kk * 32);
                q2r[kk] = *reinterpret_cast<const bf16x8*>(row + 3 * D + kk * 32);
            }
        }
        // ================= self-attention over the sequence's own tokens
        f32x4 o1[4];
        {
            bf16x8 qf[2];
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const f32x4 g0 = *reinterpret_cast<const f32x4*>(p.g_q1 + dcol + kk * 32), g1 = *reinterpret_cast<const f32x4*>(p.g_q1 + dcol + kk * 32 + 4);
                const f32x4 b0 = *reinterpret_cast<const f32x4*>(p.b_q1 + dcol + kk * 32), b1 = *reinterpret_cast<const f32x4*>(p.b_q1 + dcol + kk * 32 + 4);
                qf[kk] = ln_apply8(q1r[kk], stats[0][tok][0], stats[0][tok][1], g0, g1, b0, b1);
            }
            f32x4 s[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                s[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) s[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k1f[t][kk], qf[kk], s[t], 0, 0, 0);
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
                    const float e = __builtin_amdgcn_exp2f(s[t][r] - mx);
                    l += e;
                    pf[t * 4 + r] = (bf16)e;
                }
            l += __shfl_xor(l, 16);
            l += __shfl_xor(l, 32);
            const float inv_l = 1.0f / l;
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const int d = nt * 16 + lq;
                const bf16x8 vf = *reinterpret_cast<const bf16x8*>(&Vt[h][d * VT_LD + ((g + (d >> 3)) & 3) * 8]);
                o1[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 4; ++r) o1[nt][r] = (float)(bf16)(o1[nt][r] * inv_l);   // the unfused path stores bf16 here
            }
        }
        // ================= gated cross-attention against the env's condition rows
        {
            bf16x8 qf[2];
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const f32x4 g0 = *reinterpret_cast<const f32x4*>(p.g_q2 + dcol + kk * 32), g1 = *reinterpret_cast<const f32x4*>(p.g_q2 + dcol + kk * 32 + 4);
                const f32x4 b0 = *reinterpret_cast<const f32x4*>(p.b_q2 + dcol + kk * 32), b1 = *reinterpret_cast<const f32x4*>(p.b_q2 + dcol + kk * 32 + 4);
                qf[kk] = ln_apply8(q2r[kk], stats[2][tok][0], stats[2][tok][1], g0, g1, b0, b1);
            }
            f32x4 s[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                s[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) s[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k2f[t][kk], qf[kk], s[t], 0, 0, 0);
            }
            // keys of this lane: t*16 + g*4 + r; Lz - g*4 is the per-lane bound on t*16 + r (one compare per element, no index math)
            const int lim = p.Lz - g * 4;
            float mx = -INFINITY;
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float v = (t * 16 + r) < lim ? s[t][r] * sc : -INFINITY;
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
                    const float e = __builtin_amdgcn_exp2f(s[t][r] - mx);
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
                for (int nt = 0; nt < 4; ++nt) o2[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(v2f[sb][nt], pf, o2[nt], 0, 0, 0);
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

// ---- the same stage when the PRODUCER of the projection rows hands over the LayerNorm statistics (ina_dit_rowchain computes (mean, rstd)
// of every 384-wide segment of its output rows in its epilogue, from the fp32 accumulators): no statistics pass, no LDS exchange between
// waves and no barrier - a (sequence, head) unit is ONE single-wave workgroup with ONE exposed memory latency (every load it will ever
// need is requested at the top: the raw q1 / k1 / q2 slices of both 16-token tiles, the V1 slice, the condition K / V fragments, six
// (mean, rstd) pairs), then straight-line MFMA / VALU work on both tiles (two independent dependency chains, no branch between them). The
// projection row is read ONCE (the statistics pass read three of its four segments a second time).
// Measured (tools/native/dit_attn_probe, 64 envs x 32 samples x 32 tokens, Lz 64; profiles/r05k_*, r05o_*): 112 us with its own statistics
// pass -> 89 us as six-wave workgroups given the statistics -> 71 us as single-wave workgroups (six waves that start in lockstep also stall
// in lockstep; 3 or 2 waves per SIMD measure the same; a head's 64 columns are exactly one 128-byte line per row: heads share no lines;
// an XCD-aware unit -> env mapping changes nothing: the condition fragments are L2 hits either way) -> 60 us with the LayerNorm vectors
// through an LDS table (below) = 4.2 TB/s of the 251 MB; the engine's shape (57 envs, Lz 36: three key tiles) 104 -> 48 us = 4.7 TB/s.
// KT: 16-key tiles of the condition that can hold a real key (Lz <= 16 * KT)
template <int NH, int KT>
__global__ __launch_bounds__(64, 2) void dit_attn_stats_kernel(DitAttnArgs p) {
    constexpr int HD = 64, D = NH * HD, VT_LD = 40;
    __shared__ __attribute__((aligned(16))) bf16 Vt[HD * VT_LD];
    __shared__ __attribute__((aligned(16))) float gb[6][HD];   // LayerNorm weight / bias of q1, k1, q2 restricted to this head's 64 columns
    const int lane = threadIdx.x, h = (int)(blockIdx.x % NH);
    const int g = lane >> 4, lq = lane & 15;
    const int seq = (int)(blockIdx.x / NH), env = seq / p.seq_per_env;
    const int T = p.T;
    const bf16* __restrict__ base = reinterpret_cast<const bf16*>(p.X) + (size_t)seq * T * p.ldx;
    const float* __restrict__ st = p.stats + (size_t)seq * T * p.stats_ld;
    const bf16* __restrict__ K2 = reinterpret_cast<const bf16*>(p.K2) + (size_t)env * p.k2_bs + h * HD;
    const bf16* __restrict__ V2T = reinterpret_cast<const bf16*>(p.V2T) + ((size_t)env * NH + h) * HD * 64;
    const int dcol = h * HD + g * 8;                     // first of this lane's 8 columns inside a 32-wide k step (+ kk*32)

    // ---- every load of the wave (rows / keys beyond T or Lz come from a clamped valid row: masked to -inf / never stored)
    // the head's slice of the six LayerNorm vectors: one coalesced 4-byte load per vector (lane = column) into an LDS table the lanes then read
    // their 8-column groups from (per-lane 16-byte global loads of the same 1.5 KB - 24 per wave, each lane group repeating the others' addresses -
    // were 13 us of the 71 us kernel)
    float gbr[6];
    {
        const float* const vec[6] = {p.g_q1, p.b_q1, p.g_k1, p.b_k1, p.g_q2, p.b_q2};
#pragma unroll
        for (int i = 0; i < 6; ++i) gbr[i] = vec[i][h * HD + lane];
    }
    bf16x8 q1r[2][2], k1r[2][2], q2r[2][2], vraw[4], k2f[4][2], v2f[2][4];
    f32x2 sq1[2], sk1[2], sq2[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int tok = min(t * 16 + lq, T - 1);
        const bf16* row = base + (size_t)tok * p.ldx + dcol;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            k1r[t][kk] = *reinterpret_cast<const bf16x8*>(row + D + kk * 32);
            q1r[t][kk] = *reinterpret_cast<const bf16x8*>(row + kk * 32);
            q2r[t][kk] = *reinterpret_cast<const bf16x8*>(row + 3 * D + kk * 32);
        }
        const float* srow = st + (size_t)tok * p.stats_ld;
        sq1[t] = *reinterpret_cast<const f32x2*>(srow);
        sk1[t] = *reinterpret_cast<const f32x2*>(srow + 2);
        sq2[t] = *reinterpret_cast<const f32x2*>(srow + 6);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = (i >> 1) * 16 + (lane >> 3) * 2 + (i & 1);
        vraw[i] = *reinterpret_cast<const bf16x8*>(base + (size_t)min(row, T - 1) * p.ldx + 2 * D + h * HD + (lane & 7) * 8);
    }
#pragma unroll
    for (int t = 0; t < KT; ++t) {
        const int key = t * 16 + lq;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
            k2f[t][kk] = *reinterpret_cast<const bf16x8*>(K2 + (size_t)min(key, p.Lz - 1) * p.k2_rs + g * 8 + kk * 32);
    }
#pragma unroll
    for (int sb = 0; sb < (KT + 1) / 2; ++sb)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) v2f[sb][nt] = *reinterpret_cast<const bf16x8*>(V2T + (size_t)(nt * 16 + lq) * 64 + sb * 32 + g * 8);

#pragma unroll
    for (int i = 0; i < 6; ++i) gb[i][lane] = gbr[i];
    // ---- this wave's V1 head slice -> its own transposed, key-permuted LDS tile (layout as in dit_attn_kernel; written and read by this
    //      wave only: LDS operations of one wave complete in order, no barrier)
    {
        bf16* vt = Vt;
        const int c = lane & 7;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int pos = (dit_vt_pos(i * 16 + (lane >> 3) * 2) + 8 * c) & 31;
#pragma unroll
            for (int e = 0; e < 8; ++e)
                *reinterpret_cast<bf16x2*>(&vt[(c * 8 + e) * VT_LD + pos]) = bf16x2{vraw[2 * i][e], vraw[2 * i + 1][e]};
        }
    }

    const float sc = p.scale * 1.4426950408889634f;   // exp2 domain
    const float gate = p.head_gate ? tanhf(p.head_gate[h]) : 1.0f;
    bf16* __restrict__ O = reinterpret_cast<bf16*>(p.O) + (size_t)seq * T * p.ldo + h * HD;

    // normalised K1 fragments (keys t*16 + lq)
    bf16x8 k1f[2][2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        const f32x4 g0 = *reinterpret_cast<const f32x4*>(&gb[2][g * 8 + kk * 32]), g1 = *reinterpret_cast<const f32x4*>(&gb[2][g * 8 + kk * 32 + 4]);
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(&gb[3][g * 8 + kk * 32]), b1 = *reinterpret_cast<const f32x4*>(&gb[3][g * 8 + kk * 32 + 4]);
#pragma unroll
        for (int t = 0; t < 2; ++t) k1f[t][kk] = ln_apply8(k1r[t][kk], sk1[t][0], sk1[t][1], g0, g1, b0, b1);
    }

#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        // (both 16-token tiles unconditionally, no branch between them: rows beyond T are clamped copies that are never stored - the two
        //  tiles are independent dependency chains the scheduler may interleave)
        const int tok = qt * 16 + lq;
        const bool qok = tok < T;
        // ================= self-attention over the sequence's own tokens
        f32x4 o1[4];
        {
            bf16x8 qf[2];
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const f32x4 g0 = *reinterpret_cast<const f32x4*>(&gb[0][g * 8 + kk * 32]), g1 = *reinterpret_cast<const f32x4*>(&gb[0][g * 8 + kk * 32 + 4]);
                const f32x4 b0 = *reinterpret_cast<const f32x4*>(&gb[1][g * 8 + kk * 32]), b1 = *reinterpret_cast<const f32x4*>(&gb[1][g * 8 + kk * 32 + 4]);
                qf[kk] = ln_apply8(q1r[qt][kk], sq1[qt][0], sq1[qt][1], g0, g1, b0, b1);
            }
            f32x4 s[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                s[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) s[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k1f[t][kk], qf[kk], s[t], 0, 0, 0);
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
                    const float e = __builtin_amdgcn_exp2f(s[t][r] - mx);
                    l += e;
                    pf[t * 4 + r] = (bf16)e;
                }
            l += __shfl_xor(l, 16);
            l += __shfl_xor(l, 32);
            const float inv_l = 1.0f / l;
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const int d = nt * 16 + lq;
                const bf16x8 vf = *reinterpret_cast<const bf16x8*>(&Vt[d * VT_LD + ((g + (d >> 3)) & 3) * 8]);
                o1[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 4; ++r) o1[nt][r] = (float)(bf16)(o1[nt][r] * inv_l);   // the unfused path stores bf16 here
            }
        }
        // ================= gated cross-attention against the env's condition rows
        {
            bf16x8 qf[2];
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const f32x4 g0 = *reinterpret_cast<const f32x4*>(&gb[4][g * 8 + kk * 32]), g1 = *reinterpret_cast<const f32x4*>(&gb[4][g * 8 + kk * 32 + 4]);
                const f32x4 b0 = *reinterpret_cast<const f32x4*>(&gb[5][g * 8 + kk * 32]), b1 = *reinterpret_cast<const f32x4*>(&gb[5][g * 8 + kk * 32 + 4]);
                qf[kk] = ln_apply8(q2r[qt][kk], sq2[qt][0], sq2[qt][1], g0, g1, b0, b1);
            }
            f32x4 s[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                s[t] = f32x4{0.f, 0.f, 0.f, 0.f};      // tiles >= KT hold no key: their probabilities stay 0
                if (t < KT) {
#pragma unroll
                    for (int kk = 0; kk < 2; ++kk) s[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k2f[t][kk], qf[kk], s[t], 0, 0, 0);
                }
            }
            const int lim = p.Lz - g * 4;
            float mx = -INFINITY;
#pragma unroll
            for (int t = 0; t < KT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float v = (t * 16 + r) < lim ? s[t][r] * sc : -INFINITY;
                    s[t][r] = v;
                    mx = fmaxf(mx, v);
                }
            mx = fmaxf(mx, __shfl_xor(mx, 16));
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            float l = 0.f;
#pragma unroll
            for (int t = 0; t < KT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float e = __builtin_amdgcn_exp2f(s[t][r] - mx);
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
            for (int sb = 0; sb < (KT + 1) / 2; ++sb) {
                bf16x8 pf;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    pf[r] = (bf16)s[2 * sb][r];
                    pf[4 + r] = (bf16)s[2 * sb + 1][r];
                }
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) o2[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(v2f[sb][nt], pf, o2[nt], 0, 0, 0);
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
    if (p.stats) {
        INA_REQUIRE(p.stats_ld >= 8 && p.stats_ld % 2 == 0 && ((uintptr_t)p.stats % 8) == 0, "dit_attention: stats rows are [4 segments][mean, rstd] f32 (stats_ld=%d)", p.stats_ld);
        if (p.Lz <= 16) hipLaunchKernelGGL((dit_attn_stats_kernel<6, 1>), dim3(p.nseq * 6), dim3(64), 0, stream, p);
        else if (p.Lz <= 32) hipLaunchKernelGGL((dit_attn_stats_kernel<6, 2>), dim3(p.nseq * 6), dim3(64), 0, stream, p);
        else if (p.Lz <= 48) hipLaunchKernelGGL((dit_attn_stats_kernel<6, 3>), dim3(p.nseq * 6), dim3(64), 0, stream, p);
        else hipLaunchKernelGGL((dit_attn_stats_kernel<6, 4>), dim3(p.nseq * 6), dim3(64), 0, stream, p);
    } else {
        hipLaunchKernelGGL(dit_attn_kernel<6>, dim3(p.nseq), dim3(6 * 64), 0, stream, p);
    }
    INA_HIP_CHECK(hipGetLastError());
    return 0;
}
