// Flash-style attention forward for gfx950 (bf16 in/out, fp32 softmax + accumulation).
// Replaces flash-attn 2 / xformers / nn.MultiheadAttention SDPA on the reference hot path:
//   DINOv2 ViT-S (6 h x d64, 257 tokens), NavDP former/decoder (8 h x d48, causal T=24/32, cross M=34/132/2304),
//   NextDiT (6 h x d64, T=32 / 36), Qwen2.5-VL ViT windows (16 h x d80, varlen) and LLM prefill (GQA 28q/4kv x d128, causal).
//
// One workgroup = NW waves = NW*16 query rows of one (sequence, head); K/V are streamed in KVB-row blocks through LDS
// and shared by the NW waves. Both MFMAs are issued in the "swapped" form so that lane l always owns query row (l & 15)
// and lane group g = l >> 4 owns a fixed subset of the keys of the current block:
//   S^T[kv][q] = K . Q^T   (A operand = K fragment, B operand = Q fragment)  -> lane holds S[q][kv = t*16 + g*4 + r]
//   O^T[d][q]  = V^T . P^T (A operand = V^T fragment, B operand = P fragment) -> lane holds O[q][d = nt*16 + g*4 + r]
// P therefore never leaves its lane: the 8 probabilities a lane computed for a 32-key sub-block are exactly its PV B-operand,
// provided V^T is laid out in LDS with the matching key permutation pos(kv) (see vt_pos below). Softmax statistics are
// reduced across the 4 lane groups with two xor-shuffles.
#include "common.h"
#include "kernels.h"

namespace {

// V^T LDS image: row d holds the block's keys in the slot order vt_pos(key), with its 8-slot groups ROTATED by d >> 3 (mod KVB/8).
// The transposing 2-byte stores of one wave instruction go to rows d = c*8 + i for 8-16 different chunks c: without the rotation
// all of them fall on the same bank (row stride x 8 = 0 mod 32 dwords), a 16-way conflict on every store.
__device__ __forceinline__ int vt_pos(int kv_local) {
    // key permutation inside each 32-key sub-block: MFMA k index g*8 + j  <->  key (j<4 ? g*4+j : 16+g*4+(j-4))
    int sub = kv_local >> 5, w = kv_local & 31;
    int t = w >> 4, x = w & 15;
    return (sub << 5) + ((x >> 2) << 3) + (x & 3) + (t << 2);
}

template <int DP, int DV, int NW, int KVB, bool DROP = false>
__global__ __launch_bounds__(NW * 64) void attn_fwd_kernel(AttnArgs p) {
    constexpr int NT = NW * 64;
    constexpr int KS_LD = DP + 8;
    constexpr int VT_LD = KVB + 8;
    constexpr int KCPR = DP / 8;        // 16B chunks per K row (padded width)
    constexpr int VCPR = DV / 8;
    constexpr int NKK = DP / 32;
    constexpr int NST = KVB / 16;       // S tiles per block
    constexpr int NSB = KVB / 32;       // 32-key sub-blocks per block
    constexpr int NDT = DV / 16;        // output d tiles
    __shared__ __attribute__((aligned(16))) bf16 Ks[KVB * KS_LD];
    __shared__ __attribute__((aligned(16))) bf16 Vt[DV * VT_LD];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, lq = lane & 15;
    const int b = blockIdx.z, h = blockIdx.y;
    const int kb = b / p.kv_bdiv;
    const int kh = h / (p.H / p.Hkv);

    int q_off = 0, k_off = 0, len_q = p.Lq, len_k = p.Lk;
    if (p.cu_q) { q_off = p.cu_q[b]; len_q = p.cu_q[b + 1] - q_off; }
    if (p.cu_k) { k_off = p.cu_k[kb]; len_k = p.cu_k[kb + 1] - k_off; }
    else if (p.k_len) len_k = min(p.k_len[kb], p.Lk);
    const int qt0 = (p.causal ? gridDim.x - 1 - blockIdx.x : blockIdx.x) * (NW * 16);   // causal: longest query tiles first
    if (qt0 >= len_q) return;

    const bf16* __restrict__ Q = reinterpret_cast<const bf16*>(p.Q) + (size_t)b * p.q_bs + (size_t)q_off * p.q_rs + (size_t)h * p.q_hs;
    const bf16* __restrict__ K = reinterpret_cast<const bf16*>(p.K) + (size_t)kb * p.k_bs + (size_t)k_off * p.k_rs + (size_t)kh * p.k_hs;
    const bf16* __restrict__ V = reinterpret_cast<const bf16*>(p.V) + (size_t)kb * p.v_bs + (size_t)k_off * p.v_rs + (size_t)kh * p.v_hs;
    bf16* __restrict__ O = reinterpret_cast<bf16*>(p.O) + (size_t)b * p.o_bs + (size_t)q_off * p.o_rs + (size_t)h * p.o_hs;

    const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    const int q_abs = qt0 + wave * 16 + lq;
    bf16x8 qf[NKK];
#pragma unroll
    for (int kk = 0; kk < NKK; ++kk) {
        int d = kk * 32 + g * 8;
        qf[kk] = (q_abs < len_q && d < p.D) ? *reinterpret_cast<const bf16x8*>(Q + (size_t)q_abs * p.q_rs + d) : zero8;
    }

    f32x4 acc_o[NDT];
#pragma unroll
    for (int i = 0; i < NDT; ++i) acc_o[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m_run = -INFINITY, l_run = 0.f;
    const float sc = p.scale * 1.4426950408889634f;  // exp2 domain
    const int causal_shift = len_k - len_q;

    int kv_end = len_k;
    if (p.causal) {
        int last_q = min(qt0 + NW * 16, len_q) - 1;
        kv_end = min(len_k, last_q + causal_shift + 1);
    }
    const int kv_begin = (p.kv_start / KVB) * KVB;

    // K / V blocks travel global -> registers -> LDS; the registers of block i+1 are requested right after block i is published,
    // so the global latency runs under the MFMAs of block i
    constexpr int NKI = (KVB * KCPR + NT - 1) / NT, NVI = (KVB * VCPR + NT - 1) / NT;
    bf16x8 kreg[NKI], vreg[NVI];
    auto fetch = [&](int kv0) {
#pragma unroll
        for (int u = 0; u < NKI; ++u) {
            const int q = tid + u * NT, row = q / KCPR, c = q % KCPR, kv = kv0 + row;
            kreg[u] = (q < KVB * KCPR && kv < len_k && c * 8 < p.D) ? *reinterpret_cast<const bf16x8*>(K + (size_t)kv * p.k_rs + c * 8) : zero8;
        }
#pragma unroll
        for (int u = 0; u < NVI; ++u) {
            const int q = tid + u * NT, row = q / VCPR, c = q % VCPR, kv = kv0 + row;
            vreg[u] = (q < KVB * VCPR && kv < len_k) ? *reinterpret_cast<const bf16x8*>(V + (size_t)kv * p.v_rs + c * 8) : zero8;
        }
    };
    if (kv_begin < kv_end) fetch(kv_begin);
    for (int kv0 = kv_begin; kv0 < kv_end; kv0 += KVB) {
        __syncthreads();
        // ---- publish K block (row-major, zero padded) and V block (transposed + permuted) in LDS
#pragma unroll
        for (int u = 0; u < NKI; ++u) {
            const int q = tid + u * NT, row = q / KCPR, c = q % KCPR;
            if (q < KVB * KCPR) *reinterpret_cast<bf16x8*>(&Ks[row * KS_LD + c * 8]) = kreg[u];
        }
#pragma unroll
        for (int u = 0; u < NVI; ++u) {
            const int q = tid + u * NT, row = q / VCPR, c = q % VCPR;
            if (q < KVB * VCPR) {
                const int pos = vt_pos(row);
#pragma unroll
                for (int i = 0; i < 8; ++i) Vt[(c * 8 + i) * VT_LD + ((pos + 8 * c) & (KVB - 1))] = vreg[u][i];   // rotated rows: see vt_pos
            }
        }
        __syncthreads();
        if (kv0 + KVB < kv_end) fetch(kv0 + KVB);

        // ---- S^T = K . Q^T
        f32x4 s[NST];
#pragma unroll
        for (int t = 0; t < NST; ++t) {
            s[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < NKK; ++kk) {
                bf16x8 kf = *reinterpret_cast<const bf16x8*>(&Ks[(t * 16 + lq) * KS_LD + kk * 32 + g * 8]);
                s[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[kk], s[t], 0, 0, 0);
            }
        }
        // ---- mask + online softmax (lane owns query q_abs, keys kv0 + t*16 + g*4 + r)
        float mx = -INFINITY;
#pragma unroll
        for (int t = 0; t < NST; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                int kv = kv0 + t * 16 + g * 4 + r;
                bool ok = (kv < len_k) && (kv >= p.kv_start) && (!p.causal || kv <= q_abs + causal_shift);
                float v = ok ? s[t][r] * sc : -INFINITY;
                s[t][r] = v;
                mx = fmaxf(mx, v);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 16));
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float m_new = fmaxf(m_run, mx);
        const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
        const float alpha = exp2f(m_run - m_use);
        float rs = 0.f;
#pragma unroll
        for (int t = 0; t < NST; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float e = exp2f(s[t][r] - m_use);
                s[t][r] = e;
                rs += e;
            }
        rs += __shfl_xor(rs, 16);
        rs += __shfl_xor(rs, 32);
        l_run = l_run * alpha + rs;
        m_run = m_new;
#pragma unroll
        for (int i = 0; i < NDT; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc_o[i][r] *= alpha;

        if constexpr (DROP) {   // training: attention-probability dropout (the row sum above stays un-dropped, as in nn.MultiheadAttention)
            const uint64_t row = ((uint64_t)((size_t)b * p.H + h) * p.Lq + q_abs) * p.Lk;
#pragma unroll
            for (int t = 0; t < NST; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    s[t][r] = ina_hash(p.drop_seed + (p.drop_salt ? *p.drop_salt : 0u), row + kv0 + t * 16 + g * 4 + r) >= p.drop_thresh ? s[t][r] * p.drop_scale : 0.f;
        }
        // ---- O^T += V^T . P^T
#pragma unroll
        for (int sb = 0; sb < NSB; ++sb) {
            bf16x8 pf;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                pf[r] = (bf16)s[2 * sb][r];
                pf[4 + r] = (bf16)s[2 * sb + 1][r];
            }
#pragma unroll
            for (int nt = 0; nt < NDT; ++nt) {
                bf16x8 vf = *reinterpret_cast<const bf16x8*>(&Vt[(nt * 16 + lq) * VT_LD + ((sb * 32 + g * 8 + 8 * ((nt * 16 + lq) >> 3)) & (KVB - 1))]);
                acc_o[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf, acc_o[nt], 0, 0, 0);
            }
        }
    }

    if (q_abs >= len_q) return;
    const float inv_l = (l_run > 0.f) ? 1.0f / l_run : 0.f;
    float gate = 1.0f;
    if (p.head_gate) gate = tanhf(p.head_gate[h]);
#pragma unroll
    for (int nt = 0; nt < NDT; ++nt) {
        int d = nt * 16 + g * 4;
        if (d >= p.D) continue;
        bf16* op = O + (size_t)q_abs * p.o_rs + d;
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = acc_o[nt][r] * inv_l * gate;
        if (p.accumulate) {
            bf16x4 old = *reinterpret_cast<const bf16x4*>(op);
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] += (float)old[r];
        }
        bf16x4 o = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
        *reinterpret_cast<bf16x4*>(op) = o;
    }
}

// ------------------------------------------------------------------------------------------------ decode (few query tokens, long KV)
// Token-by-token decoding has Lq = 1 (or 1 + N_QUERY for the latent queries): one (sequence, head) pair is a single 16-row MFMA tile
// with one live row, and 196 one-wave workgroups cannot stream the KV cache at HBM rate. Two changes:
//   * GQA packing: the G = H / Hkv query heads that share one KV head (and their Lq positions) become the ROWS of the tile
//     (row R -> head kh*G + R / Lq, position R % Lq), so the K/V of a KV head are read once instead of G times;
//   * split-KV: grid.x slices the keys in chunks of DEC_CHUNK; each one-wave workgroup writes an un-normalised partial
//     (O, running max m, running sum l) to a workspace, a second kernel merges the slices (flash-decoding).
constexpr int DEC_CHUNK = 64;

template <int DP, int DV>
__global__ __launch_bounds__(64) void attn_decode_split_kernel(AttnArgs p, float* __restrict__ ws, int nsplit, int rtiles) {
    constexpr int KVB = 64, KS_LD = DP + 8, VT_LD = KVB + 8, KCPR = DP / 8, VCPR = DV / 8, NKK = DP / 32, NST = KVB / 16, NSB = KVB / 32, NDT = DV / 16;
    __shared__ __attribute__((aligned(16))) bf16 Ks[KVB * KS_LD];
    __shared__ __attribute__((aligned(16))) bf16 Vt[DV * VT_LD];
    const int lane = threadIdx.x, g = lane >> 4, lq = lane & 15;
    const int split = blockIdx.x, kh = blockIdx.y / rtiles, rt = blockIdx.y % rtiles, b = blockIdx.z;
    const int G = p.H / p.Hkv;
    const int R = rt * 16 + lq;                    // packed row of this lane
    const bool live = R < G * p.Lq;
    const int head = kh * G + (live ? R / p.Lq : 0), qpos = live ? R % p.Lq : 0;
    int len_k = p.Lk;
    if (p.k_len) len_k = min(p.k_len[b], p.Lk);
    const int kv0 = split * DEC_CHUNK;
    float* out = ws + ((((size_t)b * p.Hkv + kh) * rtiles + rt) * nsplit + split) * (size_t)(16 * (DV + 2));
    const bf16* __restrict__ Q = reinterpret_cast<const bf16*>(p.Q) + (size_t)b * p.q_bs + (size_t)qpos * p.q_rs + (size_t)head * p.q_hs;
    const bf16* __restrict__ K = reinterpret_cast<const bf16*>(p.K) + (size_t)b * p.k_bs + (size_t)kh * p.k_hs;
    const bf16* __restrict__ V = reinterpret_cast<const bf16*>(p.V) + (size_t)b * p.v_bs + (size_t)kh * p.v_hs;
    const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    float m_run = -INFINITY, l_run = 0.f;
    f32x4 acc_o[NDT];
#pragma unroll
    for (int i = 0; i < NDT; ++i) acc_o[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (kv0 < len_k) {
        bf16x8 qf[NKK];
#pragma unroll
        for (int kk = 0; kk < NKK; ++kk) {
            int d = kk * 32 + g * 8;
            qf[kk] = (live && d < p.D) ? *reinterpret_cast<const bf16x8*>(Q + d) : zero8;
        }
        // the whole K / V chunk is requested before the first LDS store: one memory latency per workgroup instead of one per 16-byte
        // piece (this one-wave kernel has nothing else to hide it behind)
        constexpr int NKI = KVB * KCPR / 64, NVI = KVB * VCPR / 64;
        static_assert((KVB * KCPR) % 64 == 0 && (KVB * VCPR) % 64 == 0, "chunk pieces must divide evenly over the wave");
        bf16x8 kreg[NKI], vreg[NVI];
#pragma unroll
        for (int u = 0; u < NKI; ++u) {
            const int q = lane + u * 64, row = q / KCPR, c = q % KCPR, kv = kv0 + row;
            kreg[u] = (kv < len_k && c * 8 < p.D) ? *reinterpret_cast<const bf16x8*>(K + (size_t)kv * p.k_rs + c * 8) : zero8;
        }
#pragma unroll
        for (int u = 0; u < NVI; ++u) {
            const int q = lane + u * 64, row = q / VCPR, c = q % VCPR, kv = kv0 + row;
            vreg[u] = (kv < len_k) ? *reinterpret_cast<const bf16x8*>(V + (size_t)kv * p.v_rs + c * 8) : zero8;
        }
#pragma unroll
        for (int u = 0; u < NKI; ++u) {
            const int q = lane + u * 64, row = q / KCPR, c = q % KCPR;
            *reinterpret_cast<bf16x8*>(&Ks[row * KS_LD + c * 8]) = kreg[u];
        }
#pragma unroll
        for (int u = 0; u < NVI; ++u) {
            const int q = lane + u * 64, row = q / VCPR, c = q % VCPR;
            const int pos = vt_pos(row);
#pragma unroll
            for (int i = 0; i < 8; ++i) Vt[(c * 8 + i) * VT_LD + ((pos + 8 * c) & (KVB - 1))] = vreg[u][i];   // rotated rows: see vt_pos
        }
        __syncthreads();
        f32x4 s[NST];
#pragma unroll
        for (int t = 0; t < NST; ++t) {
            s[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < NKK; ++kk) {
                bf16x8 kf = *reinterpret_cast<const bf16x8*>(&Ks[(t * 16 + lq) * KS_LD + kk * 32 + g * 8]);
                s[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[kk], s[t], 0, 0, 0);
            }
        }
        const float sc = p.scale * 1.4426950408889634f;
        const int causal_shift = len_k - p.Lq;
        float mx = -INFINITY;
#pragma unroll
        for (int t = 0; t < NST; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                int kv = kv0 + t * 16 + g * 4 + r;
                bool ok = live && (kv < len_k) && (!p.causal || kv <= qpos + causal_shift);
                float v = ok ? s[t][r] * sc : -INFINITY;
                s[t][r] = v;
                mx = fmaxf(mx, v);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 16));
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        m_run = mx;
        const float m_use = (mx == -INFINITY) ? 0.f : mx;
        float rs = 0.f;
#pragma unroll
        for (int t = 0; t < NST; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float e = exp2f(s[t][r] - m_use);
                s[t][r] = e;
                rs += e;
            }
        rs += __shfl_xor(rs, 16);
        rs += __shfl_xor(rs, 32);
        l_run = rs;
#pragma unroll
        for (int sb = 0; sb < NSB; ++sb) {
            bf16x8 pf;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                pf[r] = (bf16)s[2 * sb][r];
                pf[4 + r] = (bf16)s[2 * sb + 1][r];
            }
#pragma unroll
            for (int nt = 0; nt < NDT; ++nt) {
                bf16x8 vf = *reinterpret_cast<const bf16x8*>(&Vt[(nt * 16 + lq) * VT_LD + ((sb * 32 + g * 8 + 8 * ((nt * 16 + lq) >> 3)) & (KVB - 1))]);
                acc_o[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf, acc_o[nt], 0, 0, 0);
            }
        }
    }
    // partial: row lq -> [DV] un-normalised O (exp2 domain), then m, l
    float* orow = out + (size_t)lq * (DV + 2);
#pragma unroll
    for (int nt = 0; nt < NDT; ++nt) *reinterpret_cast<f32x4*>(orow + nt * 16 + g * 4) = acc_o[nt];
    if (g == 0) { orow[DV] = m_run; orow[DV + 1] = l_run; }
}

template <int DV>
__global__ __launch_bounds__(64) void attn_decode_combine_kernel(AttnArgs p, const float* __restrict__ ws, int nsplit, int rtiles) {
    // one wave per packed row: lanes cover the DV columns (DV <= 128 -> 2 per lane)
    const int lane = threadIdx.x;
    const int R = blockIdx.x, kh = blockIdx.y, b = blockIdx.z;
    const int G = p.H / p.Hkv;
    const int rt = R / 16, lq = R % 16;
    const int head = kh * G + R / p.Lq, qpos = R % p.Lq;
    const float* base = ws + (((size_t)b * p.Hkv + kh) * rtiles + rt) * nsplit * (size_t)(16 * (DV + 2)) + (size_t)lq * (DV + 2);
    // slice statistics: lane s holds (m, l) of slice s (+64, ...): one load per lane instead of a serial chain over the slices
    const size_t sstride = (size_t)16 * (DV + 2);
    float m = -INFINITY;
    for (int s0 = 0; s0 < nsplit; s0 += 64) {
        const int s = s0 + lane;
        m = fmaxf(m, s < nsplit ? base[s * sstride + DV] : -INFINITY);
    }
    m = wave_max(m);
    const float m_use = (m == -INFINITY) ? 0.f : m;
    float l = 0.f, o0 = 0.f, o1 = 0.f;
    for (int s0 = 0; s0 < nsplit; s0 += 64) {
        const int s = s0 + lane;
        const float ms = s < nsplit ? base[s * sstride + DV] : -INFINITY;
        const float wl = (ms == -INFINITY) ? 0.f : exp2f(ms - m_use);
        l += s < nsplit ? base[s * sstride + DV + 1] * wl : 0.f;
        const int cnt = min(64, nsplit - s0);
#pragma unroll 8
        for (int j = 0; j < cnt; ++j) {
            const float w = __shfl(wl, j);
            const float* q = base + (size_t)(s0 + j) * sstride;
            if (lane < DV) o0 += q[lane] * w;
            if (lane + 64 < DV) o1 += q[lane + 64] * w;
        }
    }
    l = wave_sum(l);
    const float inv = l > 0.f ? 1.0f / l : 0.f;
    bf16* O = reinterpret_cast<bf16*>(p.O) + (size_t)b * p.o_bs + (size_t)qpos * p.o_rs + (size_t)head * p.o_hs;
    if (lane < p.D) O[lane] = (bf16)(o0 * inv);
    if (lane + 64 < p.D) O[lane + 64] = (bf16)(o1 * inv);
}

// One-launch variant for key axes of up to DEC_FUSED_MAX_CHUNKS chunks (Lk <= 1024: every decode / latent-query pass of the 4-frame prompt):
// ONE 4-wave workgroup per (sequence, KV head, row tile). Wave w walks the chunks w, w + 4, ... with a running (max, sum, O) in registers -
// the chunk body is the split kernel's, on a wave-private LDS image, the next chunk's K / V pieces requested before the current one is
// multiplied - then the four partials meet in LDS and the workgroup writes the normalised rows itself. 28 workgroups and one launch per
// layer instead of 420 + 196 one-wave workgroups in two launches: what the decode chain needs while it shares the CUs with System-1
// (a launch advances when a System-1 workgroup retires, profiles/r03d_step_breakdown.log).
constexpr int DEC_FUSED_WAVES = 4, DEC_FUSED_MAX_CHUNKS = 16;

template <int DP, int DV>
__global__ __launch_bounds__(DEC_FUSED_WAVES * 64) void attn_decode_fused_kernel(AttnArgs p, int nsplit, int rtiles) {
    constexpr int KVB = 64, KS_LD = DP + 8, VT_LD = KVB + 8, KCPR = DP / 8, VCPR = DV / 8, NKK = DP / 32, NST = KVB / 16, NSB = KVB / 32, NDT = DV / 16;
    constexpr int PER_WAVE = KVB * KS_LD + DV * VT_LD;          // bf16 elements of one wave's K image + V^T image
    constexpr int PLD = DV + 2;                                 // partial row: DV x O, m, l (f32)
    static_assert(16 * PLD * 4 <= KVB * KS_LD * 2, "the partial of a wave fits its K image");
    extern __shared__ __attribute__((aligned(16))) char dec_smem[];
    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, lq = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    bf16* Ks = reinterpret_cast<bf16*>(dec_smem) + wave * PER_WAVE;
    bf16* Vt = Ks + KVB * KS_LD;
    const int kh = blockIdx.x / rtiles, rt = blockIdx.x % rtiles, b = blockIdx.y;
    const int G = p.H / p.Hkv;
    const int R = rt * 16 + lq;                    // packed row of this lane
    const bool live = R < G * p.Lq;
    const int head = kh * G + (live ? R / p.Lq : 0), qpos = live ? R % p.Lq : 0;
    int len_k = p.Lk;
    if (p.k_len) len_k = min(p.k_len[b], p.Lk);
    const bf16* __restrict__ Q = reinterpret_cast<const bf16*>(p.Q) + (size_t)b * p.q_bs + (size_t)qpos * p.q_rs + (size_t)head * p.q_hs;
    const bf16* __restrict__ K = reinterpret_cast<const bf16*>(p.K) + (size_t)b * p.k_bs + (size_t)kh * p.k_hs;
    const bf16* __restrict__ V = reinterpret_cast<const bf16*>(p.V) + (size_t)b * p.v_bs + (size_t)kh * p.v_hs;
    const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    float m_run = -INFINITY, l_run = 0.f;
    f32x4 acc_o[NDT];
#pragma unroll
    for (int i = 0; i < NDT; ++i) acc_o[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 qf[NKK];
#pragma unroll
    for (int kk = 0; kk < NKK; ++kk) {
        const int d = kk * 32 + g * 8;
        qf[kk] = (live && d < p.D) ? *reinterpret_cast<const bf16x8*>(Q + d) : zero8;
    }
    constexpr int NKI = KVB * KCPR / 64, NVI = KVB * VCPR / 64;
    static_assert((KVB * KCPR) % 64 == 0 && (KVB * VCPR) % 64 == 0, "chunk pieces must divide evenly over the wave");
    bf16x8 kreg[NKI], vreg[NVI];
    auto fetch = [&](int kv0) {
#pragma unroll
        for (int u = 0; u < NKI; ++u) {
            const int q = lane + u * 64, row = q / KCPR, c = q % KCPR, kv = kv0 + row;
            kreg[u] = (kv < len_k && c * 8 < p.D) ? *reinterpret_cast<const bf16x8*>(K + (size_t)kv * p.k_rs + c * 8) : zero8;
        }
#pragma unroll
        for (int u = 0; u < NVI; ++u) {
            const int q = lane + u * 64, row = q / VCPR, c = q % VCPR, kv = kv0 + row;
            vreg[u] = (kv < len_k) ? *reinterpret_cast<const bf16x8*>(V + (size_t)kv * p.v_rs + c * 8) : zero8;
        }
    };
    const float sc = p.scale * 1.4426950408889634f;
    const int causal_shift = len_k - p.Lq;
    int split = wave;
    if (split * DEC_CHUNK < len_k) fetch(split * DEC_CHUNK);
    for (; split * DEC_CHUNK < len_k; split += DEC_FUSED_WAVES) {
        const int kv0 = split * DEC_CHUNK;
        // registers -> this wave's LDS images (the reads of the previous chunk were consumed by its MFMAs: the LDS queue of a wave is in order)
#pragma unroll
        for (int u = 0; u < NKI; ++u) {
            const int q = lane + u * 64, row = q / KCPR, c = q % KCPR;
            *reinterpret_cast<bf16x8*>(&Ks[row * KS_LD + c * 8]) = kreg[u];
        }
#pragma unroll
        for (int u = 0; u < NVI; ++u) {
            const int q = lane + u * 64, row = q / VCPR, c = q % VCPR;
            const int pos = vt_pos(row);
#pragma unroll
            for (int i = 0; i < 8; ++i) Vt[(c * 8 + i) * VT_LD + ((pos + 8 * c) & (KVB - 1))] = vreg[u][i];   // rotated rows: see vt_pos
        }
        if ((split + DEC_FUSED_WAVES) * DEC_CHUNK < len_k) fetch((split + DEC_FUSED_WAVES) * DEC_CHUNK);      // next chunk in flight under this one's math
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        f32x4 s[NST];
#pragma unroll
        for (int t = 0; t < NST; ++t) {
            s[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < NKK; ++kk) {
                bf16x8 kf = *reinterpret_cast<const bf16x8*>(&Ks[(t * 16 + lq) * KS_LD + kk * 32 + g * 8]);
                s[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[kk], s[t], 0, 0, 0);
            }
        }
        float mx = -INFINITY;
#pragma unroll
        for (int t = 0; t < NST; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int kv = kv0 + t * 16 + g * 4 + r;
                const bool ok = live && (kv < len_k) && (!p.causal || kv <= qpos + causal_shift);
                const float v = ok ? s[t][r] * sc : -INFINITY;
                s[t][r] = v;
                mx = fmaxf(mx, v);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 16));
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float m_new = fmaxf(m_run, mx);
        const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
        const float alpha = (m_run == -INFINITY) ? 0.f : exp2f(m_run - m_use);
        float rs = 0.f;
#pragma unroll
        for (int t = 0; t < NST; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float e = exp2f(s[t][r] - m_use);
                s[t][r] = e;
                rs += e;
            }
        rs += __shfl_xor(rs, 16);
        rs += __shfl_xor(rs, 32);
        l_run = l_run * alpha + rs;
        m_run = m_new;
#pragma unroll
        for (int nt = 0; nt < NDT; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc_o[nt][r] *= alpha;
#pragma unroll
        for (int sb = 0; sb < NSB; ++sb) {
            bf16x8 pf;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                pf[r] = (bf16)s[2 * sb][r];
                pf[4 + r] = (bf16)s[2 * sb + 1][r];
            }
#pragma unroll
            for (int nt = 0; nt < NDT; ++nt) {
                bf16x8 vf = *reinterpret_cast<const bf16x8*>(&Vt[(nt * 16 + lq) * VT_LD + ((sb * 32 + g * 8 + 8 * ((nt * 16 + lq) >> 3)) & (KVB - 1))]);
                acc_o[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf, acc_o[nt], 0, 0, 0);
            }
        }
    }
    // ---- the four partials meet in LDS (each in its wave's K image): row lq -> [DV] un-normalised O (exp2 domain), m, l
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    float* part = reinterpret_cast<float*>(Ks);
#pragma unroll
    for (int nt = 0; nt < NDT; ++nt) *reinterpret_cast<f32x4*>(part + lq * PLD + nt * 16 + g * 4) = acc_o[nt];
    if (g == 0) { part[lq * PLD + DV] = m_run; part[lq * PLD + DV + 1] = l_run; }
    __syncthreads();
    // 256 threads: row tid / 16, DV / 16 consecutive columns each
    {
        constexpr int CPT = DV / 16;
        const int row = tid >> 4, c0 = (tid & 15) * CPT;
        const int Rr = rt * 16 + row;
        if (Rr < G * p.Lq) {
            const float* pw[DEC_FUSED_WAVES];
            float ms[DEC_FUSED_WAVES], m = -INFINITY;
#pragma unroll
            for (int w = 0; w < DEC_FUSED_WAVES; ++w) {
                pw[w] = reinterpret_cast<const float*>(reinterpret_cast<const bf16*>(dec_smem) + w * PER_WAVE) + row * PLD;
                ms[w] = pw[w][DV];
                m = fmaxf(m, ms[w]);
            }
            const float m_use = (m == -INFINITY) ? 0.f : m;
            float l = 0.f, o[CPT];
#pragma unroll
            for (int c = 0; c < CPT; ++c) o[c] = 0.f;
#pragma unroll
            for (int w = 0; w < DEC_FUSED_WAVES; ++w) {
                const float wl = (ms[w] == -INFINITY) ? 0.f : exp2f(ms[w] - m_use);
                l += pw[w][DV + 1] * wl;
#pragma unroll
                for (int c = 0; c < CPT; ++c) o[c] += pw[w][c0 + c] * wl;
            }
            const float inv = l > 0.f ? 1.0f / l : 0.f;
            const int hd = kh * G + Rr / p.Lq, qp = Rr % p.Lq;
            bf16* O = reinterpret_cast<bf16*>(p.O) + (size_t)b * p.o_bs + (size_t)qp * p.o_rs + (size_t)hd * p.o_hs;
#pragma unroll
            for (int c = 0; c < CPT; ++c)
                if (c0 + c < p.D) O[c0 + c] = (bf16)(o[c] * inv);
        }
    }
}

// Round 5: the same one-launch decode attention with the key axis spread over EIGHT waves and a quarter of the LDS. The 4-wave kernel above
// keeps a K image and a V^T image per wave (35 KiB each, 143 KiB per workgroup): a workgroup needs an all but empty CU before it can start -
// inside the step, where System-1's workgroups hold 67-137 KiB per CU, every layer's attention launch waited for one - and each wave walks four
// chunks one after the other (23.6 us per layer for 13 MB of K / V). Here
//   * the K fragments of a chunk are loaded straight from global memory in MFMA operand layout (lane -> key row t * 16 + lq, 16 bytes of its
//     32-wide k slice: 64 contiguous bytes per row and instruction, the weight-streaming GEMMs' access pattern) - no K image, no K store pass;
//   * V still goes through a transposed LDS image, but in two halves of the head dim (64 dims x 64 keys = 9 KiB per wave);
//   * the next chunk's K and V are requested as soon as the current chunk's S = K Q^T MFMAs have consumed the K registers (V's registers were
//     freed by the LDS store pass): the prefetch needs no second register set;
//   * 8 waves x 9 KiB = 72 KiB per workgroup, two chunks per wave at Lk <= 1024.
// The chunk arithmetic (scores, running max / sum, P V) is the 4-wave kernel's; the eight partials meet in LDS in wave order.
constexpr int DEC8_WAVES = 8;

template <int DP, int DV>
__global__ __launch_bounds__(DEC8_WAVES * 64) void attn_decode_fused8_kernel(AttnArgs p, int nsplit, int rtiles) {
    constexpr int KVB = 64, VT_LD = KVB + 8, VCPR = DV / 8, NKK = DP / 32, NST = KVB / 16, NSB = KVB / 32, NDT = DV / 16, HDT = NDT / 2, DH = DV / 2;
    constexpr int PER_WAVE = DH * VT_LD;                        // bf16 elements of one wave's half V^T image
    constexpr int PLD = DV + 2;                                 // partial row: DV x O, m, l (f32)
    static_assert(16 * PLD * 4 <= PER_WAVE * 2, "the partial of a wave fits its V^T image");
    static_assert(NDT % 2 == 0 && VCPR % 2 == 0, "the head dim splits into two halves of whole 16-column tiles");
    extern __shared__ __attribute__((aligned(16))) char dec_smem[];
    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, lq = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    bf16* Vt = reinterpret_cast<bf16*>(dec_smem) + wave * PER_WAVE;
    const int kh = blockIdx.x / rtiles, rt = blockIdx.x % rtiles, b = blockIdx.y;
    const int G = p.H / p.Hkv;
    const int R = rt * 16 + lq;                    // packed row of this lane
    const bool live = R < G * p.Lq;
    const int head = kh * G + (live ? R / p.Lq : 0), qpos = live ? R % p.Lq : 0;
    int len_k = p.Lk;
    if (p.k_len) len_k = min(p.k_len[b], p.Lk);
    const bf16* __restrict__ Q = reinterpret_cast<const bf16*>(p.Q) + (size_t)b * p.q_bs + (size_t)qpos * p.q_rs + (size_t)head * p.q_hs;
    const bf16* __restrict__ K = reinterpret_cast<const bf16*>(p.K) + (size_t)b * p.k_bs + (size_t)kh * p.k_hs;
    const bf16* __restrict__ V = reinterpret_cast<const bf16*>(p.V) + (size_t)b * p.v_bs + (size_t)kh * p.v_hs;
    const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    float m_run = -INFINITY, l_run = 0.f;
    f32x4 acc_o[NDT];
#pragma unroll
    for (int i = 0; i < NDT; ++i) acc_o[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 qf[NKK];
#pragma unroll
    for (int kk = 0; kk < NKK; ++kk) {
        const int d = kk * 32 + g * 8;
        qf[kk] = (live && d < p.D) ? *reinterpret_cast<const bf16x8*>(Q + d) : zero8;
    }
    if constexpr (DP == 128 && DV == 128) {
        if (p.rope_cos) {
            // ---- rotary embedding + KV-cache append of the NEW tokens in this launch (ina_attn_args.rope_cos: the single-token passes of the
            // decoder - no rope launch between the q|k|v projection and the attention). The Lq query tokens are the last Lq keys: token r sits at
            // cache row len_k - Lq + r. Arithmetic and rounding of rope.hip's rope_kernel (bf16 in, fp32 rotation, bf16 out).
            // q: a lane's fragments kk and kk + 2 are the two halves of the same 8 dims - the rotation is lane-local
            if (live) {
                const float* cr = p.rope_cos + ((size_t)b * p.Lq + qpos) * DP + g * 8;
                const float* sr = p.rope_sin + ((size_t)b * p.Lq + qpos) * DP + g * 8;
#pragma unroll
                for (int kk = 0; kk < NKK / 2; ++kk) {
                    const bf16x8 lo = qf[kk], hi = qf[kk + NKK / 2];
                    bf16x8 olo, ohi;
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const int d = kk * 32 + i;
                        olo[i] = (bf16)ina_rope_lo((float)lo[i], (float)hi[i], cr[d], sr[d]);
                        ohi[i] = (bf16)ina_rope_hi((float)lo[i], (float)hi[i], cr[d + DP / 2], sr[d + DP / 2]);
                    }
                    qf[kk] = olo;
                    qf[kk + NKK / 2] = ohi;
                }
            }
            // k / v of kv head kh: even waves rotate a token's key row into the cache (8 lanes x 8 dims of each half), odd waves copy its value row
            for (int r = wave >> 1; r < p.Lq; r += DEC8_WAVES / 2) {
                const size_t src = (size_t)b * p.kn_bs + (size_t)r * p.kn_rs + (size_t)kh * p.kn_hs;
                const size_t row = (size_t)(len_k - p.Lq + r);
                if ((wave & 1) == 0) {
                    if (lane < 8) {
                        const bf16* kn = reinterpret_cast<const bf16*>(p.k_new) + src + lane * 8;
                        const bf16x8 lo = *reinterpret_cast<const bf16x8*>(kn), hi = *reinterpret_cast<const bf16x8*>(kn + DP / 2);
                        const float* cr = p.rope_cos + ((size_t)b * p.Lq + r) * DP + lane * 8;
                        const float* sr = p.rope_sin + ((size_t)b * p.Lq + r) * DP + lane * 8;
                        bf16x8 olo, ohi;
#pragma unroll
                        for (int i = 0; i < 8; ++i) {
                            olo[i] = (bf16)ina_rope_lo((float)lo[i], (float)hi[i], cr[i], sr[i]);
                            ohi[i] = (bf16)ina_rope_hi((float)lo[i], (float)hi[i], cr[i + DP / 2], sr[i + DP / 2]);
                        }
                        bf16* kd = const_cast<bf16*>(K) + row * p.k_rs + lane * 8;
                        *reinterpret_cast<bf16x8*>(kd) = olo;
                        *reinterpret_cast<bf16x8*>(kd + DP / 2) = ohi;
                    }
                } else if (lane < 16) {
                    *reinterpret_cast<bf16x8*>(const_cast<bf16*>(V) + row * p.v_rs + lane * 8) =
                        *reinterpret_cast<const bf16x8*>(reinterpret_cast<const bf16*>(p.v_new) + src + lane * 8);
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the appended rows have left this CU before any wave of the workgroup reads them back
            __syncthreads();
        }
    }
    constexpr int NVI = KVB * VCPR / 64;
    static_assert((KVB * VCPR) % 64 == 0, "chunk pieces must divide evenly over the wave");
    bf16x8 kf[NST][NKK], vreg[NVI];
    auto fetch = [&](int kv0) {
#pragma unroll
        for (int t = 0; t < NST; ++t) {
            const int kv = kv0 + t * 16 + lq;
            const bf16* kr = K + (size_t)(kv < len_k ? kv : len_k - 1) * p.k_rs + g * 8;      // (keys beyond len_k: a valid row, their scores are masked)
#pragma unroll
            for (int kk = 0; kk < NKK; ++kk) kf[t][kk] = (kk * 32 + g * 8 < p.D) ? *reinterpret_cast<const bf16x8*>(kr + kk * 32) : zero8;
        }
#pragma unroll
        for (int u = 0; u < NVI; ++u) {
            const int q = lane + u * 64, row = q / VCPR, c = q % VCPR, kv = kv0 + row;
            vreg[u] = (kv < len_k) ? *reinterpret_cast<const bf16x8*>(V + (size_t)kv * p.v_rs + c * 8) : zero8;
        }
    };
    const float sc = p.scale * 1.4426950408889634f;
    const int causal_shift = len_k - p.Lq;
    const int vc = lane % VCPR;                    // this lane's 8-dim chunk of a V row (the same for every piece u: 64 % VCPR == 0)
    int split = wave;
    if (split * DEC_CHUNK < len_k) fetch(split * DEC_CHUNK);
    for (; split * DEC_CHUNK < len_k; split += DEC8_WAVES) {
        const int kv0 = split * DEC_CHUNK;
        // ---- S^T = K Q^T from the register fragments
        f32x4 s[NST];
#pragma unroll
        for (int t = 0; t < NST; ++t) {
            s[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < NKK; ++kk) s[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[t][kk], qf[kk], s[t], 0, 0, 0);
        }
        // V registers of this chunk are parked in a second set only logically: the two half images are written one after the other below,
        // so the registers stay live until the second store pass; the NEXT chunk's V therefore goes to the same registers only after it
        bf16x8 vcur[NVI];
#pragma unroll
        for (int u = 0; u < NVI; ++u) vcur[u] = vreg[u];
        float mx = -INFINITY;
#pragma unroll
        for (int t = 0; t < NST; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int kv = kv0 + t * 16 + g * 4 + r;
                const bool ok = live && (kv < len_k) && (!p.causal || kv <= qpos + causal_shift);
                const float v = ok ? s[t][r] * sc : -INFINITY;
                s[t][r] = v;
                mx = fmaxf(mx, v);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 16));
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float m_new = fmaxf(m_run, mx);
        const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
        const float alpha = (m_run == -INFINITY) ? 0.f : exp2f(m_run - m_use);
        float rs = 0.f;
#pragma unroll
        for (int t = 0; t < NST; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float e = exp2f(s[t][r] - m_use);
                s[t][r] = e;
                rs += e;
            }
        rs += __shfl_xor(rs, 16);
        rs += __shfl_xor(rs, 32);
        l_run = l_run * alpha + rs;
        m_run = m_new;
#pragma unroll
        for (int nt = 0; nt < NDT; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc_o[nt][r] *= alpha;
        bf16x8 pf[NSB];
#pragma unroll
        for (int sb = 0; sb < NSB; ++sb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                pf[sb][r] = (bf16)s[2 * sb][r];
                pf[sb][4 + r] = (bf16)s[2 * sb + 1][r];
            }
        // ---- O^T += V^T P^T, one half of the head dim at a time through the wave's half image
#pragma unroll
        for (int hv = 0; hv < 2; ++hv) {
            // (the reads of the previous half / chunk were consumed by their MFMAs: the LDS queue of a wave is in order)
            if (vc / (VCPR / 2) == hv) {
#pragma unroll
                for (int u = 0; u < NVI; ++u) {
                    const int row = (lane + u * 64) / VCPR;
                    const int pos = vt_pos(row);
#pragma unroll
                    for (int i = 0; i < 8; ++i) Vt[((vc - hv * (VCPR / 2)) * 8 + i) * VT_LD + ((pos + 8 * vc) & (KVB - 1))] = vcur[u][i];   // rotated rows: see vt_pos
                }
            }
            if (hv == 1 && (split + DEC8_WAVES) * DEC_CHUNK < len_k) fetch((split + DEC8_WAVES) * DEC_CHUNK);   // next chunk in flight under this half's math
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int sb = 0; sb < NSB; ++sb)
#pragma unroll
                for (int n2 = 0; n2 < HDT; ++n2) {
                    const int dl = n2 * 16 + lq, d = hv * DH + dl;               // row of the half image / head dim
                    const bf16x8 vf = *reinterpret_cast<const bf16x8*>(&Vt[dl * VT_LD + ((sb * 32 + g * 8 + 8 * (d >> 3)) & (KVB - 1))]);
                    acc_o[hv * HDT + n2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf[sb], acc_o[hv * HDT + n2], 0, 0, 0);
                }
        }
    }
    // ---- the eight partials meet in LDS (each in its wave's image): row lq -> [DV] un-normalised O (exp2 domain), m, l
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    float* part = reinterpret_cast<float*>(Vt);
#pragma unroll
    for (int nt = 0; nt < NDT; ++nt) *reinterpret_cast<f32x4*>(part + lq * PLD + nt * 16 + g * 4) = acc_o[nt];
    if (g == 0) { part[lq * PLD + DV] = m_run; part[lq * PLD + DV + 1] = l_run; }
    __syncthreads();
    if (tid < 256) {        // 256 threads: row tid / 16, DV / 16 consecutive columns each
        constexpr int CPT = DV / 16;
        const int row = tid >> 4, c0 = (tid & 15) * CPT;
        const int Rr = rt * 16 + row;
        if (Rr < G * p.Lq) {
            float m = -INFINITY;
#pragma unroll
            for (int w = 0; w < DEC8_WAVES; ++w)
                m = fmaxf(m, (reinterpret_cast<const float*>(reinterpret_cast<const bf16*>(dec_smem) + w * PER_WAVE) + row * PLD)[DV]);
            const float m_use = (m == -INFINITY) ? 0.f : m;
            float l = 0.f, o[CPT];
#pragma unroll
            for (int c = 0; c < CPT; ++c) o[c] = 0.f;
#pragma unroll
            for (int w = 0; w < DEC8_WAVES; ++w) {
                const float* pw = reinterpret_cast<const float*>(reinterpret_cast<const bf16*>(dec_smem) + w * PER_WAVE) + row * PLD;
                const float ms = pw[DV];
                const float wl = (ms == -INFINITY) ? 0.f : exp2f(ms - m_use);
                l += pw[DV + 1] * wl;
#pragma unroll
                for (int c = 0; c < CPT; ++c) o[c] += pw[c0 + c] * wl;
            }
            const float inv = l > 0.f ? 1.0f / l : 0.f;
            const int hd = kh * G + Rr / p.Lq, qp = Rr % p.Lq;
            bf16* O = reinterpret_cast<bf16*>(p.O) + (size_t)b * p.o_bs + (size_t)qp * p.o_rs + (size_t)hd * p.o_hs;
#pragma unroll
            for (int c = 0; c < CPT; ++c)
                if (c0 + c < p.D) O[c0 + c] = (bf16)(o[c] * inv);
        }
    }
}

template <int DP, int DV>
int launch_decode(const AttnArgs& p, hipStream_t stream) {
    const int G = p.H / p.Hkv;
    const int rtiles = (G * p.Lq + 15) / 16;
    const int nsplit = (p.Lk + DEC_CHUNK - 1) / DEC_CHUNK;
    const double keys = p.causal ? 0.5 * ((double)p.Lk + (double)(p.Lk - p.Lq) + 1.0) : (double)p.Lk;
    InaProfScope prof(INA_PROF_ATTN, 4.0 * p.B * p.H * (double)p.Lq * keys * p.D,
                      2.0 * p.D * ((double)p.B * p.H * p.Lq * 2.0 + 2.0 * (double)p.B * p.Hkv * p.Lk), stream);
    if constexpr (DV == 128 && DP == 128) {
        // the 8-wave / 72 KiB kernel (round 5) for the decoder's d = 128 heads; ina_attn_args.kernel = 3 pins the 4-wave kernel below
        if (nsplit <= DEC_FUSED_MAX_CHUNKS && p.kernel != 1 && p.kernel != 3) {
            INA_REQUIRE(!p.rope_cos || (p.rope_sin && p.k_new && p.v_new && p.D == 128 && p.causal && p.kn_rs % 8 == 0 && p.kn_hs % 8 == 0 && p.kn_bs % 8 == 0 &&
                                        ((uintptr_t)p.k_new % 16) == 0 && ((uintptr_t)p.v_new % 16) == 0 && p.k_rs % 8 == 0 && p.v_rs % 8 == 0),
                        "attention(rope): needs rope_sin, k_new, v_new, d = 128, causal decode and 16-byte aligned rows");
            constexpr size_t LDS8 = (size_t)DEC8_WAVES * (DV / 2) * (64 + 8) * sizeof(bf16);
            auto kern8 = attn_decode_fused8_kernel<DP, DV>;
            static bool attr8_done = false;
            if (!attr8_done) {
                INA_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern8), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS8));
                attr8_done = true;
            }
            hipLaunchKernelGGL(kern8, dim3(p.Hkv * rtiles, p.B), dim3(DEC8_WAVES * 64), LDS8, stream, p, nsplit, rtiles);
            INA_HIP_CHECK(hipGetLastError());
            return 0;
        }
    }
    if (nsplit <= DEC_FUSED_MAX_CHUNKS && p.kernel != 1) {      // (ina_attn_args.kernel = 1 pins the two-launch split + combine pair: parity tests compare)
        constexpr size_t LDS = (size_t)DEC_FUSED_WAVES * (64 * (DP + 8) + DV * (64 + 8)) * sizeof(bf16);
        auto kern = attn_decode_fused_kernel<DP, DV>;
        static bool attr_done = false;
        if (!attr_done) {
            INA_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS));
            attr_done = true;
        }
        hipLaunchKernelGGL(kern, dim3(p.Hkv * rtiles, p.B), dim3(DEC_FUSED_WAVES * 64), LDS, stream, p, nsplit, rtiles);
        INA_HIP_CHECK(hipGetLastError());
        return 0;
    }
    const size_t bytes = (size_t)p.B * p.Hkv * rtiles * nsplit * 16 * (DV + 2) * sizeof(float);
    float* ws = nullptr;
    if (int rc = ina_workspace(1, bytes, stream, &ws)) return rc;
    hipLaunchKernelGGL((attn_decode_split_kernel<DP, DV>), dim3(nsplit, p.Hkv * rtiles, p.B), dim3(64), 0, stream, p, ws, nsplit, rtiles);
    hipLaunchKernelGGL((attn_decode_combine_kernel<DV>), dim3(G * p.Lq, p.Hkv, p.B), dim3(64), 0, stream, p, ws, nsplit, rtiles);
    INA_HIP_CHECK(hipGetLastError());
    return 0;
}

template <int DP, int DV>
int launch_d(const AttnArgs& p, hipStream_t stream) {
    // decode shape: few query tokens against a long dense KV (GQA-packed split-KV path)
    if (!p.cu_q && !p.cu_k && p.kv_bdiv == 1 && p.kv_start == 0 && !p.accumulate && !p.head_gate && !p.drop_thresh && p.Lk >= 256 &&
        (p.H / p.Hkv) * p.Lq <= 48 && p.Lq <= 8 && DV <= 128)
        return launch_decode<DP, DV>(p, stream);
    // short query sequences: fewer waves per workgroup; short key sequences: 32-key blocks
    // (two query tiles per wave - every K / V^T fragment read feeding two MFMAs - was measured and is slower: 238 VGPRs at d = 128
    //  leave one workgroup per CU, 351 vs 184 us on the LLM prefill shape; equal at d = 80, slower at d = 64)
    const bool small_k = p.Lk <= 48;
    int nw = (p.Lq <= 16) ? 1 : (p.Lq <= 32 ? 2 : 4);
    dim3 grid((p.Lq + nw * 16 - 1) / (nw * 16), p.H, p.B);
    // algorithmic FLOPs: QK^T + PV over the unmasked keys (dense bound Lq x Lk for varlen: cu_* lengths live on the device)
    const double keys = p.causal ? 0.5 * ((double)p.Lk + (double)(p.Lk - p.Lq) + 1.0) : (double)(p.Lk - p.kv_start);
    InaProfScope prof(INA_PROF_ATTN, 4.0 * p.B * p.H * (double)p.Lq * keys * p.D,
                      2.0 * p.D * ((double)p.B * p.H * p.Lq * 2.0 + 2.0 * (double)(p.B / p.kv_bdiv) * p.Hkv * p.Lk), stream);
    // long dense shapes: 32x32 MFMA kernel (attention_wide.hip); ina_attn_args.kernel pins one of the two kernels (tests compare them)
    INA_REQUIRE(p.kernel != 3, "attention: kernel = 3 (the four-wave one-launch decode kernel) outside the decode contract (few query rows, dense K / V, Lk >= 256)");
    if (p.kernel == 2) INA_REQUIRE(ina_attention_wide_contract(p), "attention: kernel = 2 (32-rows-per-wave) outside its contract (d 64 / 80 / 128, Lq, Lk >= 128, no gate / accumulate / dropout)");
    if (p.kernel == 2 || (p.kernel == 0 && ina_attention_wide_eligible(p)))
        return ina_launch_attention_wide(p, stream);
    if (p.drop_thresh) {   // training-only variant (d <= 64 heads of the nn.Transformer layers)
        if constexpr (DP <= 64) {
#define INA_ATTN_LAUNCH_DROP(NW_, KVB_) \
    hipLaunchKernelGGL((attn_fwd_kernel<DP, DV, NW_, KVB_, true>), grid, dim3(NW_ * 64), 0, stream, p)
            if (nw == 1) { if (small_k) INA_ATTN_LAUNCH_DROP(1, 32); else INA_ATTN_LAUNCH_DROP(1, 64); }
            else if (nw == 2) { if (small_k) INA_ATTN_LAUNCH_DROP(2, 32); else INA_ATTN_LAUNCH_DROP(2, 64); }
            else { if (small_k) INA_ATTN_LAUNCH_DROP(4, 32); else INA_ATTN_LAUNCH_DROP(4, 64); }
#undef INA_ATTN_LAUNCH_DROP
            INA_HIP_CHECK(hipGetLastError());
            return 0;
        } else {
            ina_set_error("attention: dropout is built for head dims <= 64 only");
            return -2;
        }
    }
#define INA_ATTN_LAUNCH(NW_, KVB_) \
    hipLaunchKernelGGL((attn_fwd_kernel<DP, DV, NW_, KVB_>), grid, dim3(NW_ * 64), 0, stream, p)
    if (nw == 1) { if (small_k) INA_ATTN_LAUNCH(1, 32); else INA_ATTN_LAUNCH(1, 64); }
    else if (nw == 2) { if (small_k) INA_ATTN_LAUNCH(2, 32); else INA_ATTN_LAUNCH(2, 64); }
    else { if (small_k) INA_ATTN_LAUNCH(4, 32); else INA_ATTN_LAUNCH(4, 64); }
#undef INA_ATTN_LAUNCH
    INA_HIP_CHECK(hipGetLastError());
    return 0;
}

}  // namespace

int ina_launch_attention(const AttnArgs& p_in, hipStream_t stream) {
    AttnArgs p = p_in;
    if (p.kv_bdiv <= 0) p.kv_bdiv = 1;
    INA_REQUIRE(p.B > 0 && p.H > 0 && p.Hkv > 0 && p.H % p.Hkv == 0, "attention: bad B/H/Hkv (%d,%d,%d)", p.B, p.H, p.Hkv);
    INA_REQUIRE(p.Lq > 0 && p.Lk > 0, "attention: empty sequence Lq=%d Lk=%d", p.Lq, p.Lk);
    INA_REQUIRE(p.q_rs % 8 == 0 && p.k_rs % 8 == 0 && p.v_rs % 8 == 0 && p.o_rs % 4 == 0 && p.q_hs % 8 == 0 && p.k_hs % 8 == 0 &&
                    p.v_hs % 8 == 0 && p.o_hs % 4 == 0 && p.q_bs % 8 == 0 && p.k_bs % 8 == 0 && p.v_bs % 8 == 0 && p.o_bs % 4 == 0,
                "attention: strides must keep 16-byte row alignment");
    INA_REQUIRE(((uintptr_t)p.Q % 16) == 0 && ((uintptr_t)p.K % 16) == 0 && ((uintptr_t)p.V % 16) == 0 && ((uintptr_t)p.O % 8) == 0,
                "attention: misaligned pointer");
    if (p.rope_cos) {
        // the rotary embedding + KV append of the new tokens exists in the one-launch decode kernel of the d = 128 heads only: refuse - never ignore - it elsewhere
        const bool decode = !p.cu_q && !p.cu_k && p.kv_bdiv == 1 && p.kv_start == 0 && !p.accumulate && !p.head_gate && !p.drop_thresh && p.Lk >= 256 &&
                            (p.H / p.Hkv) * p.Lq <= 48 && p.Lq <= 8;
        INA_REQUIRE(decode && p.D == 128 && (p.Lk + DEC_CHUNK - 1) / DEC_CHUNK <= DEC_FUSED_MAX_CHUNKS && p.kernel == 0,
                    "attention(rope): the fused rotary embedding + KV append needs the one-launch decode kernel (d = 128, 256 <= Lk <= %d, Lq <= 8, kernel 0): Lq=%d Lk=%d D=%d",
                    DEC_CHUNK * DEC_FUSED_MAX_CHUNKS, p.Lq, p.Lk, p.D);
    }
    switch (p.D) {
        case 48: return launch_d<64, 48>(p, stream);
        case 64: return launch_d<64, 64>(p, stream);
        case 80: return launch_d<96, 80>(p, stream);
        case 128: return launch_d<128, 128>(p, stream);
        default: ina_set_error("attention: unsupported head dim %d (48/64/80/128)", p.D); return -2;
    }
}
