// Attention backward for gfx950 (bf16 in/out, fp32 statistics and accumulation): the SFT step of SURVEY.md 8(f4) needs
// d(softmax(q k^T * scale) v) for DINOv2 (6 h x d64, 257 tokens), the MemoryEncoder / QFormer nn.MultiheadAttention layers
// (d64), the NextDiT self / cross attention (d64) and, for the latent-query rows of the frozen LLM, causal GQA d128.
// Reference: torch autograd of F.scaled_dot_product_attention at dinov2_layers/attention.py:49-62, nn.MultiheadAttention inside
// internvla_n1_arch.py:76-118, diffusers LuminaAttnProcessor2_0 (nextdit_traj.py:121-178), transformers Qwen2_5_VLAttention.
//
// Same swapped-operand MFMA scheme as the forward kernel (attention.hip): the "row side" (16 rows per wave, operand fragments in
// registers) meets the "column side" streamed through LDS in 64-row blocks, and a lane always owns row (l & 15) and the columns
// t*16 + (l >> 4)*4 + r of a block, so probabilities and dS never leave their lane before they are the B operand of the next MFMA:
//   MODE 0  rows = queries:  S = Q K^T, dP = dO V^T, dS = P (dP - delta) scale, dQ += dS K      (also writes lse / delta)
//   MODE 1  rows = keys:     S^T, dP^T the same way with the roles swapped,     dK += dS^T Q,  dV += P^T dO
// The transposed column-side images (K^T for dQ; Q^T and dO^T for dK / dV) use the forward kernel's key permutation vt_pos.
#include "common.h"
#include "kernels.h"

namespace {

__device__ __forceinline__ int vt_pos(int kv_local) {
    int sub = kv_local >> 5, w = kv_local & 31;
    int t = w >> 4, x = w & 15;
    return (sub << 5) + ((x >> 2) << 3) + (x & 3) + (t << 2);
}

constexpr int CB = 64;              // column-side block rows
constexpr int T_LD = CB + 8;        // row length of a transposed image

template <int DP, int NW, int MODE>
__global__ __launch_bounds__(NW * 64) void attn_bwd_kernel(AttnBwdArgs a) {
    constexpr int NT = NW * 64;
    constexpr int R_LD = DP + 8;        // row length of a row-major image
    constexpr int CPR = DP / 8;         // 16-byte chunks per row
    constexpr int NKK = DP / 32;
    constexpr int NST = CB / 16;
    constexpr int NSB = CB / 32;
    constexpr int NDT = DP / 16;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    bf16* Y1 = reinterpret_cast<bf16*>(smem_raw);   // row-major K (MODE 0) / Q (MODE 1)
    bf16* Y2 = Y1 + CB * R_LD;                      // row-major V / dO
    bf16* Y1t = Y2 + CB * R_LD;                     // transposed K / Q
    bf16* Y2t = Y1t + DP * T_LD;                    // transposed dO (MODE 1 only)
    float* st = reinterpret_cast<float*>(MODE == 1 ? Y2t + DP * T_LD : Y1t + DP * T_LD);   // MODE 1: lse[CB], delta[CB]

    const AttnArgs& p = a.f;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, lq = lane & 15;
    const int b = blockIdx.z, h = blockIdx.y;
    const int kb = b / (p.kv_bdiv > 0 ? p.kv_bdiv : 1);
    const int kh = h / (p.H / p.Hkv);
    int len_q = p.Lq, len_k = p.Lk;
    if (p.k_len) len_k = min(p.k_len[kb], p.Lk);
    const int causal_shift = len_k - len_q;
    const float sc = p.scale * 1.4426950408889634f;

    const bf16* __restrict__ Q = reinterpret_cast<const bf16*>(p.Q) + (size_t)b * p.q_bs + (size_t)h * p.q_hs;
    const bf16* __restrict__ K = reinterpret_cast<const bf16*>(p.K) + (size_t)kb * p.k_bs + (size_t)kh * p.k_hs;
    const bf16* __restrict__ V = reinterpret_cast<const bf16*>(p.V) + (size_t)kb * p.v_bs + (size_t)kh * p.v_hs;
    const bf16* __restrict__ O = reinterpret_cast<const bf16*>(p.O) + (size_t)b * p.o_bs + (size_t)h * p.o_hs;
    const bf16* __restrict__ dO = reinterpret_cast<const bf16*>(a.dO) + (size_t)b * p.o_bs + (size_t)h * p.o_hs;
    float* __restrict__ lse = a.lse + ((size_t)b * p.H + h) * p.Lq;
    float* __restrict__ dlt = a.delta + ((size_t)b * p.H + h) * p.Lq;
    const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};

    // column-side block loaders (global -> LDS)
    auto load_rows = [&](bf16* dst, const bf16* src, size_t rs, int r0, int len) {
        for (int q = tid; q < CB * CPR; q += NT) {
            const int row = q / CPR, c = q % CPR, r = r0 + row;
            bf16x8 v = (r < len && c * 8 < p.D) ? *reinterpret_cast<const bf16x8*>(src + (size_t)r * rs + c * 8) : zero8;
            *reinterpret_cast<bf16x8*>(&dst[row * R_LD + c * 8]) = v;
        }
    };
    auto load_transposed = [&](bf16* dst, const bf16* src, size_t rs, int r0, int len) {
        for (int q = tid; q < CB * CPR; q += NT) {
            const int row = q / CPR, c = q % CPR, r = r0 + row;
            bf16x8 v = (r < len && c * 8 < p.D) ? *reinterpret_cast<const bf16x8*>(src + (size_t)r * rs + c * 8) : zero8;
            const int pos = vt_pos(row);
#pragma unroll
            for (int i = 0; i < 8; ++i) dst[(c * 8 + i) * T_LD + ((pos + 8 * c) & (CB - 1))] = v[i];
        }
    };
    auto frag = [&](const bf16* img, int t, int kk) {   // A operand: rows t*16 + lq of a row-major image
        return *reinterpret_cast<const bf16x8*>(&img[(t * 16 + lq) * R_LD + kk * 32 + g * 8]);
    };
    auto tfrag = [&](const bf16* img, int nt, int sb) {   // A operand: rows d = nt*16 + lq of a transposed image, 32-column sub-block sb
        return *reinterpret_cast<const bf16x8*>(&img[(nt * 16 + lq) * T_LD + ((sb * 32 + g * 8 + 8 * ((nt * 16 + lq) >> 3)) & (CB - 1))]);
    };

    if constexpr (MODE == 0) {
        // few query rows against a long key axis (the latent-query rows of the LLM): the keys are split over gridDim.x / q-tiles workgroups.
        //   stage 1: every split writes its partial softmax statistics (m, l);  stage 2: combine them, dS / dQ of the split's keys, fp32
        //   atomics into dq32.  stage 0 (nsplit 1) = both passes over all keys in one launch, bf16 dQ stored directly.
        const int nsplit = a.nsplit > 1 ? a.nsplit : 1;
        const int qt0 = (blockIdx.x / nsplit) * (NW * 16), split = blockIdx.x % nsplit;
        if (qt0 >= len_q) return;
        const int q_abs = qt0 + wave * 16 + lq;
        const bool live = q_abs < len_q;
        bf16x8 qf[NKK], dof[NKK];
        float dl = 0.f;
#pragma unroll
        for (int kk = 0; kk < NKK; ++kk) {
            const int d = kk * 32 + g * 8;
            const bool ok = live && d < p.D;
            qf[kk] = ok ? *reinterpret_cast<const bf16x8*>(Q + (size_t)q_abs * p.q_rs + d) : zero8;
            dof[kk] = ok ? *reinterpret_cast<const bf16x8*>(dO + (size_t)q_abs * p.o_rs + d) : zero8;
            const bf16x8 of = ok ? *reinterpret_cast<const bf16x8*>(O + (size_t)q_abs * p.o_rs + d) : zero8;
#pragma unroll
            for (int i = 0; i < 8; ++i) dl += (float)dof[kk][i] * (float)of[i];
        }
        dl += __shfl_xor(dl, 16);
        dl += __shfl_xor(dl, 32);

        int kv_end = len_k;
        if (p.causal) kv_end = min(len_k, min(qt0 + NW * 16, len_q) + causal_shift);
        int ks = 0, ke = kv_end;
        if (nsplit > 1) {
            const int nblk = (kv_end + CB - 1) / CB, per = (nblk + nsplit - 1) / nsplit;
            ks = min(kv_end, split * per * CB);
            ke = min(kv_end, ks + per * CB);
        }
        float* pm = a.part + (((size_t)b * p.H + h) * nsplit) * 2 * p.Lq;   // [split][m | l][Lq]
        // ---- pass 1: softmax statistics
        float m_run = -INFINITY, l_run = 0.f;
        if (a.stage != 2) {
          for (int kv0 = ks; kv0 < ke; kv0 += CB) {
            __syncthreads();
            load_rows(Y1, K, p.k_rs, kv0, len_k);
            __syncthreads();
            float mx = -INFINITY;
            f32x4 s[NST];
#pragma unroll
            for (int t = 0; t < NST; ++t) {
                s[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kk = 0; kk < NKK; ++kk) s[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag(Y1, t, kk), qf[kk], s[t], 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int kv = kv0 + t * 16 + g * 4 + r;
                    const bool ok = kv < len_k && (!p.causal || kv <= q_abs + causal_shift);
                    s[t][r] = ok ? s[t][r] * sc : -INFINITY;
                    mx = fmaxf(mx, s[t][r]);
                }
            }
            mx = fmaxf(mx, __shfl_xor(mx, 16));
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            const float m_new = fmaxf(m_run, mx);
            const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
            float rs = 0.f;
#pragma unroll
            for (int t = 0; t < NST; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) rs += exp2f(s[t][r] - m_use);
            rs += __shfl_xor(rs, 16);
            rs += __shfl_xor(rs, 32);
            l_run = l_run * exp2f(m_run - m_use) + rs;
            m_run = m_new;
          }
        }
        if (a.stage == 1) {
            if (live && g == 0) {
                pm[(size_t)split * 2 * p.Lq + q_abs] = m_run;
                pm[(size_t)split * 2 * p.Lq + p.Lq + q_abs] = l_run;
                if (split == 0) dlt[q_abs] = dl;
            }
            return;
        }
        if (a.stage == 2) {
            for (int sidx = 0; sidx < nsplit; ++sidx) {
                const float m2 = live ? pm[(size_t)sidx * 2 * p.Lq + q_abs] : -INFINITY;
                const float l2 = live ? pm[(size_t)sidx * 2 * p.Lq + p.Lq + q_abs] : 0.f;
                const float m_new = fmaxf(m_run, m2);
                const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
                l_run = l_run * exp2f(m_run - m_use) + l2 * exp2f(m2 - m_use);
                m_run = m_new;
            }
        }
        const float lse2 = (l_run > 0.f) ? m_run + log2f(l_run) : INFINITY;
        if (live && g == 0 && split == 0) { lse[q_abs] = lse2; dlt[q_abs] = dl; }

        // ---- pass 2: dQ
        f32x4 acc[NDT];
#pragma unroll
        for (int i = 0; i < NDT; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int kv0 = ks; kv0 < ke; kv0 += CB) {
            __syncthreads();
            load_rows(Y1, K, p.k_rs, kv0, len_k);
            load_rows(Y2, V, p.v_rs, kv0, len_k);
            load_transposed(Y1t, K, p.k_rs, kv0, len_k);
            __syncthreads();
            f32x4 s[NST], dp[NST];
#pragma unroll
            for (int t = 0; t < NST; ++t) {
                s[t] = f32x4{0.f, 0.f, 0.f, 0.f};
                dp[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kk = 0; kk < NKK; ++kk) {
                    s[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag(Y1, t, kk), qf[kk], s[t], 0, 0, 0);
                    dp[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag(Y2, t, kk), dof[kk], dp[t], 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int kv = kv0 + t * 16 + g * 4 + r;
                    const bool ok = kv < len_k && (!p.causal || kv <= q_abs + causal_shift);
                    const float pr = ok ? exp2f(s[t][r] * sc - lse2) : 0.f;
                    float dpm = dp[t][r];
                    if (p.drop_thresh)   // dropout on P: dL/dP = mask / (1 - p) * (dO . v)
                        dpm = ina_hash(p.drop_seed + (p.drop_salt ? *p.drop_salt : 0u), ((uint64_t)((size_t)b * p.H + h) * p.Lq + q_abs) * p.Lk + kv) >= p.drop_thresh ? dpm * p.drop_scale : 0.f;
                    s[t][r] = pr * (dpm - dl) * p.scale;   // dS
                }
            }
#pragma unroll
            for (int sb = 0; sb < NSB; ++sb) {
                bf16x8 df;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    df[r] = (bf16)s[2 * sb][r];
                    df[4 + r] = (bf16)s[2 * sb + 1][r];
                }
#pragma unroll
                for (int nt = 0; nt < NDT; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tfrag(Y1t, nt, sb), df, acc[nt], 0, 0, 0);
            }
        }
        if (!live) return;
        if (nsplit > 1) {
            float* dq32 = a.dq32 + (((size_t)b * p.Lq + q_abs) * p.H + h) * p.D;      // dense f32 [B, Lq, H, D]
#pragma unroll
            for (int nt = 0; nt < NDT; ++nt) {
                const int d = nt * 16 + g * 4;
                if (d >= p.D) continue;
#pragma unroll
                for (int r = 0; r < 4; ++r) atomicAdd(dq32 + d + r, acc[nt][r]);
            }
            return;
        }
        bf16* dQ = reinterpret_cast<bf16*>(a.dQ) + (size_t)b * a.dq_bs + (size_t)h * a.dq_hs + (size_t)q_abs * a.dq_rs;
#pragma unroll
        for (int nt = 0; nt < NDT; ++nt) {
            const int d = nt * 16 + g * 4;
            if (d >= p.D) continue;
            bf16x4 o = {(bf16)acc[nt][0], (bf16)acc[nt][1], (bf16)acc[nt][2], (bf16)acc[nt][3]};
            *reinterpret_cast<bf16x4*>(dQ + d) = o;
        }
    } else {
        const int row0 = a.kv_row0 < 0 ? max(0, len_k - len_q) : a.kv_row0;   // -1: the last Lq keys of every sequence (ragged k_len)
        const int kt0 = row0 + blockIdx.x * (NW * 16);
        if (kt0 >= len_k) return;
        const int k_abs = kt0 + wave * 16 + lq;
        const bool live = k_abs < len_k;
        bf16x8 kf[NKK], vf[NKK];
#pragma unroll
        for (int kk = 0; kk < NKK; ++kk) {
            const int d = kk * 32 + g * 8;
            const bool ok = live && d < p.D;
            kf[kk] = ok ? *reinterpret_cast<const bf16x8*>(K + (size_t)k_abs * p.k_rs + d) : zero8;
            vf[kk] = ok ? *reinterpret_cast<const bf16x8*>(V + (size_t)k_abs * p.v_rs + d) : zero8;
        }
        f32x4 acck[NDT], accv[NDT];
#pragma unroll
        for (int i = 0; i < NDT; ++i) { acck[i] = f32x4{0.f, 0.f, 0.f, 0.f}; accv[i] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        int q_begin = 0;
        if (p.causal) q_begin = max(0, kt0 - causal_shift) / CB * CB;   // queries before key kt0 - shift never see this tile
        for (int q0 = q_begin; q0 < len_q; q0 += CB) {
            __syncthreads();
            load_rows(Y1, Q, p.q_rs, q0, len_q);
            load_rows(Y2, dO, p.o_rs, q0, len_q);
            load_transposed(Y1t, Q, p.q_rs, q0, len_q);
            load_transposed(Y2t, dO, p.o_rs, q0, len_q);
            if (tid < CB) {
                const int q = q0 + tid;
                st[tid] = q < len_q ? lse[q] : INFINITY;
                st[CB + tid] = q < len_q ? dlt[q] : 0.f;
            }
            __syncthreads();
            f32x4 s[NST], dp[NST];
#pragma unroll
            for (int t = 0; t < NST; ++t) {
                s[t] = f32x4{0.f, 0.f, 0.f, 0.f};
                dp[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kk = 0; kk < NKK; ++kk) {
                    s[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag(Y1, t, kk), kf[kk], s[t], 0, 0, 0);
                    dp[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag(Y2, t, kk), vf[kk], dp[t], 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int ql = t * 16 + g * 4 + r, q = q0 + ql;
                    const bool ok = live && q < len_q && (!p.causal || k_abs <= q + causal_shift);
                    const float pr = ok ? exp2f(s[t][r] * sc - st[ql]) : 0.f;
                    float m = 1.f;
                    if (p.drop_thresh) m = ina_hash(p.drop_seed + (p.drop_salt ? *p.drop_salt : 0u), ((uint64_t)((size_t)b * p.H + h) * p.Lq + q) * p.Lk + k_abs) >= p.drop_thresh ? p.drop_scale : 0.f;
                    dp[t][r] = pr * (dp[t][r] * m - st[CB + ql]) * p.scale;   // dS
                    s[t][r] = pr * m;                                          // dropped P (what multiplied V in the forward pass)
                }
            }
#pragma unroll
            for (int sb = 0; sb < NSB; ++sb) {
                bf16x8 pf, df;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    pf[r] = (bf16)s[2 * sb][r];
                    pf[4 + r] = (bf16)s[2 * sb + 1][r];
                    df[r] = (bf16)dp[2 * sb][r];
                    df[4 + r] = (bf16)dp[2 * sb + 1][r];
                }
#pragma unroll
                for (int nt = 0; nt < NDT; ++nt) {
                    acck[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tfrag(Y1t, nt, sb), df, acck[nt], 0, 0, 0);
                    accv[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tfrag(Y2t, nt, sb), pf, accv[nt], 0, 0, 0);
                }
            }
        }
        if (!live) return;
        const size_t off = (size_t)b * a.dkv_bs + (size_t)h * a.dkv_hs + (size_t)(k_abs - row0) * a.dkv_rs;
        bf16* dK = reinterpret_cast<bf16*>(a.dK) + off;
        bf16* dV = reinterpret_cast<bf16*>(a.dV) + off;
#pragma unroll
        for (int nt = 0; nt < NDT; ++nt) {
            const int d = nt * 16 + g * 4;
            if (d >= p.D) continue;
            bf16x4 ok_ = {(bf16)acck[nt][0], (bf16)acck[nt][1], (bf16)acck[nt][2], (bf16)acck[nt][3]};
            bf16x4 ov_ = {(bf16)accv[nt][0], (bf16)accv[nt][1], (bf16)accv[nt][2], (bf16)accv[nt][3]};
            *reinterpret_cast<bf16x4*>(dK + d) = ok_;
            *reinterpret_cast<bf16x4*>(dV + d) = ov_;
        }
    }
}

template <int DP, int NW, int MODE>
int launch_bwd(const AttnBwdArgs& a_in, int rows, hipStream_t stream) {
    AttnBwdArgs a = a_in;
    const int nsplit = (MODE == 0 && a.nsplit > 1) ? a.nsplit : 1;
    if (MODE == 0) a.nsplit = nsplit;
    constexpr size_t lds = (size_t)(2 * CB * (DP + 8) + (MODE == 1 ? 2 : 1) * DP * T_LD) * sizeof(bf16) + (MODE == 1 ? 2 * CB * sizeof(float) : 0);
    auto kern = attn_bwd_kernel<DP, NW, MODE>;
    static bool attr_done = false;
    if (!attr_done) {
        INA_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_done = true;
    }
    dim3 grid(((rows + NW * 16 - 1) / (NW * 16)) * nsplit, a.f.H, a.f.B);
    const double fl = 2.0 * a.f.B * a.f.H * (double)a.f.Lq * a.f.Lk * DP * 4.0;
    InaProfScope prof(INA_PROF_ATTN, fl, 0.0, stream);
    if (nsplit > 1) {
        a.stage = 1;
        hipLaunchKernelGGL(kern, grid, dim3(NW * 64), lds, stream, a);
        a.stage = 2;
        hipLaunchKernelGGL(kern, grid, dim3(NW * 64), lds, stream, a);
    } else {
        a.stage = 0;
        hipLaunchKernelGGL(kern, grid, dim3(NW * 64), lds, stream, a);
    }
    INA_HIP_CHECK(hipGetLastError());
    return 0;
}

}  // namespace

int ina_launch_attention_bwd(const AttnBwdArgs& a, hipStream_t stream) {
    const AttnArgs& p = a.f;
    INA_REQUIRE(p.Q && p.K && p.V && p.O && a.dO && a.lse && a.delta, "attention_bwd: null tensor");
    INA_REQUIRE(!p.cu_q && !p.cu_k, "attention_bwd: packed (varlen) sequences are not supported");
    INA_REQUIRE(!p.head_gate && p.kv_start == 0 && !p.accumulate, "attention_bwd: head_gate / kv_start / accumulate are forward-only");
    INA_REQUIRE(!p.drop_thresh || a.nsplit <= 1, "attention_bwd: dropout with key splits is not built");
    INA_REQUIRE(p.D % 8 == 0 && p.D <= 128, "attention_bwd: head dim %d unsupported (multiple of 8, <= 128)", p.D);
    INA_REQUIRE(p.H % p.Hkv == 0, "attention_bwd: H %d not a multiple of Hkv %d", p.H, p.Hkv);
    INA_REQUIRE(a.kv_row0 >= -1 && a.kv_row0 <= p.Lk, "attention_bwd: kv_row0 %d out of range (-1 = the last Lq keys of every sequence)", a.kv_row0);
    INA_REQUIRE(a.nsplit <= 1 || (a.part && a.dq32), "attention_bwd: key splits need the `part` statistics scratch and the zeroed f32 `dq32` accumulator");
    INA_REQUIRE(a.nsplit <= 64, "attention_bwd: at most 64 key splits");
    const bool d64 = p.D <= 64;
    const bool small_q = p.Lq <= 32;
    if (a.dQ) {
        int rc = d64 ? (small_q ? launch_bwd<64, 2, 0>(a, p.Lq, stream) : launch_bwd<64, 4, 0>(a, p.Lq, stream))
                     : (small_q ? launch_bwd<128, 2, 0>(a, p.Lq, stream) : launch_bwd<128, 4, 0>(a, p.Lq, stream));
        if (rc) return rc;
    }
    if (a.dK) {
        INA_REQUIRE(a.dV != nullptr, "attention_bwd: dK without dV");
        INA_REQUIRE(a.dQ != nullptr, "attention_bwd: dK / dV need the lse / delta statistics written by the dQ pass of the same call");
        const int rows = a.kv_row0 < 0 ? min(p.Lq, p.Lk) : p.Lk - a.kv_row0;
        if (rows > 0) {
            const bool small_k = rows <= 32;
            int rc = d64 ? (small_k ? launch_bwd<64, 2, 1>(a, rows, stream) : launch_bwd<64, 4, 1>(a, rows, stream))
                         : (small_k ? launch_bwd<128, 2, 1>(a, rows, stream) : launch_bwd<128, 4, 1>(a, rows, stream));
            if (rc) return rc;
        }
    }
    return 0;
}
