// Attention forward for the long dense shapes of the path (LLM prefill: 920 tokens, GQA 28 / 4 heads x d128, causal; Qwen2.5-VL ViT
// full-attention blocks: 784 tokens x 16 heads x d80; DINOv2: 257 tokens x 6 heads x d64) on 32x32x16 MFMAs.
// Same contract as attn_fwd_kernel (attention.hip; reference call sites listed there) minus gate / accumulate / dropout; the launcher
// in attention.hip routes eligible calls here.
//
// One wave owns 32 query rows, a workgroup of NW waves shares each 64-key K / V block. Both products are issued swapped:
//   S^T[key][q] = K . Q^T    A = K fragment from LDS (ds_read_b128, rows padded by 16 B), B = Q fragment (registers, loaded once)
//   O^T[d][q]   = V^T . P^T  A = V^T fragment, B = P fragment
// so lane l always owns query row (l & 31) and, with hi = l >> 5, the keys crow(r, hi) = (r & 3) + 8 (r >> 2) + 4 hi of every 32-key
// tile: a row of P is split over just two lanes (one cross-lane max per block, the row sum is merged once at the end) and the 8
// probabilities a lane holds for a 16-key slice ARE its PV B operand - P never leaves its registers. The matching A operand
// (8 keys of one d column) comes out of a row-major V image with two ds_read_b64_tr_b16 (the hardware's 4 x 16 transpose read), which
// replaces the transposing 2-byte LDS stores of the 16-row kernel. 32 query rows per K / V^T fragment halve the LDS bytes per FLOP.
// K / V blocks travel global -> registers -> LDS into a 2-deep ring: the loads of block i + 1 are in flight under the MFMAs of
// block i, one barrier per block. The output tile leaves through LDS as whole rows.
#include "common.h"
#include "kernels.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

template <int D, int RING = 2>
struct WideCfg {
    static constexpr int KVB = 64;
    static constexpr int KLD = D + 8;                 // K row stride (elements): 16 rows of one 16-byte column hit 16 different bank quads
    static constexpr int NDT = (D + 31) / 32;         // 32-wide output column tiles (d 80: the third one is half padding)
    static constexpr int VLD = (D <= 96) ? 96 : 160;  // V row stride = 64 B (mod 256 B): the 4 rows of one transpose read tile the 64 banks
    static constexpr int NKS = D / 16;
    static constexpr int K_ELEMS = KVB * KLD, V_ELEMS = KVB * VLD;
    static constexpr size_t LDS = (size_t)RING * (K_ELEMS + V_ELEMS) * sizeof(bf16);   // RING = 1: single-block calls (window attention)
};

__device__ __forceinline__ bf16x8 tr_pair(const bf16* a0, const bf16* a1) {
    // two 4-key x 16-column transpose reads -> the 8 keys (MFMA k order) of this lane's d column
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a0));
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a1));
    s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
}

// Deferred running maximum: a row keeps its maximum until it grows by more than 2^DEFER_THR (P <= 2^DEFER_THR instead of <= 1; the
// accumulator rescale is skipped while no row of the wave moves). 3-5 % of the kernel; adopted in round 4 behind the 64-env distributional
// parity tests (the exact-maximum variant measured the same error distribution on all three System-1 heads, profiles/r04a_ab_attn.log).
constexpr float DEFER_THR = 6.0f;

template <int D, int NW, int RING = 2>
__global__ __launch_bounds__(NW * 64, 2) void attn_fwd_wide_kernel(AttnArgs p) {
    using C = WideCfg<D, RING>;
    constexpr int NT = NW * 64, KVB = C::KVB, KLD = C::KLD, VLD = C::VLD, NDT = C::NDT, NKS = C::NKS;
    constexpr int QT = NW * 32;
    constexpr int KCH = D / 8;                        // 16-byte pieces per K / V row
    constexpr int OLD = D + 8;
    static_assert((size_t)QT * OLD * sizeof(bf16) <= C::LDS, "output staging must fit in the K / V ring");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    bf16* Ks = reinterpret_cast<bf16*>(smem_raw);
    bf16* Vs = Ks + RING * C::K_ELEMS;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int hi = lane >> 5, lq = lane & 31;
    // 1-D grid. Tiles are aligned to the END of the query sequence and counted from it (tile t = rows len_q - (t + 1) QT ...): the
    // partial tile of a sequence is its FIRST rows. Causal: tile slowest - every sequence's tile 0 (the longest) first, then every
    // tile 1, ...: longest-first over the whole grid, the tail of the launch is made of the short tiles (108.8 -> 71.0 us on the LLM
    // prefill shape together with the end alignment). Not causal (equal tiles): tile fastest, so that the tiles of one (sequence, head)
    // run together and share its K / V in L2 (the tile-slowest order costs 15 % there).
    const int nbh = p.H * p.B, ntile = gridDim.x / nbh;
    const int tile = p.causal ? blockIdx.x / nbh : blockIdx.x % ntile;
    const int bh = p.causal ? blockIdx.x % nbh : blockIdx.x / ntile;
    const int b = bh / p.H, h = bh % p.H;
    const int kb = b / p.kv_bdiv;
    const int kh = h / (p.H / p.Hkv);

    int q_off = 0, k_off = 0, len_q = p.Lq, len_k = p.Lk;
    if (p.cu_q) { q_off = p.cu_q[b]; len_q = p.cu_q[b + 1] - q_off; }
    if (p.cu_k) { k_off = p.cu_k[kb]; len_k = p.cu_k[kb + 1] - k_off; }
    else if (p.k_len) len_k = min(p.k_len[kb], p.Lk);
    const int qt0 = len_q - (tile + 1) * QT;         // may be negative: rows < 0 do not exist
    if (qt0 + QT <= 0) return;

    const bf16* __restrict__ Q = reinterpret_cast<const bf16*>(p.Q) + (size_t)b * p.q_bs + (size_t)q_off * p.q_rs + (size_t)h * p.q_hs;
    const bf16* __restrict__ K = reinterpret_cast<const bf16*>(p.K) + (size_t)kb * p.k_bs + (size_t)k_off * p.k_rs + (size_t)kh * p.k_hs;
    const bf16* __restrict__ V = reinterpret_cast<const bf16*>(p.V) + (size_t)kb * p.v_bs + (size_t)k_off * p.v_rs + (size_t)kh * p.v_hs;
    bf16* __restrict__ O = reinterpret_cast<bf16*>(p.O) + (size_t)b * p.o_bs + (size_t)q_off * p.o_rs + (size_t)h * p.o_hs;

    const int wq0 = qt0 + wave * 32;                 // first query row of this wave
    const int q_abs = wq0 + lq;
    bf16x8 qf[NKS];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks)
        qf[ks] = *reinterpret_cast<const bf16x8*>(Q + (size_t)max(q_abs, 0) * p.q_rs + ks * 16 + hi * 8);   // rows < 0: never stored

    f32x16 acc[NDT];
#pragma unroll
    for (int i = 0; i < NDT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    const float sc = p.scale * 1.4426950408889634f;  // exp2 domain
    const int causal_shift = len_k - len_q;

    int kv_end = len_k;                               // workgroup: keys any of its rows can see
    if (p.causal) kv_end = min(len_k, min(qt0 + QT, len_q) - 1 + causal_shift + 1);
    const int kv_begin = (p.kv_start / KVB) * KVB;
    // wave: its own rows' bound (blocks past it are skipped, the wave still takes part in the staging and the barriers)
    const bool wave_live = wq0 + 32 > 0;
    int wave_kv_end = kv_end;
    if (p.causal) wave_kv_end = min(kv_end, min(wq0 + 32, len_q) - 1 + causal_shift + 1);

    // staging (global -> registers -> LDS): thread = one 16-byte column st_c of the image rows st_r + u * RPI. Byte offsets are 32-bit
    // (checked by the launcher) with the block's row term on the scalar unit; rows past the last key are CLAMPED to the last key
    // instead of zero-filled (no divergent loads: such keys carry P = 0 exactly). The padding columns of the V image (d 80 -> 96) are
    // never written: they only feed output rows that are never stored.
    constexpr int RPI = NT / KCH, NPASS = (KVB + RPI - 1) / RPI;
    constexpr bool ROW_CLAMP = NPASS * RPI > KVB || RPI * KCH < NT;
    const int st_c = tid % KCH, st_r = tid / KCH;
    const uint32_t k_rs_b = (uint32_t)p.k_rs * 2u, v_rs_b = (uint32_t)p.v_rs * 2u;
    const uint32_t k_lim = (uint32_t)(len_k - 1) * k_rs_b + st_c * 16, v_lim = (uint32_t)(len_k - 1) * v_rs_b + st_c * 16;
    const uint32_t k_o = st_r * k_rs_b + st_c * 16, v_o = st_r * v_rs_b + st_c * 16;
    const char* __restrict__ Kc = reinterpret_cast<const char*>(K);
    const char* __restrict__ Vc = reinterpret_cast<const char*>(V);
    bf16x8 kreg[NPASS], vreg[NPASS];
    auto fetch = [&](int kv0) {
#pragma unroll
        for (int u = 0; u < NPASS; ++u) {
            uint32_t ko, vo;
            if constexpr (ROW_CLAMP) {
                const uint32_t r = (uint32_t)(kv0 + min(st_r + u * RPI, KVB - 1));
                ko = r * k_rs_b + st_c * 16;
                vo = r * v_rs_b + st_c * 16;
            } else {
                ko = k_o + (uint32_t)(kv0 + u * RPI) * k_rs_b;
                vo = v_o + (uint32_t)(kv0 + u * RPI) * v_rs_b;
            }
            kreg[u] = *reinterpret_cast<const bf16x8*>(Kc + min(ko, k_lim));
            vreg[u] = *reinterpret_cast<const bf16x8*>(Vc + min(vo, v_lim));
        }
    };
    auto publish = [&](int buf) {
        bf16* Kb = Ks + buf * C::K_ELEMS;
        bf16* Vb = Vs + buf * C::V_ELEMS;
#pragma unroll
        for (int u = 0; u < NPASS; ++u) {
            int r = st_r + u * RPI;
            if constexpr (ROW_CLAMP) r = min(r, KVB - 1);   // duplicates rewrite the same bytes
            *reinterpret_cast<bf16x8*>(&Kb[r * KLD + st_c * 8]) = kreg[u];
            *reinterpret_cast<bf16x8*>(&Vb[r * VLD + st_c * 8]) = vreg[u];
        }
    };

    const int nblk = kv_end > kv_begin ? (kv_end - kv_begin + KVB - 1) / KVB : 0;
    if (nblk > 0) { fetch(kv_begin); publish(0); }
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): no load (Q fragments included) is pending at the loop head on ANY path, so the
    __syncthreads();                      // compiler's counters do not make Q.K^T wait for the block prefetch issued above it
    // lane-constant pieces of the V^T transpose-read address: 16-lane group gi = (hi, half), lane i of it supplies row i >> 2 of the
    // 4-key group and the 4 columns (i & 3) * 4 .. + 3, and receives column half * 16 + i = (l & 31) of the 32-wide d tile
    const int tr_off = (4 * hi + ((lane & 15) >> 2)) * VLD + ((lane >> 4) & 1) * 16 + (lane & 3) * 4;

    for (int ib = 0; ib < nblk; ++ib) {
        const int kv0 = kv_begin + ib * KVB, buf = (RING == 2) ? (ib & 1) : 0;   // RING = 1: the launcher guarantees nblk <= 1
        const bool more = (RING == 2) && ib + 1 < nblk;
        if (more) fetch(kv0 + KVB);
        if (wave_live && kv0 < wave_kv_end) {
            const bf16* Kb = Ks + buf * C::K_ELEMS;
            const bf16* Vb = Vs + buf * C::V_ELEMS;
            // ---- S^T = K . Q^T: two 32-key tiles. K fragments are read one group (2 k-slices x 2 tiles) ahead of the MFMAs that
            // consume them - pinned with scheduling barriers, the compiler otherwise sinks every read right in front of its MFMA
            f32x16 s[2];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) s[t][r] = 0.f;
            constexpr int NG = (NKS + 1) / 2;
            bf16x8 kf[2][2][2];
            const bf16* kbase = Kb + lq * KLD + hi * 8;
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int t = 0; t < 2; ++t)
                    if (j < NKS) kf[0][j][t] = *reinterpret_cast<const bf16x8*>(kbase + t * 32 * KLD + j * 16);
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                if (g + 1 < NG) {
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int t = 0; t < 2; ++t)
                            if (2 * g + 2 + j < NKS)
                                kf[(g + 1) & 1][j][t] = *reinterpret_cast<const bf16x8*>(kbase + t * 32 * KLD + (2 * g + 2 + j) * 16);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int t = 0; t < 2; ++t)
                        if (2 * g + j < NKS) s[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[g & 1][j][t], qf[2 * g + j], s[t], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            // V^T fragments of the first 16-key slice (all d tiles): requested now, consumed after the softmax
            bf16x8 vf[2][NDT];
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt) {
                const bf16* a0 = Vb + tr_off + dt * 32;
                vf[0][dt] = tr_pair(a0, a0 + 8 * VLD);
            }
            __builtin_amdgcn_sched_barrier(0);
            // ---- mask (edge / diagonal blocks only) + online softmax; lane: query q_abs, keys kv0 + 32 t + crow(r, hi)
            const bool need_mask = (kv0 + KVB > len_k) || (kv0 < p.kv_start) || (p.causal && kv0 + KVB - 1 > wq0 + causal_shift);
            if (need_mask) {
                // key of element (t, r) = kv0 + 4 hi + c(t, r), c = 32 t + (r & 3) + 8 (r >> 2): compare c with lane-relative bounds
                const int base = kv0 + 4 * hi;
                const int ub = min(len_k - 1, p.causal ? q_abs + causal_shift : 0x7fffffff) - base;
                if (kv0 < p.kv_start) {
                    const int lb = p.kv_start - base;
#pragma unroll
                    for (int t = 0; t < 2; ++t)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int c = t * 32 + (r & 3) + 8 * (r >> 2);
                            s[t][r] = (c <= ub && c >= lb) ? s[t][r] : -INFINITY;
                        }
                } else {
#pragma unroll
                    for (int t = 0; t < 2; ++t)
#pragma unroll
                        for (int r = 0; r < 16; ++r) s[t][r] = (t * 32 + (r & 3) + 8 * (r >> 2) <= ub) ? s[t][r] : -INFINITY;
                }
            }
            float mx = -INFINITY;
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[t][r]);
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            mx *= sc;
            const bool grow = mx > m_run + DEFER_THR;
            if (__builtin_amdgcn_ballot_w64(grow) != 0ull) {   // the branch is wave-wide, the decision is per row: a row's arithmetic never
                const float m_new = grow ? fmaxf(m_run, mx) : m_run;   // depends on which rows share its wave (prefix-KV reuse stays bit-exact)
                const float alpha = __builtin_amdgcn_exp2f(m_run - ((m_new == -INFINITY) ? 0.f : m_new));
                l_run *= alpha;
#pragma unroll
                for (int i = 0; i < NDT; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][r] *= alpha;
                m_run = m_new;
            }
            const float neg_m = (m_run == -INFINITY) ? 0.f : -m_run;
            bf16x8 pf[4];
            float rs = 0.f;
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float e = __builtin_amdgcn_exp2f(__builtin_fmaf(s[t][r], sc, neg_m));
                    rs += e;
                    pf[t * 2 + (r >> 3)][r & 7] = (bf16)e;
                }
            l_run += rs;
            // ---- O^T += V^T . P^T: key slices of 16 outermost, so that consecutive MFMAs go to NDT independent accumulators; the next
            // slice's V^T fragments are read under this slice's MFMAs
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                if (s4 + 1 < 4) {
#pragma unroll
                    for (int dt = 0; dt < NDT; ++dt) {
                        const bf16* a0 = Vb + tr_off + (s4 + 1) * 16 * VLD + dt * 32;
                        vf[(s4 + 1) & 1][dt] = tr_pair(a0, a0 + 8 * VLD);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int dt = 0; dt < NDT; ++dt) acc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[s4 & 1][dt], pf[s4], acc[dt], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if constexpr (RING == 2) {
            if (more) publish(buf ^ 1);
        }
        __syncthreads();
    }

    // ---- output: normalise, stage the wave's 32 x D tile in LDS (the ring is free after the last barrier), store whole rows
    l_run += __shfl_xor(l_run, 32);
    const float inv_l = (l_run > 0.f) ? 1.0f / l_run : 0.f;
    bf16* Ow = reinterpret_cast<bf16*>(smem_raw) + wave * 32 * OLD;
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
            const int d0 = dt * 32 + rg * 8 + hi * 4;
            if (d0 < D) {
                bf16x4 o = {(bf16)(acc[dt][rg * 4 + 0] * inv_l), (bf16)(acc[dt][rg * 4 + 1] * inv_l), (bf16)(acc[dt][rg * 4 + 2] * inv_l),
                            (bf16)(acc[dt][rg * 4 + 3] * inv_l)};
                *reinterpret_cast<bf16x4*>(&Ow[lq * OLD + d0]) = o;
            }
        }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < (32 * KCH + 63) / 64; ++u) {
        const int q = lane + u * 64, row = q / KCH, c = q % KCH;
        if (q < 32 * KCH && wq0 + row >= 0)
            *reinterpret_cast<bf16x8*>(O + (size_t)(wq0 + row) * p.o_rs + c * 8) = *reinterpret_cast<const bf16x8*>(&Ow[row * OLD + c * 8]);
    }
}

template <int D, int NW, int RING = 2>
int launch_wide(const AttnArgs& p, hipStream_t stream) {
    using C = WideCfg<D, RING>;
    auto kern = attn_fwd_wide_kernel<D, NW, RING>;
    static bool attr_done = false;
    if (!attr_done) {
        INA_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS));
        attr_done = true;
    }
    dim3 grid(((p.Lq + NW * 32 - 1) / (NW * 32)) * p.H * p.B);
    hipLaunchKernelGGL(kern, grid, dim3(NW * 64), C::LDS, stream, p);
    INA_HIP_CHECK(hipGetLastError());
    return 0;
}

}  // namespace

// Window attention of the Qwen2.5-VL vision tower (28 of its 32 blocks): packed sequences of at most 64 tokens, every (window, head) is ONE
// K / V block. Two waves per workgroup (64 query rows), a single-buffer image (23.5 KiB at d 80 instead of 47): no ring to fill, twice the
// workgroups per CU for a call that is pure HBM traffic (q | k | v in, o out).
static bool window_shape(const AttnArgs& p) {
    return p.cu_q && p.cu_k && p.Lq <= 64 && p.Lk <= 64 && !p.causal && p.kv_start == 0 && !p.k_len && p.kv_bdiv == 1 && p.D == 80;
}

bool ina_attention_wide_contract(const AttnArgs& p) {
    // what the kernel can run at all (ina_attn_args.kernel = 2 is rejected outside of it)
    if (p.D != 128 && p.D != 80 && p.D != 64) return false;
    if (p.accumulate || p.head_gate || p.drop_thresh) return false;
    if (p.scale <= 0.f) return false;
    if ((p.Lq < 128 || p.Lk < 128) && !window_shape(p)) return false;
    // whole-row 16-byte output stores
    if (p.o_rs % 8 || p.o_hs % 8 || p.o_bs % 8 || ((uintptr_t)p.O % 16)) return false;
    // 32-bit byte offsets inside one (batch, head) K / V plane
    if (p.k_rs <= 0 || p.v_rs <= 0 || (double)p.Lk * (double)p.k_rs * 2.0 >= 4.0e9 || (double)p.Lk * (double)p.v_rs * 2.0 >= 4.0e9) return false;
    return true;
}

bool ina_attention_wide_eligible(const AttnArgs& p) {
    // the automatic rule (ina_attn_args.kernel = 0): every long dense shape inside the kernel's contract - LLM prefill (d 128), Qwen ViT
    // full-attention blocks (d 80) and, since round 4, DINOv2 (d 64, 257 tokens: 42 vs 67 us per launch; adopted behind the 64-env
    // distributional parity test of the NavDP heads, tests/test_b64_distribution_gpu.py) and the Qwen ViT windows
    return ina_attention_wide_contract(p);
}

int ina_launch_attention_wide(const AttnArgs& p, hipStream_t stream) {
    if (window_shape(p)) return launch_wide<80, 2, 1>(p, stream);
    switch (p.D) {
        case 128: return launch_wide<128, 4>(p, stream);
        case 80: return launch_wide<80, 4>(p, stream);
        case 64: return launch_wide<64, 4>(p, stream);
        default: ina_set_error("attention (wide): unsupported head dim %d", p.D); return -2;
    }
}
