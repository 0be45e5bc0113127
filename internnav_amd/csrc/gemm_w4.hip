// 256 x 256 bf16 MFMA GEMM tile on FOUR waves (wave tile 128 x 128) - tile configs 38 / 39 (both operands through LDS-DMA staging) and 40 (round 5:
// LDS-DMA for A only, the B fragments straight from a fragment-ordered copy of W into registers; further down) of ina_gemm_bf16.
// Same stage layout, source-side XOR swizzle, K order and fused epilogue as the 8-wave kernels of gemm_glds.hip (bit-equal results).
//
// Why four waves: a 32-wide K slice costs a 128 x 128 wave tile 16 ds_read_b128 for 64 MFMAs - 64 KiB of LDS reads per workgroup and
// slice instead of the 96 KiB of the 8-wave tiles (128 x 64 wave tiles), which together with the 32 KiB of DMA writes keep the
// 128 B / clk LDS port busy for as long as the MFMAs of the slice take. The price: 64 accumulator fragments = all 256 AGPRs of a
// one-wave-per-SIMD kernel, no partner wave on the SIMD to ping-pong with, and a register allocator that cannot place the tuples (the
// builtin form of this loop spills and moves ~400 accumulator registers through VGPRs per K stage). So the K loop is written as a
// fixed instruction stream: every MFMA is an asm statement with its accumulator TIED in an AGPR tuple ("+a") and its operands in VGPRs,
// every fragment read an asm ds_read_b128 with an immediate offset, the waits are explicit. The overlap is inside the wave: while the
// 64 MFMAs of slice s issue, the 16 reads of slice s + 1 (second fragment set) and - in the second slice of a stage - the wave's 16 DMA
// instructions of stage t + 2 are threaded between them (one non-MFMA instruction after every MFMA but each third).
// One s_barrier per 64-wide K stage, between its two slices: before it every wave holds the second slice's fragments in registers (the
// buffer is free) and its share of stage t + 1 has landed; after it the DMA refills the buffer and slice 0 of stage t + 1 is read.
#include <utility>

#include "common.h"
#include "gemm_epilogue.h"
#include "kernels.h"

namespace {

template <int BM, int BN, int WM, int WN, int NS>
struct GldsCfg {
    static constexpr int NW = WM * WN, NT = NW * 64, BK = 64;
    static constexpr int TM = BM / WM, TN = BN / WN, FM = TM / 16, FN = TN / 16;
    static constexpr int A_INST = BM / 8 / NW, B_INST = BN / 8 / NW;   // 1 KiB wave-instructions per wave per stage
    static constexpr size_t LDS_BYTES = size_t(NS) * (BM + BN) * BK * sizeof(bf16);
};

__device__ __forceinline__ void glds16(const bf16* src, uint32_t lds_byte_addr) {
    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)src,
                                     (void __attribute__((address_space(3)))*)(uintptr_t)lds_byte_addr, 16, 0, 0);
}

// the same DMA piece with a scalar base + 32-bit lane offset (no vector address arithmetic per piece); M0 = LDS byte address of the piece
__device__ __forceinline__ void glds16_saddr(const void* base, uint32_t voff, uint32_t lds_byte_addr) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_byte_addr), "v"(voff), "s"(base) : "memory", "m0");
}

template <int OFF>
__device__ __forceinline__ void lds_read16(bf16x8& dst, uint32_t addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF));
}

__device__ __forceinline__ void mfma_tied(f32x4& c, const bf16x8& a, const bf16x8& b) {
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}

template <int N>
using IC = std::integral_constant<int, N>;

// f(IC<0>{}), f(IC<1>{}), ... f(IC<N - 1>{}): a fully unrolled loop whose index is a constant expression inside f (asm immediates, array slots)
template <class F, int... I>
__device__ __forceinline__ void for_seq_impl(F&& f, std::integer_sequence<int, I...>) {
    (f(IC<I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void for_seq(F&& f) {
    for_seq_impl(f, std::make_integer_sequence<int, N>{});
}

// (VAR 1 = cfg 39 is the only build: round 4's VAR 0 - the wave's DMA instructions of a stage in one clump right after the barrier - lost to it.) The DMA
// instructions of a stage are threaded between the MFMAs of the stage's second slice (after MFMAs 1, 4, 7, ...) as asm pieces with a scalar base + 32-bit lane offset
// (no vector address add per piece: 1477 vs 1440 TF/s at 8192^3, decoder chain 70.4 vs 70.9 ms, profiles/r04n_native_w4_saddr*.log); the last two
// stages (which request nothing) run in a second copy of the loop body.
template <int BM, int VAR>
__global__ __launch_bounds__(256) void gemm_bf16_w4_kernel(GemmArgs p) {
    constexpr int BN = 256;
    using C = GldsCfg<BM, BN, 2, 2, 2>;                    // BM 256: TM = TN = 128, FM = FN = 8, 8 + 8 DMA wave-instructions per wave and stage
    constexpr int FM = C::FM, NMF = FM * 8, NRD = FM + 8, NDMA = C::A_INST + C::B_INST;
    constexpr uint32_t A_STAGE = BM * 128u, B_STAGE = BN * 128u, B_BASE = 2u * A_STAGE;   // bytes
    static_assert(3 * (NRD - 1) + 2 < NMF && 3 * (NDMA - 1) + 1 < NMF, "the reads / DMA instructions of a slice are threaded between its MFMAs");
    extern __shared__ __attribute__((aligned(1024))) char smem_raw[];
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem_raw;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int tiles_n = (p.N + BN - 1) / BN, tiles_m = (p.M + BM - 1) / BM;
    const int id = xcd_remap(blockIdx.x, tiles_m * tiles_n);
    int tm, tn;
    if (p.group_m > 1) {
        const int gsz = p.group_m * tiles_n, grp = id / gsz, first = grp * p.group_m;
        const int gm = min(tiles_m - first, p.group_m), in = id - grp * gsz;
        tm = first + in % gm;
        tn = in / gm;
    } else {
        tm = id / tiles_n;
        tn = id % tiles_n;
    }
    const int m0 = tm * BM, n0 = tn * BN;
    const bf16* __restrict__ A = reinterpret_cast<const bf16*>(p.A) + (size_t)blockIdx.y * p.strideA;
    const bf16* __restrict__ W = reinterpret_cast<const bf16*>(p.W) + (size_t)blockIdx.y * p.strideW;

    // DMA: slot s of a wave covers tile rows (wave * INST + s) * 8 .. + 7; lane -> (row, physical 16-byte chunk), logical chunk = physical ^ (row & 7)
    const bf16* asrc[C::A_INST];
    const bf16* bsrc[C::B_INST];
    const int drow = lane >> 3, dcp = lane & 7;
#pragma unroll
    for (int s = 0; s < C::A_INST; ++s) {
        const int row = (wave * C::A_INST + s) * 8 + drow;
        asrc[s] = A + (size_t)min(m0 + row, p.M - 1) * p.lda + ((dcp ^ (row & 7)) << 3);
    }
#pragma unroll
    for (int s = 0; s < C::B_INST; ++s) {
        const int row = (wave * C::B_INST + s) * 8 + drow;
        bsrc[s] = W + (size_t)min(n0 + row, p.N - 1) * p.ldw + ((dcp ^ (row & 7)) << 3);
    }
    uint32_t aoff[C::A_INST], boff[C::B_INST];            // VAR 1: byte offsets of the same sources from A / W (the launcher checks they fit in 32 bits)
#pragma unroll
    for (int s = 0; s < C::A_INST; ++s) aoff[s] = (uint32_t)(reinterpret_cast<const char*>(asrc[s]) - reinterpret_cast<const char*>(A));
#pragma unroll
    for (int s = 0; s < C::B_INST; ++s) boff[s] = (uint32_t)(reinterpret_cast<const char*>(bsrc[s]) - reinterpret_cast<const char*>(W));
    uint32_t a_dma = lds0 + wave * C::A_INST * 1024u, b_dma = lds0 + B_BASE + wave * C::B_INST * 1024u;   // this wave's rows of the buffer to fill
    auto dma_one = [&](auto Dc, uint32_t la, uint32_t lb, int k0) __attribute__((always_inline)) {
        constexpr int D = decltype(Dc)::value;
        if constexpr (D < C::A_INST) glds16_saddr(A + k0, aoff[D], __builtin_amdgcn_readfirstlane(la + D * 1024u));
        else glds16_saddr(W + k0, boff[D - C::A_INST], __builtin_amdgcn_readfirstlane(lb + (D - C::A_INST) * 1024u));
    };
    auto dma_stage = [&](uint32_t la, uint32_t lb, int k0) __attribute__((always_inline)) {
        for_seq<NDMA>([&](auto Dc) __attribute__((always_inline)) { dma_one(Dc, la, lb, k0); });
    };

    f32x4 acc[2][FM][4];                                   // [64-column half of the wave tile][row fragment][column fragment]
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[h][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = p.K / 64;
    const int frow = lane & 15, g = lane >> 4, sw = lane & 7;
    // per-lane LDS byte addresses of fragment 0 of the wave tile in the CURRENT buffer, one per 32-wide slice (the swizzle is not additive in kk)
    uint32_t a_rd[2], b_rd[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        const uint32_t chunk = (uint32_t)(((kk * 4 + g) ^ sw) << 4);
        a_rd[kk] = lds0 + (uint32_t)(wm * C::TM + frow) * 128u + chunk;
        b_rd[kk] = lds0 + B_BASE + (uint32_t)(wn * C::TN + frow) * 128u + chunk;
    }
    int da = (int)A_STAGE, db = (int)B_STAGE;              // current buffer -> the other one
    bf16x8 fa[2][FM], fb[2][8];                            // two fragment sets: slice s multiplies out of one while slice s + 1 lands in the other

    // read R (0 .. NRD - 1) of fragment set SET: b0 a0 b1 a1 ... (fragment q = rows q * 16 .. + 15 of the wave tile: + q * 2 KiB)
    auto read_one = [&](auto Rc, auto Sc, uint32_t aa, uint32_t ba) __attribute__((always_inline)) {
        constexpr int R = decltype(Rc)::value, SET = decltype(Sc)::value;
        constexpr int NPAIR = 2 * FM;                      // reads 0 .. 2 FM - 1 alternate b / a, the rest (FM < 8) are b fragments
        if constexpr (R < NPAIR && (R & 1)) lds_read16<(R >> 1) * 2048>(fa[SET][R >> 1], aa);
        else {
            constexpr int Q = R < NPAIR ? (R >> 1) : (R - FM);
            lds_read16<Q * 2048>(fb[SET][Q], ba);
        }
    };
    // one slice: the MFMAs of fragment set SET with the reads of the next slice (set SET ^ 1 <- addresses aa / ba) after MFMAs 2, 5, 8, ...
    // and (DMA) this wave's DMA instructions of K offset k0 into its rows la / lb of a stage buffer after MFMAs 1, 4, 7, ...
    auto slice = [&](auto Sc, auto DMAc, uint32_t aa, uint32_t ba, uint32_t la, uint32_t lb, int k0) __attribute__((always_inline)) {
        constexpr int SET = decltype(Sc)::value;
        constexpr bool DMA = decltype(DMAc)::value != 0;
        for_seq<NMF>([&](auto Ic) __attribute__((always_inline)) {
            constexpr int I = decltype(Ic)::value, i = I >> 3, j = I & 7;
            mfma_tied(acc[j >> 2][i][j & 3], fb[SET][j], fa[SET][i]);
            if constexpr (DMA && I % 3 == 1 && I / 3 < NDMA) dma_one(IC<I / 3>{}, la, lb, k0);
            if constexpr (I % 3 == 2 && I / 3 < NRD) read_one(IC<I / 3>{}, IC<SET ^ 1>{}, aa, ba);
        });
    };

    dma_stage(a_dma, b_dma, 0);
    if (nk > 1) {
        dma_stage(a_dma + A_STAGE, b_dma + B_STAGE, 64);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA) : "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    for_seq<NRD>([&](auto Rc) __attribute__((always_inline)) { read_one(Rc, IC<0>{}, a_rd[0], b_rd[0]); });
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    // one K stage per iteration, ONE straight-line body per loop (a join of two MFMA paths makes the allocator shuffle the 256 accumulators)
    auto stage_body = [&](auto DMAc, int t) __attribute__((always_inline)) {
        slice(IC<0>{}, IC<0>{}, a_rd[1], b_rd[1], 0u, 0u, 0);        // slice 0 multiplies, slice 1 of this stage lands in set 1
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // own share of stage t + 1 landed; the current buffer is not read any more
        __builtin_amdgcn_s_barrier();
        // slice 1 multiplies, slice 0 of stage t + 1 (other buffer; stale LDS after the last stage) lands in set 0
        slice(IC<1>{}, DMAc, a_rd[0] + da, b_rd[0] + db, a_dma, b_dma, (t + 2) * 64);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        a_rd[0] += da; a_rd[1] += da; b_rd[0] += db; b_rd[1] += db;
        a_dma += da; b_dma += db;
        da = -da; db = -db;
    };
    int t = 0;
    for (; t + 2 < nk; ++t) stage_body(IC<1>{}, t);
    for (; t < nk; ++t) stage_body(IC<0>{}, t);
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");    // the last MFMAs retire before the compiler-scheduled epilogue reads the accumulators
    // LDS-transposed epilogue only (the launcher rejects outputs it cannot take): the direct register epilogue indexes the accumulator
    // array in rolled loops, which moves it to scratch memory around every asm MFMA
    __syncthreads();       // every wave is done with the stage buffers: the whole LDS becomes the per-wave transpose scratch
    float* scratch = reinterpret_cast<float*>(smem_raw) + wave * FM * 1024;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        gemm_store_tile_staged<FM, 4, C::TM, 64, true>(p, acc[h], m0, n0, wm, wn * 2 + h, lane, scratch);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
    }
}

// ---- tile config 40: the same 256 x 256 tile on four waves with the B operand NOT staged through LDS. W is stored once (ina_gemm_preshuffle)
// in MFMA fragment order - fragment (nf, ks) = W rows nf * 16 .. + 15, k ks * 32 .. + 31 is ONE contiguous KiB, lane (g, r) holding the 16
// bytes W[nf * 16 + r][ks * 32 + g * 8 .. + 7] - and every wave fetches the 8 fragments of its 128 columns per 32-wide slice straight into
// operand registers with `global_load_dwordx4 v, voff, s[base]` (a scalar base advanced per slice + eight per-fragment lane offsets).
// Why (tools/native/issue_cost_probe, profiles/r05r_*): between the MFMAs of a one-wave-per-SIMD kernel a vector load is nearly free
// (16 per stage: 1.25 -> 1.33 us), 16 LDS-DMA pieces + 32 fragment reads are not (1.25 -> 1.99 us): the stage of cfg 39 carries 16 pieces
// + 32 reads per wave, this one 8 pieces (A) + 16 vector loads + 16 reads.
// The MFMAs of a slice run fragment-major (all 8 row fragments against B fragment j, then j + 1): B fragment j is dead after MFMA 8 j + 7
// and is reloaded right there for slice s + 2 - two register sets give almost two slices (~1900 cycles) of prefetch distance. The loads share
// the in-order VMEM counter with the DMA pieces; every wait is a counted `s_waitcnt vmcnt(N)` with N derived from the fixed positions below
// (D: this stage requests a DMA refill, P: the previous stage did):
//     B reload j of a slice behind MFMA 8 j + 7, A read r behind MFMA 8 r + 3, DMA piece d (second slice) behind MFMA 8 d + 5,
//     B fragment j of slice 0 is awaited with vmcnt(15 + 8 P), of slice 1 with vmcnt(15 + (7 - j) P + j D), the DMA share of stage t + 1
//     before the barrier with vmcnt(9).
// Same products in the same K order as every other tile: bit-equal.
template <int OFF_UNUSED = 0>
__device__ __forceinline__ void gload16_saddr(bf16x8& dst, const void* base, uint32_t voff) {
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dst) : "v"(voff), "s"(base) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// positions (0 .. 6) of the A read and of the DMA piece inside a group of 8 MFMAs; the waits only assume that both precede the B reload at 7.
// Six other placements measure within +-1 % of this one on every shape (tools/native/build_w4p_variants.sh, profiles/r05z_native_w4p_positions.log).
#ifndef W4P_RD_POS
#define W4P_RD_POS 3
#endif
#ifndef W4P_DMA_POS
#define W4P_DMA_POS 5
#endif
template <int BM>
__global__ __launch_bounds__(256) void gemm_bf16_w4p_kernel(GemmArgs p) {
    constexpr int BN = 256;
    using C = GldsCfg<BM, BN, 2, 2, 2>;
    constexpr int FM = C::FM, NMF = FM * 8, NDMA = C::A_INST;
    static_assert(FM == 8 && NDMA == 8, "the positions of the threaded instructions assume 8 row fragments and 8 DMA pieces per wave");
    constexpr uint32_t A_STAGE = BM * 128u;                // bytes
    extern __shared__ __attribute__((aligned(1024))) char smem_raw[];
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem_raw;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int tiles_n = (p.N + BN - 1) / BN, tiles_m = (p.M + BM - 1) / BM;
    const int id = xcd_remap(blockIdx.x, tiles_m * tiles_n);
    int tm, tn;
    if (p.group_m > 1) {
        const int gsz = p.group_m * tiles_n, grp = id / gsz, first = grp * p.group_m;
        const int gm = min(tiles_m - first, p.group_m), in = id - grp * gsz;
        tm = first + in % gm;
        tn = in / gm;
    } else {
        tm = id / tiles_n;
        tn = id % tiles_n;
    }
    const int m0 = tm * BM, n0 = tn * BN;
    const bf16* __restrict__ A = reinterpret_cast<const bf16*>(p.A) + (size_t)blockIdx.y * p.strideA;
    const char* __restrict__ Wp = reinterpret_cast<const char*>(p.Wp);

    // A: LDS-DMA exactly as cfg 39 (slot s of a wave covers tile rows (wave * 8 + s) * 8 .. + 7; logical chunk = physical ^ (row & 7))
    uint32_t aoff[NDMA];
    const int drow = lane >> 3, dcp = lane & 7;
#pragma unroll
    for (int s = 0; s < NDMA; ++s) {
        const int row = (wave * NDMA + s) * 8 + drow;
        aoff[s] = (uint32_t)(((size_t)min(m0 + row, p.M - 1) * p.lda + ((dcp ^ (row & 7)) << 3)) * sizeof(bf16));
    }
    uint32_t a_dma = lds0 + wave * NDMA * 1024u;
    auto dma_one = [&](auto Dc, uint32_t la, int k0) __attribute__((always_inline)) {
        constexpr int D = decltype(Dc)::value;
        glds16_saddr(A + k0, aoff[D], __builtin_amdgcn_readfirstlane(la + D * 1024u));
    };
    // B: fragment j of the wave's 128 columns = W rows n0 + wn * 128 + j * 16 .. + 15 (clamped to the last fragment of W: those columns are
    // never stored); per-lane byte offset of fragment j from the slice's scalar base
    const int KS = p.K / 32, nfrag = p.N / 16;
    const int nf0 = (n0 + wn * 128) / 16;
    uint32_t boff[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) boff[j] = (uint32_t)((size_t)min(nf0 + j, nfrag - 1) * KS * 1024u + lane * 16u);
    const int nsl = 2 * (p.K / 64);                        // 32-wide slices
    auto b_base = [&](int sl) __attribute__((always_inline)) { return Wp + (size_t)min(sl, nsl - 1) * 1024u; };

    f32x4 acc[2][FM][4];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[h][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = p.K / 64;
    const int frow = lane & 15, g = lane >> 4, sw = lane & 7;
    uint32_t a_rd[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) a_rd[kk] = lds0 + (uint32_t)(wm * C::TM + frow) * 128u + (uint32_t)(((kk * 4 + g) ^ sw) << 4);
    int da = (int)A_STAGE;
    bf16x8 fa[2][FM], fb[2][8];                            // A: slice s multiplies out of one set while slice s + 1 lands in the other; B: set = slice parity

    // one slice: MFMAs fragment-major; SET = slice parity (its B set, and the A set it multiplies); the A reads of the next slice (set SET ^ 1)
    // from address aa, the B reloads of slice `sl2` (= this slice + 2) into the set being consumed, the DMA pieces of K offset k0
    auto slice = [&](auto Sc, auto Dc, auto Pc, auto DMAc, uint32_t aa, int sl2, uint32_t la, int k0) __attribute__((always_inline)) {
        constexpr int SET = decltype(Sc)::value, D = decltype(Dc)::value, P = decltype(Pc)::value;
        constexpr bool DMA = decltype(DMAc)::value != 0;
        const char* bb = b_base(sl2);
        for_seq<NMF>([&](auto Ic) __attribute__((always_inline)) {
            constexpr int I = decltype(Ic)::value, j = I >> 3, i = I & 7;
            // (one wait per fragment PAIR instead - vmcnt(14 + 8 P) / vmcnt(14 + (6 - j) P + j D) at even j - measures the same: r05v)
            if constexpr (i == 0) wait_vm<(SET == 0) ? 15 + 8 * P : 15 + (7 - j) * P + j * D>();
            mfma_tied(acc[j >> 2][i][j & 3], fb[SET][j], fa[SET][i]);
            if constexpr (i == W4P_RD_POS) lds_read16<j * 2048>(fa[SET ^ 1][j], aa);
            if constexpr (DMA && i == W4P_DMA_POS) dma_one(IC<j>{}, la, k0);
            if constexpr (i == 7) gload16_saddr(fb[SET][j], bb, boff[j]);
        });
    };

    // prologue: A of stages 0 and 1, B of slices 0 and 1
    for_seq<NDMA>([&](auto Dc) __attribute__((always_inline)) { dma_one(Dc, a_dma, 0); });
    for_seq<NDMA>([&](auto Dc) __attribute__((always_inline)) { dma_one(Dc, a_dma + A_STAGE, nk > 1 ? 64 : 0); });
    for_seq<8>([&](auto Jc) __attribute__((always_inline)) { gload16_saddr(fb[0][decltype(Jc)::value], b_base(0), boff[decltype(Jc)::value]); });
    for_seq<8>([&](auto Jc) __attribute__((always_inline)) { gload16_saddr(fb[1][decltype(Jc)::value], b_base(1), boff[decltype(Jc)::value]); });
    wait_vm<8 + 16>();                                     // stage 0 of A has landed (stage 1 and the B fragments may be in flight)
    __builtin_amdgcn_s_barrier();
    for_seq<FM>([&](auto Rc) __attribute__((always_inline)) { lds_read16<decltype(Rc)::value * 2048>(fa[0][decltype(Rc)::value], a_rd[0]); });
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

    // one K stage per call; D: it requests the refill of its buffer with stage t + 2, P: stage t - 1 did
    auto stage_body = [&](auto Dc, auto Pc, int t) __attribute__((always_inline)) {
        constexpr int D = decltype(Dc)::value;
        slice(IC<0>{}, Dc, Pc, IC<0>{}, a_rd[1], 2 * t + 2, 0u, 0);              // slice 0 multiplies, slice 1's A fragments land in set 1
        wait_vm<9>();                                                          // own DMA share of stage t + 1 landed (pieces of the previous stage's slice 1)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                      // the current A buffer is not read any more
        __builtin_amdgcn_s_barrier();
        slice(IC<1>{}, Dc, Pc, IC<D>{}, a_rd[0] + da, 2 * t + 3, a_dma, (t + 2) * 64);   // slice 1 multiplies, slice 0 of stage t + 1 lands in set 0
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        a_rd[0] += da; a_rd[1] += da;
        a_dma += da;
        da = -da;
    };
    int t = 0;
    for (; t < 1 && t + 2 < nk; ++t) stage_body(IC<1>{}, IC<0>{}, t);
    for (; t + 2 < nk; ++t) stage_body(IC<1>{}, IC<1>{}, t);
    for (; t < nk; ++t) stage_body(IC<0>{}, IC<0>{}, t);   // (P = 0 waits for more than it must when stage t - 1 did request pieces: safe)
    // The clamped tail reloads are still in flight INTO the fb registers, which the compiler considers dead behind the last MFMA: it would hand
    // them to the epilogue's first accumulator copies (placed above a plain wait statement - nothing orders register-only code against it) and
    // the late data would overwrite those. The drain therefore names every fb register as an operand: they stay allocated until it.
    asm volatile("s_waitcnt vmcnt(0)\n\ts_nop 15\n\ts_nop 15"
                 : "+v"(fb[0][0]), "+v"(fb[0][1]), "+v"(fb[0][2]), "+v"(fb[0][3]), "+v"(fb[0][4]), "+v"(fb[0][5]), "+v"(fb[0][6]), "+v"(fb[0][7]),
                   "+v"(fb[1][0]), "+v"(fb[1][1]), "+v"(fb[1][2]), "+v"(fb[1][3]), "+v"(fb[1][4]), "+v"(fb[1][5]), "+v"(fb[1][6]), "+v"(fb[1][7])
                 :
                 : "memory");
    __syncthreads();
    float* scratch = reinterpret_cast<float*>(smem_raw) + wave * FM * 1024;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        gemm_store_tile_staged<FM, 4, C::TM, 64, true>(p, acc[h], m0, n0, wm, wn * 2 + h, lane, scratch);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
    }
}

// W [N, K] (row stride ldw) -> fragment order: Wp[((n / 16) * (K / 32) + k / 32) * 512 + ((k % 32) / 8 * 16 + n % 16) * 8 + k % 8]
__global__ __launch_bounds__(256) void gemm_preshuffle_kernel(const bf16* __restrict__ W, bf16* __restrict__ Wp, int N, int K, long ldw, long total16) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total16; i += (long)gridDim.x * 256) {
        const int lane = (int)(i & 63);
        const long frag = i >> 6;
        const int KS = K / 32;
        const int ks = (int)(frag % KS);
        const long nf = frag / KS;
        const long n = nf * 16 + (lane & 15);
        const int k = ks * 32 + (lane >> 4) * 8;
        reinterpret_cast<bf16x8*>(Wp)[i] = *reinterpret_cast<const bf16x8*>(W + n * ldw + k);
    }
}

template <int BM, int VAR>
int launch_w4(const GemmArgs& p, hipStream_t stream) {
    constexpr int BN = 256;
    using C = GldsCfg<BM, BN, 2, 2, 2>;
    static_assert(C::LDS_BYTES >= size_t(4) * C::FM * 4096, "the per-wave epilogue scratch must fit in the two stage buffers");
    INA_REQUIRE(ina_gemm_w4_contract(p), "gemm(w4): tile config 39 needs K %% 64 == 0 and 16-byte aligned output (and residual) rows (K=%d ldc=%d)", p.K, p.ldc);
    static bool attr_done = false;
    auto kern = gemm_bf16_w4_kernel<BM, VAR>;
    if (!attr_done) {
        INA_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS_BYTES));
        attr_done = true;
    }
    const int tiles = ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
    GemmArgs q = p;
    if (q.group_m == 0) q.group_m = ((p.N + BN - 1) / BN >= 24 && (p.M + BM - 1) / BM >= 8) ? 8 : 1;
    const double osz = p.out_dtype == INA_DT_BF16 ? 2.0 : 4.0;
    InaProfScope prof(INA_PROF_GEMM, 2.0 * p.M * p.N * p.K * p.batch,
                      (double)p.batch * (2.0 * p.M * p.K + 2.0 * p.N * p.K + osz * p.M * (p.glu ? p.N / 2 : p.N)), stream);
    hipLaunchKernelGGL(kern, dim3(tiles, p.batch, 1), dim3(C::NT), C::LDS_BYTES, stream, q);
    INA_HIP_CHECK(hipGetLastError());
    return 0;
}

int launch_w4p(const GemmArgs& p, hipStream_t stream) {
    constexpr int BM = 256, BN = 256;
    using C = GldsCfg<BM, BN, 2, 2, 2>;
    INA_REQUIRE(ina_gemm_w4_contract(p), "gemm(w4p): tile config 40 needs K %% 64 == 0 and 16-byte aligned output (and residual) rows (K=%d ldc=%d)", p.K, p.ldc);
    INA_REQUIRE(p.Wp && ((uintptr_t)p.Wp % 16) == 0 && p.N % 16 == 0 && p.batch <= 1 && (double)p.N * p.K * 2.0 < 4.0e9,
                "gemm(w4p): tile config 40 reads the fragment-ordered copy of W (ina_gemm_preshuffle) of an unbatched GEMM with N %% 16 == 0 (N=%d)", p.N);
    static bool attr_done = false;
    auto kern = gemm_bf16_w4p_kernel<BM>;
    if (!attr_done) {
        INA_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS_BYTES));
        attr_done = true;
    }
    const int tiles = ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
    GemmArgs q = p;
    // few column tiles and a long K (the decoder's down projection: 11-15 x 14 tiles, K = 18944, 9.7 MB operand panels): row-major order hands an XCD's
    // contiguous run of ~26 tiles 2 rows x 14 columns = 16 panels; groups of 4 row-tiles make it 4 x 6.5 = 11 panels: 445 -> 435 us at 3680 rows, 437 -> 425
    // at 2760, prefill graph - 0.3 ms (profiles/r06z_native_tile_order2.log, r06z_prefill_down_tile_order.txt). Bit-equal (the order of tiles, not of K).
    if (q.group_m == 0) {
        const int tn = (p.N + BN - 1) / BN, tm = (p.M + BM - 1) / BM;
        q.group_m = (tn >= 24 && tm >= 8) ? 8 : (tm >= 8 && p.K >= 8192) ? 4 : 1;
    }
    const double osz = p.out_dtype == INA_DT_BF16 ? 2.0 : 4.0;
    ina_prof_set_sub(40);
    InaProfScope prof(INA_PROF_GEMM, 2.0 * p.M * p.N * p.K, 2.0 * p.M * p.K + 2.0 * p.N * p.K + osz * p.M * (p.glu ? p.N / 2 : p.N), stream);
    hipLaunchKernelGGL(kern, dim3(tiles, 1, 1), dim3(C::NT), C::LDS_BYTES, stream, q);   // (the epilogue scratch needs the full 128 KiB; the A ring uses half)
    INA_HIP_CHECK(hipGetLastError());
    return 0;
}

}  // namespace

int ina_launch_gemm_preshuffle(const void* W, void* Wp, int N, int K, long ldw, hipStream_t stream) {
    INA_REQUIRE(W && Wp && N > 0 && K > 0 && N % 16 == 0 && K % 32 == 0 && ldw % 8 == 0 && ((uintptr_t)W % 16) == 0 && ((uintptr_t)Wp % 16) == 0,
                "gemm_preshuffle: N %% 16, K %% 32, 16-byte aligned rows (N=%d K=%d ldw=%ld)", N, K, ldw);
    const long total16 = (long)N * K / 8;
    const int nb = (int)((total16 + 255) / 256 < 8192 ? (total16 + 255) / 256 : 8192);
    hipLaunchKernelGGL(gemm_preshuffle_kernel, dim3(nb), dim3(256), 0, stream, reinterpret_cast<const bf16*>(W), reinterpret_cast<bf16*>(Wp), N, K, ldw, total16);
    INA_HIP_CHECK(hipGetLastError());
    return 0;
}

bool ina_gemm_w4_contract(const GemmArgs& p) {
    // the kernel has the LDS-transposed epilogue only: whole 16-byte pieces of output (and residual) rows
    const size_t oes = p.out_dtype == INA_DT_BF16 ? 2 : 4, res = p.res_dtype == INA_DT_BF16 ? 2 : 4;
    if ((double)p.M * p.lda * 2.0 >= 4.0e9 || (double)p.N * p.ldw * 2.0 >= 4.0e9) return false;   // 32-bit byte offsets of the DMA sources inside A / W
    return p.K % 64 == 0 && ((uintptr_t)p.C % 16) == 0 && (p.ldc * oes) % 16 == 0 && (p.strideC * oes) % 16 == 0 &&
           (!p.R || (((uintptr_t)p.R % 16) == 0 && (p.ldr * res) % 16 == 0 && (p.strideR * res) % 16 == 0)) && ((p.glu ? p.N / 2 : p.N) % 4 == 0);
}

int ina_launch_gemm_w4(const GemmArgs& p, hipStream_t stream, int cfg) {
    switch (cfg) {
        case 39: return launch_w4<256, 1>(p, stream);   // DMA pieces threaded between the MFMAs of the second slice (asm, scalar base + lane offset)
        case 40: return launch_w4p(p, stream);          // B fragments from the fragment-ordered copy of W straight into registers, LDS for A only
        default: ina_set_error("gemm(w4): unknown tile config %d", cfg); return -2;
    }
}
