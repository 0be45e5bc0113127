// Large-K bf16 MFMA GEMM for gfx950 with direct-to-LDS staging:  C[M,N] = epilogue( A[M,K] . W[N,K]^T ),  K % 64 == 0.
//
// Same tile / wave / MFMA geometry and the same fused epilogue as gemm.hip, but the global->LDS path is the CDNA4 LDS-DMA
// (`global_load_lds_dwordx4`: 16 B per lane, 1 KiB per wave-instruction, no VGPR round trip, no ds_write pass):
//   * each stage holds A[BM][64] and W[BN][64] bf16 as plain 128-byte rows; a wave-instruction fills 8 consecutive rows;
//   * the DMA writes lane-linearly, so the bank-conflict fix is an XOR swizzle applied on the SOURCE side: the lane that lands on
//     physical 16-byte chunk cp of row r fetches logical chunk cp ^ (r & 7) (all 8 lanes of a row still cover one 128-byte line),
//     and the MFMA fragment reads apply the same involution (ds_read_b128 of chunk c ^ (r & 7));
//   * NS stages, one barrier per K tile: the first NS tiles are requested up front, then tile t+NS-1's DMA is issued right after
//     the barrier that publishes tile t and stays in flight under the MFMAs (counted `s_waitcnt vmcnt`, raw `s_barrier`).
// Rows beyond M / N are clamped to the last valid row (their accumulators are never stored) - the DMA has no predication.
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

__device__ __forceinline__ void glds16(const bf16* src, bf16* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)src,
                                     (void __attribute__((address_space(3)))*)lds_wave_base, 16, 0, 0);
}

template <int BM, int BN, int WM, int WN, int NS, bool STAGED>
__global__ __launch_bounds__(WM* WN * 64) void gemm_bf16_glds_kernel(GemmArgs p) {
    using C = GldsCfg<BM, BN, WM, WN, NS>;
    extern __shared__ __attribute__((aligned(1024))) char smem_raw[];
    bf16* As = reinterpret_cast<bf16*>(smem_raw);          // [NS][BM][64]
    bf16* Bs = As + NS * BM * 64;                          // [NS][BN][64]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int tiles_n = (p.N + BN - 1) / BN, tiles_m = (p.M + BM - 1) / BM;
    const int id = xcd_remap(blockIdx.x, tiles_m * tiles_n);
    int tm, tn;
    if (p.group_m > 1) {
        // grouped order: GM row-tiles x all column-tiles per group, row index fastest -> the ~32 workgroups an XCD runs at a time
        // cover a GM x (32/GM) block of tiles (GM + 32/GM operand panels instead of 1 + 32)
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

    // per-lane source pointers of this lane's DMA slots (k0 = 0); slot s of a wave covers tile rows (wave*INST + s)*8 .. +7
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
    auto stage = [&](int buf, int k0) {
        bf16* as = As + buf * BM * 64 + wave * C::A_INST * 512;
        bf16* bs = Bs + buf * BN * 64 + wave * C::B_INST * 512;
#pragma unroll
        for (int s = 0; s < C::A_INST; ++s) glds16(asrc[s] + k0, as + s * 512);
#pragma unroll
        for (int s = 0; s < C::B_INST; ++s) glds16(bsrc[s] + k0, bs + s * 512);
    };

    f32x4 acc[C::FM][C::FN];
#pragma unroll
    for (int i = 0; i < C::FM; ++i)
#pragma unroll
        for (int j = 0; j < C::FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = p.K / 64;
    const int frow = lane & 15, g = lane >> 4, sw = lane & 7;
    // NS-stage ring: K tiles t+1 .. t+NS-1 are in flight while tile t is multiplied. The wait is COUNTED (only the DMA of tile t
    // has to have landed: the newer tiles' INST wave-instructions may stay outstanding) and the barrier is a raw s_barrier -
    // __syncthreads() would emit vmcnt(0) for the pending LDS-DMA writes and drain the ring (guide 5, glds "span a barrier").
    constexpr int INST = C::A_INST + C::B_INST;
    if constexpr (NS == 1) {
        // single buffer, no software pipeline: 32 KiB of LDS per 128 x 128 workgroup, so 4 workgroups share a CU and hide each other's
        // DMA latency (occupancy instead of prefetch depth; for the 6-K-tile GEMMs of the d = 384 heads)
        for (int t = 0; t < nk; ++t) {
            stage(0, t * 64);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            const bf16* as = As + (wm * C::TM + frow) * 64;
            const bf16* bs = Bs + (wn * C::TN + frow) * 64;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const int chunk = ((kk * 4 + g) ^ sw) << 3;
                bf16x8 fa[C::FM], fb[C::FN];
#pragma unroll
                for (int i = 0; i < C::FM; ++i) fa[i] = *reinterpret_cast<const bf16x8*>(as + i * 16 * 64 + chunk);
#pragma unroll
                for (int j = 0; j < C::FN; ++j) fb[j] = *reinterpret_cast<const bf16x8*>(bs + j * 16 * 64 + chunk);
#pragma unroll
                for (int i = 0; i < C::FM; ++i)
#pragma unroll
                    for (int j = 0; j < C::FN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();          // everyone is done reading before the buffer is overwritten
        }
    } else {
    // all NS buffers are free at kernel start: request the first NS tiles at once (a K = 384 GEMM has only 6 tiles per output tile -
    // serialising the first two DMA latencies was ~10 % of its time); from iteration 1 on the buffer read in iteration t - 1 is refilled
#pragma unroll
    for (int s = 0; s < NS; ++s)
        if (s < nk) stage(s, s * 64);
    int buf = 0;
    for (int t = 0; t < nk; ++t) {
        // newer tiles in flight at this point: tiles 1 .. NS-1 in iteration 0, tiles t+1 .. t+NS-2 afterwards
        const int ahead = (t == 0) ? (NS - 1) : (NS - 2);
        const int newer = (nk - 1 - t) < ahead ? (nk - 1 - t) : ahead;
        if (NS >= 3 && newer >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * INST) : "memory");
        else if (newer >= 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(INST) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                        // tile t visible to all waves; all waves done with the buffer refilled below
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (t >= 1) {
            const int nt = t + NS - 1;                       // next tile to request goes into the buffer read in iteration t - 1
            int nb = buf + NS - 1;
            if (nb >= NS) nb -= NS;
            if (nt < nk) stage(nb, nt * 64);
        }
        const bf16* as = As + buf * BM * 64 + (wm * C::TM + frow) * 64;
        const bf16* bs = Bs + buf * BN * 64 + (wn * C::TN + frow) * 64;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int chunk = ((kk * 4 + g) ^ sw) << 3;
            bf16x8 fa[C::FM], fb[C::FN];
#pragma unroll
            for (int i = 0; i < C::FM; ++i) fa[i] = *reinterpret_cast<const bf16x8*>(as + i * 16 * 64 + chunk);
#pragma unroll
            for (int j = 0; j < C::FN; ++j) fb[j] = *reinterpret_cast<const bf16x8*>(bs + j * 16 * 64 + chunk);
#pragma unroll
            for (int i = 0; i < C::FM; ++i)
#pragma unroll
                for (int j = 0; j < C::FN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
        }
        if (++buf == NS) buf = 0;
    }
    }
    if constexpr (STAGED) {
        __syncthreads();   // every wave is done with the stage buffers: the whole LDS becomes the per-wave transpose scratch
        // all FM slabs of a wave at once when LDS is big enough (one copy of the epilogue math in the instruction stream)
        constexpr bool ALL = C::LDS_BYTES >= size_t(C::NW) * C::FM * 4096;
        float* scratch = reinterpret_cast<float*>(smem_raw) + wave * (ALL ? C::FM * 1024 : 1024);
        gemm_store_tile_staged<C::FM, C::FN, C::TM, C::TN, ALL>(p, acc, m0, n0, wm, wn, lane, scratch);
    } else {
        gemm_store_tile<C::FM, C::FN, C::TM, C::TN>(p, acc, m0, n0, wm, wn, lane);
    }
}

// Ping-pong variant (2 wave rows x WN wave columns, 2 stages): the two waves that share a SIMD (wave row 0 / 1 of the same wave
// column) run the same [ds_read fragments | MFMA] sequence HALF A STEP APART - while one issues its 32 MFMAs the other fetches its
// next fragments from LDS (different issue ports, so they overlap), and the roles swap at the next barrier. Time is cut into slots
// separated by one s_barrier each; with s = 2*tile + kk:
//     wave row 0:  slot 2s: R(s)  slot 2s+1: M(s)          wave row 1 (one extra barrier up front): slot 2s+1: R(s)  slot 2s+2: M(s)
// LDS buffer of tile t is read in slots 4t .. 4t+3 and may be refilled from slot 4t+4 on: every wave issues its share of tile
// t+1's DMA at the start of its R(2t) slot (4t / 4t+1; the buffer's last reader finished in slot 4t-1 and retired its reads with
// lgkmcnt(0) before that slot's barrier) and retires it (vmcnt(0)) before the barrier that ends slot 4t+3, one slot before the
// first read of tile t+1.
template <int BM, int BN, int WN, bool STAGED>
__global__ __launch_bounds__(2 * WN * 64) void gemm_bf16_pp_kernel(GemmArgs p) {
    constexpr int WM = 2, NS = 2;
    using C = GldsCfg<BM, BN, WM, WN, NS>;
    extern __shared__ __attribute__((aligned(1024))) char smem_raw[];
    bf16* As = reinterpret_cast<bf16*>(smem_raw);          // [2][BM][64]
    bf16* Bs = As + NS * BM * 64;                          // [2][BN][64]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // waves w and w + 4 land on the same SIMD (4 SIMDs, round robin): make them the two wave rows of one wave column
    const int wm = wave / WN, wn = wave % WN;
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
    auto stage = [&](int buf, int k0) {
        bf16* as = As + buf * BM * 64 + wave * C::A_INST * 512;
        bf16* bs = Bs + buf * BN * 64 + wave * C::B_INST * 512;
#pragma unroll
        for (int s = 0; s < C::A_INST; ++s) glds16(asrc[s] + k0, as + s * 512);
#pragma unroll
        for (int s = 0; s < C::B_INST; ++s) glds16(bsrc[s] + k0, bs + s * 512);
    };

    f32x4 acc[C::FM][C::FN];
#pragma unroll
    for (int i = 0; i < C::FM; ++i)
#pragma unroll
        for (int j = 0; j < C::FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = p.K / 64;
    const int frow = lane & 15, g = lane >> 4, sw = lane & 7;
    bf16x8 fa[C::FM], fb[C::FN];
    auto read_frags = [&](int buf, int kk) {
        const bf16* as = As + buf * BM * 64 + (wm * C::TM + frow) * 64;
        const bf16* bs = Bs + buf * BN * 64 + (wn * C::TN + frow) * 64;
        const int chunk = ((kk * 4 + g) ^ sw) << 3;
#pragma unroll
        for (int j = 0; j < C::FN; ++j) fb[j] = *reinterpret_cast<const bf16x8*>(bs + j * 16 * 64 + chunk);
#pragma unroll
        for (int i = 0; i < C::FM; ++i) fa[i] = *reinterpret_cast<const bf16x8*>(as + i * 16 * 64 + chunk);
    };
    auto mfmas = [&]() {
#pragma unroll
        for (int i = 0; i < C::FM; ++i)
#pragma unroll
            for (int j = 0; j < C::FN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
    };
    auto slot_end = [&]() {   // close a slot: nothing moves across, LDS reads of this slot have landed
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };

    stage(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                 // tile 0 visible (slot -1)
    if (wm == 1) __builtin_amdgcn_s_barrier();    // wave row 1 runs one slot behind
    int buf = 0;
    for (int t = 0; t < nk; ++t) {
        // R(kk = 0) slot: request tile t+1 into the other buffer, fetch the first fragment set
        if (t + 1 < nk) stage(buf ^ 1, (t + 1) * 64);
        read_frags(buf, 0);
        slot_end();
        mfmas();                                  // M(kk = 0) slot
        slot_end();
        read_frags(buf, 1);                       // R(kk = 1) slot
        if (wm == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // row 1: this barrier ends slot 4t+3
        slot_end();
        mfmas();                                  // M(kk = 1) slot
        if (wm == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // row 0: this barrier ends slot 4t+3
        slot_end();
        buf ^= 1;
    }
    if (wm == 0) __builtin_amdgcn_s_barrier();    // matching count for the extra barrier of row 1
    if constexpr (STAGED) {
        __syncthreads();
        constexpr bool ALL = C::LDS_BYTES >= size_t(C::NW) * C::FM * 4096;
        float* scratch = reinterpret_cast<float*>(smem_raw) + wave * (ALL ? C::FM * 1024 : 1024);
        gemm_store_tile_staged<C::FM, C::FN, C::TM, C::TN, ALL>(p, acc, m0, n0, wm, wn, lane, scratch);
    } else {
        gemm_store_tile<C::FM, C::FN, C::TM, C::TN>(p, acc, m0, n0, wm, wn, lane);
    }
}

template <int BM, int BN, int WM, int WN, int NS>
int launch_glds(const GemmArgs& p, hipStream_t stream) {
    using C = GldsCfg<BM, BN, WM, WN, NS>;
    // the LDS-transposed epilogue needs 16-byte aligned output (and residual) rows and 64-column wave tiles
    const size_t oes = p.out_dtype == INA_DT_BF16 ? 2 : 4, res = p.res_dtype == INA_DT_BF16 ? 2 : 4;
    const bool staged = C::TN == 64 && ((uintptr_t)p.C % 16) == 0 && (p.ldc * oes) % 16 == 0 && (p.strideC * oes) % 16 == 0 &&
                        (!p.R || (((uintptr_t)p.R % 16) == 0 && (p.ldr * res) % 16 == 0 && (p.strideR * res) % 16 == 0)) &&
                        ((p.glu ? p.N / 2 : p.N) % 4 == 0);
    static bool attr_done[2] = {false, false};
    auto kern = staged ? gemm_bf16_glds_kernel<BM, BN, WM, WN, NS, true> : gemm_bf16_glds_kernel<BM, BN, WM, WN, NS, false>;
    if (!attr_done[staged]) {
        INA_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS_BYTES));
        attr_done[staged] = true;
    }
    const int tiles = ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
    GemmArgs q = p;
    // auto tile order: with many column tiles the ~32 workgroups an XCD runs concurrently would share one A panel and stream 32
    // different W panels; groups of 8 row-tiles cut the panels per 32 tiles from 33 to 12 (PMC: FETCH_SIZE of the gate/up GEMM,
    // measured +9 % on 6440x37888x3584 and +10 % on 8192^3). Small grids keep the plain row-major order.
    if (q.group_m == 0) q.group_m = ((p.N + BN - 1) / BN >= 24 && (p.M + BM - 1) / BM >= 8) ? 8 : 1;
    const double osz = p.out_dtype == INA_DT_BF16 ? 2.0 : 4.0;
    InaProfScope prof(INA_PROF_GEMM, 2.0 * p.M * p.N * p.K * p.batch,
                      (double)p.batch * (2.0 * p.M * p.K + 2.0 * p.N * p.K + osz * p.M * (p.glu ? p.N / 2 : p.N)), stream);
    hipLaunchKernelGGL(kern, dim3(tiles, p.batch, 1), dim3(C::NT), C::LDS_BYTES, stream, q);
    INA_HIP_CHECK(hipGetLastError());
    return 0;
}

template <int BM, int BN, int WN>
int launch_pp(const GemmArgs& p, hipStream_t stream) {
    using C = GldsCfg<BM, BN, 2, WN, 2>;
    const size_t oes = p.out_dtype == INA_DT_BF16 ? 2 : 4, res = p.res_dtype == INA_DT_BF16 ? 2 : 4;
    const bool staged = C::TN == 64 && ((uintptr_t)p.C % 16) == 0 && (p.ldc * oes) % 16 == 0 && (p.strideC * oes) % 16 == 0 &&
                        (!p.R || (((uintptr_t)p.R % 16) == 0 && (p.ldr * res) % 16 == 0 && (p.strideR * res) % 16 == 0)) &&
                        ((p.glu ? p.N / 2 : p.N) % 4 == 0);
    static bool attr_done[2] = {false, false};
    auto kern = staged ? gemm_bf16_pp_kernel<BM, BN, WN, true> : gemm_bf16_pp_kernel<BM, BN, WN, false>;
    if (!attr_done[staged]) {
        INA_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS_BYTES));
        attr_done[staged] = true;
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

}  // namespace

// cfg 11: 128x128 / 4 waves / 2 stages, cfg 12: 256x128 / 8 waves / 2 stages, cfg 13: 128x256 / 8 waves / 2 stages,
// cfg 14: 256x128 / 8 waves / 3 stages, cfg 15: 128x128 / 4 waves / 3 stages, cfg 16: 128x256 / 8 waves / 3 stages,
// cfg 17: 256x256 / 8 waves / 2 stages (lock-step), cfg 18 / 19: ping-pong kernels (s_setprio(1) around their MFMA slots was
// measured and is within noise: 1351 vs 1391 TF/s at 8192^3, 1196 vs 1234 on the gate/up GEMM)
int ina_launch_gemm_glds(const GemmArgs& p, hipStream_t stream, int cfg) {
    INA_REQUIRE(p.K % 64 == 0, "gemm(glds): K=%d must be a multiple of 64", p.K);
    ina_prof_set_sub(cfg);
    switch (cfg) {
        case 11: return launch_glds<128, 128, 2, 2, 2>(p, stream);
        case 14: return launch_glds<256, 128, 4, 2, 3>(p, stream);
        case 22: return launch_glds<128, 128, 2, 2, 1>(p, stream);   // 128x128 single buffer, 4 workgroups per CU (2 waves of 128x64 wave tiles
                                                                      // instead of 4 of 64x64: 566 vs 726 TF/s at 65536x1536x384)
        case 26: return launch_glds<128, 256, 2, 4, 1>(p, stream);   // 128x256 single buffer (48 KiB LDS, 3 workgroups per CU), 85 FLOP per operand byte
        case 27: return launch_glds<256, 128, 4, 2, 1>(p, stream);   // 256x128 single buffer
        case 18: return launch_pp<256, 256, 4>(p, stream);           // 256x256, wave tile 128x64, ping-pong wave rows
        case 21: return launch_pp<192, 256, 4>(p, stream);           // 192x256, wave tile 96x64: finer row quantisation for M = 5520 / 6440
        case 33: return launch_glds<256, 256, 4, 4, 2>(p, stream);   // 256x256 lock-step with 16 waves (wave tile 64x64, 4 waves per SIMD): the fp32-residual GEMMs with a short K loop
        case 39: case 40: return ina_launch_gemm_w4(p, stream, cfg);                      // 256x256, FOUR waves of 128x128 (wave tile 128 x 128): gemm_w4.hip
        default: ina_set_error("gemm(glds): unknown tile config %d", cfg); return -2;
    }
}
