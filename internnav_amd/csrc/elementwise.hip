// Small HBM-bound kernels around the MFMA ops of the System-1 policies (gfx950):
//   patchify      NHWC frames -> normalised im2col rows of the 14x14/14 patch-embed conv (LDS-staged, coalesced both sides)
//   embed3        nn.Linear(3, C) + positional table, scattered through a row map (also plain table fill)
//   head3         final LayerNorm + nn.Linear(C, 3) + DDPM / flow-matching Euler update, one wave per token row
//   seqpool_head  LayerNorm + mean over the T tokens of a sequence + Linear(C, 1)   (NavDP critic)
//   select_traj   per-env ranking of the 32 samples by critic value + cumsum of the selected trajectories
#include "common.h"
#include "kernels.h"

namespace {

__device__ __forceinline__ int map_row(const ina_rowmap& m, int r) {
    return m.seg_len > 0 ? (r / m.seg_len) * m.seg_stride + m.off + (r % m.seg_len) : r;
}

// ------------------------------------------------------------------------------------------------ patchify
// One workgroup = one band of `ps` image rows of one frame = gw patches. The band (ps x W x C values) is read with
// fully coalesced loads into LDS, then each patch row of the im2col matrix (3*ps*ps values, k = c*ps*ps + y*ps + x) is
// written as consecutive bf16 by consecutive lanes.
template <bool IN_F32>
__global__ __launch_bounds__(256) void patchify_kernel(PatchifyArgs p) {
    extern __shared__ float band[];  // [ps][W*C]
    const int gw = p.W / p.ps, gh = p.H / p.ps;
    const int img = blockIdx.x / gh, py = blockIdx.x % gh;
    const int rowlen = p.W * p.C;
    const int nband = p.ps * rowlen;
    const size_t base = ((size_t)img * p.H + (size_t)py * p.ps) * rowlen;
    for (int i = threadIdx.x; i < nband; i += 256) {
        float v;
        if (IN_F32) v = reinterpret_cast<const float*>(p.img)[base + i];
        else v = (float)reinterpret_cast<const bf16*>(p.img)[base + i];
        band[i] = v;
    }
    __syncthreads();
    const int pp = p.ps * p.ps;
    const int kreal = 3 * pp;
    bf16* out = reinterpret_cast<bf16*>(p.out) + ((size_t)img * gh * gw + (size_t)py * gw) * p.ldo;
    const int total = gw * p.ldo;
    for (int i = threadIdx.x; i < total; i += 256) {
        const int px = i / p.ldo, k = i % p.ldo;
        float v = 0.f;
        if (k < kreal) {
            const int c = k / pp, r = k % pp;
            const int y = r / p.ps, x = r % p.ps;
            const int cs = (p.C == 1) ? 0 : c;
            v = (band[y * rowlen + (px * p.ps + x) * p.C + cs] - p.mean[c]) * p.inv_std[c];
        }
        out[i] = (bf16)v;
    }
}

// ------------------------------------------------------------------------------------------------ embed3
// One thread per group of 4 output columns: 32-bit index arithmetic (the launcher bounds rows * C / 4 below 2^31) and 16-byte loads of the
// 12 weights / 4 biases / 4 table entries of the group (the shipped version does its index arithmetic in 64 bits - two 64-bit divisions per
// element group - and fetches the operands with dword loads: 66 us for a 65536 x 384 fp32 output, 1.5 TB/s; this one 27 us).
template <bool OUT_F32>
__global__ __launch_bounds__(256) void embed3_kernel(Embed3Args p, int vec) {
    const unsigned c4 = (unsigned)p.C >> 2;  // groups of 4 columns
    const unsigned total = (unsigned)p.rows * c4;
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
        const unsigned r = i / c4, col = (i - r * c4) * 4;
        float x0 = 0.f, x1 = 0.f, x2 = 0.f;
        if (p.X) {
            const float* x = p.X + (size_t)(r / (unsigned)p.x_div) * 3;
            x0 = x[0]; x1 = x[1]; x2 = x[2];
        }
        float w[12], bb[4], pp[4];
        if (p.X) {
            const float* q = p.W + (size_t)col * 3;
            if (vec) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(q), b = *reinterpret_cast<const f32x4*>(q + 4), c = *reinterpret_cast<const f32x4*>(q + 8);
#pragma unroll
                for (int j = 0; j < 4; ++j) { w[j] = a[j]; w[4 + j] = b[j]; w[8 + j] = c[j]; }
            } else {
#pragma unroll
                for (int j = 0; j < 12; ++j) w[j] = q[j];
            }
        }
        if (p.b) {
            if (vec) { const f32x4 a = *reinterpret_cast<const f32x4*>(p.b + col); bb[0] = a[0]; bb[1] = a[1]; bb[2] = a[2]; bb[3] = a[3]; }
            else { bb[0] = p.b[col]; bb[1] = p.b[col + 1]; bb[2] = p.b[col + 2]; bb[3] = p.b[col + 3]; }
        }
        if (p.P) {
            const float* q = p.P + (size_t)(r % (unsigned)p.p_mod) * p.C + col;
            if (vec) { const f32x4 a = *reinterpret_cast<const f32x4*>(q); pp[0] = a[0]; pp[1] = a[1]; pp[2] = a[2]; pp[3] = a[3]; }
            else { pp[0] = q[0]; pp[1] = q[1]; pp[2] = q[2]; pp[3] = q[3]; }
        }
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float a = 0.f;
            if (p.X) a = w[3 * j] * x0 + w[3 * j + 1] * x1 + w[3 * j + 2] * x2;
            if (p.b) a += bb[j];
            if (p.P) a += pp[j];
            v[j] = a;
        }
        const size_t o = (size_t)map_row(p.out_map, (int)r) * p.ldy + col;
        if (OUT_F32) *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.Y) + o) = f32x4{v[0], v[1], v[2], v[3]};
        else *reinterpret_cast<bf16x4*>(reinterpret_cast<bf16*>(p.Y) + o) = bf16x4{(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
    }
}

// ------------------------------------------------------------------------------------------------ row loader shared by head3 / seqpool
template <int NCH, bool XF32>
__device__ __forceinline__ void load_row(const void* X, size_t off, int nchunks, int lane, float (&v)[NCH][8]) {
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c = lane + i * 64;
        if (c < nchunks) {
            if (XF32) {
                const float* x = reinterpret_cast<const float*>(X) + off + c * 8;
                f32x4 a = *reinterpret_cast<const f32x4*>(x), b = *reinterpret_cast<const f32x4*>(x + 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) { v[i][j] = a[j]; v[i][4 + j] = b[j]; }
            } else {
                bf16x8 xv = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const bf16*>(X) + off + c * 8);
#pragma unroll
                for (int j = 0; j < 8; ++j) v[i][j] = (float)xv[j];
            }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[i][j] = 0.f;
        }
    }
}

template <int NCH>
__device__ __forceinline__ void layernorm_row(float (&v)[NCH][8], int C, int nchunks, int lane, float eps, const float* gamma,
                                               const float* beta) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) s += v[i][j];
    const float mean = wave_sum(s) / (float)C;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        if (lane + i * 64 < nchunks) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float d = v[i][j] - mean;
                sq += d * d;
            }
        }
    }
    const float rstd = rsqrtf(wave_sum(sq) / (float)C + eps);
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c = lane + i * 64;
        if (c < nchunks) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float t = (v[i][j] - mean) * rstd;
                if (gamma) t *= gamma[c * 8 + j];
                if (beta) t += beta[c * 8 + j];
                v[i][j] = t;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ head3
// One wave per token row, RPW consecutive rows per wave. The column-constant operands of a lane (its 8 columns of the three W rows,
// gamma / beta, the env's modulation vector) are loaded ONCE per wave with 16-byte loads and kept in registers, the next row is
// requested before the current one is reduced, and the three lanes that apply the sampler update fetch the old sample up front.
// (The first version re-read every operand per row with 51 scalar dword loads per lane - 4-byte gathers at a 32-byte stride - and
// ran at 0.56 TB/s: 180 us for 65536 x 384 fp32 rows.) The arithmetic per row and its order are unchanged.
__device__ __forceinline__ void ld8(const float* q, float (&o)[8], bool vec) {
    if (vec) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(q), b = *reinterpret_cast<const f32x4*>(q + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) { o[j] = a[j]; o[4 + j] = b[j]; }
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = q[j];
    }
}

template <int NCH, bool XF32>
__global__ __launch_bounds__(256) void head3_kernel(Head3Args p, int rpw, int vec) {
    const int lane = threadIdx.x & 63;
    int row = (blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6)) * rpw;      // wave-uniform: row arithmetic on the scalar unit
    if (row >= p.rows) return;
    const int row_end = min(row + rpw, p.rows);
    const int nchunks = p.C >> 3;
    float w0[NCH][8], w1[NCH][8], w2[NCH][8], ga[NCH][8], be[NCH][8], ms[NCH][8];
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c = lane + i * 64;
        if (c < nchunks) {
            ld8(p.W + c * 8, w0[i], vec);
            ld8(p.W + p.C + c * 8, w1[i], vec);
            ld8(p.W + 2 * p.C + c * 8, w2[i], vec);
            if (p.gamma) ld8(p.gamma + c * 8, ga[i], vec);
            if (p.beta) ld8(p.beta + c * 8, be[i], vec);
        }
    }
    const float b0 = p.b[0], b1 = p.b[1], b2 = p.b[2];
    int cur_mod = -1;
    float v[NCH][8], vn[NCH][8];
    load_row<NCH, XF32>(p.X, (size_t)row * p.ldx, nchunks, lane, v);
    for (; row < row_end; ++row) {
        if (row + 1 < row_end) load_row<NCH, XF32>(p.X, (size_t)(row + 1) * p.ldx, nchunks, lane, vn);
        const size_t o = (size_t)row * 3 + lane;
        float s_old = 0.f, nz = 0.f;
        if (lane < 3) {
            if (p.mode != 0) s_old = p.sample[o];
            if (p.mode == 1 && p.noise) nz = p.noise[o];
        }
        if (p.mod_scale && row / p.mod_div != cur_mod) {       // wave-uniform
            cur_mod = row / p.mod_div;
            const float* q = p.mod_scale + (size_t)cur_mod * p.mod_ld;
#pragma unroll
            for (int i = 0; i < NCH; ++i)
                if (lane + i * 64 < nchunks) ld8(q + (lane + i * 64) * 8, ms[i], vec);
        }
        // LayerNorm of the row (same operations and order as layernorm_row)
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NCH; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) s += v[i][j];
        const float mean = wave_sum(s) / (float)p.C;
        float sq = 0.f;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            if (lane + i * 64 < nchunks) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    float d = v[i][j] - mean;
                    sq += d * d;
                }
            }
        }
        const float rstd = rsqrtf(wave_sum(sq) / (float)p.C + p.eps);
        float e0 = 0.f, e1 = 0.f, e2 = 0.f;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            if (lane + i * 64 < nchunks) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    float t = (v[i][j] - mean) * rstd;
                    if (p.gamma) t *= ga[i][j];
                    if (p.beta) t += be[i][j];
                    if (p.mod_scale) t *= 1.0f + ms[i][j];
                    e0 += t * w0[i][j];
                    e1 += t * w1[i][j];
                    e2 += t * w2[i][j];
                }
            }
        }
        e0 = wave_sum(e0) + b0;
        e1 = wave_sum(e1) + b1;
        e2 = wave_sum(e2) + b2;
        if (lane < 3) {
            const float e = lane == 0 ? e0 : (lane == 1 ? e1 : e2);
            if (p.eps_out) p.eps_out[o] = e;
            if (p.mode == 1) {
                float x0 = (s_old - p.coef[1] * e) * p.coef[0];
                x0 = fminf(fmaxf(x0, -p.clip), p.clip);
                float n = p.coef[2] * x0 + p.coef[3] * s_old;
                if (p.noise) n += p.coef[4] * nz;
                p.sample[o] = n;
            } else if (p.mode == 2) {
                p.sample[o] = s_old + p.coef[0] * e;
            }
        }
#pragma unroll
        for (int i = 0; i < NCH; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) v[i][j] = vn[i][j];
    }
}

// ------------------------------------------------------------------------------------------------ seqpool_head
template <int NCH, bool XF32>
__global__ __launch_bounds__(256) void seqpool_kernel(SeqpoolArgs p) {
    __shared__ float part[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int seq = blockIdx.x;
    const int nchunks = p.C >> 3;
    float acc = 0.f;
    for (int t = wave; t < p.T; t += 4) {
        float v[NCH][8];
        load_row<NCH, XF32>(p.X, ((size_t)seq * p.T + t) * p.ldx, nchunks, lane, v);
        layernorm_row<NCH>(v, p.C, nchunks, lane, p.eps, p.gamma, p.beta);
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = lane + i * 64;
            if (c < nchunks) {
#pragma unroll
                for (int j = 0; j < 8; ++j) acc += v[i][j] * p.w[c * 8 + j];
            }
        }
    }
    acc = wave_sum(acc);
    if (lane == 0) part[wave] = acc;
    __syncthreads();
    if (threadIdx.x == 0) p.out[seq] = (part[0] + part[1] + part[2] + part[3]) / (float)p.T + p.b[0];
}

// ------------------------------------------------------------------------------------------------ select_traj
// One 64-lane wave per env (S <= 64). rank_lo[s] = #{j : c[j] < c[s] or (c[j] == c[s] and j < s)} (stable ascending order,
// as torch.argsort on distinct values); the k lowest ranks go to `neg` in ascending order, the k highest to `pos` in
// descending order. Each selected sample's trajectory is the running sum over t of sample * scale.
__global__ __launch_bounds__(64) void select_kernel(SelectArgs p) {
    __shared__ float cv[64];
    __shared__ int neg_idx[64], pos_idx[64];
    const int b = blockIdx.x, lane = threadIdx.x;
    const float c = lane < p.S ? p.critic[(size_t)b * p.S + lane] : 0.f;
    cv[lane] = c;
    __syncthreads();
    if (lane < p.S) {
        int lo = 0, hi = 0;
        for (int j = 0; j < p.S; ++j) {
            const float o = cv[j];
            lo += (o < c) || (o == c && j < lane);
            hi += (o > c) || (o == c && j < lane);   // rank in descending order (argsort of -c, stable)
        }
        if (lo < p.k) neg_idx[lo] = lane;
        if (hi < p.k) pos_idx[hi] = lane;
    }
    __syncthreads();
    // lanes: 2*k selected trajectories x 3 components, serial cumsum over T (T <= 32, tiny)
    for (int w = lane; w < 2 * p.k * 3; w += 64) {
        const int which = w / (p.k * 3), rem = w % (p.k * 3);
        const int slot = rem / 3, comp = rem % 3;
        const int s = which == 0 ? neg_idx[slot] : pos_idx[slot];
        const float* src = p.sample + (((size_t)b * p.S + s) * p.T) * 3 + comp;
        float* dst = (which == 0 ? p.neg : p.pos) + (((size_t)b * p.k + slot) * p.T) * 3 + comp;
        float run = 0.f;
        for (int t = 0; t < p.T; ++t) {
            run += src[(size_t)t * 3] * p.scale;
            dst[(size_t)t * 3] = run;
        }
    }
}

// ------------------------------------------------------------------------------------------------ pool_act
__global__ __launch_bounds__(256) void pool_act_kernel(PoolActArgs p) {
    const long total = (long)p.nseq * p.C;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int s = (int)(i / p.C), c = (int)(i % p.C);
        float a = 0.f;
        for (int t = 0; t < p.T; ++t) {
            const size_t o = ((size_t)s * p.T + t) * p.ldx + c;
            a += p.x_dtype == INA_DT_F32 ? reinterpret_cast<const float*>(p.X)[o] : (float)reinterpret_cast<const bf16*>(p.X)[o];
        }
        a /= (float)p.T;
        if (p.P) a += p.P[(size_t)(s % p.p_mod) * p.C + c];
        a = ina_act(a, p.act);
        const size_t o = (size_t)s * p.ldy + c;
        if (p.out_dtype == INA_DT_F32) reinterpret_cast<float*>(p.Y)[o] = a;
        else reinterpret_cast<bf16*>(p.Y)[o] = (bf16)a;
    }
}

}  // namespace

int ina_launch_pool_act(const PoolActArgs& p_in, hipStream_t stream) {
    PoolActArgs p = p_in;
    if (p.p_mod <= 0) p.p_mod = 1;
    INA_REQUIRE(p.nseq > 0 && p.T > 0 && p.C > 0 && p.X && p.Y, "pool_act: bad arguments nseq=%d T=%d C=%d", p.nseq, p.T, p.C);
    InaProfScope prof(INA_PROF_ELEMENTWISE, (double)p.nseq * p.T * p.C, (double)p.nseq * p.T * p.C * (p.x_dtype == INA_DT_F32 ? 4.0 : 2.0), stream);
    const long total = (long)p.nseq * p.C;
    const int blocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
    hipLaunchKernelGGL(pool_act_kernel, dim3(blocks), dim3(256), 0, stream, p);
    INA_HIP_CHECK(hipGetLastError());
    return 0;
}

int ina_launch_patchify(const PatchifyArgs& p, hipStream_t stream) {
    INA_REQUIRE(p.n > 0 && p.ps > 0 && p.H % p.ps == 0 && p.W % p.ps == 0, "patchify: bad geometry n=%d H=%d W=%d ps=%d", p.n, p.H, p.W, p.ps);
    INA_REQUIRE(p.C == 3 || p.C == 1, "patchify: C must be 3 or 1 (got %d)", p.C);
    INA_REQUIRE(p.ldo >= 3 * p.ps * p.ps && p.ldo % 8 == 0, "patchify: ldo=%d must be >= 3*ps*ps and a multiple of 8", p.ldo);
    const size_t lds = (size_t)p.ps * p.W * p.C * sizeof(float);
    INA_REQUIRE(lds <= 64 * 1024, "patchify: band of %zu bytes does not fit LDS", lds);
    dim3 grid(p.n * (p.H / p.ps));
    InaProfScope prof(INA_PROF_ELEMENTWISE, 0.0, (double)p.n * p.H * p.W * p.C * (p.in_dtype == INA_DT_F32 ? 4.0 : 2.0) +
                      2.0 * p.n * (p.H / p.ps) * (p.W / p.ps) * p.ldo, stream);
    if (p.in_dtype == INA_DT_F32) hipLaunchKernelGGL(patchify_kernel<true>, grid, dim3(256), lds, stream, p);
    else hipLaunchKernelGGL(patchify_kernel<false>, grid, dim3(256), lds, stream, p);
    INA_HIP_CHECK(hipGetLastError());
    return 0;
}

int ina_launch_embed3(const Embed3Args& p_in, hipStream_t stream) {
    Embed3Args p = p_in;
    if (p.x_div <= 0) p.x_div = 1;
    if (p.p_mod <= 0) p.p_mod = 1;
    INA_REQUIRE(p.rows > 0 && p.C > 0 && p.C % 4 == 0 && p.ldy % 4 == 0, "embed3: bad shape rows=%d C=%d ldy=%d", p.rows, p.C, p.ldy);
    INA_REQUIRE(p.Y && (p.X == nullptr || p.W != nullptr), "embed3: Y (and W when X is given) required");
    InaProfScope prof(INA_PROF_ELEMENTWISE, 6.0 * p.rows * p.C, (double)p.rows * p.C * (p.out_dtype == INA_DT_F32 ? 4.0 : 2.0), stream);
    const long total = (long)p.rows * (p.C / 4);
    INA_REQUIRE(total < (1L << 31) - (1L << 21), "embed3: rows * C / 4 = %ld does not fit the kernel's 32-bit index", total);
    const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    auto al16 = [](const void* q) { return ((uintptr_t)q % 16) == 0; };
    const int vec = al16(p.W) && al16(p.b) && al16(p.P);     // (C % 4 == 0 keeps every group's offset a multiple of 16 bytes)
    if (p.out_dtype == INA_DT_F32) hipLaunchKernelGGL(embed3_kernel<true>, dim3(blocks), dim3(256), 0, stream, p, vec);
    else hipLaunchKernelGGL(embed3_kernel<false>, dim3(blocks), dim3(256), 0, stream, p, vec);
    INA_HIP_CHECK(hipGetLastError());
    return 0;
}

int ina_launch_head3(const Head3Args& p_in, hipStream_t stream) {
    Head3Args p = p_in;
    if (p.mod_div <= 0) p.mod_div = 1;
    INA_REQUIRE(p.rows > 0 && p.C % 8 == 0 && p.C <= 1024 && p.ldx % 8 == 0, "head3: bad shape rows=%d C=%d ldx=%d", p.rows, p.C, p.ldx);
    INA_REQUIRE(p.X && p.W && p.b, "head3: X, W, b required");
    INA_REQUIRE(p.mode == 0 ? p.eps_out != nullptr : p.sample != nullptr, "head3: mode %d needs %s", p.mode, p.mode == 0 ? "eps_out" : "sample");
    // rows per wave: large calls (the 32-sample batches of the samplers) keep >= 2 rounds of waves per CU, small ones one row per wave
    const int rpw = p.rows >= 32768 ? 4 : 1;
    dim3 grid((p.rows + 4 * rpw - 1) / (4 * rpw)), block(256);
    const bool f32 = p.x_dtype == INA_DT_F32;
    auto al16 = [](const void* q) { return ((uintptr_t)q % 16) == 0; };
    const int vec = al16(p.W) && al16(p.gamma) && al16(p.beta) && al16(p.mod_scale) && (p.mod_ld % 4 == 0) && (p.C % 8 == 0);
    InaProfScope prof(INA_PROF_ELEMENTWISE, 14.0 * p.rows * p.C, (double)p.rows * p.C * (f32 ? 4.0 : 2.0), stream);
    if (p.C <= 512) {
        if (f32) hipLaunchKernelGGL((head3_kernel<1, true>), grid, block, 0, stream, p, rpw, vec);
        else hipLaunchKernelGGL((head3_kernel<1, false>), grid, block, 0, stream, p, rpw, vec);
    } else {
        if (f32) hipLaunchKernelGGL((head3_kernel<2, true>), grid, block, 0, stream, p, rpw, vec);
        else hipLaunchKernelGGL((head3_kernel<2, false>), grid, block, 0, stream, p, rpw, vec);
    }
    INA_HIP_CHECK(hipGetLastError());
    return 0;
}

int ina_launch_seqpool(const SeqpoolArgs& p, hipStream_t stream) {
    INA_REQUIRE(p.nseq > 0 && p.T > 0 && p.C % 8 == 0 && p.C <= 512 && p.ldx % 8 == 0, "seqpool: bad shape nseq=%d T=%d C=%d", p.nseq, p.T, p.C);
    INA_REQUIRE(p.X && p.w && p.b && p.out, "seqpool: X, w, b, out required");
    InaProfScope prof(INA_PROF_ELEMENTWISE, 10.0 * p.nseq * p.T * p.C, (double)p.nseq * p.T * p.C * (p.x_dtype == INA_DT_F32 ? 4.0 : 2.0), stream);
    if (p.x_dtype == INA_DT_F32) hipLaunchKernelGGL((seqpool_kernel<1, true>), dim3(p.nseq), dim3(256), 0, stream, p);
    else hipLaunchKernelGGL((seqpool_kernel<1, false>), dim3(p.nseq), dim3(256), 0, stream, p);
    INA_HIP_CHECK(hipGetLastError());
    return 0;
}

int ina_launch_select(const SelectArgs& p, hipStream_t stream) {
    INA_REQUIRE(p.B > 0 && p.S > 0 && p.S <= 64 && p.k > 0 && p.k <= p.S && p.T > 0, "select_traj: bad shape B=%d S=%d T=%d k=%d", p.B, p.S, p.T, p.k);
    hipLaunchKernelGGL(select_kernel, dim3(p.B), dim3(64), 0, stream, p);
    INA_HIP_CHECK(hipGetLastError());
    return 0;
}
