// On-device frame pre-processing for gfx950: the byte / integer work the reference does on the host with PIL and the
// HF image processor before every System-2 / System-1 call, bit-exact:
//   * resize_u8:   one axis of PIL's ImagingResample for 8-bit images (libImaging/Resample.c, Pillow 12.x): 22-bit fixed-point
//                  filter coefficients (tables built on the host exactly as precompute_coeffs / normalize_coeffs_8bpc do), int32
//                  accumulation from 1 << 21, arithmetic shift, clip to 0..255. Image.resize = horizontal pass, then vertical pass
//                  with an 8-bit intermediate. Reference call sites: internvla_n1_policy.py:105-116 (resize_w x resize_h),
//                  internvla_n1_agent.py:309-320 (224 x 224 look-down pair), HF Qwen2VLImageProcessor.resize (smart_resize).
//   * qwen_patchify_u8: HF Qwen2VLImageProcessor rescale + normalize + patchify as one table lookup per byte (the 3 x 256 fp32
//                  table is computed on the host with the processor's own arithmetic) written in the processor's patch layout:
//                  rows (grid_h/m, grid_w/m, m, m), columns (C, T = 2 duplicated frames, ps, ps); bf16 out (the policy casts).
//   * u8_lut:      out[i] = bf16(table[in[i]])  (np.array(img) / 255.0 of the System-1 frames).
// HBM-bound byte kernels: algorithmic bytes = in + out; they run once per frame, not per sampler step.
#include "common.h"
#include "kernels.h"

namespace {

constexpr int PIL_PRECISION_BITS = 32 - 8 - 2;

__global__ __launch_bounds__(256) void resize_u8_kernel(ResizeU8Args p) {
    // tensor viewed as [outer][n_in][inner] -> [outer][n_out][inner]
    const long total = (long)p.outer * p.n_out * p.inner;
    const unsigned char* __restrict__ in = reinterpret_cast<const unsigned char*>(p.in);
    unsigned char* __restrict__ out = reinterpret_cast<unsigned char*>(p.out);
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int ii = (int)(idx % p.inner);
        const long t = idx / p.inner;
        const int xx = (int)(t % p.n_out);
        const long o = t / p.n_out;
        const int xmin = p.bounds[2 * xx], xmax = p.bounds[2 * xx + 1];
        const int* __restrict__ k = p.coefs + (size_t)xx * p.ksize;
        const unsigned char* src = in + ((size_t)o * p.n_in + xmin) * p.inner + ii;
        int ss = 1 << (PIL_PRECISION_BITS - 1);
        for (int x = 0; x < xmax; ++x) ss += (int)src[(size_t)x * p.inner] * k[x];
        ss >>= PIL_PRECISION_BITS;
        out[idx] = (unsigned char)(ss < 0 ? 0 : (ss > 255 ? 255 : ss));
    }
}

// PIL's 32-bit-float resample (mode "F": the depth frames): double accumulation in tap order with double coefficients, one rounding
// to float at the end (ImagingResampleHorizontal_32bpc / Vertical_32bpc). Products and sums are kept as separate IEEE operations
// (__dmul_rn / __dadd_rn): the host library is built without FMA contraction.
__global__ __launch_bounds__(256) void resize_f32_kernel(ResizeF32Args p) {
    const long total = (long)p.outer * p.n_out * p.inner;
    const float* __restrict__ in = reinterpret_cast<const float*>(p.in);
    float* __restrict__ out = reinterpret_cast<float*>(p.out);
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int ii = (int)(idx % p.inner);
        const long t = idx / p.inner;
        const int xx = (int)(t % p.n_out);
        const long o = t / p.n_out;
        const int xmin = p.bounds[2 * xx], xmax = p.bounds[2 * xx + 1];
        const double* __restrict__ k = p.coefs + (size_t)xx * p.ksize;
        const float* src = in + ((size_t)o * p.n_in + xmin) * p.inner + ii;
        double ss = 0.0;
        for (int x = 0; x < xmax; ++x) ss = __dadd_rn(ss, __dmul_rn((double)src[(size_t)x * p.inner], k[x]));
        out[idx] = (float)ss;
    }
}

__global__ __launch_bounds__(256) void qwen_patchify_u8_kernel(QwenPatchifyArgs p) {
    const int gh = p.H / p.ps, gw = p.W / p.ps, m = p.merge;
    const int pp = p.ps * p.ps, cols = 3 * p.tdup * pp;
    const long total = (long)p.n * gh * gw * cols;
    const unsigned char* __restrict__ img = reinterpret_cast<const unsigned char*>(p.img);
    bf16* __restrict__ out = reinterpret_cast<bf16*>(p.out);
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int col = (int)(idx % cols);
        const long r = idx / cols;
        const int i = (int)(r / (gh * gw)), rr = (int)(r % (gh * gw));
        const int cell = rr / (m * m), within = rr % (m * m);
        const int py = (cell / (gw / m)) * m + within / m, px = (cell % (gw / m)) * m + within % m;
        const int c = col / (p.tdup * pp), y = (col % pp) / p.ps, x = col % p.ps;   // the T frames are copies of the one image
        const unsigned char v = img[(((size_t)i * p.H + py * p.ps + y) * p.W + px * p.ps + x) * 3 + c];
        out[(size_t)r * p.ldo + col] = (bf16)p.lut[c * 256 + v];
    }
}

__global__ __launch_bounds__(256) void u8_lut_kernel(U8LutArgs p) {
    const unsigned char* __restrict__ in = reinterpret_cast<const unsigned char*>(p.in);
    bf16* __restrict__ out = reinterpret_cast<bf16*>(p.out);
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < p.n; idx += (long)gridDim.x * 256) out[idx] = (bf16)p.lut[in[idx]];
}

int grid_for(long total) { return (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192); }

}  // namespace

int ina_launch_resize_u8(const ResizeU8Args& p, hipStream_t stream) {
    INA_REQUIRE(p.in && p.out && p.bounds && p.coefs, "resize_u8: null pointer");
    INA_REQUIRE(p.outer > 0 && p.n_in > 0 && p.n_out > 0 && p.inner > 0 && p.ksize > 0, "resize_u8: bad geometry outer=%d n_in=%d n_out=%d inner=%d ksize=%d",
                p.outer, p.n_in, p.n_out, p.inner, p.ksize);
    const long total = (long)p.outer * p.n_out * p.inner;
    InaProfScope prof(INA_PROF_ELEMENTWISE, 2.0 * total * p.ksize, (double)p.outer * p.inner * (p.n_in + p.n_out), stream);
    hipLaunchKernelGGL(resize_u8_kernel, dim3(grid_for(total)), dim3(256), 0, stream, p);
    INA_HIP_CHECK(hipGetLastError());
    return 0;
}

int ina_launch_resize_f32(const ResizeF32Args& p, hipStream_t stream) {
    INA_REQUIRE(p.in && p.out && p.bounds && p.coefs, "resize_f32: null pointer");
    INA_REQUIRE(p.outer > 0 && p.n_in > 0 && p.n_out > 0 && p.inner > 0 && p.ksize > 0, "resize_f32: bad geometry outer=%d n_in=%d n_out=%d inner=%d ksize=%d",
                p.outer, p.n_in, p.n_out, p.inner, p.ksize);
    const long total = (long)p.outer * p.n_out * p.inner;
    InaProfScope prof(INA_PROF_ELEMENTWISE, 2.0 * total * p.ksize, 4.0 * p.outer * p.inner * (p.n_in + p.n_out), stream);
    hipLaunchKernelGGL(resize_f32_kernel, dim3(grid_for(total)), dim3(256), 0, stream, p);
    INA_HIP_CHECK(hipGetLastError());
    return 0;
}

int ina_launch_qwen_patchify_u8(const QwenPatchifyArgs& p, hipStream_t stream) {
    INA_REQUIRE(p.img && p.out && p.lut, "qwen_patchify_u8: null pointer");
    INA_REQUIRE(p.n > 0 && p.ps > 0 && p.merge > 0 && p.tdup > 0 && p.H % (p.ps * p.merge) == 0 && p.W % (p.ps * p.merge) == 0,
                "qwen_patchify_u8: H=%d W=%d must be multiples of ps*merge=%d", p.H, p.W, p.ps * p.merge);
    INA_REQUIRE(p.ldo >= 3 * p.tdup * p.ps * p.ps, "qwen_patchify_u8: ldo=%d too small", p.ldo);
    const long total = (long)p.n * (p.H / p.ps) * (p.W / p.ps) * 3 * p.tdup * p.ps * p.ps;
    InaProfScope prof(INA_PROF_ELEMENTWISE, 0.0, (double)p.n * p.H * p.W * 3 + 2.0 * total, stream);
    hipLaunchKernelGGL(qwen_patchify_u8_kernel, dim3(grid_for(total)), dim3(256), 0, stream, p);
    INA_HIP_CHECK(hipGetLastError());
    return 0;
}

int ina_launch_u8_lut(const U8LutArgs& p, hipStream_t stream) {
    INA_REQUIRE(p.in && p.out && p.lut && p.n > 0, "u8_lut: bad arguments");
    InaProfScope prof(INA_PROF_ELEMENTWISE, 0.0, 3.0 * p.n, stream);
    hipLaunchKernelGGL(u8_lut_kernel, dim3(grid_for(p.n)), dim3(256), 0, stream, p);
    INA_HIP_CHECK(hipGetLastError());
    return 0;
}
