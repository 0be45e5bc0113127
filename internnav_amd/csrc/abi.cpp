// extern "C" entry points of libinternnav_amd.so (declared in include/internnav_amd.h).
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include "common.h"
#include "kernels.h"

static thread_local char g_err[1024] = "";

void ina_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// ---- per-launch event timing -----------------------------------------------------------------------------------
#include <mutex>
#include <vector>
namespace {
struct ProfRec { int kind, sub; hipEvent_t a, b; double flops, bytes; };
thread_local int g_prof_sub = 0;
bool g_prof_on = false;
std::vector<ProfRec> g_prof;
std::mutex g_prof_mu;
}  // namespace

InaProfScope::InaProfScope(int kind, double flops, double bytes, hipStream_t s) : idx(-1), stream(s) {
    if (!g_prof_on) return;
    ProfRec r{kind, g_prof_sub, nullptr, nullptr, flops, bytes};
    g_prof_sub = 0;
    if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) return;
    (void)hipEventRecord(r.a, s);
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof.push_back(r);
    idx = (int)g_prof.size() - 1;
}
void ina_prof_set_sub(int sub) { g_prof_sub = sub; }

InaProfScope::~InaProfScope() {
    if (idx < 0) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    (void)hipEventRecord(g_prof[idx].b, stream);
}

// ---- workspace slots -----------------------------------------------------------------------------------------------
// The scratch pointer is baked into every graph captured under a slot, so a buffer that a graph has seen is NEVER freed: when a
// later (eager) launch needs more, a new buffer is allocated for new launches and the old one is retired but kept alive until the
// process ends (graphs captured earlier keep replaying into it). The current slot is per host thread, like the error string.
namespace {
thread_local int g_ws_slot = 0;
float* g_ws_ptr[INA_WS_KINDS][INA_WS_SLOTS] = {};
size_t g_ws_cap[INA_WS_KINDS][INA_WS_SLOTS] = {};
bool g_ws_captured[INA_WS_KINDS][INA_WS_SLOTS] = {};
std::vector<float*> g_ws_retired;
std::mutex g_ws_mu;
}  // namespace

int ina_workspace(int kind, size_t bytes, hipStream_t stream, float** out) {
    std::lock_guard<std::mutex> lk(g_ws_mu);
    const int slot = g_ws_slot;
    float*& ptr = g_ws_ptr[kind][slot];
    size_t& cap = g_ws_cap[kind][slot];
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(stream, &st);
    if (bytes > cap) {
        INA_REQUIRE(st == hipStreamCaptureStatusNone,
                    "workspace (kind %d, slot %d) of %zu bytes needed during graph capture: run the shape once eagerly under this slot first", kind,
                    slot, bytes);
        if (ptr) {
            if (g_ws_captured[kind][slot]) {
                g_ws_retired.push_back(ptr);              // a captured graph still points here: keep the memory alive
            } else {
                INA_HIP_CHECK(hipDeviceSynchronize());
                INA_HIP_CHECK(hipFree(ptr));
            }
            ptr = nullptr;
            cap = 0;
        }
        const size_t want = bytes < (size_t)(32u << 20) ? (size_t)(32u << 20) : bytes * 2;
        INA_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&ptr), want));
        cap = want;
        g_ws_captured[kind][slot] = false;
    }
    if (st != hipStreamCaptureStatusNone) g_ws_captured[kind][slot] = true;
    *out = ptr;
    return 0;
}

extern "C" {

int ina_workspace_retired(void) {
    std::lock_guard<std::mutex> lk(g_ws_mu);
    return (int)g_ws_retired.size();
}

int ina_set_workspace_slot(int slot) {
    INA_REQUIRE(slot >= 0 && slot < INA_WS_SLOTS, "workspace slot %d out of range [0, %d)", slot, INA_WS_SLOTS);
    g_ws_slot = slot;
    return 0;
}

int ina_prof_enable(int on) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (auto& r : g_prof) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    g_prof.clear();
    g_prof_on = on != 0;
    return 0;
}

static int prof_read_impl(int kind, int sub, double* ms_total, int64_t* launches, double* flops, double* bytes) {
    INA_REQUIRE(kind >= 0 && kind < INA_PROF_KINDS, "prof_read: bad kind %d", kind);
    std::lock_guard<std::mutex> lk(g_prof_mu);
    double ms = 0, fl = 0, by = 0;
    int64_t n = 0;
    for (auto& r : g_prof) {
        if (r.kind != kind || (sub >= 0 && r.sub != sub)) continue;
        INA_HIP_CHECK(hipEventSynchronize(r.b));
        float t = 0.f;
        INA_HIP_CHECK(hipEventElapsedTime(&t, r.a, r.b));
        ms += t; fl += r.flops; by += r.bytes; ++n;
    }
    if (ms_total) *ms_total = ms;
    if (launches) *launches = n;
    if (flops) *flops = fl;
    if (bytes) *bytes = by;
    return 0;
}

int ina_prof_read(int kind, double* ms_total, int64_t* launches, double* flops, double* bytes) {
    return prof_read_impl(kind, -1, ms_total, launches, flops, bytes);
}

int ina_prof_read_sub(int kind, int sub, double* ms_total, int64_t* launches, double* flops, double* bytes) {
    return prof_read_impl(kind, sub, ms_total, launches, flops, bytes);
}

int ina_abi_version(void) { return INA_ABI_VERSION; }

const char* ina_last_error(void) { return g_err; }

int ina_device_check(char* name, int n) {
    int dev = 0;
    INA_HIP_CHECK(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    INA_HIP_CHECK(hipGetDeviceProperties(&prop, dev));
    if (name && n > 0) {
        strncpy(name, prop.gcnArchName, (size_t)n - 1);
        name[n - 1] = 0;
    }
    INA_REQUIRE(strncmp(prop.gcnArchName, "gfx950", 6) == 0, "internnav_amd is built for gfx950 only, device is %s", prop.gcnArchName);
    return 0;
}

int ina_gemm_bf16(const ina_gemm_args* args, void* stream) {
    INA_REQUIRE(args != nullptr, "gemm: null args");
    return ina_launch_gemm(*args, reinterpret_cast<hipStream_t>(stream));
}

int ina_gemm_preshuffle(const void* W, void* Wp, int32_t N, int32_t K, int64_t ldw, void* stream) {
    return ina_launch_gemm_preshuffle(W, Wp, N, K, (long)ldw, reinterpret_cast<hipStream_t>(stream));
}

int ina_gemm_select(const ina_gemm_args* args, int* kernel) {
    INA_REQUIRE(args != nullptr && kernel != nullptr, "gemm_select: null argument");
    GemmArgs p;
    return ina_plan_gemm(*args, p, *kernel);
}

int ina_attention_bf16(const ina_attn_args* args, void* stream) {
    INA_REQUIRE(args != nullptr, "attention: null args");
    return ina_launch_attention(*args, reinterpret_cast<hipStream_t>(stream));
}

int ina_norm_bf16(const ina_norm_args* args, void* stream) {
    INA_REQUIRE(args != nullptr, "norm: null args");
    return ina_launch_norm(*args, reinterpret_cast<hipStream_t>(stream));
}

/* sizeof() of the k-th argument struct (layout check of the ctypes mirrors): 0 gemm, 1 attn, 2 norm, 3 patchify, 4 embed3, 5 head3, 6 seqpool, 7 select, 8 pool_act, 9 gather, 10 rope, 11 mrope_table, 12 argmax, 13 dit_attn, 14 resize_u8, 15 qwen_patchify, 16 u8_lut, 17 resize_f32, 18 gn_mish, 19 pad_rows, 20 ddim_step, 21 ew, 22 colsum, 23 norm_bwd, 24 transpose, 25 sparse_rows, 26 small_linear, 27 mse, 28 adamw, 29 gemm_nn, 30 attn_bwd, 31 dit_rowchain, 32 gemm_dw */
int ina_struct_size(int k) {
    switch (k) {
        case 0: return (int)sizeof(ina_gemm_args);
        case 1: return (int)sizeof(ina_attn_args);
        case 2: return (int)sizeof(ina_norm_args);
        case 3: return (int)sizeof(ina_patchify_args);
        case 4: return (int)sizeof(ina_embed3_args);
        case 5: return (int)sizeof(ina_head3_args);
        case 6: return (int)sizeof(ina_seqpool_args);
        case 7: return (int)sizeof(ina_select_args);
        case 8: return (int)sizeof(ina_pool_act_args);
        case 9: return (int)sizeof(ina_gather_args);
        case 10: return (int)sizeof(ina_rope_args);
        case 11: return (int)sizeof(ina_mrope_table_args);
        case 12: return (int)sizeof(ina_argmax_args);
        case 13: return (int)sizeof(ina_dit_attn_args);
        case 14: return (int)sizeof(ina_resize_u8_args);
        case 15: return (int)sizeof(ina_qwen_patchify_args);
        case 16: return (int)sizeof(ina_u8_lut_args);
        case 17: return (int)sizeof(ina_resize_f32_args);
        case 18: return (int)sizeof(ina_gn_mish_args);
        case 19: return (int)sizeof(ina_pad_rows_args);
        case 20: return (int)sizeof(ina_ddim_step_args);
        case 21: return (int)sizeof(ina_ew_args);
        case 22: return (int)sizeof(ina_colsum_args);
        case 23: return (int)sizeof(ina_norm_bwd_args);
        case 24: return (int)sizeof(ina_transpose_args);
        case 25: return (int)sizeof(ina_sparse_rows_args);
        case 26: return (int)sizeof(ina_small_linear_args);
        case 27: return (int)sizeof(ina_mse_args);
        case 28: return (int)sizeof(ina_adamw_args);
        case 29: return (int)sizeof(ina_gemm_nn_args);
        case 30: return (int)sizeof(ina_attn_bwd_args);
        case 31: return (int)sizeof(ina_dit_rowchain_args);
        case 32: return (int)sizeof(ina_gemm_dw_args);
        default: return -1;
    }
}

#define INA_ENTRY(name, T, launch)                                             \
    int name(const T* args, void* stream) {                                     \
        INA_REQUIRE(args != nullptr, #name ": null args");                      \
        return launch(*args, reinterpret_cast<hipStream_t>(stream));            \
    }
INA_ENTRY(ina_patchify, ina_patchify_args, ina_launch_patchify)
INA_ENTRY(ina_embed3, ina_embed3_args, ina_launch_embed3)
INA_ENTRY(ina_head3, ina_head3_args, ina_launch_head3)
INA_ENTRY(ina_seqpool_head, ina_seqpool_args, ina_launch_seqpool)
INA_ENTRY(ina_select_traj, ina_select_args, ina_launch_select)
INA_ENTRY(ina_pool_act, ina_pool_act_args, ina_launch_pool_act)
INA_ENTRY(ina_gather_rows, ina_gather_args, ina_launch_gather)
INA_ENTRY(ina_rope_bf16, ina_rope_args, ina_launch_rope)
INA_ENTRY(ina_mrope_table, ina_mrope_table_args, ina_launch_mrope_table)
INA_ENTRY(ina_dit_attention, ina_dit_attn_args, ina_launch_dit_attention)
INA_ENTRY(ina_resize_u8, ina_resize_u8_args, ina_launch_resize_u8)
INA_ENTRY(ina_qwen_patchify_u8, ina_qwen_patchify_args, ina_launch_qwen_patchify_u8)
INA_ENTRY(ina_u8_lut, ina_u8_lut_args, ina_launch_u8_lut)
INA_ENTRY(ina_resize_f32, ina_resize_f32_args, ina_launch_resize_f32)
INA_ENTRY(ina_argmax_rows, ina_argmax_args, ina_launch_argmax)
INA_ENTRY(ina_dit_rowchain, ina_dit_rowchain_args, ina_launch_dit_rowchain)
INA_ENTRY(ina_gn_mish, ina_gn_mish_args, ina_launch_gn_mish)
INA_ENTRY(ina_pad_rows, ina_pad_rows_args, ina_launch_pad_rows)
INA_ENTRY(ina_ddim_step, ina_ddim_step_args, ina_launch_ddim_step)
INA_ENTRY(ina_ew, ina_ew_args, ina_launch_ew)
INA_ENTRY(ina_colsum, ina_colsum_args, ina_launch_colsum)
INA_ENTRY(ina_norm_bwd, ina_norm_bwd_args, ina_launch_norm_bwd)
INA_ENTRY(ina_transpose, ina_transpose_args, ina_launch_transpose)
INA_ENTRY(ina_sparse_rows, ina_sparse_rows_args, ina_launch_sparse_rows)
INA_ENTRY(ina_small_linear, ina_small_linear_args, ina_launch_small_linear)
INA_ENTRY(ina_mse_masked, ina_mse_args, ina_launch_mse)
INA_ENTRY(ina_adamw, ina_adamw_args, ina_launch_adamw)
INA_ENTRY(ina_gemm_nn_bf16, ina_gemm_nn_args, ina_launch_gemm_nn)
INA_ENTRY(ina_gemm_dw, ina_gemm_dw_args, ina_launch_gemm_dw)
INA_ENTRY(ina_attention_bwd_bf16, ina_attn_bwd_args, ina_launch_attention_bwd)
#undef INA_ENTRY

}  // extern "C"
