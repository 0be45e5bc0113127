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

extern "C" {

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

int ina_attention_bf16(const ina_attn_args* args, void* stream) {
    INA_REQUIRE(args != nullptr, "attention: null args");
    return ina_launch_attention(*args, reinterpret_cast<hipStream_t>(stream));
}

int ina_norm_bf16(const ina_norm_args* args, void* stream) {
    INA_REQUIRE(args != nullptr, "norm: null args");
    return ina_launch_norm(*args, reinterpret_cast<hipStream_t>(stream));
}

}  // extern "C"
