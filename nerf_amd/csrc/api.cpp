// Library-level entry points: version, error text, device info.
#include "common.hpp"

#include <cstring>

namespace nerfx {

std::string& last_error_ref() {
    static thread_local std::string msg;
    return msg;
}

int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    last_error_ref() = buf;
    return code;
}

}  // namespace nerfx

extern "C" int nerfx_version(void) { return NERFX_VERSION; }

extern "C" const char* nerfx_last_error(void) { return nerfx::last_error_ref().c_str(); }

extern "C" int nerfx_device_info(int device, nerfx_device_info_t* out) {
    NERFX_REQUIRE(out != nullptr, "out is NULL");
    hipDeviceProp_t p;
    NERFX_HIP(hipGetDeviceProperties(&p, device));
    std::memset(out, 0, sizeof *out);
    out->compute_units = p.multiProcessorCount;
    out->wavefront = p.warpSize;
    out->lds_bytes_per_cu = (int32_t)p.maxSharedMemoryPerMultiProcessor;
    out->clock_khz = p.clockRate;
    out->hbm_bytes = (int64_t)p.totalGlobalMem;
    std::strncpy(out->arch, p.gcnArchName, sizeof out->arch - 1);
    return 0;
}
